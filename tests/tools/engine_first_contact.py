"""First contact of a command-engine change with the GPU: a few streams of the metric's make-up against the oracle, with the
engine's share of the commands.  Usage: python tests/tools/engine_first_contact.py [n_streams] [size_KiB]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util

import oracle_lib as oracle
import workloads as w


def load_pkg():
    spec = importlib.util.spec_from_file_location("rust_brotli_decompressor_amd", os.path.join(ROOT, "rust-brotli-decompressor_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec); sys.modules["rust_brotli_decompressor_amd"] = mod; spec.loader.exec_module(mod)
    return mod


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    size = (int(sys.argv[2]) if len(sys.argv) > 2 else 4096) << 10
    pkg = load_pkg()
    streams = w.make_streams("long_backref", n, size, 1000)
    datas = [s[0] for s in streams]; caps = [s[1] for s in streams]
    batch = pkg.Batch(len(datas))
    t0 = time.time()
    results, outs = batch.decode_host(datas, caps, 1)
    dt = time.time() - t0
    ms = batch.last_kernel_ms() if hasattr(batch, "last_kernel_ms") else -1
    batch.close()
    bad = 0
    for i, (d, cap) in enumerate(zip(datas, caps)):
        info, exp = oracle.decode(d, cap, 1)
        r = results[i]
        ok = (r.result, r.error_code, r.decoded_size) == (info.result, info.error_code, info.decoded_size) and outs[i] == exp and r.num_commands == info.num_commands
        first = next((k for k in range(min(len(outs[i]), len(exp))) if outs[i][k] != exp[k]), -1)
        print("stream %d: %s result %d/%d code %d/%d size %d/%d commands %d/%d engine %d first diff %d" % (
            i, "ok" if ok else "BAD", r.result, info.result, r.error_code, info.error_code, r.decoded_size, info.decoded_size, r.num_commands, info.num_commands, r.engine_commands, first))
        bad += 0 if ok else 1
    print("host call %.3f s, kernel %.3f ms; %d bad of %d" % (dt, ms, bad, n))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
