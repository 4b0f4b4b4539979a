"""Bring-up helper: decode each reference fixture alone on the GPU and diff against the oracle."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402
import oracle_lib as oracle  # noqa: E402

pkg = load_pkg()
m = [e for e in json.load(open(os.path.join(ROOT, "tests/golden/manifest.json"))) if e["name"] != "rnd_chunk.br"]
names = sys.argv[1:] or [e["name"] for e in m]
for e in m:
    if e["name"] not in names:
        continue
    d = open(os.path.join(ROOT, "tests/golden/testdata", e["name"]), "rb").read()
    cap = e.get("size", 1 << 16) + 16
    b = pkg.Batch(1)
    t = time.time()
    try:
        res, outs = b.decode_host([d], [cap], 1)
    except Exception as ex:
        print("EXC", e["name"], ex)
        continue
    dt = time.time() - t
    b.close()
    info, exp = oracle.decode(d, cap, 1)
    r = res[0]
    ok = (r.result, r.error_code, r.decoded_size, outs[0]) == (info.result, info.error_code, info.decoded_size, exp)
    first = next((i for i in range(min(len(exp), len(outs[0]))) if exp[i] != outs[0][i]), None)
    print("OK " if ok else "BAD", e["name"], "gpu", r.result, r.error_code, r.decoded_size, r.produced, "cmds", r.num_commands,
          "| oracle", info.result, info.error_code, info.decoded_size, info.num_commands, "first_diff", first, "%.1f ms" % (dt * 1e3), flush=True)
