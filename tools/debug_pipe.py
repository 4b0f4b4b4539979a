"""Debug aid for the two-engine path engine: decode a few long-back-reference streams, find the first byte that differs.
  python tools/debug_pipe.py [n_streams] [size KiB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import workloads as w
from conftest import load_pkg
pkg = load_pkg()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
size = (int(sys.argv[2]) if len(sys.argv) > 2 else 4096) << 10
raws = [w.long_backref_stream(1000 + i, size) for i in range(n)]
comp = [w.brotli_compress(r) for r in raws]
b = pkg.Batch(n)
res, outs = b.decode_host(comp, [size] * n, 1)
b.close()
for i in range(n):
    r = res[i]
    o = outs[i]
    first = next((k for k in range(min(len(o), size)) if o[k] != raws[i][k]), None)
    print("stream %d: result %d error %d decoded %d consumed %d/%d commands %d engine %d; first difference at %s" % (
        i, r.result, r.error_code, r.decoded_size, r.consumed, len(comp[i]), r.num_commands, r.engine_commands, first))
