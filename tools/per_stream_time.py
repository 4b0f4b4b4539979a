"""Kernel time of every stream of the metric's batch decoded on its own (a batch's time is its slowest stream's):
  python tools/per_stream_time.py [n_streams] [workload kind]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
import workloads as w
from conftest import load_pkg
pkg = load_pkg()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = sys.argv[2] if len(sys.argv) > 2 else "long_backref"
streams = w.make_streams(kind, n, 4 << 20, 1000 if kind == "long_backref" else 2000)
rows = []
b = pkg.Batch(1)
for i, (c, sz, sha) in enumerate(streams):
    inp = torch.frombuffer(bytearray(c), dtype=torch.uint8).cuda()
    out = torch.zeros(sz, dtype=torch.uint8, device="cuda")
    b.decode_device([inp.data_ptr()], [len(c)], [out.data_ptr()], [sz])
    r = b.wait()[0]
    assert r.result == 1 and r.decoded_size == sz
    ms = []
    for _ in range(2):
        b.relaunch(); b.wait(); ms.append(b.last_kernel_ms())
    rows.append((min(ms), i, len(c), r.num_commands, r.engine_commands))
b.close()
rows.sort()
t = [r[0] for r in rows]
print("streams %d: kernel ms alone  min %.3f  median %.3f  p90 %.3f  max %.3f" % (len(t), t[0], t[len(t) // 2], t[int(len(t) * 0.9)], t[-1]))
for r in rows[-6:]:
    print("  slow: stream %3d  %.3f ms  compressed %d  commands %d (engine %d)" % (r[1], r[0], r[2], r[3], r[4]))
for r in rows[:3]:
    print("  fast: stream %3d  %.3f ms  compressed %d  commands %d (engine %d)" % (r[1], r[0], r[2], r[3], r[4]))
