"""Where the metric's kernel time goes by part of a stream: the first `cut` KiB of each of n long-back-reference streams
(the seed part: Zipf literals, short copies) compressed on their own against the whole 4 MiB streams.
  python tools/prof_c3_parts.py [n] [cut KiB ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
import workloads as w
from conftest import load_pkg
pkg = load_pkg()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cuts = [int(a) for a in sys.argv[2:]] or [512, 1024, 2048, 4096]
raws = [w.long_backref_stream(1000 + i) for i in range(min(n, 32))]
for cut in cuts:
    comp = [w.brotli_compress(r[: cut << 10]) for r in raws]
    sz = cut << 10
    si = max((len(c) + 255) // 256 * 256 for c in comp)
    inp = torch.zeros(n * si, dtype=torch.uint8, device="cuda"); out = torch.zeros(n * sz, dtype=torch.uint8, device="cuda")
    for i in range(n):
        c = comp[i % len(comp)]
        inp[i * si: i * si + len(c)] = torch.frombuffer(bytearray(c), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    b = pkg.Batch(n)
    b.decode_device([inp.data_ptr() + i * si for i in range(n)], [len(comp[i % len(comp)]) for i in range(n)], [out.data_ptr() + i * sz for i in range(n)], [sz] * n)
    res = b.wait()
    assert all(r.result == 1 and r.decoded_size == sz for r in res), [(r.result, r.error_code) for r in res[:4]]
    ms = []
    for _ in range(3):
        b.relaunch(); b.wait(); ms.append(b.last_kernel_ms())
    print("first %4d KiB of %d streams: kernel ms %.3f  compressed %.0f KB/stream  commands %.0f (engine %.0f) metablocks %.1f" % (
        cut, n, min(ms), sum(len(c) for c in comp) / len(comp) / 1e3, sum(r.num_commands for r in res) / n, sum(r.engine_commands for r in res) / n,
        sum(r.num_metablocks for r in res) / n), flush=True)
    b.close()
