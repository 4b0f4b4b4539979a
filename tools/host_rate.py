"""PCIe-inclusive rate of the host-buffer batch entry point (BrotliAmdBatchDecodeHost) on the default bench workload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from conftest import load_pkg
import workloads as w
pkg = load_pkg()
u = w.make_streams("long_backref", 32, 4 << 20, 1000)
datas = [c for c, _, _ in u] * 8
caps = [sz for _, sz, _ in u] * 8
b = pkg.Batch(len(datas))
for it in range(3):
    t = time.perf_counter()
    res, outs = b.decode_host(datas, caps, pkg.FLAG_LARGE_WINDOW)
    dt = time.perf_counter() - t
    assert all(r.result == 1 for r in res)
    print("decode_host: %.1f ms for %.0f MiB out / %.1f MiB in -> %.1f MB/s (kernel %.1f ms)" % (dt * 1e3, sum(caps) / 2**20, sum(map(len, datas)) / 2**20, sum(caps) / dt / 1e6, b.last_kernel_ms()))
b.close()
