"""Cycle split of the command loop on a synthetic workload stream (library built with -DBROTLI_AMD_PROFILE)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
import workloads as w
from conftest import load_pkg
pkg = load_pkg()
kind = sys.argv[1] if len(sys.argv) > 1 else "long_backref"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
(c, sz, sha), = w.make_streams(kind, 1, (int(sys.argv[3]) if len(sys.argv) > 3 else 4) << 20, 1000)
src = torch.frombuffer(bytearray(c), dtype=torch.uint8).cuda()
si, so = (len(c) + 255) // 256 * 256, (sz + 255) // 256 * 256
inp = torch.zeros(n * si, dtype=torch.uint8, device="cuda"); out = torch.zeros(n * so, dtype=torch.uint8, device="cuda")
for i in range(n): inp[i * si: i * si + len(c)] = src
torch.cuda.synchronize()
b = pkg.Batch(n)
b.decode_device([inp.data_ptr() + i * si for i in range(n)], [len(c)] * n, [out.data_ptr() + i * so for i in range(n)], [sz] * n)
res = b.wait(); b.relaunch(); res = b.wait(); ms = b.last_kernel_ms()
L = pkg.load_library()
L.brotli_amd_debug_status.restype = ctypes.c_void_p; L.brotli_amd_debug_status.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
class Resume(ctypes.Structure):
    _fields_ = [("bit_pos", ctypes.c_uint64), ("out_pos", ctypes.c_uint64), ("dist_rb", ctypes.c_int32 * 4), ("idx", ctypes.c_int32), ("wb", ctypes.c_uint32), ("lw", ctypes.c_uint32), ("rb", ctypes.c_uint32), ("x", ctypes.c_uint32), ("y", ctypes.c_uint32)]
class Status(ctypes.Structure):
    _fields_ = [("result", ctypes.c_int32), ("error_code", ctypes.c_int32), ("decoded", ctypes.c_uint64), ("consumed", ctypes.c_uint64), ("produced", ctypes.c_uint64), ("nmb", ctypes.c_uint32), ("r", ctypes.c_uint32), ("ncmd", ctypes.c_uint64), ("resume", Resume)]
st = Status.from_address(L.brotli_amd_debug_status(b._h, 0))
tot, cmd, lit, dist, cp = st.resume.bit_pos, st.resume.out_pos, st.resume.dist_rb[0] << 8, st.resume.dist_rb[1] << 8, st.resume.dist_rb[2] << 8
print(kind, "n", n, "kernel ms %.3f" % ms, "result", st.result, "cmds", st.ncmd, "csize", len(c))
print("ticks total %d: cmd %.1f%%  lit %.1f%%  dist %.1f%%  copy %.1f%%  other %.1f%%" % (tot, 100 * cmd / tot, 100 * lit / tot, 100 * dist / tot, 100 * cp / tot, 100 * (tot - cmd - lit - dist - cp) / tot))
print("per command ticks: total %.0f cmd %.0f lit %.0f dist %.0f copy %.0f" % (tot / st.ncmd, cmd / st.ncmd, lit / st.ncmd, dist / st.ncmd, cp / st.ncmd))
print("fast batches", st.resume.dist_rb[3], "fast syms", st.resume.idx)
