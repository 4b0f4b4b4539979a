"""Cycle split of the command loop (needs a library built with -DBROTLI_AMD_PROFILE, see BROTLI_AMD_LIB)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_pkg
pkg = load_pkg()
name = sys.argv[1] if len(sys.argv) > 1 else "alice29.txt.compressed"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
data = open(os.path.join(ROOT, "tests/golden/testdata", name), "rb").read()
m = {e["name"]: e for e in json.load(open(os.path.join(ROOT, "tests/golden/manifest.json")))}
osz = m[name]["size"]
src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
si, so = (len(data) + 255) // 256 * 256, (osz + 255) // 256 * 256
inp = torch.zeros(n * si, dtype=torch.uint8, device="cuda"); out = torch.zeros(n * so, dtype=torch.uint8, device="cuda")
for i in range(n): inp[i * si: i * si + len(data)] = src
torch.cuda.synchronize()
b = pkg.Batch(n)
b.decode_device([inp.data_ptr() + i * si for i in range(n)], [len(data)] * n, [out.data_ptr() + i * so for i in range(n)], [osz] * n)
res = b.wait(); b.relaunch(); res = b.wait()
ms = b.last_kernel_ms()
L = pkg.load_library()
L.brotli_amd_debug_status.restype = ctypes.c_void_p
L.brotli_amd_debug_status.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
class Resume(ctypes.Structure):
    _fields_ = [("bit_pos", ctypes.c_uint64), ("out_pos", ctypes.c_uint64), ("dist_rb", ctypes.c_int32 * 4), ("dist_rb_idx", ctypes.c_int32),
                ("window_bits", ctypes.c_uint32), ("large_window", ctypes.c_uint32), ("rb", ctypes.c_uint32), ("x", ctypes.c_uint32), ("y", ctypes.c_uint32)]
class Status(ctypes.Structure):
    _fields_ = [("result", ctypes.c_int32), ("error_code", ctypes.c_int32), ("decoded", ctypes.c_uint64), ("consumed", ctypes.c_uint64),
                ("produced", ctypes.c_uint64), ("nmb", ctypes.c_uint32), ("r", ctypes.c_uint32), ("ncmd", ctypes.c_uint64), ("resume", Resume)]
st = Status.from_address(L.brotli_amd_debug_status(b._h, 0))
tot = st.resume.bit_pos
cmd, lit, dist, cp = st.resume.out_pos, st.resume.dist_rb[0] << 8, st.resume.dist_rb[1] << 8, st.resume.dist_rb[2] << 8
print(name, "n", n, "kernel ms %.3f" % ms, "result", st.result, "cmds", st.ncmd)
print("cycles total %d (%.1f MHz eff)  cmd %d  lit %d  dist %d  copy %d  other %d" % (tot, tot / ms / 1e3, cmd, lit, dist, cp, tot - cmd - lit - dist - cp))
if st.ncmd: print("per command: total %.0f cmd %.0f lit %.0f dist %.0f copy %.0f" % (tot / st.ncmd, cmd / st.ncmd, lit / st.ncmd, dist / st.ncmd, cp / st.ncmd))
print("fast batches", st.resume.dist_rb[3], "fast syms", st.resume.dist_rb_idx)
