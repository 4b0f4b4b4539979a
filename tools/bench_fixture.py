import sys, os, time, json
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import torch
from conftest import load_pkg
pkg = load_pkg()
name = sys.argv[1] if len(sys.argv) > 1 else 'alice29.txt.compressed'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
data = open('tests/golden/testdata/' + name, 'rb').read()
m = {e['name']: e for e in json.load(open('tests/golden/manifest.json'))}
osz = m[name]['size']
src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
stride_in = (len(data) + 255) // 256 * 256
stride_out = (osz + 255) // 256 * 256
inp = torch.zeros(n * stride_in, dtype=torch.uint8, device='cuda')
out = torch.zeros(n * stride_out, dtype=torch.uint8, device='cuda')
for i in range(n):
    inp[i * stride_in: i * stride_in + len(data)] = src
torch.cuda.synchronize()
b = pkg.Batch(n)
b.decode_device([inp.data_ptr() + i * stride_in for i in range(n)], [len(data)] * n, [out.data_ptr() + i * stride_out for i in range(n)], [osz] * n)
res = b.wait()
assert all(r.result == 1 and r.decoded_size == osz for r in res), [(r.result, r.error_code) for r in res[:4]]
print('first launch ms', b.last_kernel_ms())
for it in range(5):
    b.relaunch(); b.wait()
    ms = b.last_kernel_ms()
    print('iter', it, 'kernel ms %.3f' % ms, 'GB/s out %.2f' % (n * osz / ms / 1e6), 'per-stream MB/s %.1f' % (osz / ms / 1e3))
ref = out[:osz].cpu().numpy().tobytes()
import hashlib
print('sha ok', hashlib.sha256(ref).hexdigest() == m[name]['sha256'])
