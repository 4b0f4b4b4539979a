"""Throughput of the HIP decode path on text recompressed at every encoder quality (cliff hunting: more block types,
context maps and trees per metablock as the quality goes up).  python tools/quality_sweep.py [fixture] [n_streams]
The raw text is obtained by decoding the reference's fixture on the GPU and checked against the manifest's SHA-256."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
import workloads as w
from conftest import load_pkg
pkg = load_pkg()
name = sys.argv[1] if len(sys.argv) > 1 else "lcet10.txt.compressed"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
if os.path.exists(name):   # (any file of the box instead of a fixture: its first 8 MiB)
    raw = open(name, "rb").read()[: 8 << 20]; size = len(raw); sha = hashlib.sha256(raw).hexdigest(); name = os.path.basename(name)
else:
    (comp, size, sha), = w.fixture_streams(name)
    info, raw = pkg.brotli_decode(comp, size)
    assert info.result == 1 and hashlib.sha256(raw).hexdigest() == sha
for q in [int(x) for x in os.environ.get("SWEEP_Q", "0 1 2 3 4 5 6 7 8 9 10 11").split()]:
    c = w.brotli_compress(raw, q, 22)
    si, so = (len(c) + 255) // 256 * 256, (size + 255) // 256 * 256
    src = torch.frombuffer(bytearray(c), dtype=torch.uint8).cuda()
    inp = torch.zeros(n * si, dtype=torch.uint8, device="cuda"); out = torch.zeros(n * so, dtype=torch.uint8, device="cuda")
    inp.view(n, si)[:, :len(c)] = src
    torch.cuda.synchronize()
    b = pkg.Batch(n)
    b.decode_device([inp.data_ptr() + i * si for i in range(n)], [len(c)] * n, [out.data_ptr() + i * so for i in range(n)], [size] * n)
    res = b.wait(); second = b.last_second_pass_count()
    best = 1e9
    for _ in range(3):
        b.relaunch(); b.wait(); best = min(best, b.last_kernel_ms())
    ok = all(r.result == 1 and r.decoded_size == size for r in res) and hashlib.sha256(out[:size].cpu().numpy().tobytes()).hexdigest() == sha \
        and hashlib.sha256(out[(n - 1) * so:(n - 1) * so + size].cpu().numpy().tobytes()).hexdigest() == sha
    cmds = sum(r.num_commands for r in res)
    print("%s q%-2d csize %7d  first-pass kernel %7.2f ms  %7.1f MB/s  %5.2f G commands/s (%4.1f B a command, %3.0f %% by a command engine)  second pass %4d streams  spilled metablocks %d  %s" %
          (name, q, len(c), best, n * size / best / 1e3, cmds / best / 1e6, n * size / max(cmds, 1), 100.0 * sum(r.engine_commands for r in res) / max(cmds, 1), second,
           sum(r.spilled_metablocks for r in res), "bit-exact" if ok else "MISMATCH"), flush=True)
    b.close()
