"""Whole-job time of a heterogeneous batch (every decodable reference fixture, shuffled, many copies): what the order in
which blocks take streams is worth.  python tools/mixed_batch.py [copies]   (BROTLI_AMD_NO_ORDER=1: index order)"""
import hashlib, json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
from conftest import load_pkg
pkg = load_pkg()
G = os.path.join(ROOT, "tests", "golden")
m = [e for e in json.load(open(os.path.join(G, "manifest.json"))) if not e.get("must_fail") and e.get("size", 1 << 40) <= (1 << 20)]
if "--all" not in sys.argv:  # (65 537 empty metablocks each: 280 ms of metablock headers, they would be the whole job)
    m = [e for e in m if e["name"] not in ("empty.compressed.17", "empty.compressed.18")]
copies = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
items = [(open(os.path.join(G, "testdata", e["name"]), "rb").read(), e["size"], e["sha256"]) for e in m] * copies
random.Random(5).shuffle(items)
n = len(items)
in_off, out_off, a, o = [], [], 0, 0
for c, sz, _ in items:
    in_off.append(a); out_off.append(o); a += (len(c) + 255) // 256 * 256; o += (sz + 255) // 256 * 256 + 256
host_in = bytearray(a)
for (c, _, _), off in zip(items, in_off): host_in[off:off + len(c)] = c
d_in = torch.frombuffer(host_in, dtype=torch.uint8).cuda(); d_out = torch.zeros(o, dtype=torch.uint8, device="cuda")
b = pkg.Batch(n)
args = ([d_in.data_ptr() + x for x in in_off], [len(c) for c, _, _ in items], [d_out.data_ptr() + x for x in out_off], [sz for _, sz, _ in items])
best = 1e9
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b.decode_device(*args, pkg.FLAG_LARGE_WINDOW); t1 = time.perf_counter(); res = b.wait(); t2 = time.perf_counter()
    best = min(best, t2 - t0)
    split = (t1 - t0, t2 - t1, b.last_kernel_ms())
host = d_out.cpu().numpy()
bad = sum(1 for (c, sz, sha), r, off in zip(items, res, out_off) if r.result != 1 or hashlib.sha256(host[off:off + sz].tobytes()).hexdigest() != sha)
total = sum(sz for _, sz, _ in items)
print("submit %.2f ms, wait %.2f ms, first-pass kernel %.2f ms" % (split[0] * 1e3, split[1] * 1e3, split[2]))
print("%d streams of %d fixtures, %.1f MB out: %.2f ms whole job, %.1f MB/s, %d wrong, %d came back for a larger arena" %
      (n, len(m), total / 1e6, best * 1e3, total / best / 1e6, bad, b.last_second_pass_count()))
