"""GPU vs oracle on the bench workloads' 4 MiB streams: valid streams with tight output buffers, damaged streams with
roomy ones (long literal runs, copies > 1 KiB, quota and ring-buffer limits of the lean loop)."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from conftest import load_pkg
import oracle_lib as oracle, workloads as w
pkg = load_pkg()
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
streams = w.make_streams("long_backref", 4, 4 << 20, 5000) + w.make_streams("high_entropy", 2, 4 << 20, 6000) + \
          w.make_streams("long_backref", 4, 1 << 20, 7000) + w.make_streams("high_entropy", 2, 256 << 10, 8000)
datas, caps = [], []
for c, n, _ in streams:
    for cap in (n, n - 1, n // 2, n // 3 + 17, rnd.randrange(1, n), n + 1000):
        datas.append(c); caps.append(cap)
    for _ in range(24):
        d = bytearray(c)
        k = rnd.random()
        if k < 0.3: d = d[:rnd.randrange(1, len(d))]
        else:
            for _ in range(rnd.choice([1, 1, 2, 4])):
                pos = rnd.randrange(0, min(len(d), rnd.choice([64, 4096, 1 << 22]))); d[pos] ^= 1 << rnd.randrange(8)
        datas.append(bytes(d)); caps.append(8 << 20)
t0 = time.time()
b = pkg.Batch(len(datas)); res, outs = b.decode_host(datas, caps, 1); b.close()
bad = 0
for i, (d, cap) in enumerate(zip(datas, caps)):
    info, exp = oracle.decode(d, cap, 1)
    r = res[i]
    tight = cap < 8 << 20
    got = (r.result, r.error_code, r.decoded_size, outs[i]); want = (info.result, info.error_code, info.decoded_size, exp)
    if got != want:
        bad += 1
        if bad <= 8: print("MISMATCH", i, "cap", cap, "tight", tight, got[:3], want[:3], len(d))
print("big fuzz: %d streams, %d mismatches, %.0f s" % (len(datas), bad, time.time() - t0))
