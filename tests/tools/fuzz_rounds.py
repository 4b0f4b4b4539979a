"""Differential test of the helper rounds on literal-heavy streams of random make-up: segments of literals whose
distribution changes inside a stream (the encoder answers with several literal block types and trees, so block
switches fall inside long runs), copies of random lengths in between, qualities 1-9, windows 16-24, tight and roomy
output buffers.  python tests/tools/fuzz_rounds.py [seed] [n_streams]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
from conftest import load_pkg
import oracle_lib as oracle, libbrotli_ref as ref
pkg = load_pkg()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 96
rng = np.random.Generator(np.random.PCG64(seed))
datas, caps = [], []
for k in range(n):
    parts = []
    for _ in range(int(rng.integers(2, 9))):
        nsym = int(rng.choice([2, 5, 17, 64, 200, 256]))
        skew = float(rng.choice([0.0, 0.3, 1.0, 2.5]))
        p = np.arange(1, nsym + 1, dtype=np.float64) ** -skew; p /= p.sum()
        perm = rng.permutation(256)[:nsym]
        run = int(rng.choice([300, 800, 3000, 20000, 70000, 150000]))
        parts.append(perm[rng.choice(nsym, size=run, p=p)].astype(np.uint8).tobytes())
        if parts and rng.random() < 0.7:
            src = parts[int(rng.integers(0, len(parts)))]
            m = int(rng.integers(4, 2000))
            parts.append(src[:m])
    raw = b"".join(parts)
    c = ref.encode(raw, int(rng.choice([1, 3, 5, 6, 9])), int(rng.choice([16, 18, 20, 22, 24])))
    for cap in (len(raw), int(rng.integers(1, len(raw))), len(raw) + 100):
        datas.append(c); caps.append(cap)
t0 = time.time()
b = pkg.Batch(len(datas)); res, outs = b.decode_host(datas, caps, 1); b.close()
bad = 0
for i, (d, cap) in enumerate(zip(datas, caps)):
    info, exp = oracle.decode(d, cap, 1)
    r = res[i]
    if (r.result, r.error_code, r.decoded_size, outs[i]) != (info.result, info.error_code, info.decoded_size, exp) or (info.result == 1 and r.consumed != info.consumed):
        bad += 1
        if bad <= 5: print("MISMATCH", i, (r.result, r.error_code, r.decoded_size, r.consumed), (info.result, info.error_code, info.decoded_size, info.consumed), len(d), cap)
print("rounds fuzz seed %d: %d streams, %d mismatches, %.0f s" % (seed, len(datas), bad, time.time() - t0))
