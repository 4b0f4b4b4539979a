"""Differential fuzzing of the command engine (csrc/brotli_scan_engine.h) against the oracle: python tests/tools/fuzz_engine.py <first_seed> <n_seeds>
Batches of at most 200 streams (so that the host launches blocks of sixteen waves), streams whose literals do not depend
on context (qualities 0-4 never model context; Zipf-like data at any quality), whole / tight buffers / truncated / bit
flips / insertions / deletions."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
from conftest import load_pkg
import oracle_lib as oracle, libbrotli_ref as ref, workloads as w, param_corpus
pkg = load_pkg()
first, nseeds = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); total = bad = 0
for seed in range(first, first + nseeds):
    rnd = random.Random(seed)
    rng = np.random.Generator(np.random.PCG64(seed))
    def zipf(n, nsym=64, s=1.0, base=32):
        pr = np.arange(1, nsym + 1, dtype=np.float64) ** -s; pr /= pr.sum()
        return (rng.choice(nsym, size=n, p=pr) + base).astype(np.uint8).tobytes()
    raws = [w.long_backref_stream(seed * 7 + k, rnd.choice([64 << 10, 256 << 10, 1 << 20])) for k in range(3)]
    raws += [r for _, _, r in rnd.sample(param_corpus.corpus(), 6) if len(r) > 2000]
    # (round 4: a stretch of an executable -- dozens of block types at -q 5 and up, more tables than LDS holds: the LDS part as a cache
    # of the trees in use; few literals: regions bound by their closure, the scan engine's)
    exe = open(sys.executable, "rb").read()
    off = rnd.randrange(0, max(1, len(exe) - (3 << 20)))
    raws.append(exe[off:off + rnd.choice([200 << 10, 1 << 20, 3 << 20])])
    for _ in range(3):
        period = rnd.choice([1, 2, 3, 4, 7, 9, 31, 64, 77, 130, 500])
        body = bytearray(zipf(period) * (120000 // period + 1))[:120000]
        for _ in range(rnd.randrange(0, 60)): body[rnd.randrange(len(body))] = rnd.randrange(32, 96)
        raws.append(zipf(rnd.randrange(100, 5000)) + bytes(body))
    parts = [zipf(100000, rnd.choice([16, 64, 200]), rnd.choice([0.5, 1.0, 2.0]))]
    for _ in range(60):
        parts.append(zipf(rnd.choice([1, 5, 40, 64, 100, 700, 3000, 3000, 4500, 6100, 14000]), rnd.choice([64, 64, 200])))   # (round 4: runs that get regions of their own)
        src = b"".join(parts); n = rnd.choice([4, 9, 70, 600, 8200, 30000]); off = rnd.randrange(0, max(1, len(src) - n))
        parts.append(src[off:off + n])
    raws.append(b"".join(parts))
    base = []
    for raw in raws:
        c = ref.encode(raw, rnd.choice([0, 1, 2, 3, 4, 4, 5, 5, 6, 9]), rnd.choice([16, 18, 20, 22, 22, 24]))
        base.append((c, len(raw)))
    datas, caps = [], []
    while len(datas) < 200:
        c, n = rnd.choice(base)
        d = bytearray(c); k = rnd.random(); cap = n + 4096
        if k < 0.15: cap = rnd.choice([n, n - 1, n // 2, rnd.randrange(1, n + 1)])
        elif k < 0.35 and len(d) > 1: d = d[:rnd.randrange(1, len(d))]
        elif k < 0.8:
            for _ in range(rnd.choice([1, 1, 1, 2, 3])):
                pos = rnd.randrange(0, min(len(d), rnd.choice([64, 512, 4096, 1 << 22]))); d[pos] ^= 1 << rnd.randrange(8)
        elif k < 0.9:
            pos = rnd.randrange(0, len(d) + 1); d[pos:pos] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 4)))
        else:
            a = rnd.randrange(0, len(d)); b_ = min(len(d), a + rnd.randrange(1, 64)); del d[a:b_]
        if k >= 0.15 and rnd.random() < 0.3: cap = rnd.randrange(1, n + 4096)  # (damage and a buffer that may be too small: the reference's verdict, batch.h)
        datas.append(bytes(d)); caps.append(cap)
    # (FUZZ_BATCH: streams a launch -- up to 128 the host gives every stream a gang of blocks, csrc/brotli_path_engine.h PE_CFG_REMOTE; a seed's
    # 200 streams then go through several launches of one batch object)
    per = int(os.environ.get("FUZZ_BATCH", "200"))
    b = pkg.Batch(min(per, len(datas))); res, outs, gangs = [], [], set()
    for q in range(0, len(datas), per):
        r_, o_ = b.decode_host(datas[q:q + per], caps[q:q + per], 1); res += r_; outs += o_; gangs.add(b.last_gang())
    b.close()
    for i, (d, cap) in enumerate(zip(datas, caps)):
        info, exp = oracle.decode(d, cap, 1)
        r = res[i]
        if (r.result, r.error_code, r.decoded_size, outs[i]) != (info.result, info.error_code, info.decoded_size, exp) or \
           (info.result == 1 and (r.consumed != info.consumed or r.num_commands != info.num_commands)):
            bad += 1
            if bad <= 8:
                print("MISMATCH seed", seed, "i", i, (r.result, r.error_code, r.decoded_size, r.consumed, r.num_commands),
                      (info.result, info.error_code, info.decoded_size, info.consumed, info.num_commands), len(d), cap, flush=True)
                open(os.path.join(ROOT, "gpurun_out", "fuzz_engine_bad_%d_%d.br" % (seed, i)), "wb").write(d)
    total += len(datas)
    if seed % 10 == 0: print("seed", seed, "total", total, "bad", bad, "blocks a stream", sorted(gangs), "%.0fs" % (time.time() - t0), flush=True)
print("done", total, "streams", bad, "mismatches", "%.0fs" % (time.time() - t0))
