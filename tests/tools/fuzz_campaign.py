"""Differential fuzzing GPU vs oracle beyond what the test suite runs: python tests/tools/fuzz_campaign.py <first_seed> <n_seeds> [per_seed]"""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from conftest import load_pkg
import oracle_lib as oracle, param_corpus
pkg = load_pkg()
G = os.path.join(ROOT, "tests", "golden")
base = [open(os.path.join(G, "testdata", e["name"]), "rb").read() for e in json.load(open(os.path.join(G, "manifest.json")))
        if e["csize"] < 200000 and e["name"] != "rnd_chunk.br"]
base += [c for _, c, r in param_corpus.corpus() if len(c) < 60000]
first, nseeds = int(sys.argv[1]), int(sys.argv[2])
per = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
t0 = time.time(); total = bad = 0
for seed in range(first, first + nseeds):
    rnd = random.Random(seed)
    datas = []
    for _ in range(per):
        d = bytearray(rnd.choice(base))
        k = rnd.random()
        if k < 0.2 and len(d) > 1: d = d[:rnd.randrange(0, len(d))]
        elif k < 0.8:
            for _ in range(rnd.choice([1, 1, 1, 2, 3, 6])):
                if d:
                    pos = rnd.randrange(0, min(len(d), rnd.choice([8, 64, 512, 4096, 1 << 20]))); d[pos] ^= 1 << rnd.randrange(8)
        elif k < 0.9:
            pos = rnd.randrange(0, len(d) + 1); d[pos:pos] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 4)))
        else:
            a = rnd.randrange(0, len(d)); b = min(len(d), a + rnd.randrange(1, 64)); del d[a:b]
        datas.append(bytes(d))
    flags = seed & 1
    # (a quarter of the streams with a buffer that may be too small: the reference's verdict depends on what the damaged
    # stream does up to its next ring flush point, batch.h)
    caps = [1 << 20 if rnd.random() < 0.75 else rnd.randrange(1, rnd.choice([300, 20000, 200000])) for _ in datas]
    b = pkg.Batch(len(datas)); res, outs = b.decode_host(datas, caps, flags); b.close()
    for i, d in enumerate(datas):
        info, exp = oracle.decode(d, caps[i], flags)
        r = res[i]
        if (r.result, r.error_code, r.decoded_size, outs[i]) != (info.result, info.error_code, info.decoded_size, exp) or \
           (info.result == 1 and r.consumed != info.consumed):
            bad += 1
            if bad <= 5: print("MISMATCH seed", seed, "i", i, (r.result, r.error_code, r.decoded_size, r.consumed), (info.result, info.error_code, info.decoded_size, info.consumed), d[:24].hex(), len(d))
    total += len(datas)
print("fuzz: %d streams, %d mismatches, %.0f s" % (total, bad, time.time() - t0))
