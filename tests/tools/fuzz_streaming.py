"""Differential fuzzing of the streaming call: a stream fed in random pieces must end like the same stream fed in one
piece (result and error code; all of the one-piece output is a prefix of the piece-wise output, and equal to it where
the stream is valid), and valid streams must decode to what the oracle says.
python tests/tools/fuzz_streaming.py <first_seed> <n_seeds> [streams_per_seed]"""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from conftest import load_pkg
import oracle_lib as oracle, param_corpus
pkg = load_pkg()
G = os.path.join(ROOT, "tests", "golden")
base = [open(os.path.join(G, "testdata", e["name"]), "rb").read() for e in json.load(open(os.path.join(G, "manifest.json")))
        if 64 < e["csize"] < 200000 and e["name"] != "rnd_chunk.br"]
base += [c for _, c, r in param_corpus.committed() if len(c) < 60000]
E = os.path.join(G, "emitter")
base += [open(os.path.join(E, e["file"]), "rb").read() for e in json.load(open(os.path.join(E, "manifest.json")))]


def run(data, pieces, out_chunk):
    st = pkg.DecoderState(large_window=True)
    out = bytearray(); pos = 0; k = 0; pending = b""; result = 2; calls = 0
    while True:
        if not pending and result == 2:
            if pos >= len(data):
                if calls and got:   # (the input has run out inside the stream: what earlier calls had no room for still goes out, decode.rs:2835-2846)
                    result, used, got = st.decompress_stream(b"", out_chunk); out += got; calls += 1
                    continue
                break
            n = pieces[k % len(pieces)]; k += 1
            pending = data[pos:pos + n]; pos += len(pending)
        result, used, got = st.decompress_stream(pending, out_chunk)
        pending = pending[used:]; out += got; calls += 1
        if result in (0, 1) or calls > 200000: break
    code = st.error_code(); st.close()
    return result, code, bytes(out)


first, nseeds = int(sys.argv[1]), int(sys.argv[2])
per = int(sys.argv[3]) if len(sys.argv) > 3 else 40
t0 = time.time(); total = bad = 0
for seed in range(first, first + nseeds):
    rnd = random.Random(seed)
    for i in range(per):
        d = bytearray(rnd.choice(base)); k = rnd.random()
        if k < 0.4:
            for _ in range(rnd.choice([1, 1, 2])):
                p = rnd.randrange(0, len(d)); d[p] ^= 1 << rnd.randrange(8)
        elif k < 0.5 and len(d) > 1: d = d[:rnd.randrange(1, len(d))]
        d = bytes(d)
        small = len(d) < 3000
        pieces = [rnd.choice([1, 2, 5, 17] if small else [61, 300, 1000, 4096, 20000]) for _ in range(rnd.randrange(1, 4))]
        want = run(d, [len(d)], 1 << 22)
        got = run(d, pieces, rnd.choice([1 << 22, 65536, 4096] if not small else [1 << 16, 7, 1]))
        same_end = got[:2] == want[:2] or (got[:2], want[:2]) == ((0, -9), (0, -10))  # (input that runs out inside a command that
        # overshoots MLEN is BLOCK_LENGTH_1 -- the forced flush of decode.rs:2835-2846 + 1709-1711 --, the whole stream at once BLOCK_LENGTH_2)
        ok = same_end and got[2][:len(want[2])] == want[2] and (want[0] != 1 or got[2] == want[2])
        if ok and want[0] == 1:
            info, exp = oracle.decode(d, 1 << 24, 1)
            ok = info.result == 1 and exp == got[2]
        total += 1
        if not ok:
            bad += 1
            if bad <= 5:
                print("MISMATCH seed", seed, "i", i, pieces, got[:2], want[:2], len(got[2]), len(want[2]), len(d), flush=True)
                open(os.path.join(ROOT, "gpurun_out", "fuzz_streaming_bad_%d_%d.br" % (seed, i)), "wb").write(d)
print("streaming fuzz: %d streams, %d mismatches, %.0f s" % (total, bad, time.time() - t0))
