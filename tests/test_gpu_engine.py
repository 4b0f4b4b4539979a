"""The command engine (csrc/brotli_scan_engine.h) against the CPU oracle, through the C ABI (needs a real MI355X).

The engine runs in blocks of sixteen waves -- batches of at most one stream per CU, which every batch here is -- on
metablocks whose literals do not depend on context.  It parses a command at every bit position, follows the real chain,
and hands every command that needs anything unusual to the checked command loop: these tests aim at the hand-overs
(overlapping copies, literal runs of 64 .. 8191 and beyond, copies of 8 KiB and more, block switches, ring flush points
every window, output limits, truncated and damaged input, prefix codes of maximal depth) and at sizes the fixtures
never reach (one stream of many metablocks; bit positions beyond 2^32)."""
import hashlib
import os
import random
import sys

import pytest

import oracle_lib as oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu


def _enc():
    import libbrotli_ref as ref
    if not ref.encoder_available():
        pytest.fail("libbrotlienc is not available: the GPU suite needs the encoder of the image for its synthetic streams (a skip here would let a third of the suite go green unrun)")
    return ref


def _check_against_oracle(pkg, datas, caps, flags=1, what=""):
    batch = pkg.Batch(len(datas))
    results, outs = batch.decode_host(datas, caps, flags)
    batch.close()
    bad = []
    for i, (d, cap) in enumerate(zip(datas, caps)):
        info, exp = oracle.decode(d, cap, flags)
        r = results[i]
        ok = (r.result, r.error_code, r.decoded_size, outs[i]) == (info.result, info.error_code, info.decoded_size, exp)
        if ok and info.result == 1:
            ok = r.consumed == info.consumed and r.num_commands == info.num_commands and r.num_metablocks == info.num_metablocks
        if not ok:
            bad.append((i, what, (r.result, r.error_code, r.decoded_size), (info.result, info.error_code, info.decoded_size), r.consumed, info.consumed, len(d), cap))
    assert not bad, (len(bad), bad[:10])


def _variants(rnd, c, n, damaged=6):
    """a valid stream with exact, short and roomy output buffers, truncated and bit-flipped copies of it"""
    datas, caps = [], []
    for cap in (n, n - 1, n // 2, rnd.randrange(1, n), n + 1000):
        datas.append(c); caps.append(cap)
    for _ in range(damaged):
        d = bytearray(c)
        if rnd.random() < 0.4:
            d = d[:rnd.randrange(1, len(d))]
        else:
            for _ in range(rnd.choice([1, 1, 2])):
                pos = rnd.randrange(0, len(d))
                d[pos] ^= 1 << rnd.randrange(8)
        datas.append(bytes(d)); caps.append(n + 4096)
    return datas, caps


def test_one_large_stream_of_many_metablocks(pkg):
    """BASELINE config 3 as written, at 64 MiB: ONE stream, window 22, many metablocks, long back-references; whole,
    with a buffer one byte short, truncated, and with a flipped bit deep inside"""
    sys.path.insert(0, ROOT)
    import workloads as w
    if not w.encoder_available():
        pytest.fail("libbrotlienc is not available: the GPU suite needs the encoder of the image for its synthetic streams (a skip here would let a third of the suite go green unrun)")
    raw = w.long_backref_stream(4321, 64 << 20)
    c = w.brotli_compress(raw, 5, 22)
    info, out = oracle.decode(c, len(raw), 1)
    assert info.result == 1 and out == raw and info.num_metablocks > 10
    d_cut = c[: len(c) * 3 // 5]
    d_flip = bytearray(c); d_flip[len(c) // 2] ^= 0x10
    _check_against_oracle(pkg, [c, c, d_cut, bytes(d_flip)], [len(raw), len(raw) - 1, len(raw), len(raw)], 1, "64 MiB stream")


def test_bit_positions_beyond_32_bits(pkg):
    """A stream whose compressed size exceeds 2^32 bits (544 MiB of high-entropy literals): reader positions, the engine's
    origin and the output offset all pass 4 Gi; checked by SHA-256 and by the oracle's status words"""
    import numpy as np
    sys.path.insert(0, ROOT)
    import workloads as w
    if not w.encoder_available():
        pytest.fail("libbrotlienc is not available: the GPU suite needs the encoder of the image for its synthetic streams (a skip here would let a third of the suite go green unrun)")
    rng = np.random.Generator(np.random.PCG64(77))
    p = np.arange(1, 257, dtype=np.float64) ** -0.6
    p /= p.sum()
    perm = rng.permutation(256).astype(np.uint8)
    parts = [perm[rng.choice(256, size=32 << 20, p=p)].tobytes() for _ in range(17)]
    raw = b"".join(parts)
    del parts
    c = w.brotli_compress(raw, 5, 22)
    assert len(c) * 8 > (1 << 32)
    info, out = oracle.decode(c, len(raw), 1)
    assert info.result == 1 and hashlib.sha256(out).digest() == hashlib.sha256(raw).digest()
    del out
    batch = pkg.Batch(1)
    results, outs = batch.decode_host([c], [len(raw)], 1)
    batch.close()
    r = results[0]
    assert (r.result, r.error_code, r.decoded_size, r.consumed) == (1, 1, len(raw), len(c))
    assert hashlib.sha256(outs[0]).digest() == hashlib.sha256(raw).digest()


@pytest.mark.parametrize("seed", [1, 2])
def test_streams_that_stress_the_engines_hand_overs(pkg, seed):
    import numpy as np
    ref = _enc()
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    rnd = random.Random(seed)

    def zipf(n, nsym=64, s=1.0, base=32):
        pr = np.arange(1, nsym + 1, dtype=np.float64) ** -s
        pr /= pr.sum()
        return (rng.choice(nsym, size=n, p=pr) + base).astype(np.uint8).tobytes()

    raws = []
    # periodic data: overlapping copies of every small distance, some interrupted by fresh symbols
    for period in (1, 2, 3, 5, 7, 13, 63, 64, 65, 100, 300, 1100):
        pat = zipf(period)
        body = bytearray(pat * (200000 // period + 1))[:200000]
        for _ in range(rnd.randrange(0, 40)):
            body[rnd.randrange(len(body))] = rnd.randrange(32, 96)
        raws.append(zipf(3000) + bytes(body) + zipf(500))
    # text-like seed with short matches, then literal runs of 64 .. 9000 between copies of every size
    for _ in range(3):
        parts = [zipf(300000)]
        for _ in range(120):
            parts.append(zipf(rnd.choice([64, 65, 100, 500, 767, 768, 2000, 5000, 9000]) + rnd.randrange(3)))
            src = b"".join(parts)
            n = rnd.choice([4, 9, 70, 600, 8191, 8192, 8193, 20000, 100000])
            off = rnd.randrange(0, max(1, len(src) - n))
            parts.append(src[off:off + n])
        raws.append(b"".join(parts))
    # a prefix code of maximal depth for the literals (lengths 6, 9, 12, 15) with short copies in between
    syms = rng.permutation(256)[:253].astype(np.uint8)
    block = np.concatenate([np.repeat(syms[:60], 512), np.repeat(syms[60:89], 64), np.repeat(syms[89:93], 8), syms[93:253]])
    deep = b"".join(rng.permutation(block).tobytes() for _ in range(6))
    parts = []
    for k in range(0, len(deep) - 40, 40):
        parts.append(deep[k:k + 40])
        if k > 4000:
            o = rnd.randrange(0, k - 100)
            parts.append(deep[o:o + rnd.randrange(4, 30)])
    raws.append(b"".join(parts))
    # the bench's own long-back-reference make-up, small
    sys.path.insert(0, ROOT)
    import workloads as w
    raws.append(w.long_backref_stream(9000 + seed, 2 << 20))

    datas, caps = [], []
    for raw in raws:
        q = rnd.choice([2, 4, 5, 5, 6, 9])
        lgwin = rnd.choice([16, 18, 20, 22, 22, 24])
        c = ref.encode(raw, q, lgwin)
        d, cp = _variants(rnd, c, len(raw), damaged=4)
        datas += d; caps += cp
    _check_against_oracle(pkg, datas, caps, 1, "hand-overs")


def test_regions_put_together_in_lds(pkg):
    """Round 3: a region whose output fits J1's room is put together in LDS and written out in one piece; copies that read the
    region's own output go through LDS there (csrc/brotli_path_engine.h, execute).  Synthetic LZ77 data whose copies are short
    and come from a few bytes to a few hundred bytes back -- overlapping themselves, reading each other's output, reaching
    back over the region's first byte -- next to stretches of long copies from far back, whose regions do not fit the stage:
    both kinds of region follow each other in one stream."""
    import numpy as np
    ref = _enc()
    rng = np.random.Generator(np.random.PCG64(4711))
    rnd = random.Random(4711)
    pr = np.arange(1, 65, dtype=np.float64) ** -1.0
    pr /= pr.sum()

    def zipf(n):
        return (rng.choice(64, size=n, p=pr) + 32).astype(np.uint8).tobytes()

    raws = []
    for near in (8, 40, 300, 5000):
        out = bytearray(zipf(2000))
        while len(out) < 700000:
            k = rnd.random()
            if k < 0.45:
                out += zipf(rnd.choice([1, 2, 3, 5, 9, 17, 40, 70, 130]))
            elif k < 0.95:  # a copy from close by (byte by byte: it may overlap itself)
                d = rnd.randrange(1, min(len(out), near) + 1)
                for _ in range(rnd.choice([3, 4, 5, 8, 15, 16, 17, 31, 33, 63, 64, 65, 200])):
                    out.append(out[-d])
            else:  # a stretch of long copies from far back
                for _ in range(rnd.randrange(1, 6)):
                    n = rnd.randrange(1000, 30000)
                    o = rnd.randrange(0, max(1, len(out) - n))
                    out += out[o:o + n]
        raws.append(bytes(out))
    datas, caps = [], []
    for raw in raws:
        for q, lgwin in ((5, 22), (rnd.choice([2, 4, 6, 9]), rnd.choice([18, 20, 24]))):
            c = ref.encode(raw, q, lgwin)
            d, cp = _variants(rnd, c, len(raw), damaged=3)
            datas += d; caps += cp
    _check_against_oracle(pkg, datas, caps, 1, "lds stage")


@pytest.mark.parametrize("alphabet", [2, 5, 64, 256])
def test_regions_of_long_literal_runs(pkg, alphabet):
    """Round 4: a command whose literal run is longer than a region's path gets regions of its own, decoded a stretch of the
    stream a lane out of registers (csrc/brotli_path_engine.h, run_region).  Runs of every length around the hand-over points
    (what a region's path holds, PE_RUN_MIN, several regions) with short copies between them, under literal codes from one bit
    a symbol (a lane's stretch is hundreds of literals: the region's literal room is what ends it) to fifteen (code words
    beyond the wide table's eleven bits); whole, short of output inside a run, truncated inside a run, damaged."""
    import numpy as np
    ref = _enc()
    rng = np.random.Generator(np.random.PCG64(8800 + alphabet))
    rnd = random.Random(8800 + alphabet)
    if alphabet == 256:
        # lengths 6, 9, 12 and 15: four groups of symbols whose frequencies are 512 : 64 : 8 : 1
        syms = rng.permutation(256)[:253].astype(np.uint8)
        block = np.concatenate([np.repeat(syms[:60], 512), np.repeat(syms[60:89], 64), np.repeat(syms[89:93], 8), syms[93:253]])

        def lits(n):
            return np.tile(rng.permutation(block), n // len(block) + 1)[:n].tobytes()
    else:
        pr = np.arange(1, alphabet + 1, dtype=np.float64) ** (-1.2 if alphabet > 5 else -0.3)
        pr /= pr.sum()
        base = rng.permutation(200)[:alphabet].astype(np.uint8) + 20

        def lits(n):
            return base[rng.choice(alphabet, size=n, p=pr)].tobytes()

    length_sets = ((1000, 1023, 1024, 1025, 2000, 3000, 4000, 4300, 4400, 5000, 5999, 6000, 6001, 6500, 6656, 6657, 7000),
                   (13000, 13312, 13313, 20000, 26000, 40000, 66000, 131072, 300000),
                   None)
    datas, caps = [], []
    if alphabet <= 5:
        # (an encoder finds copies everywhere in data of so few symbols: the commands are written by the repository's own
        # emitter, tools/brotli_emit.py, insert lengths as they are wanted)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import brotli_emit as be
        for lengths in length_sets:
            hist, cmds = bytearray(), []
            for k in range(14 if lengths and lengths[0] > 7000 else 30):
                n = rnd.choice(lengths) + rnd.randrange(-2, 3) if lengths else rnd.randrange(900, 30000)
                ins = lits(n)
                hist += ins
                m = rnd.choice([2, 4, 7, 20, 300, 5000])
                dist = rnd.randrange(1, min(len(hist), 60000) + 1)
                cmds.append((ins, m, dist))
                for _ in range(m):
                    hist.append(hist[-dist])
            tail = lits(rnd.choice([1, 700, 6100]))
            cmds.append((tail, 0, 0)); hist += tail
            w = be.BitWriter()
            be.write_stream_header(w, 22)
            made = be.emit_compressed(w, cmds, be.Plan(), True)
            c = w.finish()
            assert bytes(made) == bytes(hist)
            info, exp = oracle.decode(c, len(hist), 1)
            assert info.result == 1 and exp == bytes(hist)
            d, cp = _variants(rnd, c, len(hist), damaged=4)
            datas += d; caps += cp
    else:
        for lengths in length_sets:
            out = bytearray(lits(40000))
            for k in range(40):
                n = rnd.choice(lengths) + rnd.randrange(-2, 3) if lengths else rnd.randrange(900, 30000)
                out += lits(n)
                for _ in range(rnd.randrange(1, 4)):   # copies the encoder will find: far back and close by
                    m = rnd.choice([4, 7, 20, 300, 5000])
                    o = rnd.randrange(0, len(out) - m)
                    out += out[o:o + m]
            raw = bytes(out)
            for q, lgwin in ((5, 22), (rnd.choice([2, 4, 6, 9]), rnd.choice([18, 20, 24]))):
                c = ref.encode(raw, q, lgwin)
                d, cp = _variants(rnd, c, len(raw), damaged=4)
                datas += d; caps += cp
    _check_against_oracle(pkg, datas, caps, 1, "literal runs, alphabet %d" % alphabet)


def test_words_of_the_static_dictionary_in_the_engine(pkg):
    """Round 4: text at the qualities servers use (-q 4 .. 9: literals without context, a word of the static dictionary every
    33 to 87 commands; tools/eligibility_survey.py) had the engine stop in front of every such word.  Now a pass of its resolve
    ends behind the command's literals, wave 0 puts the word behind them (decode.rs:2593-2640) and the commands behind it get
    the next pass (csrc/brotli_path_engine.h, PE_DICT).  The reference's four texts recompressed at several qualities and
    windows (16 and 18: the window is full and the word's number no longer depends on the position; 22: it does), whole /
    short of output / truncated / damaged against the oracle; the whole streams must go through the engine (>= 90 % of their
    commands), or the test says nothing about the passes."""
    ref = _enc()
    rnd = random.Random(2593)
    gold = os.path.join(ROOT, "tests", "golden", "testdata")
    texts = [oracle.decode(open(os.path.join(gold, n + ".compressed"), "rb").read(), 1 << 20, 1)[1] for n in ("alice29.txt", "asyoulik.txt", "lcet10.txt", "plrabn12.txt")]
    datas, caps, whole = [], [], []
    for t in texts:
        for q, lgwin in ((5, 22), (rnd.choice([4, 6, 9]), rnd.choice([16, 18])), (rnd.choice([7, 8, 9]), 24)):
            c = ref.encode(t, q, lgwin)
            whole.append(len(datas))
            d, cp = _variants(rnd, c, len(t), damaged=8)
            datas += d; caps += cp
    batch = pkg.Batch(len(datas))
    results, outs = batch.decode_host(datas, caps, 1)
    batch.close()
    for i in whole:
        r = results[i]
        assert r.result == 1 and r.engine_commands >= 0.9 * r.num_commands, (i, r.result, r.engine_commands, r.num_commands)
    _check_against_oracle(pkg, datas, caps, 1, "dictionary words")


_CACHE_SCRIPT = r"""
import importlib.util, json, os, sys, hashlib
ROOT = sys.argv[1]
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libbrotli_ref as ref
spec = importlib.util.spec_from_file_location("rust_brotli_decompressor_amd", os.path.join(ROOT, "rust-brotli-decompressor_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec); sys.modules["rust_brotli_decompressor_amd"] = pkg; spec.loader.exec_module(pkg)
raw = open(sys.executable, "rb").read()[: 3 << 20]
datas = [ref.encode(raw, 5, 22), ref.encode(raw, 9, 22)]
datas.append(datas[0][: len(datas[0]) * 2 // 3])
bad = bytearray(datas[1]); bad[len(bad) // 2] ^= 16; datas.append(bytes(bad))
caps = [len(raw), len(raw), len(raw), len(raw)]
n = int(sys.argv[2])
datas = datas * n; caps = caps * n
b = pkg.Batch(len(datas))
res, outs = b.decode_host(datas, caps, 1)
b.close()
print(json.dumps([[r.result, r.error_code, r.decoded_size, r.consumed, r.num_commands, r.engine_commands, r.spilled_metablocks, hashlib.sha256(o).hexdigest()] for r, o in zip(res, outs)]))
"""


@pytest.mark.parametrize("copies", [1, 300])
def test_more_tables_than_lds_holds(pkg, copies):
    """Round 4: a metablock whose prefix-code tables do not fit the LDS part of the table arena (an executable at -q 5 / 9: dozens
    of block types, a tree each) but whose literals do not depend on context runs with the LDS part as a cache of the trees in
    use (run_commands / cached_tree in csrc/brotli_kernels.hip), in blocks of sixteen waves (4 streams) and in small blocks (1200).
    Against the oracle, and against the same batch with the tables read where they lie (BROTLI_AMD_ENGINE=nocache, a fresh
    process each); without the cache every such metablock counts as spilled, with it none of the whole streams' do."""
    import json
    import subprocess
    _enc()
    rows = {}
    for name, env in (("cache", {}), ("nocache", {"BROTLI_AMD_ENGINE": "nocache"})):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, "-c", _CACHE_SCRIPT, ROOT, str(copies)], env=e, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        rows[name] = json.loads(out.stdout.strip().splitlines()[-1])
    strip = lambda rs: [r[:5] + r[7:] for r in rs]   # everything but engine_commands and spilled_metablocks
    assert strip(rows["cache"]) == strip(rows["nocache"])
    ref = _enc()
    raw = open(sys.executable, "rb").read()[: 3 << 20]
    for i, q in ((0, 5), (1, 9)):
        info, exp = oracle.decode(ref.encode(raw, q, 22), len(raw), 1)
        assert exp == raw
        for r in rows["cache"][i::4]:
            assert r[:5] == [info.result, info.error_code, info.decoded_size, info.consumed, info.num_commands] and r[7] == hashlib.sha256(raw).hexdigest(), r
    assert any(r[6] != 0 for r in rows["nocache"][:2]), rows["nocache"][:2]   # (else the test streams do not test what they are for)


def test_many_block_types(pkg):
    """literal, command and distance statistics that change every few KiB: the encoder answers with many block types and
    short blocks (block switches every few dozen commands: the engine's part ends at each of them)"""
    import numpy as np
    ref = _enc()
    rng = np.random.Generator(np.random.PCG64(31337))
    rnd = random.Random(5)
    datas, caps = [], []
    for _ in range(6):
        parts = []
        for seg in range(60):
            nsym = int(rng.choice([4, 16, 64]))
            base = int(rng.choice([0, 48, 97, 160]))
            pr = np.arange(1, nsym + 1, dtype=np.float64) ** -float(rng.choice([0.5, 1.0, 2.0]))
            pr /= pr.sum()
            seg_raw = (rng.choice(nsym, size=int(rng.integers(2000, 20000)), p=pr) + base).astype(np.uint8).tobytes()
            if seg % 3 == 2 and parts:  # a stretch made of copies only
                src = b"".join(parts)
                seg_raw = b"".join(src[o:o + n] for o, n in ((rnd.randrange(0, len(src) - 300), rnd.randrange(5, 300)) for _ in range(80)))
            parts.append(seg_raw)
        raw = b"".join(parts)
        c = ref.encode(raw, rnd.choice([4, 5, 6]), 22)
        info, out = oracle.decode(c, len(raw), 1)
        assert info.result == 1 and out == raw
        d, cp = _variants(rnd, c, len(raw), damaged=3)
        datas += d; caps += cp
    _check_against_oracle(pkg, datas, caps, 1, "block types")


def test_emitter_vectors(pkg):
    """the repository's own emitter's streams (tests/golden/emitter/): 40 / 60 / 256 literal block types, every literal
    context mode with chosen context maps, block-type codes 0 and 1, NPOSTFIX / NDIRECT != 0, a stream of compressed,
    metadata, stored and empty metablocks; whole, with exact, short and half buffers, truncated and damaged"""
    import json
    d = os.path.join(ROOT, "tests", "golden", "emitter")
    rnd = random.Random(11)
    datas, caps = [], []
    for e in json.load(open(os.path.join(d, "manifest.json"))):
        comp = open(os.path.join(d, e["file"]), "rb").read()
        dd, cc = _variants(rnd, comp, e["size"], damaged=8)
        datas += dd; caps += cc
    _check_against_oracle(pkg, datas, caps, 0, "emitter vectors")
    _check_against_oracle(pkg, datas, caps, 1, "emitter vectors, large windows allowed")


def test_engine_blocks_serving_more_streams_than_cus(pkg):
    """A batch of more large streams than the device has CUs (up to three per CU) gets engine blocks that take the streams
    one after the other; streams whose metablocks the engine cannot take (context-modelled text) come back and continue in
    a launch of small blocks.  Mixed batch: the bench's long-back-reference make-up, the reference's text fixtures, both
    whole, with short buffers, truncated and damaged."""
    import json
    import torch
    sys.path.insert(0, ROOT)
    import workloads as w
    if not w.encoder_available():
        pytest.fail("libbrotlienc is not available: the GPU suite needs the encoder of the image for its synthetic streams (a skip here would let a third of the suite go green unrun)")
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rnd = random.Random(2024)
    m = {e["name"]: e for e in json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))}
    pool = []
    for k in range(6):
        raw = w.long_backref_stream(7000 + k, 768 << 10)
        pool.append((w.brotli_compress(raw, 5, 22), len(raw)))
    for name in ("alice29.txt.compressed", "asyoulik.txt.compressed", "lcet10.txt.compressed", "plrabn12.txt.compressed"):
        pool.append((open(os.path.join(ROOT, "tests", "golden", "testdata", name), "rb").read(), m[name]["size"]))
    assert sum(len(c) for c, _ in pool) // len(pool) >= 32768
    n = cus + cus // 4
    datas, caps = [], []
    for i in range(n):
        c, size = pool[i % len(pool)]
        kind = rnd.random()
        if kind < 0.7:
            datas.append(c); caps.append(size)
        elif kind < 0.8:
            datas.append(c); caps.append(rnd.randrange(1, size))
        elif kind < 0.9:
            datas.append(c[:rnd.randrange(1, len(c))]); caps.append(size)
        else:
            d = bytearray(c); d[rnd.randrange(len(d))] ^= 1 << rnd.randrange(8)
            datas.append(bytes(d)); caps.append(size)
    batch = pkg.Batch(n)
    results, outs = batch.decode_host(datas, caps, 1)
    came_back = batch.last_second_pass_count()
    batch.close()
    if not os.environ.get("BROTLI_AMD_NO_SCAN") and not os.environ.get("BROTLI_AMD_NO_ENGINE_QUEUE"):
        assert came_back >= n // 4, came_back  # (the text streams; fewer would mean the batch did not get engine blocks)
    memo = {}
    bad = []
    for i, (d, cap) in enumerate(zip(datas, caps)):
        key = (d, cap)
        if key not in memo:
            memo[key] = oracle.decode(d, cap, 1)
        info, exp = memo[key]
        r = results[i]
        ok = (r.result, r.error_code, r.decoded_size, outs[i]) == (info.result, info.error_code, info.decoded_size, exp)
        if ok and info.result == 1:
            ok = r.consumed == info.consumed and r.num_commands == info.num_commands and r.num_metablocks == info.num_metablocks
        if not ok:
            bad.append((i, (r.result, r.error_code, r.decoded_size), (info.result, info.error_code, info.decoded_size), len(d), cap))
    assert not bad, (len(bad), bad[:10])


def _metric_streams(n, size=1 << 20):
    sys.path.insert(0, ROOT)
    import workloads as w
    if not w.encoder_available():
        pytest.fail("libbrotlienc is not available: the GPU suite needs the encoder of the image for its synthetic streams (a skip here would let a third of the suite go green unrun)")
    return w.make_streams("long_backref", n, size, 1000)


def test_the_engine_takes_the_commands_of_the_streams_it_is_built_for(pkg):
    """Which path ran?  `engine_commands` of the status word: in a batch of at most one stream per CU, streams of the metric's
    make-up (literals that do not depend on context) must go through a command engine for at least 90 % of their commands.
    A launch that fell back to blocks without the engine (or an engine that hands everything back) fails here, loudly."""
    streams = _metric_streams(6)
    datas = [s[0] for s in streams]; caps = [s[1] for s in streams]
    batch = pkg.Batch(len(datas))
    results, outs = batch.decode_host(datas, caps, 1)
    batch.close()
    for (c, n, sha), r, out in zip(streams, results, outs):
        info, exp = oracle.decode(c, n, 1)
        assert (r.result, r.error_code, r.decoded_size, r.num_commands) == (1, 1, n, info.num_commands) and out == exp
        assert r.engine_commands >= 0.9 * r.num_commands, (r.engine_commands, r.num_commands)


_AB_SCRIPT = r"""
import importlib.util, json, os, sys
ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
import workloads as w
spec = importlib.util.spec_from_file_location("rust_brotli_decompressor_amd", os.path.join(ROOT, "rust-brotli-decompressor_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec); sys.modules["rust_brotli_decompressor_amd"] = pkg; spec.loader.exec_module(pkg)
import hashlib
streams = w.make_streams("long_backref", 4, 1 << 20, 1000)
datas = [s[0] for s in streams]
datas.append(datas[0][: len(datas[0]) // 2])                      # truncated
bad = bytearray(datas[1]); bad[len(bad) // 3] ^= 4; datas.append(bytes(bad))  # damaged
caps = [s[1] for s in streams] + [streams[0][1], streams[1][1]]
caps[2] -= 1                                                      # one byte short
b = pkg.Batch(len(datas))
res, outs = b.decode_host(datas, caps, 1)
b.close()
print(json.dumps([[r.result, r.error_code, r.decoded_size, r.consumed, r.num_commands, r.engine_commands, hashlib.sha256(o).hexdigest()] for r, o in zip(res, outs)]))
"""


def test_the_engines_and_the_one_wave_path_agree(pkg):
    """The same batch (whole, truncated, damaged, one byte short) four times in fresh processes: default (path engine), two
    engines of eight waves a block taking regions in turn (BROTLI_AMD_ENGINE=path2, round 4: kept as an opt-in), the scan
    engine only (BROTLI_AMD_ENGINE=scan), no engine blocks but the command records and their hand-written run wherever a block has helper waves
    (BROTLI_AMD_NO_SCAN=1; round 6: metablocks without context too), nothing but the one-wave loops (and BROTLI_AMD_ENGINE=norecall) -- same status words and bytes;
    extra to, not instead of, the comparison with the oracle above."""
    import json
    import subprocess
    _metric_streams(1)  # (skips without an encoder)
    rows = {}
    for name, env in (("path", {}), ("path2", {"BROTLI_AMD_ENGINE": "path2"}), ("scan", {"BROTLI_AMD_ENGINE": "scan"}), ("records", {"BROTLI_AMD_NO_SCAN": "1"}),
                      ("none", {"BROTLI_AMD_NO_SCAN": "1", "BROTLI_AMD_ENGINE": "norecall"})):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, "-c", _AB_SCRIPT, ROOT], env=e, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        rows[name] = json.loads(out.stdout.strip().splitlines()[-1])
    strip = lambda rs: [r[:5] + r[6:] for r in rs]  # everything but engine_commands
    assert strip(rows["path"]) == strip(rows["path2"]) == strip(rows["scan"]) == strip(rows["records"]) == strip(rows["none"])
    assert all(r[5] == 0 for r in rows["none"]), rows["none"]
    for name in ("path", "path2", "scan"):
        assert all(r[5] >= (0.8 if name == "path2" else 0.9) * r[4] for r in rows[name][:2]), (name, rows[name])   # (two engines: a region whose closure is full leaves its metablock to the one-wave loop)


def test_large_window_streams(pkg):
    """Streams with the large-window extension (decode.rs:152-187: distance codes of up to 62 extra bits).  Round 4: the path
    engine takes them -- a distance code of more than 24 extra bits is the checked loop's, like every command the engine's fields
    do not hold --, so a large-window stream of the metric's make-up must show engine_commands >= 90 %;
    a text stream comes out right as well.  Windows
    of 2^26 and 2^30, whole / short of output / truncated / damaged against the oracle."""
    ref = _enc()
    sys.path.insert(0, ROOT)
    import workloads as w
    rnd = random.Random(2630)
    raw = w.long_backref_stream(77, 1 << 20)
    text = oracle.decode(open(os.path.join(ROOT, "tests", "golden", "testdata", "alice29.txt.compressed"), "rb").read(), 1 << 20, 1)[1]
    datas, raws = [], []
    for r in (raw, text):
        datas.append(ref.encode_stream([r], {ref.PARAM_QUALITY: 5, ref.PARAM_LARGE_WINDOW: 1, ref.PARAM_LGWIN: 26})); raws.append(r)
    batch = pkg.Batch(len(datas))
    results, outs = batch.decode_host(datas, [len(r) for r in raws], 1)
    batch.close()
    for d, r, res, out in zip(datas, raws, results, outs):
        info, exp = oracle.decode(d, len(r), 1)
        assert exp == r and (res.result, res.error_code, res.decoded_size, res.num_commands) == (1, 1, len(r), info.num_commands) and out == r
    assert results[0].engine_commands >= 0.9 * results[0].num_commands, (results[0].engine_commands, results[0].num_commands)
    # (the text at -q 5: its literals do not depend on context with this encoder, so the engine takes what it can of it too)
    # (without the flag the same streams are refused, as the reference's plain instances refuse them: ffi/mod.rs:127)
    batch = pkg.Batch(len(datas))
    results, _ = batch.decode_host(datas, [len(r) for r in raws], 0)
    batch.close()
    assert all((res.result, res.error_code) == (0, -13) for res in results)
    # whole, short, truncated and damaged copies, two window sizes, two make-ups the engine takes
    vd, vc = [], []
    for r in (raw, w.long_backref_stream(78, 3 << 19)):
        for lgwin, q in ((26, 5), (30, 4)):
            c = ref.encode_stream([r], {ref.PARAM_QUALITY: q, ref.PARAM_LARGE_WINDOW: 1, ref.PARAM_LGWIN: lgwin})
            d, cp = _variants(rnd, c, len(r), damaged=8)
            vd += d; vc += cp
    _check_against_oracle(pkg, vd, vc, 1, "large window")


_CTX_SCRIPT = r"""
import importlib.util, json, os, sys, hashlib
ROOT = sys.argv[1]
spec = importlib.util.spec_from_file_location("rust_brotli_decompressor_amd", os.path.join(ROOT, "rust-brotli-decompressor_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec); sys.modules["rust_brotli_decompressor_amd"] = pkg; spec.loader.exec_module(pkg)
gold = os.path.join(ROOT, "tests", "golden", "testdata")
names = ["alice29.txt.compressed", "asyoulik.txt.compressed", "lcet10.txt.compressed", "plrabn12.txt.compressed"]
datas = [open(os.path.join(gold, n), "rb").read() for n in names]
caps = [1 << 20] * 4
datas.append(datas[0][: len(datas[0]) // 2]); caps.append(1 << 20)                      # truncated
bad = bytearray(datas[1]); bad[len(bad) // 3] ^= 4; datas.append(bytes(bad)); caps.append(1 << 20)  # damaged
datas.append(datas[2]); caps.append(100000)                                                # output buffer too small
datas = datas * int(sys.argv[2]); caps = caps * int(sys.argv[2])
b = pkg.Batch(len(datas))
res, outs = b.decode_host(datas, caps, 1)
b.close()
print(json.dumps([[r.result, r.error_code, r.decoded_size, r.consumed, r.num_commands, r.engine_commands, hashlib.sha256(o).hexdigest()] for r, o in zip(res, outs)]))
"""


@pytest.mark.parametrize("copies", [1, 160])
def test_context_modelled_streams_with_and_without_the_helper_waves(pkg, copies):
    """Metablocks whose literals depend on context (the reference's text fixtures: whole, truncated, damaged, output buffer too
    small), in blocks of sixteen waves (7 streams) and of four (1120 streams), three times in fresh processes: default (wave 2
    parses command records ahead of the decoding wave: rec_wave / lean_rec_commands), BROTLI_AMD_ENGINE=split (wave 1 also
    executes what the decoding wave parses: copier_wave / lean_split_commands), BROTLI_AMD_ENGINE=norec (the decoding wave
    alone).  Same status words and bytes, equal to the oracle's; `engine_commands` says which path ran."""
    import json
    import subprocess
    rows = {}
    for name, env in (("records", {}), ("split", {"BROTLI_AMD_ENGINE": "split"}), ("alone", {"BROTLI_AMD_ENGINE": "norec"})):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, "-c", _CTX_SCRIPT, ROOT, str(copies)], env=e, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        rows[name] = json.loads(out.stdout.strip().splitlines()[-1])
    strip = lambda rs: [r[:5] + r[6:] for r in rs]  # everything but engine_commands
    assert strip(rows["records"]) == strip(rows["split"]) == strip(rows["alone"])
    gold = os.path.join(ROOT, "tests", "golden", "testdata")
    names = ["alice29.txt.compressed", "asyoulik.txt.compressed", "lcet10.txt.compressed", "plrabn12.txt.compressed"]
    for i, n in enumerate(names):
        info, exp = oracle.decode(open(os.path.join(gold, n), "rb").read(), 1 << 20, 1)
        for r in rows["records"][i::7]:
            assert r[:5] == [info.result, info.error_code, info.decoded_size, info.consumed, info.num_commands] and r[6] == hashlib.sha256(exp).hexdigest(), (n, r)
    assert all(r[5] == 0 for r in rows["alone"]), rows["alone"][:7]
    for mode in ("records", "split"):
        assert all(r[5] >= 0.9 * r[4] for i, r in enumerate(rows[mode]) if i % 7 < 4), (mode, rows[mode][:7])


def test_text_at_every_quality_in_blocks_of_four_waves(pkg):
    """The reference's text fixtures encoded again at -q 1, 5, 9 and 11 (window 22; at 11 a metablock's tables leave LDS no room for a
    whole ring of command records: wave 2 writes half a ring, XW_RING_MASK), a valid copy, a truncated and a damaged one each, more streams
    than four a CU: blocks of four waves whose wave 2 parses records (rec_wave / lean_rec_commands).  Status words and bytes the oracle's."""
    ref = _enc()
    gold = os.path.join(ROOT, "tests", "golden", "testdata")
    rnd = random.Random(77)
    uniq, caps = [], []
    for n in ("lcet10.txt.compressed", "plrabn12.txt.compressed"):
        _, raw = oracle.decode(open(os.path.join(gold, n), "rb").read(), 1 << 20, 1)
        for q in (1, 5, 9, 11):
            c = ref.encode(raw, q, 22)
            bad = bytearray(c); bad[rnd.randrange(len(c) // 4, len(c))] ^= 1 << rnd.randrange(8)
            uniq += [c, c[: rnd.randrange(len(c) // 2, len(c))], bytes(bad)]; caps += [len(raw)] * 3
    want = [oracle.decode(d, cap, 1) for d, cap in zip(uniq, caps)]
    copies = 48                                                 # 24 x 48 = 1152 streams > 4 x 256 CUs
    b = pkg.Batch(len(uniq) * copies)
    res, outs = b.decode_host(uniq * copies, caps * copies, 1)
    b.close()
    wrong = []
    for i, (r, o) in enumerate(zip(res, outs)):
        info, exp = want[i % len(uniq)]
        if (r.result, r.error_code, r.decoded_size, o) != (info.result, info.error_code, info.decoded_size, exp) or \
                (info.result == 1 and (r.consumed, r.num_commands, r.num_metablocks) != (info.consumed, info.num_commands, info.num_metablocks)):
            wrong.append((i, r.result, r.error_code, r.decoded_size, info.result, info.error_code, info.decoded_size))
    assert not wrong, (len(wrong), wrong[:8])
    valid = [r for i, r in enumerate(res) if i % 3 == 0]
    assert sum(r.engine_commands for r in valid) >= 0.9 * sum(r.num_commands for r in valid)


def test_batches_and_one_shot_calls_from_several_threads(pkg):
    """Contexts of different block shapes launched from several threads at once (engine blocks of sixteen waves, small
    batches of one-wave blocks, one-shot calls): setting the kernel's LDS attribute and launching is one critical section
    (a launch between another thread's set and launch got the wrong attribute)."""
    import threading
    streams = _metric_streams(3, 256 << 10)
    gold = os.path.join(ROOT, "tests", "golden", "testdata")
    alice = open(os.path.join(gold, "alice29.txt.compressed"), "rb").read()
    a_info, a_exp = oracle.decode(alice, 200000, 1)
    errors = []

    def big():
        try:
            for _ in range(6):
                b = pkg.Batch(3)
                res, outs = b.decode_host([s[0] for s in streams], [s[1] for s in streams], 1)
                b.close()
                for (c, n, sha), r, o in zip(streams, res, outs):
                    if (r.result, r.decoded_size) != (1, n) or hashlib.sha256(o).hexdigest() != sha or r.engine_commands < 0.9 * r.num_commands:
                        errors.append(("big", r.result, r.error_code, r.decoded_size, r.engine_commands, r.num_commands))
        except Exception as ex:  # noqa: BLE001
            errors.append(("big", repr(ex)))

    def many():
        try:
            for _ in range(6):
                b = pkg.Batch(2048)
                res, outs = b.decode_host([alice] * 2048, [200000] * 2048, 1)
                b.close()
                if any((r.result, r.decoded_size) != (1, a_info.decoded_size) for r in res) or outs[7] != a_exp or outs[2047] != a_exp:
                    errors.append(("many", [(r.result, r.error_code) for r in res if r.result != 1][:3]))
        except Exception as ex:  # noqa: BLE001
            errors.append(("many", repr(ex)))

    def oneshot():
        try:
            for _ in range(40):
                info, out = pkg.brotli_decode(alice, 200000)
                if info.result != 1 or out != a_exp:
                    errors.append(("oneshot", info.result, info.code))
        except Exception as ex:  # noqa: BLE001
            errors.append(("oneshot", repr(ex)))

    ts = [threading.Thread(target=f) for f in (big, many, oneshot, big, oneshot)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:5]


def test_the_device_says_which_streams_get_engine_blocks(pkg):
    """More streams than CUs, few enough for blocks of sixteen waves to pay (round 5: a probe launch reads every stream's first
    metablock header and the host routes by what it says -- engine-shaped streams to engine blocks, context-modelled ones to the
    launch of small blocks behind it; round 4 guessed from the batch's size).  Three batches of 600 streams through the device-pointer
    entry point: all of the metric's make-up (every stream through a command engine), all context-modelled text (none), and both
    mixed -- every stream's status words against the oracle's and every output by SHA-256."""
    import torch
    gold = os.path.join(ROOT, "tests", "golden", "testdata")
    alice = open(os.path.join(gold, "alice29.txt.compressed"), "rb").read()
    a_info, a_exp = oracle.decode(alice, 200000, 1)
    a_item = (alice, len(a_exp), hashlib.sha256(a_exp).hexdigest())
    metric = _metric_streams(8, 1 << 20)

    def run(items):
        n = len(items)
        in_off, out_off, a, o = [], [], 0, 0
        for c, sz, _ in items:
            in_off.append(a); out_off.append(o); a += (len(c) + 255) // 256 * 256; o += (sz + 255) // 256 * 256 + 256
        host_in = bytearray(a)
        for (c, _, _), off in zip(items, in_off):
            host_in[off:off + len(c)] = c
        d_in = torch.frombuffer(host_in, dtype=torch.uint8).cuda(); d_out = torch.zeros(o, dtype=torch.uint8, device="cuda")
        b = pkg.Batch(n)
        b.decode_device([d_in.data_ptr() + x for x in in_off], [len(c) for c, _, _ in items], [d_out.data_ptr() + x for x in out_off],
                        [sz for _, sz, _ in items], pkg.FLAG_LARGE_WINDOW)
        res = b.wait()
        second = b.last_second_pass_count()
        b.close()
        host = d_out.cpu().numpy()
        bad = [i for i, ((c, sz, sha), r, off) in enumerate(zip(items, res, out_off))
               if r.result != 1 or r.decoded_size != sz or hashlib.sha256(host[off:off + sz].tobytes()).hexdigest() != sha]
        assert not bad, (len(bad), bad[:5])
        return res, second

    # real text at -q 5 -- words of the static dictionary: a block's engine goes from its lean form to the general one and, with the
    # block's next stream, back (round 5: the waves that stay inside the engine between invocations stayed in the wrong one)
    ref = _enc()
    lcet = open(os.path.join(gold, "lcet10.txt.compressed"), "rb").read()
    _, lraw = oracle.decode(lcet, 1 << 20, 1)
    l_item = (ref.encode(lraw, 5, 22), len(lraw), hashlib.sha256(lraw).hexdigest())
    # (round 6: the probe tells text -- short commands -- from the metric's make-up: the texts are deferred to the launch of small blocks, where the command
    # records and their hand-written run take them at 2.4 times what an engine block does; they count as engine commands there too)
    res, second = run([metric[i % 8] if i % 4 else l_item for i in range(520)])   # (three quarters of the streams, more than half of the bytes, the engines' kind: engine blocks)
    assert second == 130 and all(r.engine_commands >= 0.9 * r.num_commands for r in res), (second, [(r.engine_commands, r.num_commands) for r in res[:4]])
    res, second = run([l_item] * 520)
    assert second == 0 and all(r.engine_commands >= 0.9 * r.num_commands for r in res), (second, [(r.engine_commands, r.num_commands) for r in res[:4]])   # (small blocks for all: nobody was sent back)
    res, second = run([metric[i % 8] for i in range(600)])
    assert second == 0 and all(r.engine_commands >= 0.9 * r.num_commands for r in res), (second, [(r.engine_commands, r.num_commands) for r in res[:4]])
    res, second = run([a_item] * 600)
    assert second == 0 and all(r.num_commands == a_info.num_commands for r in res)   # (small blocks for all of them: nobody was sent back)
    items = [metric[i % 8] if i % 3 else a_item for i in range(600)]
    res, second = run(items)
    assert second == 200, second   # (the texts: deferred to the launch of small blocks)
    assert all(r.engine_commands >= 0.9 * r.num_commands for i, r in enumerate(res) if i % 3), [(i, r.engine_commands, r.num_commands) for i, r in enumerate(res[:6])]
