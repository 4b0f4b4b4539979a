"""Several CUs on one stream (needs a real MI355X): batches of fewer streams than half the device's CUs give every stream a GANG of
blocks -- its owner and one, three or seven helper blocks that take the path engine's regions in turns with it
(csrc/brotli_path_engine.h, PE_CFG_REMOTE; the words they exchange: GC_* in csrc/brotli_kernels.hip).

What the reference does for one stream is one serial loop (src/decode.rs:2330-2744, ProcessCommandsInternal); whatever the number of
blocks on a stream, its bytes and status words must be the oracle's."""
import hashlib
import os
import random
import sys

import pytest

import oracle_lib as oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu

sys.path.insert(0, ROOT)


def _w():
    import workloads as w
    if not w.encoder_available():
        pytest.fail("libbrotlienc is not available: the GPU suite needs the encoder of the image for its synthetic streams")
    return w


def _decode(pkg, datas, caps, flags=1):
    b = pkg.Batch(len(datas))
    res, outs = b.decode_host(datas, caps, flags)
    gang = b.last_gang()
    b.close()
    return res, outs, gang


def _against_oracle(res, outs, datas, caps, flags=1):
    bad = []
    for i, (d, cap) in enumerate(zip(datas, caps)):
        info, exp = oracle.decode(d, cap, flags)
        r = res[i]
        ok = (r.result, r.error_code, r.decoded_size, outs[i]) == (info.result, info.error_code, info.decoded_size, exp)
        if ok and info.result == 1:
            ok = r.consumed == info.consumed and r.num_commands == info.num_commands and r.num_metablocks == info.num_metablocks
        if not ok:
            bad.append((i, (r.result, r.error_code, r.decoded_size, r.consumed, r.num_commands), (info.result, info.error_code, info.decoded_size, info.consumed, info.num_commands), len(d), cap))
    assert not bad, (len(bad), bad[:8])


def _pool(w, rnd):
    """streams of every kind the command loop knows, each with its raw size: the metric's make-up at several sizes and qualities, its
    survey variant (a quarter seed), high-entropy literals (long literal runs: regions of their own, the one-block form's), real text at
    -q 5 (words of the static dictionary), a context-modelled fixture (no engine at all), an executable (dozens of block types)"""
    pool = []
    for k, size in enumerate((4 << 20, 256 << 10, 1 << 20, 1 << 20)):   # (the first one large: a batch of small streams gets no gangs, see below)
        raw = w.long_backref_stream(7000 + k, size)
        pool.append((w.brotli_compress(raw, rnd.choice([4, 5, 5, 9]), rnd.choice([18, 22, 22, 24])), len(raw)))
    raw = w.long_backref_stream(7100, 2 << 20, seed_shift=2)   # (a quarter of it seed)
    pool.append((w.brotli_compress(raw, 5, 22), len(raw)))
    raw = w.high_entropy_stream(7200, 1 << 20)
    pool.append((w.brotli_compress(raw, 5, 22), len(raw)))
    gold = os.path.join(ROOT, "tests", "golden", "testdata")
    alice = open(os.path.join(gold, "alice29.txt.compressed"), "rb").read()
    info, text = oracle.decode(alice, 200000, 1)
    assert info.result == 1
    pool.append((alice, len(text)))
    pool.append((w.brotli_compress(text, 5, 22), len(text)))
    exe = open(sys.executable, "rb").read()[: 1 << 20]
    pool.append((w.brotli_compress(exe, 5, 22), len(exe)))
    return pool


@pytest.mark.parametrize("n,blocks", [(1, 8), (3, 8), (8, 8), (9, 8), (32, 8), (33, 0), (64, 0), (65, 0), (128, 0), (129, 0)])
def test_every_stream_of_a_small_batch_has_a_gang_of_blocks(pkg, n, blocks):
    """n streams of very different sizes -- whole, with a buffer one byte short, truncated, with a flipped bit -- against the oracle; the host says
    how many blocks each stream had (BrotliAmdBatchLastGang): eight up to 32 streams; beyond that such a batch is a POOL (blocks 0 here:
    BrotliAmdBatchLastPool) -- the blocks without a stream of their own help the largest streams still being decoded"""
    w = _w()
    rnd = random.Random(20261001 + n)
    pool = _pool(w, rnd)
    datas, caps = [], []
    for i in range(n):
        c, size = pool[i % len(pool)] if i < len(pool) else rnd.choice(pool)
        d, cap = c, size
        k = rnd.random()
        if i >= len(pool) and k < 0.15:
            cap = rnd.choice([size - 1, size // 2, rnd.randrange(1, size)])
        elif i >= len(pool) and k < 0.3:
            d = c[: rnd.randrange(1, len(c))]
        elif i >= len(pool) and k < 0.5:
            t = bytearray(c)
            t[rnd.randrange(len(t))] ^= 1 << rnd.randrange(8)
            d = bytes(t)
        datas.append(d)
        caps.append(cap)
    b = pkg.Batch(len(datas))
    res, outs = b.decode_host(datas, caps, 1)
    gang, pool = b.last_gang(), b.last_pool()
    b.close()
    assert (gang, pool) == ((blocks, False) if blocks else (1, True)), (n, gang, pool)
    _against_oracle(res, outs, datas, caps)


@pytest.mark.parametrize("n,blocks", [(33, 4), (64, 4), (65, 2), (128, 2), (129, 1)])
def test_batches_of_equal_streams_get_gangs_of_four_and_two(pkg, n, blocks):
    """streams of a size: four blocks a stream up to 64 streams, two up to 128, one beyond (no pool: they end together)"""
    w = _w()
    us = w.make_streams("long_backref", 4, 1 << 20, 1000)
    datas, caps = [us[i % 4][0] for i in range(n)], [us[i % 4][1] for i in range(n)]
    b = pkg.Batch(n)
    res, outs = b.decode_host(datas, caps, 1)
    gang, pool = b.last_gang(), b.last_pool()
    b.close()
    assert (gang, pool) == (blocks, False), (n, gang, pool)
    for i, (r, o) in enumerate(zip(res, outs)):
        assert r.result == 1 and hashlib.sha256(o).hexdigest() == us[i % 4][2], i


def test_a_gang_and_one_block_agree_and_the_gang_takes_the_commands(pkg):
    """the same eight streams by gangs of eight blocks and (BROTLI_AMD_GANG=0) by one block each: the same bytes and status words, and in both
    runs the command engines take (nearly) all commands of the streams they are built for"""
    w = _w()
    us = w.make_streams("long_backref", 6, 4 << 20, 1000) + w.make_streams("survey_mix", 2, 4 << 20, 4000)
    datas, caps = [c for c, _, _ in us], [sz for _, sz, _ in us]
    old = os.environ.get("BROTLI_AMD_GANG")
    try:
        os.environ.pop("BROTLI_AMD_GANG", None)
        res_g, outs_g, gang_g = _decode(pkg, datas, caps)
        os.environ["BROTLI_AMD_GANG"] = "0"
        res_1, outs_1, gang_1 = _decode(pkg, datas, caps)
    finally:
        if old is None:
            os.environ.pop("BROTLI_AMD_GANG", None)
        else:
            os.environ["BROTLI_AMD_GANG"] = old
    assert (gang_g, gang_1) == (8, 1)
    for i, (_, sz, sha) in enumerate(us):
        for r, o in ((res_g[i], outs_g[i]), (res_1[i], outs_1[i])):
            assert (r.result, r.decoded_size) == (1, sz) and hashlib.sha256(o).hexdigest() == sha, i
            assert r.engine_commands >= 0.99 * r.num_commands, (i, r.engine_commands, r.num_commands)
        assert (res_g[i].consumed, res_g[i].num_commands, res_g[i].num_metablocks) == (res_1[i].consumed, res_1[i].num_commands, res_1[i].num_metablocks)


def test_one_batch_object_through_gang_launches_and_others(pkg):
    """one batch object, launches of 1, 200, 5 and 40 streams in turn (gangs of eight, a pool, eight, a pool): the gangs' control blocks are
    the object's, zeroed before every launch that has any"""
    w = _w()
    rnd = random.Random(77)
    pool = _pool(w, rnd)
    b = pkg.Batch(200)
    for n, blocks in ((1, 8), (200, 0), (5, 8), (40, 0), (1, 8)):   # (0: a pool -- streams of very different sizes, more than 32 of them)
        picks = [pool[(i * 5) % len(pool)] for i in range(n)]   # (the first one the 4 MiB stream: see test_batches_of_small_streams_get_no_gangs)
        datas, caps = [c for c, _ in picks], [sz for _, sz in picks]
        res, outs = b.decode_host(datas, caps, 1)
        assert (b.last_gang(), b.last_pool()) == ((blocks, False) if blocks else (1, True)), (n, b.last_gang(), b.last_pool())
        _against_oracle(res, outs, datas, caps)
    b.close()


def test_helpers_that_never_turn_up(pkg):
    """BROTLI_AMD_GANG_NO_HELPERS: the gangs' helper blocks leave at once, as blocks the device has no CU for would never start; the owners wait
    for them once (a millisecond or two), dissolve their gangs and decode their streams alone -- same bytes, same status words"""
    w = _w()
    us = w.make_streams("long_backref", 3, 1 << 20, 1000)
    datas, caps = [c for c, _, _ in us], [sz for _, sz, _ in us]
    old = os.environ.get("BROTLI_AMD_GANG_NO_HELPERS")
    try:
        os.environ["BROTLI_AMD_GANG_NO_HELPERS"] = "1"
        res, outs, gang = _decode(pkg, datas, caps)
    finally:
        if old is None:
            os.environ.pop("BROTLI_AMD_GANG_NO_HELPERS", None)
        else:
            os.environ["BROTLI_AMD_GANG_NO_HELPERS"] = old
    assert gang == 8
    _against_oracle(res, outs, datas, caps)
    for r in res:
        assert r.engine_commands >= 0.9 * r.num_commands, (r.engine_commands, r.num_commands)


def _chained_copies(seed, size):
    """data whose copies BUILD ON EACH OTHER: units of 9 .. 40 bytes, each the unit before it with a byte or two changed -- a copy from one unit back
    and a literal, hundreds in a row (every copy reads what the copy before it wrote: levels far beyond PE_DEP_ROUNDS, the execute's in-order tail) --,
    stretches of units that repeat units from a few KiB back (copies that read the REGION BEFORE's output: a gang's lagging copies, and what builds on
    them), and long runs of one short pattern (copies that repeat themselves)"""
    rnd = random.Random(seed)
    out = bytearray(rnd.randbytes(256))
    while len(out) < size:
        k = rnd.random()
        if k < 0.6:
            u = rnd.randrange(9, 41)
            unit = bytearray(out[-u:])
            for _ in range(rnd.randrange(20, 400)):
                unit[rnd.randrange(u)] = rnd.randrange(256)
                if rnd.random() < 0.3:
                    unit[rnd.randrange(u)] = rnd.randrange(256)
                out += unit
        elif k < 0.9:
            back = rnd.randrange(2000, 60000)
            if back < len(out):
                for _ in range(rnd.randrange(5, 60)):
                    a = len(out) - back + rnd.randrange(-200, 200)
                    a = max(0, min(a, len(out) - 64))
                    out += out[a:a + rnd.randrange(8, 64)]
                    if rnd.random() < 0.5:
                        out.append(rnd.randrange(256))
        else:
            pat = rnd.randbytes(rnd.randrange(1, 9))
            out += pat * rnd.randrange(20, 300)
    return bytes(out[:size])


@pytest.mark.parametrize("blocks", ["16", "8", "0"])
def test_copies_that_build_on_each_other(pkg, blocks):
    """The execute's levels (csrc/brotli_path_engine.h, `dependent_copies`; the reference copies byte by byte, decode.rs:2641-2720): chains of copies
    far deeper than the levels go, copies that read the region before's output and copies that build on those, copies that repeat themselves -- one long
    stream by sixteen blocks, by eight and by one, and a batch of shorter ones, whole and with buffers that end inside a chain."""
    w = _w()
    rnd = random.Random(5)
    datas, caps = [], []
    raw = _chained_copies(900, 6 << 20)
    for q in (5, 9):
        c = w.brotli_compress(raw, q, 22)
        datas.append(c); caps.append(len(raw))
    for k in range(4):
        r = _chained_copies(910 + k, rnd.randrange(300 << 10, 1 << 20))
        c = w.brotli_compress(r, rnd.choice([4, 5, 6]), rnd.choice([18, 20, 22]))
        datas += [c, c]; caps += [len(r), rnd.randrange(len(r) // 2, len(r))]
    old = os.environ.get("BROTLI_AMD_GANG")
    os.environ["BROTLI_AMD_GANG"] = blocks
    try:
        res, outs, gang = _decode(pkg, datas[:1], caps[:1])      # (one long stream: the gang asked for)
        assert gang == (int(blocks) or 1)
        _against_oracle(res, outs, datas[:1], caps[:1])
        assert res[0].engine_commands >= 0.9 * res[0].num_commands
        res, outs, gang = _decode(pkg, datas, caps)
        _against_oracle(res, outs, datas, caps)
    finally:
        if old is None:
            os.environ.pop("BROTLI_AMD_GANG", None)
        else:
            os.environ["BROTLI_AMD_GANG"] = old


def test_batches_of_small_streams_get_no_gangs(pkg):
    """a gang has something to divide from 64 KiB of compressed data on (a dozen regions) and costs a launch ten microseconds: batches whose
    largest stream is smaller are launched with one block a stream, whatever their number; one large stream among them brings the gangs back"""
    w = _w()
    small = [w.brotli_compress(w.long_backref_stream(7300 + k, 128 << 10), 5, 22) for k in range(4)]
    assert max(len(c) for c in small) < 65536
    res, outs, gang = _decode(pkg, small, [128 << 10] * 4)
    assert gang == 1
    _against_oracle(res, outs, small, [128 << 10] * 4)
    big = w.brotli_compress(w.long_backref_stream(7310, 2 << 20), 5, 22)
    assert len(big) >= 65536
    res, outs, gang = _decode(pkg, small + [big], [128 << 10] * 4 + [2 << 20])
    assert gang == 8
    _against_oracle(res, outs, small + [big], [128 << 10] * 4 + [2 << 20])


def test_a_pool_of_blocks_helps_the_streams_that_last(pkg):
    """more streams than half the CUs, fewer than CUs, one of them sixteen times the others: a pool launch -- the blocks without a stream of their
    own join the streams still being decoded, the large one first (BrotliAmdBatchLastPool) -- against the oracle, damaged streams among them;
    the same with BROTLI_AMD_POOL=0 (no pool): the same bytes and status words"""
    w = _w()
    rnd = random.Random(4242)
    big_raw = w.long_backref_stream(7400, 16 << 20)
    big = w.brotli_compress(big_raw, 5, 22)
    smalls = [(w.brotli_compress(w.long_backref_stream(7410 + k, 1 << 20), 5, 22), 1 << 20) for k in range(6)]
    datas, caps = [big], [len(big_raw)]
    for i in range(149):
        c, sz = smalls[i % len(smalls)]
        d, cap = c, sz
        k = rnd.random()
        if k < 0.1:
            d = c[: rnd.randrange(1, len(c))]
        elif k < 0.2:
            t = bytearray(c)
            t[rnd.randrange(len(t))] ^= 1 << rnd.randrange(8)
            d = bytes(t)
        elif k < 0.3:
            cap = rnd.randrange(1, sz)
        datas.append(d)
        caps.append(cap)
    b = pkg.Batch(len(datas))
    res, outs = b.decode_host(datas, caps, 1)
    assert b.last_pool() and b.last_gang() == 1
    _against_oracle(res, outs, datas, caps)
    assert res[0].engine_commands >= 0.99 * res[0].num_commands
    old = os.environ.get("BROTLI_AMD_POOL")
    try:
        os.environ["BROTLI_AMD_POOL"] = "0"
        b2 = pkg.Batch(len(datas))
        res2, outs2 = b2.decode_host(datas, caps, 1)
        assert not b2.last_pool()
        b2.close()
    finally:
        if old is None:
            os.environ.pop("BROTLI_AMD_POOL", None)
        else:
            os.environ["BROTLI_AMD_POOL"] = old
    b.close()
    assert [(r.result, r.error_code, r.decoded_size, r.consumed, r.num_commands) for r in res] == [(r.result, r.error_code, r.decoded_size, r.consumed, r.num_commands) for r in res2]
    assert outs == outs2
