"""Parity of the HIP decode path against the CPU oracle, through the C ABI (needs a real MI355X)."""
import hashlib
import json
import os
import random

import pytest

import oracle_lib as oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")


def _manifest():
    return json.load(open(os.path.join(GOLD, "manifest.json")))


def _data(name):
    return open(os.path.join(GOLD, "testdata", name), "rb").read()


def _check_against_oracle(pkg, datas, caps, flags=1, what=""):
    batch = pkg.Batch(len(datas))
    results, outs = batch.decode_host(datas, caps, flags)
    batch.close()
    bad = []
    for i, (d, cap) in enumerate(zip(datas, caps)):
        info, exp = oracle.decode(d, cap, flags)
        r = results[i]
        got = (r.result, r.error_code, r.decoded_size, outs[i])
        want = (info.result, info.error_code, info.decoded_size, exp)
        ok = got == want
        if ok and info.result == 1:  # (num_metablocks: also across passes with larger arenas, behind runs of metadata blocks)
            ok = r.consumed == info.consumed and r.num_metablocks == info.num_metablocks and r.num_commands == info.num_commands
        if not ok:
            bad.append((i, what, got[:3], want[:3], r.consumed, info.consumed, len(d), cap))
    assert not bad, (len(bad), bad[:10])


def test_reference_fixtures_bit_exact(pkg):
    """every testdata/*.compressed* of the reference decodes to its original (SHA-256), borked fails"""
    m = [e for e in _manifest() if e["name"] != "rnd_chunk.br"]
    datas = [_data(e["name"]) for e in m]
    caps = [e.get("size", 1 << 16) + 16 for e in m]
    batch = pkg.Batch(len(m))
    results, outs = batch.decode_host(datas, caps, pkg.FLAG_LARGE_WINDOW)
    batch.close()
    for e, r, out in zip(m, results, outs):
        if e.get("must_fail"):
            assert r.result != 1, e["name"]
            continue
        assert (r.result, r.error_code) == (1, 1), (e["name"], r.result, r.error_code)
        assert len(out) == e["size"], e["name"]
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e["name"]
        assert r.consumed == e["csize"], e["name"]


def test_fixtures_match_oracle_status(pkg):
    m = [e for e in _manifest() if e["name"] != "rnd_chunk.br"]
    _check_against_oracle(pkg, [_data(e["name"]) for e in m], [e.get("size", 1 << 16) + 16 for e in m], 1, "fixtures")


def test_large_window_fixture(pkg):
    """rnd_chunk.br: 100 011 280 bytes, prefix / 1e8 zeros / postfix (src/bin/integration_tests.rs:997-1006)"""
    edges = json.load(open(os.path.join(GOLD, "rnd_chunk_edges.json")))
    batch = pkg.Batch(1)
    results, outs = batch.decode_host([_data("rnd_chunk.br")], [edges["size"] + 64], pkg.FLAG_LARGE_WINDOW)
    batch.close()
    r, out = results[0], outs[0]
    assert (r.result, r.decoded_size) == (1, edges["size"])
    pre, post = bytes.fromhex(edges["prefix_hex"]), bytes.fromhex(edges["postfix_hex"])
    assert out[:len(pre)] == pre and out[-len(post):] == post
    assert out[len(pre):len(pre) + edges["zero_count"]].count(0) == edges["zero_count"]
    # without the large-window flag the same stream is rejected (ffi instances, ffi/mod.rs:127)
    batch = pkg.Batch(1)
    results, _ = batch.decode_host([_data("rnd_chunk.br")], [1 << 16], 0)
    batch.close()
    assert (results[0].result, results[0].error_code) == (0, -13)


def test_inline_vectors(pkg):
    vec = json.load(open(os.path.join(GOLD, "inline_vectors.json")))
    datas = [bytes.fromhex(v["input_hex"]) for v in vec]
    caps = [1 << 18] * len(vec)
    batch = pkg.Batch(len(vec))
    results, outs = batch.decode_host(datas, caps, pkg.FLAG_LARGE_WINDOW)
    batch.close()
    for v, r, out, d in zip(vec, results, outs, datas):
        if "result" in v:
            if v["result"] == 1:
                assert r.result == 1, v["name"]
            else:
                assert r.result != 1, v["name"]
        if "error_code" in v:
            assert r.error_code == v["error_code"], v["name"]
        if "output_hex" in v:
            assert out.hex() == v["output_hex"], v["name"]
        if "output_sha256" in v:
            assert hashlib.sha256(out).hexdigest() == v["output_sha256"], v["name"]
        if "result_is_success" in v:
            assert (r.result == 1) == v["result_is_success"], v["name"]
        if v.get("consumed_all"):
            assert r.consumed == len(d), v["name"]
    _check_against_oracle(pkg, datas, caps, 1, "inline")


def _mutations(seed, count):
    rnd = random.Random(seed)
    names = [e["name"] for e in _manifest() if e["csize"] < 200000 and e["name"] != "rnd_chunk.br"]
    base = {n: _data(n) for n in names}
    datas = []
    for _ in range(count):
        d = bytearray(base[rnd.choice(names)])
        k = rnd.random()
        if k < 0.3 and len(d) > 1:
            d = d[:rnd.randrange(0, len(d))]
        elif k < 0.8:
            for _ in range(rnd.choice([1, 1, 1, 2, 3])):
                if not d:
                    break
                pos = rnd.randrange(0, min(len(d), rnd.choice([8, 64, 512, 1 << 20])))
                d[pos] ^= 1 << rnd.randrange(8)
        else:
            pos = rnd.randrange(0, len(d) + 1)
            d[pos:pos] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 4)))
        datas.append(bytes(d))
    return datas


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_mutated_streams_match_oracle(pkg, seed):
    """truncations, bit flips and insertions: result, error code, delivered size and bytes equal the oracle's"""
    datas = _mutations(seed, 1500)
    flags = 1 if seed != 3 else 0
    _check_against_oracle(pkg, datas, [1 << 20] * len(datas), flags, "mutated seed %d" % seed)


def test_unaligned_device_inputs(pkg):
    """every input alignment mod 4 (the bit reader rounds the base pointer down) and odd output addresses"""
    torch = pytest.importorskip("torch")
    alice = _data("alice29.txt.compressed")
    exp = oracle.decode(alice, 200000, 1)[1]
    out = torch.zeros(4 * 160000, dtype=torch.uint8, device="cuda")
    src = torch.frombuffer(bytearray(alice), dtype=torch.uint8).cuda()
    batch = pkg.Batch(4)
    in_ptrs, out_ptrs, bufs = [], [], []
    for a in range(4):
        b = torch.zeros(len(alice) + 64, dtype=torch.uint8, device="cuda")
        b[a:a + len(alice)] = src
        bufs.append(b)
        in_ptrs.append(b.data_ptr() + a)
        out_ptrs.append(out.data_ptr() + a * 160000 + a)
    torch.cuda.synchronize()
    batch.decode_device(in_ptrs, [len(alice)] * 4, out_ptrs, [152089] * 4, pkg.FLAG_LARGE_WINDOW)
    res = batch.wait()
    batch.close()
    host = out.cpu().numpy().tobytes()
    for a in range(4):
        assert (res[a].result, res[a].decoded_size) == (1, 152089)
        assert host[a * 160000 + a: a * 160000 + a + 152089] == exp


def test_tiny_inputs(pkg):
    _check_against_oracle(pkg, [b"", b"\x06", b"\x00", b"\x01"], [16, 16, 16, 0], 1, "tiny")


def test_parameterised_encoder_corpus(pkg):
    """streams with NPOSTFIX/NDIRECT != 0, every context mode, many block types, every window size (standard and
    large), flushes and metadata blocks: bit-exact, and every status field equal to the oracle's"""
    import param_corpus
    streams = param_corpus.corpus()
    if not streams:
        pytest.fail("libbrotlienc is not available: the GPU suite needs the encoder of the image for its synthetic streams (a skip here would let a third of the suite go green unrun)")
    datas = [c for _, c, _ in streams]
    caps = [len(r) + 16 for _, _, r in streams]
    batch = pkg.Batch(len(streams))
    results, outs = batch.decode_host(datas, caps, pkg.FLAG_LARGE_WINDOW)
    batch.close()
    for (label, comp, raw), r, out in zip(streams, results, outs):
        assert (r.result, r.error_code, r.decoded_size, r.consumed) == (1, 1, len(raw), len(comp)), label
        assert out == raw, label
    _check_against_oracle(pkg, datas, caps, 1, "corpus")
    # exact-fit and too-small output buffers
    _check_against_oracle(pkg, datas, [len(r) for _, _, r in streams], 1, "corpus exact fit")
    _check_against_oracle(pkg, datas, [len(r) // 2 for _, _, r in streams], 1, "corpus half")


@pytest.mark.parametrize("arena", [256, 4096])
def test_tables_that_do_not_fit_lds(pkg, arena):
    """the spill policy: with an LDS arena this small most prefix-code tables of a metablock live in the block's
    global scratch (the generic instantiation of the command loop); results must not change"""
    import param_corpus
    m = [e for e in _manifest() if e["name"] != "rnd_chunk.br"]
    datas = [_data(e["name"]) for e in m]
    caps = [e.get("size", 1 << 16) + 16 for e in m]
    for label, comp, raw in param_corpus.corpus()[::3]:
        datas.append(comp)
        caps.append(len(raw) + 16)
    expected = [oracle.decode(d, cap, 1) for d, cap in zip(datas, caps)]
    # (a) spilling in place: the generic instantiation of the command loop runs
    batch = pkg.Batch(len(datas), lds_arena_bytes=arena)
    results, outs = batch.decode_host(datas, caps, pkg.FLAG_LARGE_WINDOW | pkg.FLAG_SPILL_IN_PLACE)
    batch.close()
    assert sum(r.spilled_metablocks for r in results) > 0
    for i, (info, exp) in enumerate(expected):
        r = results[i]
        assert (r.result, r.error_code, r.decoded_size, outs[i]) == (info.result, info.error_code, info.decoded_size, exp), (i, arena)
    # (b) the default policy: those streams continue, from the metablock boundary they stopped at, in a second launch
    # whose blocks have the largest arena; same results, and (these tables all fit 58 KiB) nothing spills
    batch = pkg.Batch(len(datas), lds_arena_bytes=arena)
    results, outs = batch.decode_host(datas, caps, pkg.FLAG_LARGE_WINDOW)
    batch.close()
    assert sum(r.spilled_metablocks for r in results) == 0
    for i, (info, exp) in enumerate(expected):
        r = results[i]
        assert (r.result, r.error_code, r.decoded_size, r.consumed, outs[i]) == \
               (info.result, info.error_code, info.decoded_size, info.consumed if info.result == 1 else r.consumed, exp), (i, arena)


@pytest.mark.parametrize("seed", [11, 12])
def test_mutated_corpus_streams_match_oracle(pkg, seed):
    """truncations, bit flips and insertions applied to the parameterised corpus (every window size, NPOSTFIX/NDIRECT,
    context modes, flushes, metadata): result, error code, delivered size and bytes equal the oracle's"""
    import param_corpus
    streams = [(c, len(r)) for _, c, r in param_corpus.corpus() if len(c) < 60000]
    if not streams:
        pytest.fail("libbrotlienc is not available: the GPU suite needs the encoder of the image for its synthetic streams (a skip here would let a third of the suite go green unrun)")
    rnd = random.Random(seed)
    datas, caps = [], []
    for _ in range(1200):
        c, n = rnd.choice(streams)
        d = bytearray(c)
        k = rnd.random()
        if k < 0.25 and len(d) > 1:
            d = d[:rnd.randrange(0, len(d))]
        elif k < 0.85:
            for _ in range(rnd.choice([1, 1, 2, 3])):
                pos = rnd.randrange(0, min(len(d), rnd.choice([16, 128, 2048, 1 << 20])))
                d[pos] ^= 1 << rnd.randrange(8)
        else:
            pos = rnd.randrange(0, len(d) + 1)
            d[pos:pos] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 4)))
        datas.append(bytes(d))
        # (room for whatever a damaged stream produces: with a buffer that fills up the batch entry points report
        # NEEDS_MORE_OUTPUT at once, the reference at its next ring-buffer flush -- DESIGN.md section 5)
        caps.append(1 << 20)
    _check_against_oracle(pkg, datas, caps, 1, "mutated corpus seed %d" % seed)


@pytest.mark.parametrize("seed", [21, 22])
def test_damaged_streams_with_buffers_that_are_too_small(pkg, seed):
    """What the reference reports for a stream that runs out of output depends on what the stream does up to its next
    ring-buffer flush point (an error or the end of the input in front of it wins over NEEDS_MORE_OUTPUT): damaged and
    truncated streams with output buffers of every size get the oracle's verdict from the batch entry points; with
    BROTLI_AMD_BATCH_EAGER_OUTPUT_LIMIT they are reported NEEDS_MORE_OUTPUT where the buffer ends."""
    import param_corpus
    streams = [(c, len(r)) for _, c, r in param_corpus.committed() if 200 < len(c) < 60000]
    m = [e for e in _manifest() if e["name"] != "rnd_chunk.br" and 2000 < e["csize"] < 200000 and e.get("size", 0) > 100]
    streams += [(_data(e["name"]), e["size"]) for e in m]
    rnd = random.Random(seed)
    datas, caps = [], []
    for _ in range(1500):
        c, n = rnd.choice(streams)
        d = bytearray(c)
        k = rnd.random()
        if k < 0.45:
            d = d[:rnd.randrange(1, len(d))]
        elif k < 0.9:
            for _ in range(rnd.choice([1, 1, 2])):
                pos = rnd.randrange(len(d) // 8, len(d))
                d[pos] ^= 1 << rnd.randrange(8)
        datas.append(bytes(d))
        caps.append(rnd.choice([rnd.randrange(1, n + 1), rnd.randrange(1, n + 1), n - 1, n // 2, max(1, n - rnd.randrange(1, 70000))]))
    batch = pkg.Batch(len(datas))
    results, outs = batch.decode_host(datas, caps, 1)
    eager, _ = batch.decode_host(datas, caps, 1 | pkg.FLAG_EAGER_OUTPUT_LIMIT)
    batch.close()
    bad, settled = [], 0
    for i, (d, cap) in enumerate(zip(datas, caps)):
        info, exp = oracle.decode(d, cap, 1)
        r = results[i]
        if (r.result, r.error_code, r.decoded_size, outs[i]) != (info.result, info.error_code, info.decoded_size, exp):
            bad.append((i, (r.result, r.error_code, r.decoded_size), (info.result, info.error_code, info.decoded_size), len(d), cap))
        if eager[i].result == 3 and r.result != 3:
            settled += 1
            assert eager[i].decoded_size == cap
        else:
            assert (eager[i].result, eager[i].error_code, eager[i].decoded_size) == (r.result, r.error_code, r.decoded_size), i
    assert not bad, (len(bad), bad[:10])
    assert settled > 20, settled  # (the case exists in this sample)


def test_error_right_behind_a_flush_point(pkg):
    """found by the engine fuzzer: the first metablock of this (damaged) stream fills the 64 KiB ring exactly, the header
    behind it is invalid.  The reference flushes as soon as the ring is full, so with less than 64 KiB of output it
    reports NEEDS_MORE_OUTPUT, with 64 KiB or more the header error (after delivering 64 KiB); the batch entry points'
    second decode and the one-shot entry point stop one byte short of the flush point to tell the two apart."""
    d = open(os.path.join(ROOT, "tests", "golden", "regress", "ring_full_then_header_error.br"), "rb").read()
    caps = [1, 42011, 65535, 65536, 65537, 100000]
    want = [oracle.decode(d, cap, 1) for cap in caps]
    assert [(i.result, i.error_code) for i, _ in want] == [(3, 3)] * 3 + [(0, -7)] * 3
    _check_against_oracle(pkg, [d] * len(caps), caps, 1, "ring full, then a header error")
    for cap, (oinfo, exp) in zip(caps, want):
        info, out = pkg.brotli_decode(d, cap)
        assert (info.result, info.code, info.decoded_size, out) == (oinfo.result, oinfo.error_code, oinfo.decoded_size, exp), cap


def test_bench_workload_streams_tight_buffers_and_damage(pkg):
    """the bench workloads' own streams (long literal runs, copies of more than 1 KiB: the lean loop's limits): valid
    streams with output buffers that are exact, one short, half, ...; damaged streams with roomy buffers"""
    import sys
    sys.path.insert(0, ROOT)
    import workloads as w
    if not w.encoder_available():
        pytest.fail("libbrotlienc is not available: the GPU suite needs the encoder of the image for its synthetic streams (a skip here would let a third of the suite go green unrun)")
    rnd = random.Random(7)
    streams = w.make_streams("long_backref", 2, 4 << 20, 5000) + w.make_streams("high_entropy", 1, 4 << 20, 6000) + \
        w.make_streams("long_backref", 2, 1 << 20, 7000) + w.make_streams("high_entropy", 1, 256 << 10, 8000)
    datas, caps = [], []
    for c, n, _ in streams:
        for cap in (n, n - 1, n // 2, n // 3 + 17, rnd.randrange(1, n), n + 1000):
            datas.append(c)
            caps.append(cap)
        for _ in range(16):
            d = bytearray(c)
            if rnd.random() < 0.3:
                d = d[:rnd.randrange(1, len(d))]
            else:
                for _ in range(rnd.choice([1, 1, 2, 4])):
                    pos = rnd.randrange(0, min(len(d), rnd.choice([64, 4096, 1 << 22])))
                    d[pos] ^= 1 << rnd.randrange(8)
            datas.append(bytes(d))
            caps.append(8 << 20)
    _check_against_oracle(pkg, datas, caps, 1, "bench workload streams")


def test_sharded_decode_single_rank(pkg):
    """sharding.decode_sharded with the real device decode as its per-rank function (world size 1 here; the 2-rank
    partition/collective logic runs on CPU under gloo in test_sharding_gloo.py)"""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("brotli_amd_sharding", os.path.join(ROOT, "rust-brotli-decompressor_amd", "sharding.py"))
    sharding = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharding)
    m = [e for e in _manifest() if e["name"] != "rnd_chunk.br" and not e.get("must_fail")][:20]
    datas = [_data(e["name"]) for e in m]
    caps = [e["size"] + 16 for e in m]

    def decode_fn(streams, out_caps):
        batch = pkg.Batch(max(1, len(streams)))
        results, outs = batch.decode_host(streams, out_caps, pkg.FLAG_LARGE_WINDOW)
        batch.close()
        return np.array([[r.result, r.error_code, r.decoded_size, r.consumed] for r in results], dtype=np.int64).reshape(-1, 4), outs

    mine, outs, status = sharding.decode_sharded(datas, caps, decode_fn, weights=[e["size"] for e in m])
    assert mine == list(range(len(m)))
    for e, out, row in zip(m, outs, status):
        assert (int(row[0]), int(row[2]), int(row[3])) == (1, e["size"], e["csize"]), e["name"]
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e["name"]


def test_helper_rounds_with_a_code_that_never_resynchronises(pkg):
    """Long literal runs are decoded in rounds whose later chunks the helper waves decode speculatively from a guessed
    literal boundary, counting on prefix codes to re-synchronise.  This data's optimal code has lengths 6, 9, 12 and 15
    only (253 symbols with probabilities 2^-6 ... 2^-15), so a chain that starts off the grid of multiples of three
    never meets the true one: every round must end after its first chunk, and the output must not care.  A second
    stream (lengths 7..9) is the usual case where the helpers' chunks do fall in."""
    import numpy as np
    import libbrotli_ref as ref
    if not ref.encoder_available():
        pytest.fail("libbrotlienc is not available: the GPU suite needs the encoder of the image for its synthetic streams (a skip here would let a third of the suite go green unrun)")
    rng = np.random.Generator(np.random.PCG64(99))
    syms = rng.permutation(256)[:253].astype(np.uint8)
    block = np.concatenate([np.repeat(syms[:60], 512), np.repeat(syms[60:89], 64), np.repeat(syms[89:93], 8), syms[93:253]])
    assert len(block) == 32768
    grid3 = np.concatenate([rng.permutation(block) for _ in range(8)]).tobytes()  # 256 KiB of literals, no repeats to speak of
    usual = rng.choice(256, size=300000, p=(np.arange(1, 257) ** -0.3) / np.sum(np.arange(1, 257) ** -0.3)).astype(np.uint8).tobytes()
    datas, raws = [], []
    for raw in (grid3, usual):
        raw = raw + raw[:50000]  # a long copy at the end
        datas.append(ref.encode(raw, 5, 22))
        raws.append(raw)
        info, out = oracle.decode(datas[-1], len(raw) + 16, 1)
        assert info.result == 1 and out == raw and info.num_literals > 200000 and info.num_commands < 64  # long runs indeed
    batch = pkg.Batch(2)
    results, outs = batch.decode_host(datas, [len(r) + 16 for r in raws], pkg.FLAG_LARGE_WINDOW)
    batch.close()
    for r, out, raw, d in zip(results, outs, raws, datas):
        assert (r.result, r.error_code, r.decoded_size, r.consumed) == (1, 1, len(raw), len(d))
        assert out == raw


@pytest.mark.parametrize("n", [2304, 700])
def test_batches_larger_than_the_resident_grid(pkg, n):
    """The host picks the block shape by batch size: eight waves (seven helpers) up to two blocks per CU, four waves up
    to four blocks per CU (700 streams on a 256-CU device), one-wave blocks beyond (2304 streams: no helper waves, up to
    eight blocks per CU with a smaller table arena; streams whose tables need the large arena come back in the second
    pass).  Same results in every shape, including a stream with literal runs long enough for helper rounds."""
    import numpy as np
    import libbrotli_ref as ref
    m = [e for e in _manifest() if not e.get("must_fail") and e.get("size", 1 << 30) <= 300000][:24]
    datas = [_data(e["name"]) for e in m]
    want = [(e["size"], e["sha256"]) for e in m]
    if ref.encoder_available():
        rng = np.random.Generator(np.random.PCG64(7))
        raw = rng.choice(256, size=200000, p=(np.arange(1, 257) ** -0.3) / np.sum(np.arange(1, 257) ** -0.3)).astype(np.uint8).tobytes()
        datas.append(ref.encode(raw, 5, 22))
        want.append((len(raw), hashlib.sha256(raw).hexdigest()))
    idx = [i % len(datas) for i in range(n)]
    batch = pkg.Batch(n)
    results, outs = batch.decode_host([datas[i] for i in idx], [want[i][0] for i in idx], pkg.FLAG_LARGE_WINDOW)
    batch.close()
    for k, i in enumerate(idx):
        r = results[k]
        assert (r.result, r.decoded_size, r.consumed) == (1, want[i][0], len(datas[i])), (k, i)
        assert hashlib.sha256(outs[k]).hexdigest() == want[i][1], (k, i)


def test_literal_runs_of_many_lengths(pkg):
    """Literal runs from a few hundred to tens of thousands of bytes, at entropies from under two bits to nearly eight
    per literal, separated by copies: rounds of the helper waves that end in the middle of a chunk because the run does
    (the last literal's bit position comes from the recorded start masks), runs too short for a round, runs that end
    exactly at a window or chunk boundary by chance.  Also decoded into buffers that end inside a run."""
    import numpy as np
    import libbrotli_ref as ref
    if not ref.encoder_available():
        pytest.fail("libbrotlienc is not available: the GPU suite needs the encoder of the image for its synthetic streams (a skip here would let a third of the suite go green unrun)")
    rng = np.random.Generator(np.random.PCG64(2024))
    datas, raws = [], []
    for skew, nsym in ((0.2, 256), (1.0, 256), (2.0, 64), (3.0, 8), (0.0, 200)):
        p = np.arange(1, nsym + 1, dtype=np.float64) ** -skew
        p /= p.sum()
        perm = rng.permutation(256)[:nsym]
        parts, filler = [], rng.integers(0, 256, size=4096, dtype=np.uint8).tobytes()
        parts.append(filler)
        for run in (700, 767, 768, 769, 1100, 2047, 2048, 2049, 3000, 4095, 5000, 8191, 8256, 9000, 12000, 16383, 16385, 20000, 33000):
            run += int(rng.integers(0, 3))
            parts.append(perm[rng.choice(nsym, size=run, p=p)].astype(np.uint8).tobytes())
            parts.append(filler[: int(rng.integers(16, 600))])  # a copy from the start of the stream
        raw = b"".join(parts)
        c = ref.encode(raw, 5, 22)
        info, out = oracle.decode(c, len(raw), 1)
        assert info.result == 1 and out == raw
        datas.append(c); raws.append(raw)
    all_d, all_caps = [], []
    for c, raw in zip(datas, raws):
        for cap in (len(raw), len(raw) - 1, len(raw) // 2, len(raw) // 3, 4096 + 700 + 300, 40000, 100001):
            all_d.append(c); all_caps.append(cap)
    _check_against_oracle(pkg, all_d, all_caps, 1, "literal runs")
    # the same through four-wave blocks (a batch of more than two and at most four blocks per CU)
    reps = 600 // len(all_d) + 1
    batch = pkg.Batch(len(all_d))
    r1, o1 = batch.decode_host(all_d, all_caps, 1)
    batch.close()
    batch = pkg.Batch(len(all_d) * reps)
    r4, o4 = batch.decode_host(all_d * reps, all_caps * reps, 1)
    batch.close()
    for k in range(len(all_d) * reps):
        a, b = r1[k % len(all_d)], r4[k]
        assert (a.result, a.error_code, a.decoded_size, a.consumed) == (b.result, b.error_code, b.decoded_size, b.consumed), k
        assert o1[k % len(all_d)] == o4[k], k


def test_long_literal_runs_across_ring_flush_points(pkg):
    """Literal runs much longer than the ring buffer (windows of 1 KiB to 256 KiB, runs of 10^5 literals): the reference
    flushes its ring every window's worth of output, which bounds how far a command may go without a check; the lean
    loop does not take such a run, the checked path decodes it part by part between flush points -- with helper rounds
    where a part is long enough -- and the delivered sizes at every output limit must still be the reference's."""
    import numpy as np
    import libbrotli_ref as ref
    if not ref.encoder_available():
        pytest.fail("libbrotlienc is not available: the GPU suite needs the encoder of the image for its synthetic streams (a skip here would let a third of the suite go green unrun)")
    rng = np.random.Generator(np.random.PCG64(4242))
    p = np.arange(1, 257, dtype=np.float64) ** -0.5
    p /= p.sum()
    perm = rng.permutation(256)
    raw = perm[rng.choice(256, size=700000, p=p)].astype(np.uint8).tobytes()
    raw = raw + raw[1000:3000] + raw[:50]
    datas, caps = [], []
    for lgwin in (10, 12, 16, 18, 22):
        for q in (1, 5):
            c = ref.encode(raw, q, lgwin)
            for cap in (len(raw), len(raw) - 1, 300000, 65536, 65537, 4097, 262144 + 5):
                datas.append(c); caps.append(cap)
    _check_against_oracle(pkg, datas, caps, 1, "ring flush points")


@pytest.mark.parametrize("seed", [101, 102])
def test_literal_heavy_streams_of_random_makeup(pkg, seed):
    """Segments of literals whose distribution changes inside a stream (the encoder answers with several literal block
    types and trees, so block switches fall inside long runs), copies in between, qualities 1-9, windows 16-24, tight
    and roomy buffers: the helper rounds next to block switches (tests/tools/fuzz_rounds.py runs this at length)."""
    import numpy as np
    import libbrotli_ref as ref
    if not ref.encoder_available():
        pytest.fail("libbrotlienc is not available: the GPU suite needs the encoder of the image for its synthetic streams (a skip here would let a third of the suite go green unrun)")
    rng = np.random.Generator(np.random.PCG64(seed))
    datas, caps = [], []
    for _ in range(20):
        parts = []
        for _ in range(int(rng.integers(2, 9))):
            nsym = int(rng.choice([2, 5, 17, 64, 200, 256]))
            p = np.arange(1, nsym + 1, dtype=np.float64) ** -float(rng.choice([0.0, 0.3, 1.0, 2.5]))
            p /= p.sum()
            perm = rng.permutation(256)[:nsym]
            parts.append(perm[rng.choice(nsym, size=int(rng.choice([300, 800, 3000, 20000, 70000, 150000])), p=p)].astype(np.uint8).tobytes())
            if rng.random() < 0.7:
                parts.append(parts[int(rng.integers(0, len(parts)))][: int(rng.integers(4, 2000))])
        raw = b"".join(parts)
        c = ref.encode(raw, int(rng.choice([1, 3, 5, 6, 9])), int(rng.choice([16, 18, 20, 22, 24])))
        for cap in (len(raw), int(rng.integers(1, len(raw))), len(raw) + 100):
            datas.append(c); caps.append(cap)
    _check_against_oracle(pkg, datas, caps, 1, "random make-up")


def test_prefix_code_headers_cut_at_every_byte_and_repeat_code_runs(pkg):
    """The symbol code lengths of a prefix code are read sixty-four stream bits a step (round 5, read_symbol_lengths_wide: the chain of
    code-length words by pointer doubling, repeat codes and code space by scans) -- same words, same order, same verdicts as the
    reference's loop (decode.rs:661-797).  Streams whose headers exercise it: every fixture cut at EVERY byte of its first 700 (the input
    ends inside the code-length code, inside a word, inside a repeat code's extra bits ...), and data whose alphabets make long runs of
    repeat codes (sixteens behind sixteens, seventeens behind seventeens: the run length is a number in base four / eight)."""
    import libbrotli_ref as ref
    gold = os.path.join(ROOT, "tests", "golden", "testdata")
    datas, caps = [], []
    for name in ("alice29.txt.compressed", "asyoulik.txt.compressed", "mapsdatazrh.compressed", "ukkonooa.compressed", "compressed_file.compressed", "plrabn12.txt.compressed"):
        if not os.path.exists(os.path.join(gold, name)):
            continue
        c = open(os.path.join(gold, name), "rb").read()
        for cut in range(1, min(len(c), 700)):
            datas.append(c[:cut]); caps.append(1 << 16)
    if ref.encoder_available():
        rnd = random.Random(77)
        for k in range(24):
            # few distinct byte values far apart: literal codes with long runs of zero lengths (seventeens), or many symbols of one
            # length (sixteens); a sprinkling of others so that runs of different kinds meet
            vals = rnd.sample(range(256), rnd.choice([2, 3, 5, 17, 40, 200, 256]))
            raw = bytes(rnd.choice(vals) for _ in range(rnd.choice([300, 5000, 70000])))
            c = ref.encode(raw + bytes(range(256)) * rnd.choice([0, 1]), rnd.choice([1, 5, 9, 11]), 18)
            datas.append(c); caps.append(1 << 18)
            for _ in range(6):
                d = bytearray(c); pos = rnd.randrange(0, min(len(d), 200)); d[pos] ^= 1 << rnd.randrange(8)
                datas.append(bytes(d)); caps.append(1 << 18)
    for i in range(0, len(datas), 2000):
        _check_against_oracle(pkg, datas[i:i + 2000], caps[i:i + 2000], 1, "headers")
