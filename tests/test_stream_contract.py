"""The per-call contract of the streaming entry point (SURVEY 8b / VERDICT r2 item 7): what EVERY call returns --
(result, consumed, produced) -- under the chunkings of the reference's integration tests (src/bin/integration_tests.rs:122-216).

tests/stream_model.py restates the reference's resumable driver (src/decode.rs:2779-2911) on top of the oracle's trace.  The
Rust reference cannot be run here; what pins the model is Google's libbrotlidec 1.0.9 -- the C decoder whose driver the
reference's is a line-by-line port of -- fed the same schedules (CPU, below), on streams that fit the ring buffer under both
decoders' sizing rules.  The GPU tests compare the product with the model: call for call on streams that fit their ring buffer;
on streams that wrap it, every byte, the totals and the final result (this decoder keeps no ring: where inside such a stream a
NEEDS_MORE_OUTPUT falls is its own)."""
import os

import pytest

import libbrotli_ref as ref
import oracle_lib as oracle
import stream_model as sm
from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden", "testdata")
CHUNKINGS = [(65536, 65536), (1, 65536), (65536, 1), (1, 1), (3, 3), (12, 1)]
SMALL = ["10x10y.compressed", "64x.compressed", "ukkonooa.compressed", "monkey.compressed", "x.compressed.03", "xyzzy.compressed",
         "quickfox.compressed", "ends_with_truncated_dictionary.compressed", "empty.compressed", "random1024.br", "fuzz502.compressed",
         "quickfox_repeated.compressed"]
BIG = ["alice29.txt.compressed", "mapsdatazrh.compressed"]
WRAP = ["metablock_reset.compressed", "zeros.compressed", "backward65536.compressed", "compressed_repeated.compressed", "random_org_10k.bin.compressed"]


def _data(name):
    return open(os.path.join(GOLD, name), "rb").read()


def _model_seq(data, ic, oc):
    m = sm.ReferenceStream(data)
    return sm.run_schedule(lambda pending, cap: m.call(len(pending), cap), data, ic, oc)


@pytest.mark.skipif(not ref.available(), reason="libbrotlidec not available")
@pytest.mark.parametrize("chunks", CHUNKINGS)
def test_model_of_the_reference_driver_against_libbrotlidec(chunks):
    ic, oc = chunks
    names = SMALL + (BIG if ic > 1 and oc > 1 else BIG[:1] if (ic, oc) != (1, 1) else [])
    for name in names:
        data = _data(name)
        for d in (data, data[: max(1, len(data) * 2 // 3)]):      # whole, and cut short (ends in NEEDS_MORE_INPUT)
            dec = ref.StreamDecoder()
            got = sm.run_schedule(lambda pending, cap: (lambda r: (r[0], r[1], len(r[2])))(dec.step(pending, cap)), d, ic, oc)
            dec.close()
            want = _model_seq(d, ic, oc)
            assert got == want, (name, len(d), chunks, next((i, g, w) for i, (g, w) in enumerate(zip(got + [None], want + [None])) if g != w))


@pytest.mark.skipif(not ref.available(), reason="libbrotlidec not available")
def test_model_against_libbrotlidec_where_the_ring_wraps():
    """streams longer than their ring buffer (1 KiB, 16 KiB and window-sized rings): the stops at a full ring, call by call"""
    for name in WRAP:
        data = _data(name)
        for ic, oc in [(65536, 65536), (4096, 517), (65536, 1000), (3, 70000)]:
            dec = ref.StreamDecoder()
            got = sm.run_schedule(lambda pending, cap: (lambda r: (r[0], r[1], len(r[2])))(dec.step(pending, cap)), data, ic, oc)
            dec.close()
            assert got == _model_seq(data, ic, oc), (name, ic, oc)


def test_model_totals():
    """whatever the schedule: the produced bytes add up to the decoded size, the consumed bytes to the stream's length"""
    for name in SMALL + BIG[:1]:
        data = _data(name)
        info, _, _ = sm.trace(data)
        for ic, oc in [(65536, 65536), (7, 5), (65536, 1)] if len(data) < 1000 else [(65536, 65536), (4096, 517)]:
            seq = _model_seq(data, ic, oc)
            assert seq[-1][0] == sm.RESULT_SUCCESS
            assert sum(s[2] for s in seq) == info.produced and sum(s[1] for s in seq) == info.consumed


def _product_seq(pkg, data, ic, oc):
    st = pkg.DecoderState(large_window=True)
    outs = []

    def step(pending, cap):
        r = st.decompress_stream(pending, cap)
        outs.append(r[2])
        return r[0], r[1], len(r[2])
    seq = sm.run_schedule(step, data, ic, oc, drain=True)
    st.close()
    return seq, b"".join(outs)


@pytest.mark.gpu
@pytest.mark.parametrize("chunks", CHUNKINGS)
def test_streaming_calls_against_the_model(pkg, chunks):
    """The product, call by call, against the reference's contract (src/decode.rs:2779-2911), on streams that fit their ring
    buffer (every one here), whole and cut short: every call returns the (result, consumed, produced) the model returns --
    including the calls whose output has no room: what fits is written, the call's input is taken, the answer is
    NEEDS_MORE_INPUT (decode.rs:2835-2846), and the rest goes out with the calls that follow."""
    ic, oc = chunks
    names = SMALL + (BIG if ic > 1 and oc > 1 else BIG[:1] if (ic, oc) != (1, 1) else [])
    for name in names:
        data = _data(name)
        for d in (data, data[: max(1, len(data) * 2 // 3)]):
            got, out = _product_seq(pkg, d, ic, oc)
            m = sm.ReferenceStream(d)
            want = sm.run_schedule(lambda pending, cap: m.call(len(pending), cap), d, ic, oc, drain=True)
            assert got == want, (name, len(d), chunks, next((i, g, w) for i, (g, w) in enumerate(zip(got + [None], want + [None])) if g != w))
            assert out == oracle.decode(d, 1 << 22, 1)[1]


@pytest.mark.gpu
def test_streaming_calls_where_the_ring_wraps(pkg):
    """Streams longer than their ring buffer.  The reference stops at every full ring and hands its unread input back; the
    product decodes flat and owes output only when the caller's buffer is full (include/brotli/decode.h).  What must agree:
    every byte, the final result, the totals, and the checkpoints at the end of the input; what the product may do
    differently: where inside the stream a NEEDS_MORE_OUTPUT falls and how much of a call's input it has consumed by then
    (never more than the call gave it, never input behind the end of the stream)."""
    import oracle_lib as oracle
    for name in WRAP:
        data = _data(name)
        info, exp = oracle.decode(data, 1 << 24, 1)
        for ic, oc in [(65536, 65536), (4096, 517), (65536, 1000)]:
            got, out = _product_seq(pkg, data, ic, oc)
            assert got[-1][0] == sm.RESULT_SUCCESS and out == exp, (name, ic, oc, got[-1])
            assert sum(g[2] for g in got) == info.produced and sum(g[1] for g in got) == info.consumed, (name, ic, oc)
            assert all(g[2] <= oc for g in got)
