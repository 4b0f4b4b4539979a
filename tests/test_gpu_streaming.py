"""The streaming and one-shot entry points of the C ABI on a real MI355X (reference: src/ffi/mod.rs,
src/bin/integration_tests.rs, src/bin/tests.rs, src/bin/error_handling_tests.rs, src/reader.rs, src/writer.rs)."""
import hashlib
import io
import json
import os
import subprocess
import sys

import pytest

import oracle_lib as oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")
MANIFEST = {e["name"]: e for e in json.load(open(os.path.join(GOLD, "manifest.json")))}


def _data(name):
    return open(os.path.join(GOLD, "testdata", name), "rb").read()


def _stream_decode(pkg, data, in_chunk, out_chunk, large_window=False, max_calls=600000, counted=None):
    """the loop of the reference's decompress_internal (src/bin/integration_tests.rs:122-216)"""
    st = pkg.DecoderState(large_window=large_window)
    out = bytearray()
    pos, calls = 0, 0
    pending = b""
    result = pkg.RESULT_NEEDS_MORE_INPUT
    while True:
        if not pending and result == pkg.RESULT_NEEDS_MORE_INPUT:
            if pos >= len(data):
                if calls and got:   # (a cut stream: what earlier calls had no room for still goes out, decode.rs:2835-2846)
                    result, used, got = st.decompress_stream(b"", out_chunk)
                    out += got; calls += 1
                    continue
                break
            pending = data[pos:pos + in_chunk]
            pos += len(pending)
        result, used, got = st.decompress_stream(pending, out_chunk)
        pending = pending[used:]
        out += got
        calls += 1
        assert calls < max_calls
        if result in (pkg.RESULT_SUCCESS, pkg.RESULT_ERROR):
            break
    code = st.error_code()
    finished = st.is_finished()
    if counted is not None:
        counted["device_commands"] = st.device_commands(); counted["calls"] = calls
    st.close()
    return result, code, bytes(out), finished, pos - len(pending)


BIG = ["alice29.txt.compressed", "metablock_reset.compressed", "mapsdatazrh.compressed", "random_then_unicode.compressed"]
SMALL = ["10x10y.compressed", "64x.compressed", "ukkonooa.compressed", "monkey.compressed", "x.compressed.03", "xyzzy.compressed",
         "quickfox.compressed", "ends_with_truncated_dictionary.compressed", "empty.compressed", "empty.compressed.16", "random1024.br",
         "zeros.compressed", "fuzz502.compressed"]


@pytest.mark.parametrize("name", BIG)
@pytest.mark.parametrize("chunks", [(65536, 65536), (4096, 517), (517, 65536), (12345, 181)])
def test_streaming_big_fixtures(pkg, name, chunks):
    data = _data(name)
    result, code, out, finished, consumed = _stream_decode(pkg, data, *chunks)
    assert (result, code) == (1, 1) and finished
    assert hashlib.sha256(out).hexdigest() == MANIFEST[name]["sha256"]
    assert consumed == len(data)


@pytest.mark.parametrize("name", SMALL)
@pytest.mark.parametrize("chunks", [(65536, 65536), (1, 65536), (65536, 1), (1, 1), (3, 3), (12, 1)])
def test_streaming_small_fixtures_adversarial_chunks(pkg, name, chunks):
    """the buffer-size pairs of src/bin/integration_tests.rs:528-1006"""
    data = _data(name)
    result, code, out, finished, consumed = _stream_decode(pkg, data, *chunks)
    assert (result, code) == (1, 1) and finished
    assert hashlib.sha256(out).hexdigest() == MANIFEST[name]["sha256"]


def test_streaming_errors_latch_and_trailing_input(pkg):
    # corrupt stream: same code as the one-shot path; later calls keep failing (decode.rs:2796-2798)
    bad = bytes.fromhex("1b3000e08dd4592d39ffb5024810952a9aea420e51a416b9cbf5f85c64b92fc96a3fb1dca8e03507")
    st = pkg.DecoderState()
    r, used, out = st.decompress_stream(bad, 4096)
    assert (r, st.error_code(), st.error_string()) == (0, -8, "ERROR_FORMAT_CONTEXT_MAP_REPEAT")
    assert st.decompress_stream(b"", 16)[0] == 0
    st.close()
    # "hello\n" followed by garbage: 10 of 18 bytes consumed (src/reader.rs:359, src/writer.rs:385)
    hello = pkg.load_library()  # noqa: F841
    import libbrotli_ref as ref
    if ref.encoder_available():
        comp = ref.encode(b"hello\n", 5, 22)
        st = pkg.DecoderState()
        r, used, out = st.decompress_stream(comp + b"\x01\x02\x03garbage", 64)
        assert (r, out, used) == (1, b"hello\n", len(comp))
        assert st.is_finished() and st.is_used()
        st.close()
    # large-window streams are rejected by instances until the parameter is set (ffi/mod.rs:127, 743-750)
    lw = _data("rnd_chunk.br")
    st = pkg.DecoderState(large_window=False)
    assert st.decompress_stream(lw, 16)[0] == 0 and st.error_code() == -13
    st.close()
    # parameters can only be set before the first byte
    L = pkg.load_library()
    st = pkg.DecoderState()
    st.decompress_stream(b"\x0b", 16)
    assert L.BrotliDecoderSetParameter(st._h, pkg.PARAM_LARGE_WINDOW, 1) == 0
    st.close()


def test_one_byte_streams_through_the_reader(pkg):
    """src/bin/tests.rs:76-98: exactly these one-byte inputs are complete streams"""
    ok = []
    for b in range(256):
        try:
            if pkg.Decompressor(io.BytesIO(bytes([b])), 8, large_window=False).read() == b"":
                ok.append(b)
        except ValueError:
            pass
    assert ok == [6, 26, 51, 53, 55, 57, 59, 61, 63]


def test_reader_and_writer_adapters(pkg):
    """src/bin/integration_tests.rs:294-415: 178-byte reads with 181/121/8192-byte buffers; 517-byte writer buffer"""
    data, want = _data("alice29.txt.compressed"), MANIFEST["alice29.txt.compressed"]["sha256"]
    for bufsize in (181, 8192):
        r = pkg.Decompressor(io.BytesIO(data), bufsize)
        got = bytearray()
        while True:
            c = r.read(178 if bufsize == 8192 else 4000)
            if not c:
                break
            got += c
        assert hashlib.sha256(got).hexdigest() == want
    sink = io.BytesIO()
    w = pkg.DecompressorWriter(sink, 517)
    for i in range(0, len(data), 512):
        assert w.write(data[i:i + 512]) == len(data[i:i + 512])
    w.close()
    assert hashlib.sha256(sink.getvalue()).hexdigest() == want
    # truncated input: into_inner()/close() is an error
    w = pkg.DecompressorWriter(io.BytesIO(), 517)
    w.write(data[:1000])
    with pytest.raises(ValueError):
        w.close()
    with pytest.raises(ValueError):
        pkg.Decompressor(io.BytesIO(data[:1000]), 4096).read()


def test_one_shot_matches_oracle_including_small_outputs(pkg):
    """BrotliDecoderDecompressWithReturnInfo == the reference's brotli_decode (lib.rs:447-468), also when the output
    buffer is too small: what is reported then depends on the stream up to the next ring-buffer flush point"""
    cases = []
    for name in ("alice29.txt.compressed", "zeros.compressed", "metablock_reset.compressed", "64x.compressed", "borked.compressed"):
        d = _data(name)
        size = MANIFEST[name].get("size", 64)
        for cap in sorted({0, 1, 10, size // 2, max(size - 1, 0), size, size + 100}):
            cases.append((name, d, cap))
    # corrupt tail after a too-small buffer: the error wins over NEEDS_MORE_OUTPUT iff it comes before the flush point
    d = bytearray(_data("alice29.txt.compressed"))
    d[40000] ^= 0x10
    cases.append(("alice29 corrupt", bytes(d), 1000))
    for name, d, cap in cases:
        info, out = pkg.brotli_decode(d, cap)
        oinfo, oout = oracle.decode(d, cap, oracle.FLAG_LARGE_WINDOW)
        assert (info.result, info.code, info.decoded_size, out) == (oinfo.result, oinfo.error_code, oinfo.decoded_size, oout), (name, cap)
        assert info.error.decode() == pkg.load_library().BrotliDecoderErrorString(info.code).decode()


def _native(target):
    path = os.path.join(ROOT, "tests", "native", target)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "native")])
    return path


def test_c_acceptance_program(pkg):
    exe = _native("abi_acceptance")
    data = _data("alice29.txt.compressed")
    p = subprocess.run([exe, "--stream"], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    assert hashlib.sha256(p.stdout).hexdigest() == MANIFEST["alice29.txt.compressed"]["sha256"]
    p = subprocess.run([exe, "--stream"], input=data[:20000], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 2 and b"Unexpected EOF" in p.stderr


def test_cpp_reader_writer_adapters(pkg, tmp_path):
    exe = _native("wrappers_test")
    for name, read_size, buf in (("alice29.txt.compressed", 178, 181), ("alice29.txt.compressed", 4096, 8192), ("monkey.compressed", 1, 1),
                                 ("quickfox_repeated.compressed", 3, 121)):
        comp = os.path.join(GOLD, "testdata", name)
        raw = tmp_path / (name + ".raw")
        info, out = oracle.decode(_data(name), MANIFEST[name]["size"] + 16)
        raw.write_bytes(out)
        p = subprocess.run([exe, comp, str(raw), str(read_size), str(buf)], stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, (name, p.stderr.decode())


def test_streaming_parameterised_corpus_random_chunks(pkg):
    """every fifth stream of the parameterised corpus (window sizes, NPOSTFIX/NDIRECT, flushes, metadata blocks, large
    windows) through BrotliDecoderDecompressStream with random input and output chunk sizes"""
    import random
    import param_corpus
    streams = param_corpus.corpus()[::5]
    if not streams:
        pytest.fail("libbrotlienc is not available: the GPU suite needs the encoder of the image for its synthetic streams (a skip here would let a third of the suite go green unrun)")
    rnd = random.Random(5)
    for label, comp, raw in streams:
        chunks = (rnd.choice([1, 7, 64, 517, 4096, 65536]), rnd.choice([1, 13, 181, 4096, 65536]))
        if len(raw) > 30000 and chunks[1] < 100:
            chunks = (chunks[0], 4096)
        if len(comp) > 30000 and chunks[0] < 64:
            chunks = (517, chunks[1])
        result, code, out, finished, consumed = _stream_decode(pkg, comp, *chunks, large_window=label.startswith("large-"))
        assert (result, code, finished, consumed) == (1, 1, True, len(comp)), (label, chunks)
        assert out == raw, (label, chunks)


def test_command_line_tool(pkg, tmp_path):
    """tools/cli: the reference's brotli-decompressor binary (src/bin/brotli-decompressor.rs) over reader.hpp"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools", "cli")])
    exe = os.path.join(ROOT, "tools", "cli", "brotli-decompressor")
    out = tmp_path / "alice29.txt"
    subprocess.check_call([exe, os.path.join(GOLD, "testdata", "alice29.txt.compressed"), str(out)], timeout=300)
    assert hashlib.sha256(out.read_bytes()).hexdigest() == MANIFEST["alice29.txt.compressed"]["sha256"]
    p = subprocess.run([exe], input=_data("alice29.txt.compressed")[:1000], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 1 and b"Unexpected EOF" in p.stderr


def test_less_common_entry_points(pkg):
    """BrotliDecoderDecompressPrealloc (ffi/mod.rs:179), BrotliDecoderDecompressStreaming (:467), HasMoreOutput /
    TakeOutput (:546-565) on the reference's alice29 fixture"""
    import ctypes
    L = pkg.load_library()
    data, size, sha = _data("alice29.txt.compressed"), MANIFEST["alice29.txt.compressed"]["size"], MANIFEST["alice29.txt.compressed"]["sha256"]
    # Prealloc: same result as the one-shot function; the scratch arrays are only validated
    L.BrotliDecoderDecompressPrealloc.restype = pkg.ReturnInfo
    L.BrotliDecoderDecompressPrealloc.argtypes = [ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                  ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    out = ctypes.create_string_buffer(size + 16)
    s8, s32, shc = ctypes.create_string_buffer(1 << 20), (ctypes.c_uint32 * 4096)(), ctypes.create_string_buffer(4 * 65536)
    info = L.BrotliDecoderDecompressPrealloc(len(data), data, size + 16, out, len(s8), s8, 4096, s32, 65536, shc)
    assert (info.result, info.code, info.decoded_size) == (1, 1, size)
    assert hashlib.sha256(out.raw[:size]).hexdigest() == sha
    # the scratch slices are accounted the way the reference's allocators consume them (src/lib.rs:374-401): a request
    # they cannot serve is ERROR_UNREACHABLE with nothing decoded (ffi/mod.rs:686-713, prealloc_catches_scratch_exhaustion)
    info = L.BrotliDecoderDecompressPrealloc(0, None, 0, None, 0, None, 0, None, 0, None)
    assert (info.result, info.code, info.decoded_size) == (0, -31, 0)
    # alice29's ring buffer is 256 KiB (decode.rs:1843-1850): 64 KiB of u8 scratch cannot hold it
    info = L.BrotliDecoderDecompressPrealloc(len(data), data, size + 16, out, 1 << 16, s8, 4096, s32, 65536, shc)
    assert (info.result, info.code, info.decoded_size) == (0, -31, 0)
    # the context-map code, block-type and block-length trees alone are 7 x 1080 HuffmanCode cells (state.rs:395, decode.rs:2958-2969)
    info = L.BrotliDecoderDecompressPrealloc(len(data), data, size + 16, out, len(s8), s8, 4096, s32, 7 * 1080, shc)
    assert (info.result, info.code, info.decoded_size) == (0, -31, 0)
    info = L.BrotliDecoderDecompressPrealloc(len(data), data, size + 16, out, len(s8), s8, 0, None, 65536, shc)
    assert (info.result, info.code, info.decoded_size) == (0, -31, 0)
    # DecompressStreaming: pointers by value, counters by reference
    L.BrotliDecoderCreateInstance.restype = ctypes.c_void_p
    L.BrotliDecoderCreateInstance.argtypes = [ctypes.c_void_p] * 3
    L.BrotliDecoderDestroyInstance.argtypes = [ctypes.c_void_p]
    L.BrotliDecoderDecompressStreaming.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
    st = L.BrotliDecoderCreateInstance(None, None, None)
    got, pos = bytearray(), 0
    obuf = ctypes.create_string_buffer(8192)
    while True:
        chunk = data[pos:pos + 3000]
        ain, aout = ctypes.c_size_t(len(chunk)), ctypes.c_size_t(len(obuf))
        r = L.BrotliDecoderDecompressStreaming(st, ctypes.byref(ain), chunk, ctypes.byref(aout), obuf)
        pos += len(chunk) - ain.value
        got += obuf.raw[:len(obuf) - aout.value]
        assert r != 0
        if r == 1:
            break
    L.BrotliDecoderDestroyInstance(st)
    assert hashlib.sha256(got).hexdigest() == sha and pos == len(data)
    # TakeOutput: feed everything with no output room, then take what the decoder holds (ffi/mod.rs:552-565)
    L.BrotliDecoderDecompressStream.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_char_p),
                                                ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    L.BrotliDecoderHasMoreOutput.argtypes = [ctypes.c_void_p]
    L.BrotliDecoderTakeOutput.restype = ctypes.c_void_p
    L.BrotliDecoderTakeOutput.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    st = L.BrotliDecoderCreateInstance(None, None, None)
    ain, nin = ctypes.c_size_t(len(data)), ctypes.c_char_p(data)
    aout, nout = ctypes.c_size_t(0), ctypes.c_void_p(None)
    r = L.BrotliDecoderDecompressStream(st, ctypes.byref(ain), ctypes.byref(nin), ctypes.byref(aout), ctypes.byref(nout), None)
    assert r == 3 and L.BrotliDecoderHasMoreOutput(st)
    got = bytearray()
    for _ in range(10000):
        if not L.BrotliDecoderHasMoreOutput(st):
            r = L.BrotliDecoderDecompressStream(st, ctypes.byref(ain), ctypes.byref(nin), ctypes.byref(aout), ctypes.byref(nout), None)
            if r == 1 and not L.BrotliDecoderHasMoreOutput(st):
                break
            continue
        n = ctypes.c_size_t(4096)  # at most this much; 0 would mean "all there is"
        p = L.BrotliDecoderTakeOutput(st, ctypes.byref(n))
        got += ctypes.string_at(p, n.value)
    L.BrotliDecoderDestroyInstance(st)
    # (the package's own prototype back: the library object is shared by every test of the session)
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    L.BrotliDecoderDecompressStream.argtypes = [vp, ctypes.POINTER(sz), ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(vp), ctypes.POINTER(sz)]
    assert hashlib.sha256(got).hexdigest() == sha


def test_output_that_does_not_fit(pkg):
    """decode.rs:2835-2846: a call whose input ends inside the stream writes what fits, takes all of its input and answers
    NEEDS_MORE_INPUT; the rest goes out with later calls.  At the end of the stream the output is OWED (decode.rs:3382-3397):
    NEEDS_MORE_OUTPUT, and bytes offered behind the end of the stream stay with the caller."""
    import ctypes
    L = pkg.load_library()
    data, sha = _data("alice29.txt.compressed"), MANIFEST["alice29.txt.compressed"]["sha256"]
    L.BrotliDecoderCreateInstance.restype = ctypes.c_void_p
    L.BrotliDecoderCreateInstance.argtypes = [ctypes.c_void_p] * 3
    L.BrotliDecoderDestroyInstance.argtypes = [ctypes.c_void_p]
    L.BrotliDecoderDecompressStreaming.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
    st = L.BrotliDecoderCreateInstance(None, None, None)
    third = len(data) // 3
    obuf = ctypes.create_string_buffer(1000)
    got = bytearray()
    for piece in (data[:third], data[third:2 * third]):
        ain, aout = ctypes.c_size_t(len(piece)), ctypes.c_size_t(1000)
        r = L.BrotliDecoderDecompressStreaming(st, ctypes.byref(ain), piece, ctypes.byref(aout), obuf)
        assert r == 2 and ain.value == 0 and aout.value == 0  # the piece is taken, 1000 bytes of what it held are written
        got += obuf.raw[:1000]
    ain, aout = ctypes.c_size_t(0), ctypes.c_size_t(1000)   # no input: what there is goes on coming
    r = L.BrotliDecoderDecompressStreaming(st, ctypes.byref(ain), b"", ctypes.byref(aout), obuf)
    assert r == 2 and aout.value == 0
    got += obuf.raw[:1000]
    rest = data[2 * third:] + b"trailing"
    ain, aout = ctypes.c_size_t(len(rest)), ctypes.c_size_t(1000)
    r = L.BrotliDecoderDecompressStreaming(st, ctypes.byref(ain), rest, ctypes.byref(aout), obuf)
    assert r == 3 and ain.value == len(b"trailing") and aout.value == 0  # the stream's end: the output is owed now
    got += obuf.raw[:1000]
    rest = rest[len(rest) - ain.value:]
    for _ in range(3):  # while it is owed nothing more is consumed
        ain, aout = ctypes.c_size_t(len(rest)), ctypes.c_size_t(1000)
        r = L.BrotliDecoderDecompressStreaming(st, ctypes.byref(ain), rest, ctypes.byref(aout), obuf)
        assert r == 3 and ain.value == len(rest) and aout.value == 0
        got += obuf.raw[:1000]
    big = ctypes.create_string_buffer(1 << 20)
    ain, aout = ctypes.c_size_t(len(rest)), ctypes.c_size_t(len(big))
    r = L.BrotliDecoderDecompressStreaming(st, ctypes.byref(ain), rest, ctypes.byref(aout), big)
    got += big.raw[:len(big) - aout.value]
    assert r == 1 and ain.value == len(rest)
    L.BrotliDecoderDestroyInstance(st)
    assert hashlib.sha256(got).hexdigest() == sha


def test_streaming_memory_stays_bounded(pkg):
    """A 48 MiB stream of many metablocks through one instance in 512 KiB pieces: what the instance keeps on the device is
    the window and the metablock in flight, not the stream (src/state.rs: the reference keeps a ring buffer of one window)."""
    import ctypes
    import sys
    sys.path.insert(0, ROOT)
    import workloads as w
    if not w.encoder_available():
        pytest.fail("libbrotlienc is not available: the GPU suite needs the encoder of the image for its synthetic streams (a skip here would let a third of the suite go green unrun)")
    raw = w.long_backref_stream(777, 48 << 20)
    c = w.brotli_compress(raw, 5, 22)
    L = pkg.load_library()
    L.brotli_amd_debug_stream_device_bytes.restype = ctypes.c_size_t
    L.brotli_amd_debug_stream_device_bytes.argtypes = [ctypes.c_void_p]
    st = pkg.DecoderState(large_window=False)
    h = hashlib.sha256()
    total, peak = 0, 0
    for pos in range(0, len(c), 512 << 10):
        pending = c[pos:pos + (512 << 10)]
        while True:
            result, used, got = st.decompress_stream(pending, 1 << 20)
            pending = pending[used:]
            h.update(got); total += len(got)
            peak = max(peak, L.brotli_amd_debug_stream_device_bytes(st._h))
            assert result != pkg.RESULT_ERROR
            if result != pkg.RESULT_NEEDS_MORE_OUTPUT and not pending:
                break
    assert result == pkg.RESULT_SUCCESS and st.is_finished()
    st.close()
    assert total == len(raw) and h.digest() == hashlib.sha256(raw).digest()
    assert peak < (24 << 20), peak  # (the whole stream would be 48 MiB of output and 6 MiB of input)


def test_streaming_goes_on_inside_a_metablock(pkg):
    """A stream fed in small pieces goes on from the command boundary the call before got to (BrotliAmdResume: mid_*),
    not from the metablock's first command: streams with block switches in all three categories (the emitter's vectors,
    mapsdatazrh), context-modelled text, long back-references, each in pieces of 1 .. 4096 bytes, whole and with a
    flipped bit (the piece-wise result must be the one-piece result); and the cost: a 4 MiB stream in 4 KiB pieces takes
    a hundred launches of a few KiB each, not a hundred of up to 4 MiB."""
    import json
    import random
    import time
    sys.path.insert(0, ROOT)
    import workloads as w
    rnd = random.Random(99)
    cases = []
    d = os.path.join(ROOT, "tests", "golden", "emitter")
    for e in json.load(open(os.path.join(d, "manifest.json"))):
        cases.append((e["file"], open(os.path.join(d, e["file"]), "rb").read()))
    for name in ("mapsdatazrh.compressed", "alice29.txt.compressed", "metablock_reset.compressed", "random_then_unicode.compressed"):
        cases.append((name, _data(name)))
    if w.encoder_available():
        cases.append(("long_backref 1 MiB", w.brotli_compress(w.long_backref_stream(31, 1 << 20), 5, 22)))
        cases.append(("long_backref 256 KiB q9 w18", w.brotli_compress(w.long_backref_stream(32, 256 << 10), 9, 18)))
    for label, comp in cases:
        variants = [comp]
        bad = bytearray(comp); bad[rnd.randrange(len(bad) // 2, len(bad))] ^= 1 << rnd.randrange(8)
        variants.append(bytes(bad))
        for v in variants:
            want = _stream_decode(pkg, v, len(v), 1 << 24)
            for piece in (rnd.choice([1, 2, 3]) if len(v) < 3000 else rnd.choice([61, 97]), rnd.choice([256, 517, 1000]), 4096):
                got = _stream_decode(pkg, v, piece, 1 << 16)
                if want[0] == 1:
                    assert got[:4] == want[:4], (label, piece, got[:2], want[:2], len(got[2]), len(want[2]))
                else:  # (every call that ends for want of input delivers what has been decoded: more than one call does)
                    assert got[:2] == want[:2] and got[2][:len(want[2])] == want[2], (label, piece, got[:2], want[:2], len(got[2]), len(want[2]))
    if w.encoder_available():
        raw = w.long_backref_stream(4242, 4 << 20)
        comp = w.brotli_compress(raw, 5, 22)
        # the cost of a call must not grow with how far into the metablock the stream is: a call is a launch from the last command
        # boundary reached, so over all of a stream's calls the device decodes little more than the stream's own commands (every
        # call from the metablock's first command would be hundreds of times that).  A counter, not a clock.
        counted = {}
        result, code, out, finished, _ = _stream_decode(pkg, comp, 4096, 1 << 20, counted=counted)
        assert (result, code, finished) == (1, 1, True) and out == raw
        info, _ = oracle.decode(comp, len(raw) + 64, 1)
        assert info.num_commands <= counted["device_commands"] <= 2 * info.num_commands + 64 * counted["calls"], (counted, info.num_commands)
