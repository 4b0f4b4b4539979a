"""The CPU oracle against the reference's own golden data (runs without a GPU)."""
import ctypes
import hashlib
import json
import os
import random

import pytest

import libbrotli_ref as ref
import oracle_lib as oracle
from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))
VECTORS = json.load(open(os.path.join(GOLD, "inline_vectors.json")))
TABLES = json.load(open(os.path.join(GOLD, "huffman_tables.json")))


def _data(name):
    return open(os.path.join(GOLD, "testdata", name), "rb").read()


@pytest.mark.parametrize("entry", [e for e in MANIFEST if e["name"] != "rnd_chunk.br"], ids=lambda e: e["name"])
def test_fixture(entry):
    """testdata/X.compressed* -> X, byte for byte (src/bin/integration_tests.rs:528-1006)"""
    data = _data(entry["name"])
    info, out = oracle.decode(data, entry.get("size", 1 << 16) + 16)
    if entry.get("must_fail"):
        assert info.result == oracle.RESULT_ERROR
        return
    assert (info.result, info.error_code) == (1, 1)
    assert len(out) == entry["size"]
    assert hashlib.sha256(out).hexdigest() == entry["sha256"]
    assert info.consumed == entry["csize"]


def test_large_window_fixture():
    """rnd_chunk.br (src/bin/integration_tests.rs:997-1006)"""
    edges = json.load(open(os.path.join(GOLD, "rnd_chunk_edges.json")))
    info, out = oracle.decode(_data("rnd_chunk.br"), edges["size"] + 16)
    assert (info.result, info.decoded_size) == (1, edges["size"])
    pre, post = bytes.fromhex(edges["prefix_hex"]), bytes.fromhex(edges["postfix_hex"])
    assert out[:len(pre)] == pre and out[-len(post):] == post
    assert out[len(pre):len(pre) + edges["zero_count"]].count(0) == edges["zero_count"]
    info, _ = oracle.decode(_data("rnd_chunk.br"), 1 << 16, flags=0)  # FFI instances: large window off
    assert (info.result, info.error_code) == (0, -13)


@pytest.mark.parametrize("vec", VECTORS[:18], ids=lambda v: v["name"])
def test_inline_vector(vec):
    data = bytes.fromhex(vec["input_hex"])
    info, out = oracle.decode(data, 1 << 18)
    if vec.get("result") == 1:
        assert info.result == 1
    elif vec.get("result") == 0:
        assert info.result != 1
    if "error_code" in vec:
        assert info.error_code == vec["error_code"]
    if "output_hex" in vec:
        assert out.hex() == vec["output_hex"]
    if "output_sha256" in vec:
        assert hashlib.sha256(out).hexdigest() == vec["output_sha256"]
    if vec.get("consumed_all"):
        assert info.consumed == len(data)


def test_one_byte_streams():
    """src/bin/tests.rs:76-98: exactly {6,26,51,53,55,57,59,61,63} are complete streams"""
    ok = []
    for vec in VECTORS:
        if vec["name"].startswith("one_byte_"):
            info, _ = oracle.decode(bytes.fromhex(vec["input_hex"]), 64)
            assert (info.result == 1) == vec["result_is_success"], vec["name"]
            if info.result == 1:
                ok.append(int(vec["input_hex"], 16))
    assert ok == [6, 26, 51, 53, 55, 57, 59, 61, 63]


@pytest.mark.parametrize("rec", TABLES, ids=lambda r: r["name"])
def test_huffman_known_answer_tables(rec):
    """src/huffman/tests.rs: every entry of every table"""
    L = oracle.lib()
    vals, bits = (ctypes.c_uint16 * 4096)(), (ctypes.c_uint8 * 4096)()
    if rec["kind"] == "full":
        cl = (ctypes.c_uint8 * len(rec["code_lengths"]))(*rec["code_lengths"])
        n = L.brotli_oracle_build_huffman(cl, len(rec["code_lengths"]), rec["root_bits"], vals, bits)
        assert n == rec["size"]
        assert [[bits[i], vals[i]] for i in range(n)] == rec["table"]
    elif rec["kind"] == "code_lengths":
        cl = (ctypes.c_uint8 * 18)(*rec["code_lengths"])
        L.brotli_oracle_build_code_lengths(cl, vals, bits)
        want = rec["table"] if len(rec["table"]) == 32 else rec["table"] * 32
        assert [[bits[i], vals[i]] for i in range(32)] == want
    else:
        syms = (ctypes.c_uint16 * 5)(*(rec["symbols"] + [0] * 5)[:5])
        n = L.brotli_oracle_build_simple(syms, rec["num_symbols"], 8, vals, bits)
        assert n == 256
        assert [[bits[i], vals[i]] for i in range(n)] == rec["table"]


def test_bit_reader_values():
    """value semantics of src/bit_reader/mod.rs:323-338 on a fixed byte array: LSB-first fields"""
    L = oracle.lib()
    data = bytes(range(1, 33))
    big = int.from_bytes(data, "little")
    widths = [1, 3, 7, 16, 24, 5, 32, 0, 11, 15]
    n = (ctypes.c_uint32 * len(widths))(*widths)
    out = (ctypes.c_uint32 * len(widths))()
    L.brotli_oracle_read_bits(data, len(data), n, len(widths), out)
    pos = 0
    for w, got in zip(widths, out):
        assert got == (big >> pos) & ((1 << w) - 1)
        pos += w


def test_cmd_lut_formula():
    """kCmdLut regenerated from RFC 7932 section 5; spot values from src/prefix.rs:124-140, 5720-5755"""
    L = oracle.lib()
    out = (ctypes.c_int32 * 6)()

    def lut(c):
        L.brotli_oracle_cmd_lut(c, out)
        return list(out)
    assert lut(0) == [0, 0, 0, 0, 0, 2]
    assert lut(1) == [0, 0, 0, 1, 0, 3]
    assert lut(2)[3] == 2 and lut(3)[3] == 3
    assert lut(703) == [24, 24, -1, 3, 22594, 2118]
    assert lut(128)[2] == -1 and lut(127)[2] == 0


def test_transforms_against_libbrotli():
    """all 121 transforms on a few words equal libbrotlicommon's BrotliTransformDictionaryWord"""
    try:
        common = ctypes.CDLL("libbrotlicommon.so.1")
    except OSError:
        pytest.skip("libbrotlicommon not present")
    common.BrotliGetTransforms.restype = ctypes.c_void_p
    tr = common.BrotliGetTransforms()
    common.BrotliTransformDictionaryWord.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L = oracle.lib()
    words = [b"time", b"\xc3\xa9cole\xe4\xb8\xad", b"abcdefghijklmnopqrstuvwx", b"\xe4\xb8\xad\xe6\x96\x87\xc3\xa9", b"down"]
    for w in words:
        for t in range(121):
            a, b = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
            src = ctypes.create_string_buffer(w + b"\0" * 8)
            na = L.brotli_oracle_transform(a, src, len(w), t)
            nb = common.BrotliTransformDictionaryWord(b, src, len(w), tr, t)
            assert na == nb and a.raw[:max(na, 0)] == b.raw[:max(nb, 0)], (w, t)


@pytest.mark.skipif(not ref.available(), reason="libbrotlidec not present")
@pytest.mark.parametrize("seed", [11, 12])
def test_differential_vs_libbrotlidec(seed):
    """mutated fixtures: result, error code and delivered bytes equal libbrotlidec 1.0.9 (the C decoder the
    reference is a port of), except for the documented divergences of the reference."""
    rnd = random.Random(seed)
    names = [e["name"] for e in MANIFEST if e["csize"] < 200000 and e["name"] != "rnd_chunk.br"]
    base = {n: _data(n) for n in names}
    diffs = []
    for _ in range(1200):
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
        d = bytes(d)
        cap = rnd.choice([1 << 20, 1 << 20, 1 << 20, 5000, 100, 0])
        lw = rnd.random() < 0.7
        info, out = oracle.decode(d, cap, 1 if lw else 0)
        res, code, rout, used = ref.decode(d, cap, lw)
        if code == -14:
            code = -15  # the reference reports header padding as PADDING_2 (src/decode.rs:2990-2994)
        same = (info.result, info.error_code, out) == (res, code, rout)
        if same and info.result == 1:
            same = info.consumed == used
        # The reference sizes its ring buffer differently from the C decoder (src/decode.rs:1843-1850: twice the
        # metablock length plus slack, against the C decoder's tight power of two).  That only shows after a fatal
        # error: it decides how much had been flushed to the caller by then, and it moves commands that overshoot
        # MLEN between BLOCK_LENGTH_1 and BLOCK_LENGTH_2.  Everything else must agree exactly.
        if not same and info.result == 0 and res == 0:
            codes_ok = info.error_code == code or {info.error_code, code} == {-9, -10}
            prefix_ok = out == rout[:len(out)] or rout == out[:len(rout)]
            same = codes_ok and prefix_ok
        if not same:
            diffs.append((len(d), cap, lw, info.result, info.error_code, info.decoded_size, res, code, len(rout)))
    assert not diffs, repr(diffs[:5])


def test_parameterised_encoder_corpus():
    """NPOSTFIX/NDIRECT != 0, all context modes, many block types, all window sizes, flushes and metadata blocks
    (what the reference's fixtures leave unpinned): the oracle reproduces the raw data"""
    import param_corpus
    streams = param_corpus.corpus()
    if not streams:
        pytest.skip("libbrotlienc not available")
    seen_np = False
    for label, comp, raw in streams:
        info, out = oracle.decode(comp, len(raw) + 16, oracle.FLAG_LARGE_WINDOW)
        assert (info.result, info.error_code, info.decoded_size) == (1, 1, len(raw)), label
        assert out == raw, label
        assert info.consumed == len(comp), label
        # standard-window streams are also accepted without the large-window flag, large-window ones are not
        info2, _ = oracle.decode(comp, len(raw) + 16, 0)
        assert (info2.result == 1) == (not label.startswith("large-")), label
        seen_np = seen_np or label.startswith("np3-")
    assert seen_np


def test_committed_parameterised_corpus():
    """the edition of that corpus that travels with the repository (tests/golden/param_corpus, tools/make_param_corpus.py):
    the oracle reproduces raw data with the manifest's sizes and SHA-256, so the GPU tests that use the corpus never skip"""
    import param_corpus
    streams = param_corpus.committed()
    assert len(streams) >= 100
    labels = [l for l, _, _ in streams]
    assert any(l.startswith("np3-") for l in labels) and any(l.startswith("large-") for l in labels) and "metadata" in labels
    for label, comp, raw in streams:
        info, out = oracle.decode(comp, len(raw), oracle.FLAG_LARGE_WINDOW)  # exact fit
        assert (info.result, info.decoded_size, info.consumed) == (1, len(raw), len(comp)) and out == raw, label


def test_custom_dictionary_vectors():
    """BrotliState::new_with_custom_dictionary (src/state.rs:400-411): the reference's two known-answer tests
    (src/test.rs:438-520, tests/golden/custom_dict_vectors.json).  The GPU path has no custom dictionaries (not part of the
    C ABI); the restatement is pinned here so that it stays a restatement of the whole decode path."""
    import ctypes
    L = oracle.lib()
    L.brotli_oracle_decode_dict.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32,
                                            ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(oracle.OracleInfo)]
    for v in json.load(open(os.path.join(GOLD, "custom_dict_vectors.json"))):
        comp, d, exp = bytes.fromhex(v["compressed"]), bytes.fromhex(v["dictionary"]), bytes.fromhex(v["expected"])
        out = ctypes.create_string_buffer(len(exp) + 64)
        info = oracle.OracleInfo()
        L.brotli_oracle_decode_dict(comp, len(comp), out, len(out), 1, d, len(d), ctypes.byref(info))
        assert (info.result, info.decoded_size) == (1, len(exp)), v["name"]
        assert out.raw[:len(exp)] == exp, v["name"]
        # without the dictionary the same bytes do not decode to the same data
        info2, out2 = oracle.decode(comp, len(exp) + 64, 1)
        assert info2.result != 1 or out2 != exp, v["name"]


def test_emitter_vectors():
    """streams of the repository's own emitter (tools/brotli_emit.py, tests/golden/emitter/): 40 / 60 / 256 literal block
    types, every context mode with chosen context maps, NPOSTFIX / NDIRECT != 0, compressed / metadata / stored / empty
    metablocks in one stream -- what libbrotlienc cannot be steered to ("unpinned")"""
    d = os.path.join(GOLD, "emitter")
    man = json.load(open(os.path.join(d, "manifest.json")))
    assert max(e["max_block_types"] for e in man) == 256 and any(e["metablocks"] >= 6 for e in man)
    for e in man:
        comp = open(os.path.join(d, e["file"]), "rb").read()
        info, out = oracle.decode(comp, e["size"], 0)  # exact fit, standard windows
        assert (info.result, info.decoded_size, info.consumed) == (1, e["size"], len(comp)), e["label"]
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e["label"]
        assert (info.num_metablocks, info.num_commands, info.max_block_types) == (e["metablocks"], e["commands"], e["max_block_types"]), e["label"]
        if ref.available():
            r = ref.decode(comp, e["size"] + 16, False)
            assert r[0] == 1 and hashlib.sha256(r[2]).hexdigest() == e["sha256"], e["label"]


def test_emitter_is_deterministic():
    """tools/make_emitter_vectors.py writes the committed bytes again (the emitter uses no library and no clock)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_emitter_vectors as m
    man = {e["label"]: e for e in json.load(open(os.path.join(GOLD, "emitter", "manifest.json")))}
    for label, comp, raw in m.vectors():
        assert comp == open(os.path.join(GOLD, "emitter", man[label]["file"]), "rb").read(), label
        assert hashlib.sha256(raw).hexdigest() == man[label]["sha256"], label
