"""The device's prefix-code table builder on its own, and BASELINE config 2 at full size with every output hashed
(needs a real MI355X).

The builder (build_tree in csrc/brotli_kernels.hip) restates src/huffman/mod.rs:273-386 with a table layout of its own, so
the known-answer tables of src/huffman/tests.rs cannot be compared entry by entry.  What can be: the symbol and the length
every possible fifteen-bit input decodes to (reference: DecodeSymbol, src/decode.rs:378-398).  BrotliAmdDebugBuildTree
(include/brotli/batch.h) runs the device builder alone and returns exactly that."""
import ctypes
import hashlib
import json
import os
import random

import pytest

import oracle_lib as oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu

TABLES = [t for t in json.load(open(os.path.join(ROOT, "tests", "golden", "huffman_tables.json"))) if t["kind"] == "full"]


def _device_decode(pkg, lengths):
    L = pkg.load_library()
    L.BrotliAmdDebugBuildTree.restype = ctypes.c_int
    L.BrotliAmdDebugBuildTree.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    dec = (ctypes.c_uint16 * 32768)()
    n = ctypes.c_uint32(0)
    rc = L.BrotliAmdDebugBuildTree(bytes(lengths), len(lengths), dec, ctypes.byref(n))
    assert rc == 0, pkg.last_error()
    return [(d >> 4, d & 15) for d in dec], n.value


def _reference_decode(table, v):
    """DecodeSymbol (src/decode.rs:378-398) over a table in the reference's layout: entries [bits, value], root of 8 bits"""
    bits, value = table[v & 0xFF]
    if bits > 8:
        bits2, value = table[(v & 0xFF) + value + ((v >> 8) & ((1 << (bits - 8)) - 1))]
        return value, 8 + bits2
    return value, bits


@pytest.mark.parametrize("rec", TABLES, ids=lambda r: r["name"])
def test_device_table_builder_known_answers(pkg, rec):
    """src/huffman/tests.rs: the code lengths of each known-answer table through the DEVICE builder; every fifteen-bit value
    must decode to the symbol and length the reference's table gives, and the table must be as large as the reference's"""
    assert rec["root_bits"] == 8
    got, entries = _device_decode(pkg, rec["code_lengths"])
    want = [_reference_decode(rec["table"], v) for v in range(32768)]
    bad = [(v, g, w) for v, (g, w) in enumerate(zip(got, want)) if g != w]
    assert not bad, (len(bad), bad[:8])
    assert entries == rec["size"]


def _random_complete_code(rnd, alphabet, used, max_len=15):
    """code lengths of a complete prefix code over `used` of `alphabet` symbols: leaves split at random"""
    leaves = [0]
    while len(leaves) < used:
        cand = [i for i, d in enumerate(leaves) if d < max_len]
        i = rnd.choice(cand)
        d = leaves.pop(i)
        leaves += [d + 1, d + 1]
    lengths = [0] * alphabet
    for sym, d in zip(rnd.sample(range(alphabet), used), leaves):
        lengths[sym] = d
    return lengths


def _skewed_code(alphabet, max_len=15):
    """lengths 1, 2, 3 .. max_len, max_len: the deepest code there is (second-level tables of every depth)"""
    lengths = [0] * alphabet
    for k in range(max_len):
        lengths[(k * 37) % alphabet] = min(k + 1, max_len)
    lengths[(max_len * 37) % alphabet] = max_len
    return lengths


def test_device_table_builder_against_the_oracle(pkg):
    """alphabets the fixtures of src/huffman/tests.rs do not have (704 command symbols, 520 and 1128 distance symbols, small ones),
    codes of every depth: the device builder against the oracle's (which the known-answer tables pin entry by entry, test_oracle.py)"""
    O = oracle.lib()
    rnd = random.Random(20260930)
    cases = [_skewed_code(a) for a in (256, 704, 1128)]
    for alphabet in (18, 26, 64, 256, 258, 520, 704, 1128):
        for used in sorted({2, 3, min(alphabet, 17), min(alphabet, 200), alphabet}):
            for _ in range(3):
                cases.append(_random_complete_code(rnd, alphabet, used))
    vals, bits = (ctypes.c_uint16 * 4096)(), (ctypes.c_uint8 * 4096)()
    for lengths in cases:
        cl = (ctypes.c_uint8 * len(lengths))(*lengths)
        n = O.brotli_oracle_build_huffman(cl, len(lengths), 8, vals, bits)
        table = [[bits[i], vals[i]] for i in range(n)]
        got, entries = _device_decode(pkg, lengths)
        bad = [(v, got[v], _reference_decode(table, v)) for v in range(32768) if got[v] != _reference_decode(table, v)]
        assert not bad, (len(lengths), sum(1 for x in lengths if x), len(bad), bad[:8])
        assert entries == n, (len(lengths), entries, n)


def test_1024_copies_of_alice29_every_output_hashed(pkg):
    """BASELINE config 2 at full size: 1024 separate copies of the reference's alice29 fixture in one batch; every stream's
    status words against the oracle's and every one of the 1024 outputs by SHA-256 (bench.py checks the same before it times)"""
    gold = os.path.join(ROOT, "tests", "golden", "testdata")
    alice = open(os.path.join(gold, "alice29.txt.compressed"), "rb").read()
    info, exp = oracle.decode(alice, 200000, 1)
    manifest = {e["name"]: e for e in json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))}["alice29.txt.compressed"]
    assert info.result == 1 and hashlib.sha256(exp).hexdigest() == manifest["sha256"] and len(exp) == manifest["size"]   # (the reference's own alice29.txt)
    want = hashlib.sha256(exp).hexdigest()
    b = pkg.Batch(1024)
    res, outs = b.decode_host([alice] * 1024, [200000] * 1024, 1)
    b.close()
    bad = [(i, r.result, r.error_code, r.decoded_size, r.consumed, r.num_commands) for i, r in enumerate(res)
           if (r.result, r.error_code, r.decoded_size, r.consumed, r.num_commands, r.num_metablocks) != (1, info.error_code, info.decoded_size, info.consumed, info.num_commands, info.num_metablocks)]
    assert not bad, (len(bad), bad[:5])
    wrong = [i for i, o in enumerate(outs) if hashlib.sha256(o).hexdigest() != want]
    assert not wrong, (len(wrong), wrong[:10])
