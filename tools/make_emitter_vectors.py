#!/usr/bin/env python3
"""tests/golden/emitter/: streams written by tools/brotli_emit.py that an encoder library cannot be steered to
 -- up to 256 literal block types, every literal context mode (MSB6 included) with chosen context
maps, block-type codes 0 / 1, NPOSTFIX / NDIRECT != 0, sequences of compressed, stored, metadata and empty metablocks.
Every stream is checked with Google's libbrotlidec (and the oracle) before it is written; the manifest carries size and
SHA-256 of the raw data."""
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import brotli_emit as E  # noqa: E402
import libbrotli_ref as ref  # noqa: E402
import oracle_lib as oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "emitter")


def text(rnd, n, alphabet):
    words = ["".join(rnd.choice(alphabet) for _ in range(rnd.randrange(2, 9))) for _ in range(200)]
    s = []
    while sum(len(x) + 1 for x in s) < n:
        s.append(rnd.choice(words))
    return (" ".join(s)).encode("latin1")[:n]


def split(rnd, n, ntypes, lo, hi, cycle=False):
    """block splits [(type, count)] covering n symbols, first block of type 0, every type used"""
    blocks, total, k = [], 0, 0
    while total < n or k < ntypes:
        t = 0 if not blocks else (k % ntypes if cycle or k < ntypes else rnd.randrange(ntypes))
        if blocks and t == blocks[-1][0]:
            t = (t + 1) % ntypes
        c = rnd.randrange(lo, hi)
        blocks.append((t, c)); total += c; k += 1
    return blocks


def counts(cmds):
    return sum(len(i) for i, _, _ in cmds), len(cmds), sum(1 for _, c, _ in cmds if c)


def vectors():
    rnd = random.Random(20260929)
    out = []
    # 1. forty literal block types, all four context modes, a context map that sends contexts to 23 trees, ring codes
    data = text(rnd, 60000, "etaoinshrdlucmfw") + bytes(rnd.randrange(256) for _ in range(3000)) + text(rnd, 20000, "ETAOINSHRDLU0123456789")
    cmds = E.greedy_commands(data)
    nl, nc, nd = counts(cmds)
    nbt0 = 40
    plan = E.Plan(lit_blocks=split(rnd, nl, nbt0, 200, 3000), cmd_blocks=split(rnd, nc, 3, 50, 800), dist_blocks=split(rnd, nd, 5, 30, 600),
                  modes=[t % 4 for t in range(nbt0)], lit_map=[(t * 7 + (c >> 3)) % 23 for t in range(nbt0) for c in range(64)],
                  dist_map=[(t + c) % 4 for t in range(5) for c in range(4)], npostfix=2, ndirect=12 << 2, type_codes="ring")
    w = E.BitWriter(); E.write_stream_header(w, 22)
    assert E.emit_compressed(w, cmds, plan, True) == data
    out.append(("40-literal-block-types-all-modes", w.finish(), data))
    # 2. context-free literals with 60 block types (the command engine's side of the kernel), NPOSTFIX 1, many switches
    data = text(rnd, 150000, "abcdefghijklmnopqrstuvwxyz .,")
    cmds = E.greedy_commands(data)
    nl, nc, nd = counts(cmds)
    plan = E.Plan(lit_blocks=split(rnd, nl, 60, 100, 1500, cycle=True), cmd_blocks=split(rnd, nc, 7, 20, 300), dist_blocks=split(rnd, nd, 9, 20, 300),
                  lit_map=[t % 11 for t in range(60) for _ in range(64)], dist_map=[t % 3 for t in range(9) for _ in range(4)], npostfix=1, ndirect=6 << 1)
    w = E.BitWriter(); E.write_stream_header(w, 18)
    assert E.emit_compressed(w, cmds, plan, True) == data
    out.append(("60-block-types-context-free", w.finish(), data))
    # 3. compressed, metadata, stored, empty metadata, compressed (copies reach into the stored block), last-empty
    a, b, c = text(rnd, 20000, "xyz012 "), bytes(rnd.randrange(256) for _ in range(5000)), text(rnd, 30000, "xyz012 ")
    w = E.BitWriter(); E.write_stream_header(w, 16)
    o1 = E.emit_compressed(w, E.greedy_commands(a), E.Plan(modes=[1], lit_map=[(ctx >> 4) for ctx in range(64)]), False)
    E.emit_metadata(w, b"metadata block: skipped by every decoder")
    E.emit_stored(w, b)
    E.emit_metadata(w, b"")
    cmds = E.greedy_commands(c + b[:700], history=a + b)
    o3 = E.emit_compressed(w, cmds, E.Plan(modes=[3], lit_map=[ctx % 5 for ctx in range(64)], npostfix=3, ndirect=15 << 3), False, prev=a + b)
    E.emit_last_empty(w)
    assert o1 == a and o3 == c + b[:700]
    out.append(("compressed-metadata-stored-mix", w.finish(), a + b + c + b[:700]))
    # 4. 256 literal block types with a tree each: tables far beyond LDS (the spill / larger-arena passes)
    data = b"".join(bytes(rnd.choice(range(t, 256, 7)) for _ in range(300)) for t in range(256))
    cmds = [(data, 0, 0)] if False else E.greedy_commands(data, min_match=6)
    nl, nc, nd = counts(cmds)
    blocks = [(t, 300) for t in range(256)]
    blocks[-1] = (255, 300 + nl)  # (the last block takes what is left)
    plan = E.Plan(lit_blocks=blocks, modes=[2] * 256, lit_map=[t for t in range(256) for _ in range(64)])
    w = E.BitWriter(); E.write_stream_header(w, 20)
    assert E.emit_compressed(w, cmds, plan, True) == data
    out.append(("256-literal-block-types", w.finish(), data))
    # 5. MSB6 for every block, simple prefix codes (one to four symbols) everywhere
    data = bytes(rnd.choice(b"ab") for _ in range(4000)) + bytes(rnd.choice(b"abc") for _ in range(4000)) + b"a" * 3000 + bytes(rnd.choice(b"abcd") for _ in range(4000))
    cmds = E.greedy_commands(data, min_match=12)
    nl, nc, nd = counts(cmds)
    plan = E.Plan(lit_blocks=[(0, 4000), (1, 4000), (2, 3000), (3, 1 << 20)], modes=[1, 1, 1, 1], lit_map=[t for t in range(4) for _ in range(64)])
    w = E.BitWriter(); E.write_stream_header(w, 17)
    assert E.emit_compressed(w, cmds, plan, True) == data
    out.append(("msb6-simple-codes", w.finish(), data))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    for f in os.listdir(OUT):
        os.remove(os.path.join(OUT, f))
    manifest = []
    for i, (label, comp, raw) in enumerate(vectors()):
        if ref.available():
            r = ref.decode(comp, len(raw) + 16, False)
            assert r[0] == 1 and r[2] == raw, (label, r[0], r[1])
        info, out = oracle.decode(comp, len(raw) + 16, 0)
        assert info.result == 1 and out == raw, (label, info.result, info.error_code)
        name = "%02d-%s.br" % (i, label)
        open(os.path.join(OUT, name), "wb").write(comp)
        manifest.append({"label": label, "file": name, "csize": len(comp), "size": len(raw), "sha256": hashlib.sha256(raw).hexdigest(),
                         "metablocks": info.num_metablocks, "commands": info.num_commands, "max_block_types": info.max_block_types,
                         "max_literal_trees": info.max_literal_trees})
        print(manifest[-1])
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
