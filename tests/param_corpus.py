"""Synthetic streams that reach what the reference's own fixtures leave unpinned: NPOSTFIX /
NDIRECT != 0, every literal context mode, many block types, every window size including large windows, several
metablocks, flushes and metadata blocks.  Made with Google's libbrotlienc 1.0.9 where the image has it; every stream is
known to decode to its raw data with libbrotlidec, so the expected output needs no oracle.  A smaller edition of the
same corpus is committed under tests/golden/param_corpus/ (tools/make_param_corpus.py; SHA-256 of every raw input in
its manifest), so that the tests that use it never skip: corpus() falls back to it on a box without the encoder.
-> list of (label, compressed, raw)."""
import hashlib
import json
import os

import numpy as np

import libbrotli_ref as ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _raws(scale=1.0):
    rng = np.random.Generator(np.random.PCG64(20260928))
    import oracle_lib as oracle
    alice = open(os.path.join(ROOT, "tests", "golden", "testdata", "alice29.txt.compressed"), "rb").read()
    text = oracle.decode(alice, 200000, 1)[1]
    words = text.split()
    utf = " ".join(words[i].decode("latin1") + ("é" if i % 7 == 0 else "ß" if i % 11 == 0 else "") for i in range(6000)).encode("utf8")
    ints = (np.cumsum(rng.integers(-3, 4, 20000)).astype("<i4")).tobytes()                      # structured binary: signed deltas
    floats = np.sin(np.arange(16000) * 0.01).astype("<f4").tobytes()
    zipf = (rng.choice(96, size=60000, p=(1.0 / np.arange(1, 97)) / np.sum(1.0 / np.arange(1, 97))) + 32).astype(np.uint8).tobytes()
    noise = rng.integers(0, 256, 30000, dtype=np.uint8).tobytes()
    runs = b"".join(bytes([int(b)]) * int(n) for b, n in zip(rng.integers(0, 256, 400), rng.integers(1, 300, 400)))
    mixed = text[:30000] + ints[:20000] + noise[:5000] + text[30000:50000] + runs[:10000] + zipf[:20000]
    raws = {"text": text[:120000], "utf8": utf, "ints": ints, "floats": floats, "zipf": zipf, "noise": noise, "runs": runs, "mixed": mixed}
    raws = {k: v[:max(64, int(len(v) * scale)) & ~3] for k, v in raws.items()}
    raws.update({"tiny": b"abcabcabcabc", "empty": b""})
    return raws


COMMITTED = os.path.join(ROOT, "tests", "golden", "param_corpus")


def committed():
    """the committed edition: compressed streams from the repository, raw data from the checker, pinned by the manifest's SHA-256"""
    import oracle_lib as oracle
    out = []
    for e in json.load(open(os.path.join(COMMITTED, "manifest.json"))):
        comp = open(os.path.join(COMMITTED, e["file"]), "rb").read()
        info, raw = oracle.decode(comp, e["size"] + 16, 1)
        assert info.result == 1 and len(raw) == e["size"] and hashlib.sha256(raw).hexdigest() == e["sha256"], e["label"]
        out.append((e["label"], comp, raw))
    return out


def corpus(limit=None, scale=1.0):
    if not ref.encoder_available():
        c = committed()
        return c[:limit] if limit else c
    raws = _raws(scale)
    out = []

    def add(label, raw, params, chunks=None, ops=None):
        comp = ref.encode_stream(chunks if chunks is not None else [raw], params, ops)
        out.append((label, comp, raw))

    P = ref
    # qualities x data kinds (greedy / hashing / zopfli back ends; block splitting and context modelling from q >= 5 / 10)
    for q in (0, 1, 2, 4, 5, 6, 9, 10, 11):
        for name in ("text", "utf8", "ints", "zipf", "mixed"):
            add("q%d-%s" % (q, name), raws[name], {P.PARAM_QUALITY: q, P.PARAM_LGWIN: 22})
    # distance code parameters (RFC 7932 section 4): every NPOSTFIX with several NDIRECT
    for npostfix in (0, 1, 2, 3):
        for nd in (0, 1, 5, 15):
            ndirect = nd << npostfix
            for name in ("ints", "mixed"):
                add("np%d-nd%d-%s" % (npostfix, ndirect, name), raws[name],
                    {P.PARAM_QUALITY: 9, P.PARAM_LGWIN: 20, P.PARAM_NPOSTFIX: npostfix, P.PARAM_NDIRECT: ndirect})
    # modes: generic / text (UTF-8 context) / font (distance parameters chosen by the encoder)
    for mode in (0, 1, 2):
        for name in ("utf8", "floats", "mixed"):
            add("mode%d-%s" % (mode, name), raws[name], {P.PARAM_QUALITY: 11, P.PARAM_LGWIN: 18, P.PARAM_MODE: mode})
    # every window size, standard and large-window encodings
    for lgwin in range(10, 25):
        add("lgwin%d" % lgwin, raws["mixed"], {P.PARAM_QUALITY: 6, P.PARAM_LGWIN: lgwin})
    for lgwin in (10, 16, 24, 25, 28, 30):
        add("large-lgwin%d" % lgwin, raws["text"], {P.PARAM_QUALITY: 5, P.PARAM_LGWIN: lgwin, P.PARAM_LARGE_WINDOW: 1})
    # small input blocks: many metablocks; flushes; metadata blocks; literal context modelling off
    add("lgblock16", raws["mixed"] * 3, {P.PARAM_QUALITY: 9, P.PARAM_LGWIN: 18, P.PARAM_LGBLOCK: 16})
    add("no-ctx", raws["text"], {P.PARAM_QUALITY: 11, P.PARAM_LGWIN: 22, P.PARAM_NO_LITERAL_CONTEXT: 1})
    t = raws["text"]
    add("flushes", t[:50000], {P.PARAM_QUALITY: 5, P.PARAM_LGWIN: 16}, [t[:10], t[10:20000], b"", t[20000:20001], t[20001:50000]],
        [P.OP_FLUSH, P.OP_FLUSH, P.OP_FLUSH, P.OP_FLUSH, P.OP_PROCESS])
    add("metadata", t[:30000], {P.PARAM_QUALITY: 9, P.PARAM_LGWIN: 22}, [b"0123456789abcdef", t[:15000], b"", b"xyz", t[15000:30000]],
        [P.OP_EMIT_METADATA, P.OP_FLUSH, P.OP_EMIT_METADATA, P.OP_EMIT_METADATA, P.OP_PROCESS])
    for name in ("noise", "runs", "tiny", "empty"):
        for q in (1, 9):
            add("q%d-%s" % (q, name), raws[name], {P.PARAM_QUALITY: q, P.PARAM_LGWIN: 22})
    return out[:limit] if limit else out
