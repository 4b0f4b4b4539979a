"""Synthetic Brotli workloads of BASELINE.json -- inputs for bench.py and the GPU tests.

Raw data comes from a fixed-seed generator (numpy PCG64); it is compressed with Google's libbrotlienc 1.0.9
through ctypes (the same image runs on the GPU box).  Expected outputs are never stored: the generator is
re-run.  When no encoder library can be loaded, callers fall back to the committed reference fixtures.
"""
import ctypes
import hashlib
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))

_enc = None
for _cand in ("libbrotlienc.so.1", "/usr/lib/x86_64-linux-gnu/libbrotlienc.so.1", "/opt/conda/lib/libbrotlienc.so.1"):
    try:
        _enc = ctypes.CDLL(_cand)
        _enc.BrotliEncoderCompress.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p,
                                               ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
        _enc.BrotliEncoderMaxCompressedSize.argtypes = [ctypes.c_size_t]
        _enc.BrotliEncoderMaxCompressedSize.restype = ctypes.c_size_t
        break
    except OSError:
        _enc = None


def encoder_available():
    return _enc is not None


def brotli_compress(data: bytes, quality=5, lgwin=22) -> bytes:
    cap = _enc.BrotliEncoderMaxCompressedSize(len(data)) or (len(data) + 1024)
    out = ctypes.create_string_buffer(cap)
    n = ctypes.c_size_t(cap)
    if not _enc.BrotliEncoderCompress(quality, lgwin, 0, len(data), data, ctypes.byref(n), out):
        raise RuntimeError("BrotliEncoderCompress failed")
    return out.raw[:n.value]


def long_backref_stream(seed: int, size: int = 4 << 20, seed_shift: int = 3) -> bytes:
    """C3 of SURVEY.md section 8d, per 4 MiB-window stream: a Zipf(s=1) text-like seed region over 64 symbols,
    then 90 % copies of U[32,4096) bytes from far back (U[64 KiB, pos) bytes, at most the 4 MiB window) and
    10 % short runs of fresh symbols -- many long back-references, few literals."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ranks = np.arange(1, 65, dtype=np.float64)
    p = (1.0 / ranks) / np.sum(1.0 / ranks)
    out = np.empty(size, dtype=np.uint8)
    seed_len = min(size, max(1024, size >> seed_shift))  # 512 KiB of a 4 MiB stream (seed_shift 2: a quarter, as 4 MiB of SURVEY 8(a1)'s 16 MiB prototype)
    out[:seed_len] = (rng.choice(64, size=seed_len, p=p) + 32).astype(np.uint8)
    pos = seed_len
    max_back = (4 << 20) - 16
    while pos < size:
        if rng.random() < 0.9:
            n = int(rng.integers(32, 4096))
            hi = min(pos, max_back)
            d = int(rng.integers(min(64 << 10, hi // 2), hi)) if hi > 1 else 1
            n = min(n, size - pos, d)
            out[pos:pos + n] = out[pos - d:pos - d + n]
        else:
            n = min(int(rng.integers(1, 64)), size - pos)
            out[pos:pos + n] = (rng.choice(64, size=n, p=p) + 32).astype(np.uint8)
        pos += n
    return out.tobytes()


def high_entropy_stream(seed: int, size: int = 4 << 20) -> bytes:
    """C4 of SURVEY.md section 8d: i.i.d. bytes, p(rank r) ~ r^-0.6 over 256 symbols through a fixed permutation
    (about 7.5 bits/byte): Huffman-coded literals, almost no copies."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ranks = np.arange(1, 257, dtype=np.float64)
    p = ranks ** -0.6
    p /= p.sum()
    perm = np.random.Generator(np.random.PCG64(12345)).permutation(256).astype(np.uint8)
    return perm[rng.choice(256, size=size, p=p)].tobytes()


_made = {}


def make_streams(kind: str, n_unique: int, size: int, seed0: int, quality=5, lgwin=22, threads=None):
    """-> list of (compressed bytes, raw size, sha256 of raw).  kind in {'long_backref', 'high_entropy', 'survey_mix'}."""
    gen = long_backref_stream if kind == "long_backref" else (lambda sd, sz: long_backref_stream(sd, sz, 2)) if kind == "survey_mix" else high_entropy_stream
    key = (kind, n_unique, size, seed0, quality, lgwin)
    if key in _made:  # (bench legs that reuse the headline's streams)
        return _made[key]

    def one(i):
        raw = gen(seed0 + i, size)
        return brotli_compress(raw, quality, lgwin), len(raw), hashlib.sha256(raw).hexdigest()

    threads = threads or min(n_unique, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=threads) as ex:
        _made[key] = list(ex.map(one, range(n_unique)))
    return _made[key]


def fixture_streams(name="alice29.txt.compressed"):
    """The reference's own fixture, for boxes without an encoder (and for BASELINE config 1)."""
    import json
    m = {e["name"]: e for e in json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))}
    data = open(os.path.join(ROOT, "tests", "golden", "testdata", name), "rb").read()
    return [(data, m[name]["size"], m[name]["sha256"])]
