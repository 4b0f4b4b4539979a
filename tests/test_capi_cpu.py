"""Host-side checks that need no GPU: the C-ABI library loads, exports every declared symbol, and fails
loudly (never silently) when no HIP device is usable."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, load_pkg


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="module")
def lib():
    pkg = load_pkg()
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build()
    return pkg.load_library()


def test_exports_every_declared_symbol(lib):
    pkg = load_pkg()
    for header, names in (("decode.h", pkg.DECODE_H_SYMBOLS), ("batch.h", pkg.BATCH_H_SYMBOLS)):
        text = open(os.path.join(ROOT, "include", "brotli", header)).read()
        declared = set(re.findall(r"BROTLI_DEC_API[^;(]*?\b(Brotli\w+)\s*\(", text))
        assert declared == set(names), (header, declared ^ set(names))
        for n in names:
            assert hasattr(lib, n), n


def test_version_and_error_strings(lib):
    assert lib.BrotliDecoderVersion() == 0x1000f00  # src/ffi/mod.rs:588-590
    assert lib.BrotliDecoderErrorString(-8) == b"ERROR_FORMAT_CONTEXT_MAP_REPEAT"
    assert lib.BrotliDecoderErrorString(-6) == b"ERROR_FORMAT_FL_SPACE"  # src/state.rs:547
    assert lib.BrotliDecoderErrorString(2) == b"NEEDS_MORE_INPUT"


def test_return_info_layout():
    pkg = load_pkg()
    assert ctypes.sizeof(pkg.ReturnInfo) == 272  # src/lib.rs:336-342


def test_instance_lifecycle_without_decoding(lib):
    st = lib.BrotliDecoderCreateInstance(None, None, None)
    assert st
    assert lib.BrotliDecoderIsUsed(st) == 0 and lib.BrotliDecoderIsFinished(st) == 0
    assert lib.BrotliDecoderSetParameter(st, 1, 1) == 1
    assert lib.BrotliDecoderGetErrorCode(st) == 1
    p = lib.BrotliDecoderMallocU8(st, 64)
    assert p
    lib.BrotliDecoderFreeU8(st, p, 64)
    lib.BrotliDecoderDestroyInstance(st)
    # one callback without the other is rejected (src/ffi/mod.rs:132-135)
    alloc_t = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)
    cb = alloc_t(lambda opaque, n: None)
    assert not lib.BrotliDecoderCreateInstance(ctypes.cast(cb, ctypes.c_void_p), None, None)


def test_invalid_arguments(lib):
    pkg = load_pkg()
    info = lib.BrotliDecoderDecompressWithReturnInfo(4, None, 16, None)
    assert (info.result, info.code, info.decoded_size) == (0, -20, 0)  # src/ffi/mod.rs:606-640
    st = lib.BrotliDecoderCreateInstance(None, None, None)
    assert lib.BrotliDecoderDecompressStream(st, None, None, None, None, None) == 0
    assert lib.BrotliDecoderGetErrorCode(st) == -20
    lib.BrotliDecoderDestroyInstance(st)
    assert isinstance(pkg.last_error(), str)


@pytest.mark.skipif(_has_gpu(), reason="checks the behaviour on a box without a GPU")
def test_fails_loudly_without_a_device(lib):
    pkg = load_pkg()
    info, out = pkg.brotli_decode(b"\x06", 16)
    assert (info.result, info.code) == (0, -31) and out == b""
    assert b"HIP" in info.error
    with pytest.raises(RuntimeError):
        pkg.Batch(4)
