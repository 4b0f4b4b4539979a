"""Host-side checks that need no GPU: the C-ABI library loads, exports every declared symbol, and fails
loudly (never silently) when no HIP device is usable."""
import ctypes
import os
import re
import sys

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


def _plan(lib, sizes, cus=256, gang_env=-1, pool_env=-1):
    lib.BrotliAmdDebugPlanGangs.restype = ctypes.c_uint32
    lib.BrotliAmdDebugPlanGangs.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_size_t), ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]
    arr = (ctypes.c_size_t * len(sizes))(*sizes)
    grid = ctypes.c_uint32(0)
    kind = lib.BrotliAmdDebugPlanGangs(len(sizes), cus, arr, gang_env, pool_env, ctypes.byref(grid))
    return kind, grid.value


def test_how_many_blocks_a_stream_gets(lib):
    """the host's plan for several blocks on a stream (csrc/brotli_capi.cpp: plan_gangs), a pure function of the batch's compressed
    sizes and the device's CUs: gangs of 8 / 4 / 2 blocks a stream for batches of equal streams up to an eighth / a quarter / half the CUs'
    number (sixteen where the device has them and a stream is long: 2 MiB compressed), a pool (0x108) where more than 32 streams differ widely in
    size, nothing for small streams or where the environment says so"""
    MB = 400_000   # (a 4 MiB stream of the metric's, compressed)
    assert _plan(lib, [MB]) == (8, 64)                    # one stream: a gang of eight, eight streams' worth of blocks
    assert _plan(lib, [MB] * 8) == (8, 64)
    assert _plan(lib, [MB] * 9) == (8, 128)
    assert _plan(lib, [MB] * 32) == (8, 256)
    assert _plan(lib, [MB] * 33) == (4, 160)
    assert _plan(lib, [MB] * 64) == (4, 256)
    assert _plan(lib, [MB] * 65) == (2, 144)
    assert _plan(lib, [MB] * 128) == (2, 256)
    assert _plan(lib, [MB] * 129) == (0, 129)             # one block a stream, as many blocks as streams
    assert _plan(lib, [MB] * 256) == (0, 256)
    assert _plan(lib, [MB] * 257) == (0, 257)             # (more streams than CUs: not this function's)
    # long streams, few enough for sixteen blocks each (round 6: eight blocks on one long stream are busy, not waiting for one another)
    assert _plan(lib, [8_000_000]) == (16, 128)
    assert _plan(lib, [8_000_000] * 16) == (16, 256)
    assert _plan(lib, [8_000_000] * 17) == (8, 192)
    assert _plan(lib, [8_000_000], cus=120) == (8, 64)
    assert _plan(lib, [8_000_000], gang_env=8) == (8, 64)
    assert _plan(lib, [MB], gang_env=16) == (16, 128)
    # small streams: nothing to divide
    assert _plan(lib, [60_000] * 8) == (0, 8)
    assert _plan(lib, [60_000] * 7 + [70_000]) == (8, 64)
    # very different sizes: a pool from 33 streams on (up to 32 every stream has eight blocks anyway), as many blocks as CUs
    big = 8_000_000
    assert _plan(lib, [big] + [100_000] * 31) == (8, 256)
    assert _plan(lib, [big] + [100_000] * 39) == (0x108, 256)
    assert _plan(lib, [big] + [100_000] * 199) == (0x108, 256)
    assert _plan(lib, [big] + [100_000] * 255) == (0x108, 256)
    assert _plan(lib, [big] + [100_000] * 255, cus=304) == (0x108, 304)
    assert _plan(lib, [200_000] + [60_000] * 99) == (2, 208)   # (the long one is no long pole: under 256 KiB compressed)
    assert _plan(lib, [MB] * 100 + [2 * MB]) == (2, 208)       # (not more than twice the median)
    # the environment: BROTLI_AMD_GANG=0 nothing, =2 / 4 gangs of at most that many and no pool; BROTLI_AMD_POOL=0 no pool, =2 a pool whatever the sizes
    assert _plan(lib, [MB] * 8, gang_env=0) == (0, 8)
    assert _plan(lib, [MB] * 8, gang_env=1) == (0, 8)
    assert _plan(lib, [MB] * 8, gang_env=4) == (4, 32)
    assert _plan(lib, [big] + [100_000] * 199, gang_env=8) == (0, 200)
    assert _plan(lib, [big] + [100_000] * 199, pool_env=0) == (0, 200)
    assert _plan(lib, [big] + [100_000] * 39, pool_env=0) == (4, 160)
    assert _plan(lib, [MB] * 200, pool_env=2) == (0x108, 256)
    assert _plan(lib, [MB] * 100, pool_env=2) == (2, 208)      # (where there are gangs, a forced pool does not replace them)


def test_the_hand_written_run_is_what_its_generator_writes(tmp_path):
    """csrc/brotli_rec_run_asm.h (LEAN_REC_RUN_ASM: the record loop's plain commands) is GENERATED by tools/gen_rec_asm.py: the committed
    header and the script must not drift apart"""
    import subprocess
    out = tmp_path / "run.h"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_rec_asm.py"), "--out=" + str(out)], stdout=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(ROOT, "rust-brotli-decompressor_amd", "csrc", "brotli_rec_run_asm.h")).read()
