"""ctypes binding of oracle/libbrotli_oracle.so -- the CPU checker (test infrastructure only)."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "libbrotli_oracle.so")

FLAG_LARGE_WINDOW = 1
FLAG_NO_CANNY = 2

RESULT_ERROR, RESULT_SUCCESS, RESULT_NEEDS_MORE_INPUT, RESULT_NEEDS_MORE_OUTPUT = 0, 1, 2, 3


class OracleInfo(ctypes.Structure):
    _fields_ = [("result", ctypes.c_int32), ("error_code", ctypes.c_int32), ("decoded_size", ctypes.c_uint64),
                ("consumed", ctypes.c_uint64), ("produced", ctypes.c_uint64), ("window_bits", ctypes.c_uint32),
                ("num_metablocks", ctypes.c_uint32), ("num_commands", ctypes.c_uint64), ("num_literals", ctypes.c_uint64),
                ("num_context_literals", ctypes.c_uint64), ("max_literal_trees", ctypes.c_uint32), ("max_block_types", ctypes.c_uint32)]


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ROOT, "oracle", "brotli_oracle.c")
        if not os.path.exists(_SO) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.brotli_oracle_decode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                              ctypes.c_uint32, ctypes.POINTER(OracleInfo)]
        _lib.brotli_oracle_decode.restype = ctypes.c_int
    return _lib


def decode(data: bytes, out_cap: int, flags: int = FLAG_LARGE_WINDOW):
    """-> (info, output bytes delivered)"""
    L = lib()
    info = OracleInfo()
    inbuf = (ctypes.c_uint8 * max(1, len(data))).from_buffer_copy(data.ljust(1, b"\0"))
    out = (ctypes.c_uint8 * max(1, out_cap))()
    L.brotli_oracle_decode(inbuf, len(data), out, out_cap, flags, ctypes.byref(info))
    return info, bytes(memoryview(out)[:info.decoded_size])
