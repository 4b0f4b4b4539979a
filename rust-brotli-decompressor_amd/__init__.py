"""MI355X-native Brotli decode path: thin Python binding of libbrotli_decompressor.so (ctypes).

The product is the shared library (HIP kernels + the reference's C ABI, see include/brotli/decode.h and
include/brotli/batch.h).  This module only loads it and gives tests and bench.py a convenient handle; it
contains no decoder and no CPU fallback -- if the library is missing or no HIP device is usable, calls fail.

Names follow the reference: `Decompressor` is the pull adapter of src/reader.rs (io::Read), `DecompressorWriter`
the push adapter of src/writer.rs (io::Write), `brotli_decode` the one-shot helper of src/lib.rs:447-468.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BROTLI_AMD_LIB") or os.path.join(_HERE, "libbrotli_decompressor.so")

RESULT_ERROR, RESULT_SUCCESS, RESULT_NEEDS_MORE_INPUT, RESULT_NEEDS_MORE_OUTPUT = 0, 1, 2, 3
FLAG_LARGE_WINDOW, FLAG_NO_CANNY = 1, 2
FLAG_SPILL_IN_PLACE = 16  # BROTLI_AMD_BATCH_SPILL_IN_PLACE: no second launch with a larger LDS arena
FLAG_EAGER_OUTPUT_LIMIT = 64  # BROTLI_AMD_BATCH_EAGER_OUTPUT_LIMIT: NEEDS_MORE_OUTPUT where the buffer ends, no second look (batch.h)
PARAM_DISABLE_RING_BUFFER_REALLOCATION, PARAM_LARGE_WINDOW = 0, 1

# every symbol include/brotli/decode.h and include/brotli/batch.h declare
DECODE_H_SYMBOLS = [
    "BrotliDecoderSetParameter", "BrotliDecoderCreateInstance", "BrotliDecoderDestroyInstance", "BrotliDecoderDecompress",
    "BrotliDecoderDecompressWithReturnInfo", "BrotliDecoderDecompressPrealloc", "BrotliDecoderDecompressStream",
    "BrotliDecoderDecompressStreaming", "BrotliDecoderHasMoreOutput", "BrotliDecoderTakeOutput", "BrotliDecoderIsUsed",
    "BrotliDecoderIsFinished", "BrotliDecoderGetErrorCode", "BrotliDecoderGetErrorString", "BrotliDecoderErrorString",
    "BrotliDecoderVersion", "BrotliDecoderMallocU8", "BrotliDecoderFreeU8", "BrotliDecoderMallocUsize", "BrotliDecoderFreeUsize",
]
BATCH_H_SYMBOLS = [
    "BrotliAmdBatchCreate", "BrotliAmdBatchDestroy", "BrotliAmdBatchDecodeDevice", "BrotliAmdBatchRelaunch", "BrotliAmdBatchWait",
    "BrotliAmdBatchDecodeHost", "BrotliAmdBatchLastKernelMs", "BrotliAmdBatchLastSecondPassCount", "BrotliAmdBatchLastGang", "BrotliAmdBatchLastPool", "BrotliAmdBatchLastProbeMs", "BrotliAmdDebugPlanGangs", "BrotliAmdLastError", "BrotliAmdLastNote", "BrotliAmdDebugBuildTree", "BrotliAmdDecoderDeviceCommands",
]


class ReturnInfo(ctypes.Structure):  # BrotliDecoderReturnInfo, reference src/lib.rs:336-342
    _fields_ = [("decoded_size", ctypes.c_size_t), ("error", ctypes.c_char * 256), ("result", ctypes.c_int), ("code", ctypes.c_int)]


class BatchResult(ctypes.Structure):  # BrotliAmdResult
    _fields_ = [("result", ctypes.c_int32), ("error_code", ctypes.c_int32), ("decoded_size", ctypes.c_uint64),
                ("consumed", ctypes.c_uint64), ("produced", ctypes.c_uint64), ("num_metablocks", ctypes.c_uint32),
                ("spilled_metablocks", ctypes.c_uint32), ("num_commands", ctypes.c_uint64),
                ("engine_commands", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


def build(force=False):
    """Compile the HIP kernels for gfx950 and link the C-ABI library in-tree."""
    if force:
        subprocess.check_call(["make", "-s", "-C", _HERE, "clean"])
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def load_library():
    """The HIP extension.  Raises if it has not been built -- there is nothing to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32
    L.BrotliDecoderCreateInstance.restype = vp
    L.BrotliDecoderCreateInstance.argtypes = [vp, vp, vp]
    L.BrotliDecoderDestroyInstance.argtypes = [vp]
    L.BrotliDecoderSetParameter.argtypes = [vp, ctypes.c_int, u32]
    L.BrotliDecoderDecompress.argtypes = [sz, vp, ctypes.POINTER(sz), vp]
    L.BrotliDecoderDecompressWithReturnInfo.restype = ReturnInfo
    L.BrotliDecoderDecompressWithReturnInfo.argtypes = [sz, vp, sz, vp]
    L.BrotliDecoderDecompressPrealloc.restype = ReturnInfo
    L.BrotliDecoderDecompressPrealloc.argtypes = [sz, vp, sz, vp, sz, vp, sz, vp, sz, vp]
    L.BrotliDecoderDecompressStream.argtypes = [vp, ctypes.POINTER(sz), ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(vp),
                                                ctypes.POINTER(sz)]
    L.BrotliDecoderDecompressStreaming.argtypes = [vp, ctypes.POINTER(sz), vp, ctypes.POINTER(sz), vp]
    for name in ("BrotliDecoderHasMoreOutput", "BrotliDecoderIsUsed", "BrotliDecoderIsFinished", "BrotliDecoderGetErrorCode"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = ctypes.c_int
    L.BrotliDecoderTakeOutput.restype = vp
    L.BrotliDecoderTakeOutput.argtypes = [vp, ctypes.POINTER(sz)]
    L.BrotliDecoderGetErrorString.restype = ctypes.c_char_p
    L.BrotliDecoderGetErrorString.argtypes = [vp]
    L.BrotliDecoderErrorString.restype = ctypes.c_char_p
    L.BrotliDecoderErrorString.argtypes = [ctypes.c_int]
    L.BrotliDecoderVersion.restype = u32
    L.BrotliDecoderMallocU8.restype = vp
    L.BrotliDecoderMallocU8.argtypes = [vp, sz]
    L.BrotliDecoderFreeU8.argtypes = [vp, vp, sz]
    L.BrotliDecoderMallocUsize.restype = vp
    L.BrotliDecoderMallocUsize.argtypes = [vp, sz]
    L.BrotliDecoderFreeUsize.argtypes = [vp, vp, sz]
    L.BrotliAmdBatchCreate.restype = vp
    L.BrotliAmdBatchCreate.argtypes = [u32, u32, u32]
    L.BrotliAmdBatchDestroy.argtypes = [vp]
    L.BrotliAmdBatchDecodeDevice.argtypes = [vp, u32, vp, vp, vp, vp, u32, vp]
    L.BrotliAmdBatchRelaunch.argtypes = [vp, vp]
    L.BrotliAmdBatchWait.argtypes = [vp, vp]
    L.BrotliAmdBatchDecodeHost.argtypes = [vp, u32, vp, vp, vp, vp, u32, vp]
    L.BrotliAmdBatchLastKernelMs.restype = ctypes.c_float
    L.BrotliAmdBatchLastKernelMs.argtypes = [vp]
    L.BrotliAmdBatchLastSecondPassCount.restype = ctypes.c_uint32
    L.BrotliAmdBatchLastSecondPassCount.argtypes = [vp]
    if hasattr(L, "BrotliAmdBatchLastGang"):   # (libraries of earlier rounds, for A/B runs: tools/ab.sh)
        L.BrotliAmdBatchLastGang.restype = ctypes.c_uint32
        L.BrotliAmdBatchLastGang.argtypes = [vp]
    if hasattr(L, "BrotliAmdBatchLastProbeMs"):
        L.BrotliAmdBatchLastProbeMs.restype = ctypes.c_float
        L.BrotliAmdBatchLastProbeMs.argtypes = [vp]
    if hasattr(L, "BrotliAmdBatchLastPool"):
        L.BrotliAmdBatchLastPool.restype = ctypes.c_uint32
        L.BrotliAmdBatchLastPool.argtypes = [vp]
    L.BrotliAmdLastError.restype = ctypes.c_char_p
    if hasattr(L, "BrotliAmdLastNote"):   # (an older build of the library, loaded through BROTLI_AMD_LIB for an A/B, has no such symbol)
        L.BrotliAmdLastNote.restype = ctypes.c_char_p
    _lib = L
    return L


def last_error():
    return load_library().BrotliAmdLastError().decode()


# ------------------------------------------------------------------ one-shot (src/lib.rs:447-468)
def brotli_decode(data: bytes, out_cap: int):
    """-> (ReturnInfo, delivered bytes).  Large-window streams accepted, like the reference's brotli_decode."""
    L = load_library()
    out = ctypes.create_string_buffer(max(1, out_cap))
    src = ctypes.create_string_buffer(bytes(data), max(1, len(data)))
    info = L.BrotliDecoderDecompressWithReturnInfo(len(data), ctypes.addressof(src), out_cap, ctypes.addressof(out))
    return info, out.raw[:info.decoded_size]


# ------------------------------------------------------------------ batch (include/brotli/batch.h)
class Batch:
    """Owns one BrotliAmdBatch on the current HIP device."""

    def __init__(self, max_streams, lds_arena_bytes=0, grid_blocks=0):
        self._L = load_library()
        self._h = self._L.BrotliAmdBatchCreate(max_streams, lds_arena_bytes, grid_blocks)
        if not self._h:
            raise RuntimeError("BrotliAmdBatchCreate failed: " + last_error())
        self.max_streams = max_streams
        self.n = 0

    def close(self):
        if self._h:
            self._L.BrotliAmdBatchDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode_device(self, in_ptrs, in_sizes, out_ptrs, out_caps, flags=FLAG_LARGE_WINDOW, stream=None):
        """Device pointers in, asynchronous launch on `stream` (a hipStream_t handle as int, None = default)."""
        n = len(in_ptrs)
        a_in = (ctypes.c_void_p * n)(*in_ptrs)
        a_is = (ctypes.c_size_t * n)(*in_sizes)
        a_out = (ctypes.c_void_p * n)(*out_ptrs)
        a_oc = (ctypes.c_size_t * n)(*out_caps)
        if self._L.BrotliAmdBatchDecodeDevice(self._h, n, a_in, a_is, a_out, a_oc, flags, stream) != 0:
            raise RuntimeError("BrotliAmdBatchDecodeDevice failed: " + last_error())
        self.n = n

    def relaunch(self, stream=None):
        if self._L.BrotliAmdBatchRelaunch(self._h, stream) != 0:
            raise RuntimeError("BrotliAmdBatchRelaunch failed: " + last_error())

    def wait(self):
        res = (BatchResult * max(1, self.n))()
        if self._L.BrotliAmdBatchWait(self._h, res) != 0:
            raise RuntimeError("BrotliAmdBatchWait failed: " + last_error())
        return list(res)[:self.n]

    def last_kernel_ms(self):
        return float(self._L.BrotliAmdBatchLastKernelMs(self._h))

    def last_second_pass_count(self):
        return int(self._L.BrotliAmdBatchLastSecondPassCount(self._h))

    def last_gang(self):
        """blocks (CUs) a stream of the last launch: 1, or 2 / 4 / 8 where each stream had a gang of blocks"""
        return int(self._L.BrotliAmdBatchLastGang(self._h)) if hasattr(self._L, "BrotliAmdBatchLastGang") else 1

    def last_probe_ms(self):
        """host milliseconds the last decode call spent asking the device what kind the batch's streams are (batch.h; 0: no probe)"""
        return float(self._L.BrotliAmdBatchLastProbeMs(self._h)) if hasattr(self._L, "BrotliAmdBatchLastProbeMs") else 0.0

    def last_pool(self):
        """whether the last launch was a pool: blocks without a stream of their own help the largest stream still being decoded"""
        return bool(self._L.BrotliAmdBatchLastPool(self._h)) if hasattr(self._L, "BrotliAmdBatchLastPool") else False

    def decode_host(self, datas, out_caps, flags=FLAG_LARGE_WINDOW):
        """Host bytes in, (results, outputs) out: upload, decode, download."""
        n = len(datas)
        ins = [ctypes.create_string_buffer(bytes(d), max(1, len(d))) for d in datas]
        outs = [ctypes.create_string_buffer(max(1, c)) for c in out_caps]
        a_in = (ctypes.c_void_p * n)(*[ctypes.addressof(b) for b in ins])
        a_is = (ctypes.c_size_t * n)(*[len(d) for d in datas])
        a_out = (ctypes.c_void_p * n)(*[ctypes.addressof(b) for b in outs])
        a_oc = (ctypes.c_size_t * n)(*out_caps)
        res = (BatchResult * max(1, n))()
        if self._L.BrotliAmdBatchDecodeHost(self._h, n, a_in, a_is, a_out, a_oc, flags, res) != 0:
            raise RuntimeError("BrotliAmdBatchDecodeHost failed: " + last_error())
        self.n = n
        results = list(res)[:n]
        return results, [outs[i].raw[:min(results[i].decoded_size, out_caps[i])] for i in range(n)]

    def decode_host_raw(self, in_ptrs, in_sizes, out_ptrs, out_caps, flags=FLAG_LARGE_WINDOW):
        """BrotliAmdBatchDecodeHost on buffers the caller owns (host addresses as integers): nothing is copied on the Python side"""
        n = len(in_ptrs)
        a_in = (ctypes.c_void_p * n)(*in_ptrs)
        a_is = (ctypes.c_size_t * n)(*in_sizes)
        a_out = (ctypes.c_void_p * n)(*out_ptrs)
        a_oc = (ctypes.c_size_t * n)(*out_caps)
        res = (BatchResult * max(1, n))()
        if self._L.BrotliAmdBatchDecodeHost(self._h, n, a_in, a_is, a_out, a_oc, flags, res) != 0:
            raise RuntimeError("BrotliAmdBatchDecodeHost failed: " + last_error())
        self.n = n
        return list(res)[:n]


# ------------------------------------------------------------------ streaming state (src/ffi/mod.rs:390-463)
class DecoderState:
    """BrotliDecoderState through the C ABI."""

    def __init__(self, large_window=False):
        self._L = load_library()
        self._h = self._L.BrotliDecoderCreateInstance(None, None, None)
        if not self._h:
            raise MemoryError("BrotliDecoderCreateInstance")
        if large_window:
            self._L.BrotliDecoderSetParameter(self._h, PARAM_LARGE_WINDOW, 1)

    def close(self):
        if self._h:
            self._L.BrotliDecoderDestroyInstance(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decompress_stream(self, data: bytes, out_cap: int):
        """One BrotliDecoderDecompressStream call -> (result, bytes consumed, output bytes)."""
        L = self._L
        src = ctypes.create_string_buffer(bytes(data), max(1, len(data)))
        out = ctypes.create_string_buffer(max(1, out_cap))
        avail_in, avail_out = ctypes.c_size_t(len(data)), ctypes.c_size_t(out_cap)
        next_in, next_out = ctypes.c_void_p(ctypes.addressof(src)), ctypes.c_void_p(ctypes.addressof(out))
        total = ctypes.c_size_t(0)
        r = L.BrotliDecoderDecompressStream(self._h, ctypes.byref(avail_in), ctypes.byref(next_in), ctypes.byref(avail_out),
                                            ctypes.byref(next_out), ctypes.byref(total))
        return r, len(data) - avail_in.value, out.raw[:out_cap - avail_out.value]

    def error_code(self):
        return self._L.BrotliDecoderGetErrorCode(self._h)

    def error_string(self):
        return self._L.BrotliDecoderGetErrorString(self._h).decode()

    def is_finished(self):
        return bool(self._L.BrotliDecoderIsFinished(self._h))

    def is_used(self):
        return bool(self._L.BrotliDecoderIsUsed(self._h))

    def device_commands(self):
        """commands the device has decoded for this stream in all launches together (batch.h: BrotliAmdDecoderDeviceCommands)"""
        self._L.BrotliAmdDecoderDeviceCommands.restype = ctypes.c_uint64
        self._L.BrotliAmdDecoderDeviceCommands.argtypes = [ctypes.c_void_p]
        return int(self._L.BrotliAmdDecoderDeviceCommands(self._h))

    def has_more_output(self):
        return bool(self._L.BrotliDecoderHasMoreOutput(self._h))


class Decompressor:
    """Pull adapter, reference src/reader.rs:91-182 (`Decompressor<R>: io::Read`): wraps a readable object that
    yields compressed bytes; read(n) returns decompressed bytes, b'' at the end of the stream.  A failure of the
    decoder, or input that ends before the stream does, raises ValueError (io::ErrorKind::InvalidData /
    UnexpectedEof in the reference, reader.rs:335-346)."""

    def __init__(self, reader, buffer_size=4096, large_window=True):
        self._r = reader
        self._bufsize = max(1, buffer_size)
        self._st = DecoderState(large_window=large_window)  # native constructors default to large_window (state.rs:394)
        self._pending = b""
        self._done = False
        self._eof = False

    def read(self, n=-1):
        if n is None or n < 0:
            chunks = []
            while True:
                c = self.read(65536)
                if not c:
                    return b"".join(chunks)
                chunks.append(c)
        out = b""
        while not out and not self._done:
            if not self._pending and not self._eof:
                self._pending = self._r.read(self._bufsize) or b""  # an exception from the reader passes through
                if not self._pending:
                    self._eof = True
            r, used, out = self._st.decompress_stream(self._pending, n)
            self._pending = self._pending[used:]
            if r == RESULT_ERROR:
                raise ValueError("Invalid Data: " + self._st.error_string())
            if r == RESULT_SUCCESS:
                self._done = True
            elif r == RESULT_NEEDS_MORE_INPUT and self._eof and not out:
                raise ValueError("Unexpected EOF")
        return out

    def into_inner(self):
        return self._r


class DecompressorWriter:
    """Push adapter, reference src/writer.rs:104-199 (`DecompressorWriter<W>: io::Write`): write() takes compressed
    bytes and forwards decompressed bytes to the wrapped writer; close() drains and fails if the stream is
    incomplete (writer.rs:257-289)."""

    def __init__(self, writer, buffer_size=4096, large_window=True):
        self._w = writer
        self._bufsize = max(1, buffer_size)
        self._st = DecoderState(large_window=large_window)
        self._done = False

    def write(self, data: bytes):
        data = bytes(data)
        off = 0
        while True:
            r, used, out = self._st.decompress_stream(data[off:], self._bufsize)
            off += used
            if out:
                self._w.write(out)
            if r == RESULT_ERROR:
                raise ValueError("Invalid Data: " + self._st.error_string())
            if r == RESULT_SUCCESS:
                self._done = True
                return off  # bytes after the end of the stream are not consumed (writer.rs:383-398)
            if r == RESULT_NEEDS_MORE_INPUT:
                return len(data)

    def close(self):
        while not self._done:
            r, _, out = self._st.decompress_stream(b"", self._bufsize)
            if out:
                self._w.write(out)
            if r == RESULT_ERROR:
                raise ValueError("Invalid Data: " + self._st.error_string())
            if r == RESULT_SUCCESS:
                self._done = True
            elif r == RESULT_NEEDS_MORE_INPUT:
                raise ValueError("Unexpected EOF")
        return self._w
