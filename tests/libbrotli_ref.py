"""ctypes binding of Google's libbrotlidec / libbrotlienc 1.0.9 when the image has them.

Not the reference and not the oracle: an independent implementation (the C decoder the reference is a
port of) used for differential checks of error codes, and the encoder used to make synthetic streams.
Every user must cope with `available() == False`.
"""
import ctypes

_dec = _enc = None
for _cand in ("libbrotlidec.so.1", "/usr/lib/x86_64-linux-gnu/libbrotlidec.so.1", "/opt/conda/lib/libbrotlidec.so.1"):
    try:
        _dec = ctypes.CDLL(_cand)
        break
    except OSError:
        pass
for _cand in ("libbrotlienc.so.1", "/usr/lib/x86_64-linux-gnu/libbrotlienc.so.1", "/opt/conda/lib/libbrotlienc.so.1"):
    try:
        _enc = ctypes.CDLL(_cand)
        break
    except OSError:
        pass

if _dec is not None:
    _dec.BrotliDecoderCreateInstance.restype = ctypes.c_void_p
    _dec.BrotliDecoderCreateInstance.argtypes = [ctypes.c_void_p] * 3
    _dec.BrotliDecoderDestroyInstance.argtypes = [ctypes.c_void_p]
    _dec.BrotliDecoderSetParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32]
    _dec.BrotliDecoderGetErrorCode.argtypes = [ctypes.c_void_p]
    _dec.BrotliDecoderGetErrorCode.restype = ctypes.c_int
    _dec.BrotliDecoderDecompressStream.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t),
                                                   ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                                                   ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
if _enc is not None:
    _enc.BrotliEncoderCompress.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p,
                                           ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
    _enc.BrotliEncoderMaxCompressedSize.argtypes = [ctypes.c_size_t]
    _enc.BrotliEncoderMaxCompressedSize.restype = ctypes.c_size_t


def available():
    return _dec is not None


def encoder_available():
    return _enc is not None


def decode(data: bytes, out_cap: int, large_window: bool = True):
    """one BrotliDecoderDecompressStream call with all input -> (result, error_code, output, consumed)"""
    st = _dec.BrotliDecoderCreateInstance(None, None, None)
    if large_window:
        _dec.BrotliDecoderSetParameter(st, 1, 1)
    inbuf = ctypes.create_string_buffer(data, max(1, len(data)))
    out = ctypes.create_string_buffer(max(1, out_cap))
    avail_in = ctypes.c_size_t(len(data))
    next_in = ctypes.c_void_p(ctypes.addressof(inbuf))
    avail_out = ctypes.c_size_t(out_cap)
    next_out = ctypes.c_void_p(ctypes.addressof(out))
    total = ctypes.c_size_t(0)
    res = _dec.BrotliDecoderDecompressStream(st, ctypes.byref(avail_in), ctypes.byref(next_in), ctypes.byref(avail_out),
                                             ctypes.byref(next_out), ctypes.byref(total))
    code = _dec.BrotliDecoderGetErrorCode(st)
    _dec.BrotliDecoderDestroyInstance(st)
    produced = out_cap - avail_out.value
    return res, code, out.raw[:produced], len(data) - avail_in.value


def encode(data: bytes, quality: int = 5, lgwin: int = 22, mode: int = 0) -> bytes:
    cap = _enc.BrotliEncoderMaxCompressedSize(len(data)) or (len(data) + 1024)
    out = ctypes.create_string_buffer(cap)
    n = ctypes.c_size_t(cap)
    src = ctypes.create_string_buffer(data, max(1, len(data)))
    ok = _enc.BrotliEncoderCompress(quality, lgwin, mode, len(data), src, ctypes.byref(n), out)
    if not ok:
        raise RuntimeError("BrotliEncoderCompress failed")
    return out.raw[:n.value]


# ---- streaming encoder with explicit parameters (brotli/encode.h of 1.0.9) ----
PARAM_MODE, PARAM_QUALITY, PARAM_LGWIN, PARAM_LGBLOCK, PARAM_NO_LITERAL_CONTEXT, PARAM_SIZE_HINT, PARAM_LARGE_WINDOW, PARAM_NPOSTFIX, \
    PARAM_NDIRECT = range(9)
OP_PROCESS, OP_FLUSH, OP_FINISH, OP_EMIT_METADATA = range(4)

if _enc is not None:
    _enc.BrotliEncoderCreateInstance.restype = ctypes.c_void_p
    _enc.BrotliEncoderCreateInstance.argtypes = [ctypes.c_void_p] * 3
    _enc.BrotliEncoderDestroyInstance.argtypes = [ctypes.c_void_p]
    _enc.BrotliEncoderSetParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32]
    _enc.BrotliEncoderCompressStream.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_void_p),
                                                 ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    _enc.BrotliEncoderIsFinished.argtypes = [ctypes.c_void_p]
    _enc.BrotliEncoderHasMoreOutput.argtypes = [ctypes.c_void_p]


def encode_stream(chunks, params=None, ops=None) -> bytes:
    """Compress `chunks` (list of bytes) with the streaming encoder.  params: {PARAM_*: value}.  ops: one operation per
    chunk (OP_PROCESS / OP_FLUSH / OP_EMIT_METADATA; a metadata chunk must be at most 16 bytes); the stream is always
    finished at the end.  Flushes give several metablocks, metadata chunks give metadata metablocks."""
    st = _enc.BrotliEncoderCreateInstance(None, None, None)
    try:
        for k, v in (params or {}).items():
            if not _enc.BrotliEncoderSetParameter(st, k, v):
                raise ValueError("BrotliEncoderSetParameter(%d, %d) refused" % (k, v))
        out = bytearray()
        obuf = ctypes.create_string_buffer(1 << 16)

        def pump(op, data):
            ibuf = ctypes.create_string_buffer(data, max(1, len(data)))
            avail_in = ctypes.c_size_t(len(data))
            next_in = ctypes.c_void_p(ctypes.addressof(ibuf))
            while True:
                avail_out = ctypes.c_size_t(len(obuf))
                next_out = ctypes.c_void_p(ctypes.addressof(obuf))
                if not _enc.BrotliEncoderCompressStream(st, op, ctypes.byref(avail_in), ctypes.byref(next_in), ctypes.byref(avail_out),
                                                        ctypes.byref(next_out), None):
                    raise RuntimeError("BrotliEncoderCompressStream failed")
                out.extend(obuf.raw[:len(obuf) - avail_out.value])
                if avail_in.value == 0 and not _enc.BrotliEncoderHasMoreOutput(st):
                    break
        ops = ops or [OP_PROCESS] * len(chunks)
        for c, op in zip(chunks, ops):
            pump(op, c)
        pump(OP_FINISH, b"")
        return bytes(out)
    finally:
        _enc.BrotliEncoderDestroyInstance(st)


class StreamDecoder:
    """A libbrotlidec instance fed call by call: step(pending bytes, out_cap) -> (result, consumed, produced bytes)"""

    def __init__(self, large_window=True):
        self.st = _dec.BrotliDecoderCreateInstance(None, None, None)
        if large_window:
            _dec.BrotliDecoderSetParameter(self.st, 1, 1)

    def step(self, data: bytes, out_cap: int):
        inbuf = ctypes.create_string_buffer(data, max(1, len(data)))
        out = ctypes.create_string_buffer(max(1, out_cap))
        avail_in = ctypes.c_size_t(len(data)); next_in = ctypes.c_void_p(ctypes.addressof(inbuf))
        avail_out = ctypes.c_size_t(out_cap); next_out = ctypes.c_void_p(ctypes.addressof(out))
        total = ctypes.c_size_t(0)
        res = _dec.BrotliDecoderDecompressStream(self.st, ctypes.byref(avail_in), ctypes.byref(next_in), ctypes.byref(avail_out),
                                                 ctypes.byref(next_out), ctypes.byref(total))
        return res, len(data) - avail_in.value, out.raw[:out_cap - avail_out.value]

    def close(self):
        if self.st:
            _dec.BrotliDecoderDestroyInstance(self.st)
            self.st = None
