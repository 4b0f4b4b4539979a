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
