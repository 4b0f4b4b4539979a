"""The per-call contract of the reference's resumable driver, restated on top of the oracle (test infrastructure only).

`BrotliDecompressStream` (reference src/decode.rs:2779-2911) is a state machine that decodes into a ring buffer and stops
for exactly three reasons; what a call returns follows from WHERE the decode stands, which the one-shot oracle knows
(oracle/brotli_oracle.c: brotli_oracle_decode_trace gives, after every unit that produces output, the stream's bit
position and the bytes decoded so far):

* NEEDS_MORE_INPUT (decode.rs:2835-2896): the next unit's bits are not all there.  Whatever is decoded and not delivered
  yet is written out as far as the output has room (WriteRingBuffer with force, decode.rs:2837-2846; a partial write is
  not an error here), then the unread tail of the input -- less than eight bytes -- is copied into the state's own buffer:
  the call consumes ALL of its input.
* NEEDS_MORE_OUTPUT (decode.rs:1693-1738, 3299-3344, 3382-3397): the ring buffer is full (the decoded position reaches a
  multiple of the ring size) or the stream is complete, and what has to be written does not fit.  The bit reader gives
  the whole bytes it has not used back (BrotliBitReaderUnload, decode.rs:2909-2911): the call consumes up to the byte
  that holds the last bit of the last unit parsed.
* SUCCESS: everything delivered; consumed up to the end of the stream (trailing bytes stay with the caller).

Only streams that decode without a format error are modelled (complete ones, or cut short: those end in NEEDS_MORE_INPUT).
"""
import ctypes

import oracle_lib as oracle

RESULT_ERROR, RESULT_SUCCESS, RESULT_NEEDS_MORE_INPUT, RESULT_NEEDS_MORE_OUTPUT = 0, 1, 2, 3


def trace(data: bytes, flags: int = oracle.FLAG_LARGE_WINDOW):
    """-> (info, units [(bit position behind the unit, bytes decoded)], ring size)"""
    L = oracle.lib()
    L.brotli_oracle_decode_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32,
                                             ctypes.POINTER(oracle.OracleInfo), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                             ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_uint64)]
    cap_out = 1 << 16
    while True:  # (room for all of it: with less, a ring that fills ends the decode in NEEDS_MORE_OUTPUT)
        info0, _ = oracle.decode(data, cap_out, flags)
        if info0.result != RESULT_NEEDS_MORE_OUTPUT or cap_out >= (1 << 31):
            break
        cap_out <<= 2
    cap_out = max(1, int(info0.produced) + 64)
    inbuf = (ctypes.c_uint8 * max(1, len(data))).from_buffer_copy(data.ljust(1, b"\0"))
    out = (ctypes.c_uint8 * cap_out)()
    info = oracle.OracleInfo()
    n = ctypes.c_size_t(0)
    rb = ctypes.c_uint64(0)
    L.brotli_oracle_decode_trace(inbuf, len(data), out, cap_out, flags, ctypes.byref(info), None, None, 0, ctypes.byref(n), ctypes.byref(rb))
    cap = n.value
    bits = (ctypes.c_uint64 * max(1, cap))()
    ps = (ctypes.c_uint64 * max(1, cap))()
    L.brotli_oracle_decode_trace(inbuf, len(data), out, cap_out, flags, ctypes.byref(info), bits, ps, cap, ctypes.byref(n), ctypes.byref(rb))
    return info, [(int(bits[i]), int(ps[i])) for i in range(cap)], int(rb.value)


class ReferenceStream:
    """One stream fed call by call: call(avail_in, avail_out) -> (result, consumed, produced).  The caller presents the
    stream's bytes in order and presents unconsumed bytes again, as the reference's harness does."""

    def __init__(self, data: bytes, flags: int = oracle.FLAG_LARGE_WINDOW):
        self.info, self.units, self.rb = trace(data, flags)
        assert self.info.result in (RESULT_SUCCESS, RESULT_NEEDS_MORE_INPUT), "only streams without format errors are modelled"
        self.complete = self.info.result == RESULT_SUCCESS
        self.end_bytes = int(self.info.consumed)   # a complete stream ends here
        self.size = len(data)
        self.fed = 0            # bytes of the stream the decoder has consumed
        self.k = 0              # units applied
        self.P = 0              # bytes decoded (into the ring)
        self.delivered = 0
        self.owed_to = None     # a ring boundary (or the end) everything up to which has to be written before decoding goes on
        self.stop_bits = 0      # bit position behind the last unit parsed
        self.finished = False

    def _boundary_after(self, p):
        return (p // self.rb + 1) * self.rb if self.rb else None

    def call(self, avail_in: int, avail_out: int):
        n_avail = self.fed + avail_in      # bytes of the stream the decoder can see
        room = avail_out
        produced = 0
        while True:
            if self.owed_to is not None:
                w = min(self.owed_to - self.delivered, room)
                self.delivered += w; room -= w; produced += w
                if self.delivered < self.owed_to:
                    used = max(0, (self.stop_bits + 7) // 8 - self.fed)
                    self.fed += used
                    return RESULT_NEEDS_MORE_OUTPUT, used, produced
                self.owed_to = None
                if self.finished:
                    used = max(0, self.end_bytes - self.fed)
                    self.fed += used
                    return RESULT_SUCCESS, used, produced
            if self.k == len(self.units):
                if self.complete and self.end_bytes <= n_avail:
                    # DONE: the last write (decode.rs:3382-3397)
                    self.finished = True
                    self.stop_bits = self.end_bytes * 8
                    self.owed_to = self.P
                    continue
                break  # the rest of the stream (headers, padding) is not there yet
            bits_end, p_after = self.units[self.k]
            if bits_end > 8 * n_avail:
                break
            nb = self._boundary_after(self.P)
            if nb is not None and p_after >= nb and self.rb == (1 << self.info.window_bits):
                # the ring fills inside this unit: written out before the decoder goes on (COMMAND_*_WRITE states)
                self.P = nb
                self.stop_bits = bits_end
                self.owed_to = nb
                if p_after == nb:
                    self.k += 1
                continue
            self.P = p_after
            self.stop_bits = bits_end
            self.k += 1
        # NEEDS_MORE_INPUT: what there is goes out as far as there is room, the whole input is taken
        w = min(self.P - self.delivered, room)
        self.delivered += w; produced += w
        self.fed = n_avail
        return RESULT_NEEDS_MORE_INPUT, avail_in, produced


def run_schedule(step, data: bytes, in_chunk: int, out_chunk: int, max_calls=2000000, drain=False):
    """The loop of the reference's decompress_internal (src/bin/integration_tests.rs:122-216) around `step(pending bytes,
    out_chunk) -> (result, consumed, produced)`: new input only on NEEDS_MORE_INPUT with nothing pending.
    -> list of (result, consumed, produced)"""
    seq = []
    pos, pending = 0, b""
    result = RESULT_NEEDS_MORE_INPUT
    while len(seq) < max_calls:
        if not pending and result == RESULT_NEEDS_MORE_INPUT:
            if pos >= len(data):
                if drain and seq and seq[-1][2] != 0:   # (no more input: calls without input while they still deliver something)
                    result, used, got = step(b"", out_chunk)
                    seq.append((result, used, got))
                    continue
                break
            pending = data[pos:pos + in_chunk]
            pos += len(pending)
        result, used, got = step(pending, out_chunk)
        pending = pending[used:]
        seq.append((result, used, got))
        if result in (RESULT_SUCCESS, RESULT_ERROR):
            break
    return seq
