//! UNTESTED SOURCE -- this image has no Rust toolchain; nothing here has been compiled.
//!
//! Bindings to `libbrotli_decompressor.so` of this repository: the reference's own C ABI
//! (`src/ffi/mod.rs` of dropbox/rust-brotli-decompressor, header `c/brotli/decode.h`; here
//! `include/brotli/decode.h`) implemented on MI355X, and on top of it the two adapters a user of the
//! reference crate knows: [`Decompressor`] (`src/reader.rs:91-182`) and [`DecompressorWriter`]
//! (`src/writer.rs:104-199`), with the reference's constructor signatures and error kinds.
//!
//! What does not exist on this path: custom (LZ77 prefix) dictionaries (`new_with_custom_dict`), custom
//! allocators for the decoder's tables (they live in the GPU's LDS), `no_std`.
#![allow(non_camel_case_types, non_snake_case)]

use libc::{c_char, c_int, c_void, size_t};
use std::io::{self, Error, ErrorKind, Read, Write};

// ---------------------------------------------------------------------------------------------------
// raw ABI (include/brotli/decode.h; each item cites reference src/ffi/mod.rs line)
// ---------------------------------------------------------------------------------------------------
#[repr(C)]
pub struct BrotliDecoderState {
    _private: [u8; 0],
}

pub type brotli_alloc_func = Option<unsafe extern "C" fn(opaque: *mut c_void, size: size_t) -> *mut c_void>;
pub type brotli_free_func = Option<unsafe extern "C" fn(opaque: *mut c_void, address: *mut c_void)>;

/// src/ffi/interface.rs:17-22
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum BrotliDecoderResult {
    BROTLI_DECODER_RESULT_ERROR = 0,
    BROTLI_DECODER_RESULT_SUCCESS = 1,
    BROTLI_DECODER_RESULT_NEEDS_MORE_INPUT = 2,
    BROTLI_DECODER_RESULT_NEEDS_MORE_OUTPUT = 3,
}

/// src/ffi/interface.rs (BrotliDecoderParameter)
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum BrotliDecoderParameter {
    BROTLI_DECODER_PARAM_DISABLE_RING_BUFFER_REALLOCATION = 0,
    BROTLI_DECODER_PARAM_LARGE_WINDOW = 1,
}

/// src/state.rs:22-65; carried as a plain integer so that codes this crate does not name cannot be UB
pub type BrotliDecoderErrorCode = c_int;
pub const BROTLI_DECODER_NO_ERROR: c_int = 0;
pub const BROTLI_DECODER_SUCCESS: c_int = 1;
pub const BROTLI_DECODER_NEEDS_MORE_INPUT: c_int = 2;
pub const BROTLI_DECODER_NEEDS_MORE_OUTPUT: c_int = 3;
pub const BROTLI_DECODER_ERROR_INVALID_ARGUMENTS: c_int = -20;
pub const BROTLI_DECODER_ERROR_UNREACHABLE: c_int = -31;

/// src/huffman/mod.rs:28-33
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct HuffmanCode {
    pub value: u16,
    pub bits: u8,
}

/// src/lib.rs:336-342
#[repr(C)]
#[derive(Clone, Copy)]
pub struct BrotliDecoderReturnInfo {
    pub decoded_size: size_t,
    pub error_string: [c_char; 256],
    pub result: BrotliDecoderResult,
    pub error_code: BrotliDecoderErrorCode,
}

extern "C" {
    /// ffi/mod.rs:108
    pub fn BrotliDecoderCreateInstance(alloc_func: brotli_alloc_func, free_func: brotli_free_func, opaque: *mut c_void) -> *mut BrotliDecoderState;
    /// ffi/mod.rs:533
    pub fn BrotliDecoderDestroyInstance(state: *mut BrotliDecoderState);
    /// ffi/mod.rs:156
    pub fn BrotliDecoderSetParameter(state: *mut BrotliDecoderState, param: BrotliDecoderParameter, value: u32) -> c_int;
    /// ffi/mod.rs:263
    pub fn BrotliDecoderDecompress(encoded_size: size_t, encoded_buffer: *const u8, decoded_size: *mut size_t, decoded_buffer: *mut u8) -> BrotliDecoderResult;
    /// ffi/mod.rs:246
    pub fn BrotliDecoderDecompressWithReturnInfo(encoded_size: size_t, encoded_buffer: *const u8, decoded_size: size_t, decoded_buffer: *mut u8) -> BrotliDecoderReturnInfo;
    /// ffi/mod.rs:179
    pub fn BrotliDecoderDecompressPrealloc(
        encoded_size: size_t, encoded_buffer: *const u8, decoded_size: size_t, decoded_buffer: *mut u8,
        scratch_u8_size: size_t, scratch_u8_buffer: *mut u8, scratch_u32_size: size_t, scratch_u32_buffer: *mut u32,
        scratch_hc_size: size_t, scratch_hc_buffer: *mut HuffmanCode,
    ) -> BrotliDecoderReturnInfo;
    /// ffi/mod.rs:390
    pub fn BrotliDecoderDecompressStream(
        state: *mut BrotliDecoderState, available_in: *mut size_t, next_in: *mut *const u8,
        available_out: *mut size_t, next_out: *mut *mut u8, total_out: *mut size_t,
    ) -> BrotliDecoderResult;
    /// ffi/mod.rs:467
    pub fn BrotliDecoderDecompressStreaming(
        state: *mut BrotliDecoderState, available_in: *mut size_t, next_in: *const u8, available_out: *mut size_t, next_out: *mut u8,
    ) -> BrotliDecoderResult;
    /// ffi/mod.rs:493-530
    pub fn BrotliDecoderMallocU8(state: *mut BrotliDecoderState, size: size_t) -> *mut u8;
    pub fn BrotliDecoderFreeU8(state: *mut BrotliDecoderState, data: *mut u8, size: size_t);
    pub fn BrotliDecoderMallocUsize(state: *mut BrotliDecoderState, size: size_t) -> *mut size_t;
    pub fn BrotliDecoderFreeUsize(state: *mut BrotliDecoderState, data: *mut size_t, size: size_t);
    /// ffi/mod.rs:546-580
    pub fn BrotliDecoderHasMoreOutput(state: *const BrotliDecoderState) -> c_int;
    pub fn BrotliDecoderTakeOutput(state: *mut BrotliDecoderState, size: *mut size_t) -> *const u8;
    pub fn BrotliDecoderIsUsed(state: *const BrotliDecoderState) -> c_int;
    pub fn BrotliDecoderIsFinished(state: *const BrotliDecoderState) -> c_int;
    pub fn BrotliDecoderGetErrorCode(state: *const BrotliDecoderState) -> BrotliDecoderErrorCode;
    pub fn BrotliDecoderGetErrorString(state: *const BrotliDecoderState) -> *const c_char;
    /// ffi/mod.rs:582-590
    pub fn BrotliDecoderErrorString(c: BrotliDecoderErrorCode) -> *const c_char;
    pub fn BrotliDecoderVersion() -> u32;
}

// ---------------------------------------------------------------------------------------------------
// batches (include/brotli/batch.h; not in the reference: many independent streams in one launch, buffers in device or
// host memory) -- untested source like the rest of this crate
// ---------------------------------------------------------------------------------------------------
#[repr(C)]
pub struct BrotliAmdBatch {
    _private: [u8; 0],
}

/// batch.h: BrotliAmdResult
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct BrotliAmdResult {
    pub result: i32,
    pub error_code: i32,
    pub decoded_size: u64,
    pub consumed: u64,
    pub produced: u64,
    pub num_metablocks: u32,
    pub spilled_metablocks: u32,
    pub num_commands: u64,
    pub engine_commands: u32,
    pub reserved: u32,
}

pub const BROTLI_AMD_BATCH_LARGE_WINDOW: u32 = 1;
pub const BROTLI_AMD_BATCH_NO_CANNY: u32 = 2;
pub const BROTLI_AMD_BATCH_SPILL_IN_PLACE: u32 = 16;
pub const BROTLI_AMD_BATCH_EAGER_OUTPUT_LIMIT: u32 = 64;

extern "C" {
    pub fn BrotliAmdBatchCreate(max_streams: u32, lds_arena_bytes: u32, grid_blocks: u32) -> *mut BrotliAmdBatch;
    pub fn BrotliAmdBatchDestroy(batch: *mut BrotliAmdBatch);
    /// device pointers; asynchronous on `hip_stream` (a `hipStream_t`, may be null) until `BrotliAmdBatchWait`
    pub fn BrotliAmdBatchDecodeDevice(
        batch: *mut BrotliAmdBatch, n: u32, d_in: *const *const c_void, in_sizes: *const size_t, d_out: *const *mut c_void,
        out_caps: *const size_t, flags: u32, hip_stream: *mut c_void,
    ) -> c_int;
    pub fn BrotliAmdBatchRelaunch(batch: *mut BrotliAmdBatch, hip_stream: *mut c_void) -> c_int;
    pub fn BrotliAmdBatchWait(batch: *mut BrotliAmdBatch, results: *mut BrotliAmdResult) -> c_int;
    /// host pointers; synchronous
    pub fn BrotliAmdBatchDecodeHost(
        batch: *mut BrotliAmdBatch, n: u32, input: *const *const u8, in_sizes: *const size_t, output: *const *mut u8,
        out_caps: *const size_t, flags: u32, results: *mut BrotliAmdResult,
    ) -> c_int;
    pub fn BrotliAmdBatchLastKernelMs(batch: *mut BrotliAmdBatch) -> f32;
    pub fn BrotliAmdBatchLastSecondPassCount(batch: *mut BrotliAmdBatch) -> u32;
    pub fn BrotliAmdBatchLastGang(batch: *mut BrotliAmdBatch) -> u32;
    pub fn BrotliAmdBatchLastPool(batch: *mut BrotliAmdBatch) -> u32;
    pub fn BrotliAmdBatchLastProbeMs(batch: *mut BrotliAmdBatch) -> f32;
    pub fn BrotliAmdLastError() -> *const c_char;
    pub fn BrotliAmdLastNote() -> *const c_char;
}

/// Decodes `inputs[i]` into `outputs[i]` (host memory) in one launch; `None` when the device or the runtime failed
/// (`BrotliAmdLastError`).
pub fn decode_batch(inputs: &[&[u8]], outputs: &mut [&mut [u8]], flags: u32) -> Option<Vec<BrotliAmdResult>> {
    assert_eq!(inputs.len(), outputs.len());
    let n = inputs.len();
    let in_ptrs: Vec<*const u8> = inputs.iter().map(|s| s.as_ptr()).collect();
    let in_sizes: Vec<size_t> = inputs.iter().map(|s| s.len()).collect();
    let out_ptrs: Vec<*mut u8> = outputs.iter_mut().map(|s| s.as_mut_ptr()).collect();
    let out_caps: Vec<size_t> = outputs.iter().map(|s| s.len()).collect();
    let mut results = vec![BrotliAmdResult::default(); n];
    unsafe {
        let b = BrotliAmdBatchCreate(n.max(1) as u32, 0, 0);
        if b.is_null() {
            return None;
        }
        let rc = BrotliAmdBatchDecodeHost(b, n as u32, in_ptrs.as_ptr(), in_sizes.as_ptr(), out_ptrs.as_ptr(), out_caps.as_ptr(), flags, results.as_mut_ptr());
        BrotliAmdBatchDestroy(b);
        if rc != 0 {
            return None;
        }
    }
    Some(results)
}

// ---------------------------------------------------------------------------------------------------
// the crate's one-shot function (src/lib.rs:447-468): BrotliResult + bytes written
// ---------------------------------------------------------------------------------------------------
/// `brotli_decode(input, output)` of the reference: decodes a whole stream into `output`
pub fn brotli_decode(input: &[u8], output: &mut [u8]) -> BrotliDecoderReturnInfo {
    unsafe { BrotliDecoderDecompressWithReturnInfo(input.len(), input.as_ptr(), output.len(), output.as_mut_ptr()) }
}

// ---------------------------------------------------------------------------------------------------
// an owned decoder instance
// ---------------------------------------------------------------------------------------------------
struct State(*mut BrotliDecoderState);
// one state per stream, never used from two threads at once (CAllocator is Send in the reference)
unsafe impl Send for State {}
impl State {
    fn new() -> io::Result<State> {
        let p = unsafe { BrotliDecoderCreateInstance(None, None, std::ptr::null_mut()) };
        if p.is_null() {
            return Err(Error::new(ErrorKind::Other, "BrotliDecoderCreateInstance failed"));
        }
        // the reference's native constructors accept large-window streams (src/state.rs:394); the C ABI's default does not
        unsafe { BrotliDecoderSetParameter(p, BrotliDecoderParameter::BROTLI_DECODER_PARAM_LARGE_WINDOW, 1) };
        Ok(State(p))
    }
    /// one BrotliDecoderDecompressStream call -> (result, input consumed, output produced)
    fn step(&mut self, input: &[u8], output: &mut [u8]) -> (BrotliDecoderResult, usize, usize) {
        let mut avail_in = input.len();
        let mut next_in = input.as_ptr();
        let mut avail_out = output.len();
        let mut next_out = output.as_mut_ptr();
        let r = unsafe { BrotliDecoderDecompressStream(self.0, &mut avail_in, &mut next_in, &mut avail_out, &mut next_out, std::ptr::null_mut()) };
        (r, input.len() - avail_in, output.len() - avail_out)
    }
}
impl Drop for State {
    fn drop(&mut self) {
        unsafe { BrotliDecoderDestroyInstance(self.0) }
    }
}

// ---------------------------------------------------------------------------------------------------
// Decompressor<R: Read>  (src/reader.rs:91-182; read loop 295-350)
// ---------------------------------------------------------------------------------------------------
pub struct Decompressor<R: Read> {
    input: R,
    buffer: Vec<u8>,
    offset: usize,
    len: usize,
    state: State,
    done: bool,
}

impl<R: Read> Decompressor<R> {
    /// `buffer_size` bytes of internal input buffer, 4096 when 0 (reader.rs:106-118)
    pub fn new(r: R, buffer_size: usize) -> Self {
        Decompressor {
            input: r,
            buffer: vec![0u8; if buffer_size == 0 { 4096 } else { buffer_size }],
            offset: 0,
            len: 0,
            state: State::new().expect("no HIP device: this decode path has no CPU fallback"),
            done: false,
        }
    }
    pub fn get_ref(&self) -> &R {
        &self.input
    }
    pub fn get_mut(&mut self) -> &mut R {
        &mut self.input
    }
    pub fn into_inner(self) -> R {
        self.input
    }
}

impl<R: Read> Read for Decompressor<R> {
    /// Ok(n > 0) while the stream yields bytes, Ok(0) once it is complete, InvalidData for a decoder error or for
    /// bytes behind the end of the stream that are still in the buffer, UnexpectedEof when the source ends first
    /// (reader.rs:280-350)
    fn read(&mut self, buf: &mut [u8]) -> io::Result<usize> {
        if buf.is_empty() {
            return Ok(0);
        }
        loop {
            let (r, used, got) = {
                let (b, st) = (&self.buffer[self.offset..self.len], &mut self.state);
                st.step(b, buf)
            };
            self.offset += used;
            match r {
                BrotliDecoderResult::BROTLI_DECODER_RESULT_ERROR => return Err(Error::new(ErrorKind::InvalidData, "Invalid Data")),
                BrotliDecoderResult::BROTLI_DECODER_RESULT_SUCCESS => {
                    if got == 0 {
                        if !self.done {
                            self.done = true;
                        } else if self.len != self.offset {
                            return Err(Error::new(ErrorKind::InvalidData, "Invalid Data"));  // did not consume the entire input
                        }
                    }
                    return Ok(got);
                }
                BrotliDecoderResult::BROTLI_DECODER_RESULT_NEEDS_MORE_OUTPUT => return Ok(got),
                BrotliDecoderResult::BROTLI_DECODER_RESULT_NEEDS_MORE_INPUT => {
                    if got != 0 {
                        return Ok(got);  // hand over what there is rather than risk an error of the source (reader.rs:305-311)
                    }
                    if self.offset == self.len {
                        self.offset = 0;
                        self.len = 0;
                    } else if self.len == self.buffer.len() {
                        // a full buffer with an unconsumed tail (the decoder keeps less than eight bytes back): move the tail to
                        // the front so that there is room to read into (reader.rs:258-269, copy_to_front)
                        self.buffer.copy_within(self.offset..self.len, 0);
                        self.len -= self.offset;
                        self.offset = 0;
                    }
                    let n = self.input.read(&mut self.buffer[self.len..])?;
                    if n == 0 {
                        return Err(Error::new(ErrorKind::UnexpectedEof, "Unexpected EOF"));
                    }
                    self.len += n;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// DecompressorWriter<W: Write>  (src/writer.rs:104-199; write loop 337-383)
// ---------------------------------------------------------------------------------------------------
pub struct DecompressorWriter<W: Write> {
    output: Option<W>,
    buffer: Vec<u8>,
    state: State,
    finished: bool,
}

impl<W: Write> DecompressorWriter<W> {
    pub fn new(w: W, buffer_size: usize) -> Self {
        DecompressorWriter {
            output: Some(w),
            buffer: vec![0u8; if buffer_size == 0 { 4096 } else { buffer_size }],
            state: State::new().expect("no HIP device: this decode path has no CPU fallback"),
            finished: false,
        }
    }
    pub fn get_ref(&self) -> &W {
        self.output.as_ref().unwrap()
    }
    pub fn get_mut(&mut self) -> &mut W {
        self.output.as_mut().unwrap()
    }
    /// drains what the decoder still holds; an incomplete stream is an error (writer.rs:257-289)
    pub fn close(&mut self) -> io::Result<()> {
        loop {
            let (r, _, got) = self.state.step(&[], &mut self.buffer);
            if got != 0 {
                self.output.as_mut().unwrap().write_all(&self.buffer[..got])?;
            }
            match r {
                BrotliDecoderResult::BROTLI_DECODER_RESULT_SUCCESS => {
                    self.finished = true;
                    return Ok(());
                }
                BrotliDecoderResult::BROTLI_DECODER_RESULT_NEEDS_MORE_OUTPUT => continue,
                BrotliDecoderResult::BROTLI_DECODER_RESULT_NEEDS_MORE_INPUT => return Err(Error::new(ErrorKind::UnexpectedEof, "Unexpected EOF")),
                BrotliDecoderResult::BROTLI_DECODER_RESULT_ERROR => return Err(Error::new(ErrorKind::InvalidData, "Invalid Data")),
            }
        }
    }
    /// Ok(writer) when the stream was complete, Err(writer) otherwise (writer.rs:297-303)
    pub fn into_inner(mut self) -> Result<W, W> {
        let ok = self.close().is_ok();
        let w = self.output.take().unwrap();
        if ok { Ok(w) } else { Err(w) }
    }
}

impl<W: Write> Write for DecompressorWriter<W> {
    /// feeds `buf` to the decoder and passes what comes out on; bytes behind the end of the stream are InvalidData
    /// (writer.rs:337-368)
    fn write(&mut self, buf: &[u8]) -> io::Result<usize> {
        let mut off = 0usize;
        loop {
            let (r, used, got) = self.state.step(&buf[off..], &mut self.buffer);
            off += used;
            if got != 0 {
                self.output.as_mut().unwrap().write_all(&self.buffer[..got])?;
            }
            match r {
                BrotliDecoderResult::BROTLI_DECODER_RESULT_NEEDS_MORE_INPUT => {
                    debug_assert_eq!(off, buf.len());
                    return Ok(buf.len());
                }
                BrotliDecoderResult::BROTLI_DECODER_RESULT_NEEDS_MORE_OUTPUT => continue,
                BrotliDecoderResult::BROTLI_DECODER_RESULT_SUCCESS => {
                    if off != buf.len() {
                        return Err(Error::new(ErrorKind::InvalidData, "Invalid Data"));
                    }
                    self.finished = true;
                    return Ok(buf.len());
                }
                BrotliDecoderResult::BROTLI_DECODER_RESULT_ERROR => return Err(Error::new(ErrorKind::InvalidData, "Invalid Data")),
            }
        }
    }
    fn flush(&mut self) -> io::Result<()> {
        self.output.as_mut().unwrap().flush()
    }
}

impl<W: Write> Drop for DecompressorWriter<W> {
    fn drop(&mut self) {
        if self.output.is_some() && !self.finished {
            let _ = self.close();  // (writer.rs:305-312: errors of a drop are swallowed)
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// what a maintainer would run first (the reference's own vectors: c/main.c:17-33, src/reader.rs:359)
// ---------------------------------------------------------------------------------------------------
#[cfg(test)]
mod tests {
    use super::*;

    const BROADCAST: [u8; 40] = [
        0x1b, 0x30, 0x00, 0xe0, 0x8d, 0xd4, 0x59, 0x2d, 0x39, 0x37, 0xb5, 0x02, 0x48, 0x10, 0x95, 0x2a, 0x9a, 0xea, 0x42, 0x0e, 0x51, 0xa4, 0x16, 0xb9, 0xcb, 0xf5,
        0xf8, 0x5c, 0x64, 0xb9, 0x2f, 0xc9, 0x6a, 0x3f, 0xb1, 0xdc, 0xa8, 0xe0, 0x35, 0x07,
    ];

    #[test]
    fn one_shot() {
        let mut out = [0u8; 256];
        let info = brotli_decode(&BROADCAST, &mut out);
        assert_eq!(info.result, BrotliDecoderResult::BROTLI_DECODER_RESULT_SUCCESS);
        assert_eq!(&out[..info.decoded_size], &b"THIS IS A TEST OF THE EMERGENCY BROADCAST SYSTEM\n"[..]);
    }

    #[test]
    fn reader_and_writer() {
        let mut r = Decompressor::new(&BROADCAST[..], 7);
        let mut got = Vec::new();
        r.read_to_end(&mut got).unwrap();
        assert_eq!(&got[..], &b"THIS IS A TEST OF THE EMERGENCY BROADCAST SYSTEM\n"[..]);
        let mut w = DecompressorWriter::new(Vec::new(), 5);
        w.write_all(&BROADCAST).unwrap();
        assert_eq!(&w.into_inner().unwrap()[..], &got[..]);
    }

    #[test]
    fn prealloc_scratch_exhaustion_is_unreachable() {
        let info = unsafe {
            BrotliDecoderDecompressPrealloc(0, std::ptr::null(), 0, std::ptr::null_mut(), 0, std::ptr::null_mut(), 0, std::ptr::null_mut(), 0, std::ptr::null_mut())
        };
        assert_eq!(info.decoded_size, 0);
        assert_eq!(info.error_code, BROTLI_DECODER_ERROR_UNREACHABLE);
    }
}
