// brotli/reader.hpp -- pull adapter over the C ABI, the C++ counterpart of the reference's
// `Decompressor<R: io::Read>` (src/reader.rs:91-182, 258-350): wraps a source of compressed bytes and yields
// decompressed bytes through read().  Header-only; link against libbrotli_decompressor.so.
//
// Semantics kept from the reference: the internal input buffer has a caller-chosen size (reader.rs:106-118);
// read() returns 0 only at the end of the stream; a decoder failure or input that ends before the stream does is
// an error (io::ErrorKind::InvalidData / UnexpectedEof, reader.rs:335-346); bytes after the end of the stream are
// left unread in the buffer (reader.rs:353-421); an error of the source passes through and decoding can be resumed
// afterwards (error_handling_tests.rs:34-44).
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "brotli/decode.h"

namespace brotli_amd {

struct InvalidData : std::runtime_error { using std::runtime_error::runtime_error; };
struct UnexpectedEof : std::runtime_error { using std::runtime_error::runtime_error; };

// R needs: size_t read(uint8_t* dst, size_t n)   -- 0 = end of input, may throw
// every call of the streaming ABI is a kernel launch: a buffer of this size amortises it (the default here; the reference's own default is 4096)
#ifndef BROTLI_AMD_RECOMMENDED_BUFFER
#define BROTLI_AMD_RECOMMENDED_BUFFER
constexpr size_t kRecommendedBufferSize = 1u << 20;
#endif
template <class R>
class Decompressor {
 public:
  Decompressor(R source, size_t buffer_size = kRecommendedBufferSize, bool large_window = true)  // (0: the reference's 4096, src/reader.rs)
      : src_(std::move(source)), buf_(buffer_size ? buffer_size : 4096u), state_(BrotliDecoderCreateInstance(nullptr, nullptr, nullptr)) {
    if (!state_) throw std::bad_alloc();
    // native constructors of the reference accept large-window streams (src/state.rs:394)
    if (large_window) BrotliDecoderSetParameter(state_, BROTLI_DECODER_PARAM_LARGE_WINDOW, 1);
  }
  Decompressor(const Decompressor&) = delete;
  Decompressor& operator=(const Decompressor&) = delete;
  ~Decompressor() { BrotliDecoderDestroyInstance(state_); }

  size_t read(uint8_t* dst, size_t n) {
    if (n == 0 || done_) return 0;
    for (;;) {
      if (begin_ == end_ && !eof_) {
        begin_ = 0;
        end_ = src_.read(buf_.data(), buf_.size());  // exceptions of the source pass through
        if (end_ == 0) eof_ = true;
      }
      size_t avail_in = end_ - begin_, avail_out = n;
      const uint8_t* next_in = buf_.data() + begin_;
      uint8_t* next_out = dst;
      BrotliDecoderResult r = BrotliDecoderDecompressStream(state_, &avail_in, &next_in, &avail_out, &next_out, nullptr);
      begin_ = end_ - avail_in;
      size_t produced = n - avail_out;
      if (r == BROTLI_DECODER_RESULT_ERROR) throw InvalidData(std::string("Invalid Data: ") + BrotliDecoderGetErrorString(state_));
      if (r == BROTLI_DECODER_RESULT_SUCCESS) { done_ = true; return produced; }
      if (produced) return produced;
      if (r == BROTLI_DECODER_RESULT_NEEDS_MORE_INPUT && eof_ && begin_ == end_) throw UnexpectedEof("Unexpected EOF");
    }
  }
  // compressed bytes read from the source but not consumed by the decoder (trailing data after the stream)
  size_t unread() const { return end_ - begin_; }
  R& get_ref() { return src_; }
  R into_inner() { return std::move(src_); }

 private:
  R src_;
  std::vector<uint8_t> buf_;
  size_t begin_ = 0, end_ = 0;
  bool eof_ = false, done_ = false;
  BrotliDecoderState* state_;
};

}  // namespace brotli_amd
