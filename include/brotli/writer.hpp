// brotli/writer.hpp -- push adapter over the C ABI, the C++ counterpart of the reference's
// `DecompressorWriter<W: io::Write>` (src/writer.rs:104-199, 257-368): compressed bytes go in through write(),
// decompressed bytes are forwarded to the wrapped sink.  Header-only; link against libbrotli_decompressor.so.
//
// Semantics kept from the reference: write() reports how many compressed bytes it took -- bytes after the end of
// the stream are not taken (writer.rs:383-398); close() drains what is left and fails with UnexpectedEof when the
// stream is incomplete (writer.rs:257-289); a decoder failure is InvalidData.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "brotli/decode.h"
#include "brotli/reader.hpp"

namespace brotli_amd {

// W needs: void write_all(const uint8_t* p, size_t n)   -- may throw
// every call of the streaming ABI is a kernel launch: a buffer of this size amortises it (the default here; the reference's own default is 4096)
#ifndef BROTLI_AMD_RECOMMENDED_BUFFER
#define BROTLI_AMD_RECOMMENDED_BUFFER
constexpr size_t kRecommendedBufferSize = 1u << 20;
#endif
template <class W>
class DecompressorWriter {
 public:
  DecompressorWriter(W sink, size_t buffer_size = kRecommendedBufferSize, bool large_window = true)  // (0: the reference's 4096, src/reader.rs)
      : sink_(std::move(sink)), buf_(buffer_size ? buffer_size : 4096u), state_(BrotliDecoderCreateInstance(nullptr, nullptr, nullptr)) {
    if (!state_) throw std::bad_alloc();
    if (large_window) BrotliDecoderSetParameter(state_, BROTLI_DECODER_PARAM_LARGE_WINDOW, 1);
  }
  DecompressorWriter(const DecompressorWriter&) = delete;
  DecompressorWriter& operator=(const DecompressorWriter&) = delete;
  ~DecompressorWriter() { BrotliDecoderDestroyInstance(state_); }

  size_t write(const uint8_t* p, size_t n) {
    size_t avail_in = n;
    const uint8_t* next_in = p;
    for (;;) {
      size_t avail_out = buf_.size();
      uint8_t* next_out = buf_.data();
      BrotliDecoderResult r = BrotliDecoderDecompressStream(state_, &avail_in, &next_in, &avail_out, &next_out, nullptr);
      if (avail_out != buf_.size()) sink_.write_all(buf_.data(), buf_.size() - avail_out);
      if (r == BROTLI_DECODER_RESULT_ERROR) throw InvalidData(std::string("Invalid Data: ") + BrotliDecoderGetErrorString(state_));
      if (r == BROTLI_DECODER_RESULT_SUCCESS) { done_ = true; return n - avail_in; }
      if (r == BROTLI_DECODER_RESULT_NEEDS_MORE_INPUT) return n - avail_in;
    }
  }
  W close() {
    while (!done_) {
      size_t avail_in = 0, avail_out = buf_.size();
      const uint8_t* next_in = nullptr;
      uint8_t* next_out = buf_.data();
      BrotliDecoderResult r = BrotliDecoderDecompressStream(state_, &avail_in, &next_in, &avail_out, &next_out, nullptr);
      if (avail_out != buf_.size()) sink_.write_all(buf_.data(), buf_.size() - avail_out);
      if (r == BROTLI_DECODER_RESULT_ERROR) throw InvalidData(std::string("Invalid Data: ") + BrotliDecoderGetErrorString(state_));
      if (r == BROTLI_DECODER_RESULT_SUCCESS) done_ = true;
      else if (r == BROTLI_DECODER_RESULT_NEEDS_MORE_INPUT) throw UnexpectedEof("Unexpected EOF");
    }
    return std::move(sink_);
  }
  W& get_ref() { return sink_; }

 private:
  W sink_;
  std::vector<uint8_t> buf_;
  bool done_ = false;
  BrotliDecoderState* state_;
};

}  // namespace brotli_amd
