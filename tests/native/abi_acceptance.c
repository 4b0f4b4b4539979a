/* C acceptance program for libbrotli_decompressor.so: the calls a C client of the reference makes (its c/main.c
 * exercises the same three things): one-shot decode of a known vector, one-shot with return info of a corrupt
 * vector (error code + text), and a streaming decode of stdin to stdout with custom allocator callbacks. */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "brotli/decode.h"

static int alloc_tag = 0, allocs = 0, frees = 0;
static void* my_alloc(void* opaque, size_t size) { assert(opaque == &alloc_tag); allocs++; return malloc(size); }
static void my_free(void* opaque, void* p) { assert(opaque == &alloc_tag); if (p) frees++; free(p); }

static const unsigned char kVector[] = {0x1b, 0x30, 0x00, 0xe0, 0x8d, 0xd4, 0x59, 0x2d, 0x39, 0x37, 0xb5, 0x02, 0x48, 0x10,
                                        0x95, 0x2a, 0x9a, 0xea, 0x42, 0x0e, 0x51, 0xa4, 0x16, 0xb9, 0xcb, 0xf5, 0xf8, 0x5c,
                                        0x64, 0xb9, 0x2f, 0xc9, 0x6a, 0x3f, 0xb1, 0xdc, 0xa8, 0xe0, 0x35, 0x07};
static const char kText[] = "THIS IS A TEST OF THE EMERGENCY BROADCAST SYSTEM\n";

int main(int argc, char** argv) {
  unsigned char out[256];
  size_t n = sizeof out;
  assert(BrotliDecoderVersion() == 0x1000f00);
  /* one-shot */
  assert(BrotliDecoderDecompress(sizeof kVector, kVector, &n, out) == BROTLI_DECODER_RESULT_SUCCESS);
  assert(n == strlen(kText) && memcmp(out, kText, n) == 0);
  BrotliDecoderReturnInfo info = BrotliDecoderDecompressWithReturnInfo(sizeof kVector, kVector, sizeof out, out);
  assert(info.result == BROTLI_DECODER_RESULT_SUCCESS && info.decoded_size == strlen(kText));
  /* output too small: the one-shot entry reports an error, nothing is lost silently */
  n = 10;
  assert(BrotliDecoderDecompress(sizeof kVector, kVector, &n, out) == BROTLI_DECODER_RESULT_ERROR);
  /* corrupt vector: byte 9 flipped -> context map repeat error, with its text */
  unsigned char bad[sizeof kVector];
  memcpy(bad, kVector, sizeof bad);
  bad[9] = 0xff;
  info = BrotliDecoderDecompressWithReturnInfo(sizeof bad, bad, sizeof out, out);
  assert(info.result == BROTLI_DECODER_RESULT_ERROR && info.code == BROTLI_DECODER_ERROR_FORMAT_CONTEXT_MAP_REPEAT);
  assert(strcmp(info.error, "ERROR_FORMAT_CONTEXT_MAP_REPEAT") == 0);
  /* same through an instance with custom allocators */
  BrotliDecoderState* st = BrotliDecoderCreateInstance(my_alloc, my_free, &alloc_tag);
  assert(st && allocs >= 1);
  {
    size_t avail_in = sizeof bad, avail_out = 0, total = 0;
    const unsigned char* ip = bad;
    unsigned char* op = out;
    assert(BrotliDecoderDecompressStream(st, &avail_in, &ip, &avail_out, &op, &total) == BROTLI_DECODER_RESULT_ERROR);
    assert(strcmp(BrotliDecoderGetErrorString(st), "ERROR_FORMAT_CONTEXT_MAP_REPEAT") == 0);
    assert(BrotliDecoderGetErrorCode(st) == BROTLI_DECODER_ERROR_FORMAT_CONTEXT_MAP_REPEAT);
    /* errors latch */
    avail_in = 0;
    assert(BrotliDecoderDecompressStream(st, &avail_in, &ip, &avail_out, &op, &total) == BROTLI_DECODER_RESULT_ERROR);
  }
  BrotliDecoderDestroyInstance(st);
  assert(frees >= 1);
  if (argc > 1 && strcmp(argv[1], "--stream") == 0) {
    /* stdin -> stdout, 4096-byte buffers on both sides */
    st = BrotliDecoderCreateInstance(my_alloc, my_free, &alloc_tag);
    unsigned char ibuf[4096], obuf[4096];
    size_t total = 0;
    BrotliDecoderResult r = BROTLI_DECODER_RESULT_NEEDS_MORE_INPUT;
    for (;;) {
      size_t avail_in = fread(ibuf, 1, sizeof ibuf, stdin);
      int eof = avail_in == 0;
      const unsigned char* ip = ibuf;
      for (;;) {
        unsigned char* op = obuf;
        size_t avail_out = sizeof obuf;
        r = BrotliDecoderDecompressStream(st, &avail_in, &ip, &avail_out, &op, &total);
        if (op != obuf) fwrite(obuf, 1, (size_t)(op - obuf), stdout);
        if (r != BROTLI_DECODER_RESULT_NEEDS_MORE_OUTPUT) break;
      }
      if (r == BROTLI_DECODER_RESULT_NEEDS_MORE_INPUT && eof) { fprintf(stderr, "Unexpected EOF\n"); return 2; }
      if (r == BROTLI_DECODER_RESULT_SUCCESS || r == BROTLI_DECODER_RESULT_ERROR) break;
    }
    assert(BrotliDecoderIsFinished(st) == (r == BROTLI_DECODER_RESULT_SUCCESS));
    BrotliDecoderDestroyInstance(st);
    if (r != BROTLI_DECODER_RESULT_SUCCESS) return 3;
  }
  fprintf(stderr, "abi acceptance ok\n");
  return 0;
}
