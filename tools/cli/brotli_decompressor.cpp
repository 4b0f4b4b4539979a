// brotli-decompressor [in [out]] -- the reference's command-line tool (src/bin/brotli-decompressor.rs:325-359) on top
// of include/brotli/reader.hpp: stdin/stdout by default, 64 KiB buffers, "Invalid Data"/"Unexpected EOF" on stderr.
// (The reference's -dict= option is a custom dictionary: not part of the C ABI, not supported.)
#include <cstdio>
#include <cstring>
#include <vector>

#include "brotli/reader.hpp"

namespace {
struct FileSource {
  FILE* f;
  size_t read(uint8_t* dst, size_t n) { return fread(dst, 1, n, f); }
};
}  // namespace

int main(int argc, char** argv) {
  FILE* in = stdin;
  FILE* out = stdout;
  if (argc > 1 && std::strncmp(argv[1], "-dict=", 6) == 0) { std::fprintf(stderr, "custom dictionaries are not supported\n"); return 2; }
  if (argc > 1 && !(in = std::fopen(argv[1], "rb"))) { std::perror(argv[1]); return 1; }
  if (argc > 2 && !(out = std::fopen(argv[2], "wb"))) { std::perror(argv[2]); return 1; }
  try {
    brotli_amd::Decompressor<FileSource> r(FileSource{in}, 65536);
    std::vector<uint8_t> buf(65536);
    for (;;) {
      size_t n = r.read(buf.data(), buf.size());
      if (n == 0) break;
      if (std::fwrite(buf.data(), 1, n, out) != n) { std::perror("write"); return 1; }
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  if (out != stdout) std::fclose(out);
  return 0;
}
