// Reader / writer adapters (include/brotli/reader.hpp, writer.hpp) on files given on the command line:
//   wrappers_test <compressed> <expected> <read_size> <buffer_size>
// mirrors the reference's wrapper tests (src/bin/integration_tests.rs:294-415): odd read sizes, small internal
// buffers, truncated input must fail, trailing bytes are left unread.
#include <cassert>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <vector>

#include "brotli/reader.hpp"
#include "brotli/writer.hpp"

struct MemSource {
  const std::vector<uint8_t>* v; size_t pos = 0;
  size_t read(uint8_t* dst, size_t n) { size_t k = std::min(n, v->size() - pos); memcpy(dst, v->data() + pos, k); pos += k; return k; }
};
struct MemSink {
  std::vector<uint8_t> v;
  void write_all(const uint8_t* p, size_t n) { v.insert(v.end(), p, p + n); }
};
static std::vector<uint8_t> slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc < 5) return 64;
  std::vector<uint8_t> comp = slurp(argv[1]), want = slurp(argv[2]);
  size_t read_size = (size_t)atol(argv[3]), buffer_size = (size_t)atol(argv[4]);
  {  // pull
    brotli_amd::Decompressor<MemSource> d(MemSource{&comp}, buffer_size);
    std::vector<uint8_t> got, chunk(read_size);
    for (;;) { size_t n = d.read(chunk.data(), chunk.size()); if (!n) break; got.insert(got.end(), chunk.begin(), chunk.begin() + n); }
    assert(got == want);
  }
  {  // push, with 3 bytes of trailing garbage that must not be consumed
    std::vector<uint8_t> in = comp; in.push_back(1); in.push_back(2); in.push_back(3);
    brotli_amd::DecompressorWriter<MemSink> w(MemSink{}, buffer_size);
    size_t off = 0;
    while (off < in.size()) {
      size_t n = std::min(read_size, in.size() - off);
      size_t took = w.write(in.data() + off, n);
      off += took;
      if (took < n) break;  // end of stream inside this piece
    }
    assert(off == comp.size());
    MemSink s = w.close();
    assert(s.v == want);
  }
  if (comp.size() > 4) {  // truncated input: both adapters must fail
    std::vector<uint8_t> cut(comp.begin(), comp.begin() + comp.size() / 2);
    bool failed = false;
    try {
      brotli_amd::Decompressor<MemSource> d(MemSource{&cut}, buffer_size);
      std::vector<uint8_t> chunk(read_size);
      while (d.read(chunk.data(), chunk.size())) {}
    } catch (const brotli_amd::UnexpectedEof&) { failed = true; } catch (const brotli_amd::InvalidData&) { failed = true; }
    assert(failed);
    failed = false;
    try {
      brotli_amd::DecompressorWriter<MemSink> w(MemSink{}, buffer_size);
      w.write(cut.data(), cut.size());
      w.close();
    } catch (const brotli_amd::UnexpectedEof&) { failed = true; } catch (const brotli_amd::InvalidData&) { failed = true; }
    assert(failed);
  }
  fprintf(stderr, "wrappers ok\n");
  return 0;
}
