// Does a wave see its own global stores in later loads when the line is already in its L1?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(1))) uint8_t gu8;
__global__ void k(uint8_t* buf_, int* bad, int rounds) {
  gu8* buf = (gu8*)(uintptr_t)buf_;
  int lane = threadIdx.x;
  int nbad = 0;
  for (int r = 0; r < rounds; r++) {
    uint64_t base = (uint64_t)r * 256;
    // 1. every lane loads from the line (brings it into L1)
    uint32_t a = buf[base + lane];
    // 2. lanes store new values into the same line (pattern-fill like: few lanes)
    if (lane < 4) buf[base + 64 + lane] = (uint8_t)(r + 7 + lane);
    if (lane < 4) buf[base + lane] = (uint8_t)(r + 11 + lane);
    // 3. uniform loads of the freshly stored bytes
    uint32_t x = buf[base + 64 + 3], y = buf[base + 2];
    if (x != (uint8_t)(r + 7 + 3) || y != (uint8_t)(r + 11 + 2)) nbad++;
    (void)a;
  }
  if (lane == 0) *bad = nbad;
}
int main() {
  uint8_t* d; hipMalloc(&d, 1 << 22); hipMemset(d, 0, 1 << 22);
  int* bad; hipMalloc(&bad, 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, bad, 10000);
  int h = -1; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("stale reads: %d of 10000\n", h);
  return 0;
}
