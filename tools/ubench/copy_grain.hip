// What a CU pays for a copy of 128 bytes from one unaligned place of a 4 MiB window to another, by the granule its lanes move:
// sixteen waves a CU (one block of 1024 threads, 256 blocks), every wave 512 copies one after the other, four in flight.
//   A: a byte a lane (two loads and two stores of 64 lanes)        B: a dword a lane, source and destination dword-aligned (32 lanes)
//   C: sixteen bytes a lane, aligned (8 lanes)                       D: a dword a lane from an UNALIGNED source (32 lanes; the hardware splits)
//   E: a byte a lane, 64 bytes a copy (one load, one store)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }
template <int MODE> __global__ void __launch_bounds__(1024) k(uint8_t* buf, uint64_t* out, int iters) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint8_t* const win = buf + (size_t)blockIdx.x * (4u << 20);
  uint32_t s = blockIdx.x * 977u + wave * 131u + 7u;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i += 4) {
    uint32_t so[4], dq[4];
    for (int q = 0; q < 4; q++) { so[q] = __builtin_amdgcn_readfirstlane(rnd(s)) & ((4u << 20) - 512u); dq[q] = __builtin_amdgcn_readfirstlane(rnd(s)) & ((4u << 20) - 512u); }
    if (MODE == 0) {
      uint32_t a[4], b[4];
      for (int q = 0; q < 4; q++) { a[q] = win[so[q] + lane]; b[q] = win[so[q] + 64u + lane]; }
      for (int q = 0; q < 4; q++) { win[dq[q] + lane] = (uint8_t)a[q]; win[dq[q] + 64u + lane] = (uint8_t)b[q]; }
    } else if (MODE == 1 || MODE == 3) {
      uint32_t a[4];
      for (int q = 0; q < 4; q++) { const uint32_t o_ = MODE == 1 ? so[q] & ~3u : so[q] | 1u; if (lane < 32u) a[q] = *reinterpret_cast<uint32_t*>(win + o_ + lane * 4u); }
      for (int q = 0; q < 4; q++) if (lane < 32u) *reinterpret_cast<uint32_t*>(win + (dq[q] & ~3u) + lane * 4u) = a[q];
    } else if (MODE == 2) {
      u32x4 a[4];
      for (int q = 0; q < 4; q++) if (lane < 8u) a[q] = *reinterpret_cast<u32x4*>(win + (so[q] & ~15u) + lane * 16u);
      for (int q = 0; q < 4; q++) if (lane < 8u) *reinterpret_cast<u32x4*>(win + (dq[q] & ~15u) + lane * 16u) = a[q];
    } else {
      uint32_t a[4];
      for (int q = 0; q < 4; q++) a[q] = win[so[q] + lane];
      for (int q = 0; q < 4; q++) win[dq[q] + lane] = (uint8_t)a[q];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* what, uint8_t* buf, uint64_t* d, int blocks) {
  const int iters = 512; uint64_t h[256];
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, buf, d, iters); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, buf, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, d, 8 * blocks, hipMemcpyDeviceToHost);
  double sum = 0; for (int i = 0; i < blocks; i++) sum += h[i];
  printf("%-58s %d blocks: %.3f ms; %.0f clocks a copy a wave, %.1f a copy a CU (sixteen waves)\n", what, blocks, ms, sum / blocks / iters, sum / blocks / iters / 16.0);
}
int main() {
  uint8_t* buf; uint64_t* d; hipMalloc(&buf, (size_t)256 * (4u << 20)); hipMemset(buf, 1, (size_t)256 * (4u << 20)); hipMalloc(&d, 8 * 256);
  for (int blocks : {256, 16}) {
    run<0>("A: 128 bytes, a byte a lane", buf, d, blocks);
    run<1>("B: 128 bytes, a dword a lane, aligned", buf, d, blocks);
    run<3>("D: 128 bytes, a dword a lane, source unaligned", buf, d, blocks);
    run<2>("C: 128 bytes, sixteen bytes a lane, aligned", buf, d, blocks);
    run<4>("E: 64 bytes, a byte a lane", buf, d, blocks);
  }
  return 0;
}
