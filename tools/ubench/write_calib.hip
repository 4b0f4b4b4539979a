// What do FETCH_SIZE / WRITE_SIZE say about accesses like the decode kernel's?  (MI355X_MICROARCH.md: only wide coalesced reads
// are calibrated.)  Each kernel moves the same 256 MiB with one access pattern; run under
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -- tools/ubench/bin/write_calib     (and once more with FETCH_SIZE)
// and compare the counters per kernel with 256 MiB (tools/ubench/run_write_calib.sh prints the ratios).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr size_t N = 256u << 20;
// a wave stores 1 KiB a step, sixteen bytes a lane, at `shift` bytes from a sixteen-byte boundary
__global__ void k_store16(uint8_t* dst, uint32_t shift) {
  const size_t per_block = N / gridDim.x;
  uint8_t* p = dst + shift + blockIdx.x * per_block;
  const u32x4 v = {threadIdx.x, 1u, 2u, 3u};
  for (size_t o = threadIdx.x * 16u; o + 16u <= per_block - 16u; o += blockDim.x * 16u) __builtin_memcpy(p + o, &v, 16);
}
__global__ void k_store16_aligned(uint8_t* dst, uint32_t) {
  const size_t per_block = N / gridDim.x;
  u32x4* p = (u32x4*)(dst + blockIdx.x * per_block);
  const u32x4 v = {threadIdx.x, 1u, 2u, 3u};
  for (size_t o = threadIdx.x; o < per_block / 16u; o += blockDim.x) p[o] = v;
}
// a byte a lane, the wave's 64 bytes side by side
__global__ void k_store1(uint8_t* dst, uint32_t) {
  const size_t per_block = N / gridDim.x;
  uint8_t* p = dst + blockIdx.x * per_block;
  for (size_t o = threadIdx.x; o < per_block; o += blockDim.x) p[o] = (uint8_t)o;
}
// short pieces at odd places: every lane stores 1..16 bytes of its own sixteen (the engine's lane-per-command stores)
__global__ void k_store_pieces(uint8_t* dst, uint32_t) {
  const size_t per_block = N / gridDim.x;
  uint8_t* p = dst + blockIdx.x * per_block;
  for (size_t o = threadIdx.x * 16u; o + 16u <= per_block; o += blockDim.x * 16u) {
    const uint32_t cut = 1u + ((uint32_t)(o >> 4) * 7u) % 15u;   // two stores a lane: [0, cut) and [cut, 16), byte by byte
    for (uint32_t k = 0; k < 16u; k++) p[o + k] = (uint8_t)(k < cut ? 1 : 2);
  }
}
// copy: sixteen bytes a lane from an odd source to an odd destination (a wave per long copy)
__global__ void k_copy16(uint8_t* dst, const uint8_t* src, uint32_t shift) {
  const size_t per_block = N / gridDim.x;
  uint8_t* p = dst + shift + blockIdx.x * per_block;
  const uint8_t* q = src + 5 + blockIdx.x * per_block;
  for (size_t o = threadIdx.x * 16u; o + 16u <= per_block - 16u; o += blockDim.x * 16u) { u32x4 v; __builtin_memcpy(&v, q + o, 16); __builtin_memcpy(p + o, &v, 16); }
}
// the same 256 MiB written twice by the same block, 32 KiB at a time: first as bytes, then as sixteen-byte lines (is a
// line that is written again counted again?)
__global__ void k_store_twice(uint8_t* dst, uint32_t) {
  const size_t per_block = N / gridDim.x;
  uint8_t* p = dst + blockIdx.x * per_block;
  const u32x4 v = {threadIdx.x, 1u, 2u, 3u};
  for (size_t base = 0; base < per_block; base += 32768u) {
    for (size_t o = threadIdx.x; o < 32768u; o += blockDim.x) p[base + o] = (uint8_t)o;
    __syncthreads();
    for (size_t o = threadIdx.x * 16u; o < 32768u; o += blockDim.x * 16u) __builtin_memcpy(p + base + o, &v, 16);
    __syncthreads();
  }
}
int main() {
  uint8_t *dst, *src;
  hipMalloc(&dst, N + 4096); hipMalloc(&src, N + 4096);
  hipMemset(src, 7, N + 4096); hipMemset(dst, 0, N + 4096);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k_store16_aligned, dim3(256), dim3(1024), 0, 0, dst, 0u);
    hipLaunchKernelGGL(k_store16, dim3(256), dim3(1024), 0, 0, dst, 1u);
    hipLaunchKernelGGL(k_store16, dim3(256), dim3(1024), 0, 0, dst, 8u);
    hipLaunchKernelGGL(k_store1, dim3(256), dim3(1024), 0, 0, dst, 0u);
    hipLaunchKernelGGL(k_store_pieces, dim3(256), dim3(1024), 0, 0, dst, 0u);
    hipLaunchKernelGGL(k_copy16, dim3(256), dim3(1024), 0, 0, dst, src, 3u);
    hipLaunchKernelGGL(k_store_twice, dim3(256), dim3(1024), 0, 0, dst, 0u);
    hipDeviceSynchronize();
  }
  printf("done\n");
  return 0;
}
