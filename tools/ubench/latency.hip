// Micro-benchmarks of the primitives the wave-uniform decode chain is made of (one wave, gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
extern __shared__ uint8_t smem[];
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane(v); }

__global__ void k_lds_chain(uint64_t* out, int iters, uint32_t seed) {
  uint32_t* t = (uint32_t*)smem;
  for (int i = threadIdx.x; i < 4096; i += 64) t[i] = (i * 2654435761u + seed) & 4095;
  __syncthreads();
  uint32_t idx = rfl(seed & 4095);
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) idx = rfl(t[idx]);   // dependent LDS lookup through v_readfirstlane
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = idx; }
}
__global__ void k_salu_chain(uint64_t* out, int iters, uint32_t seed) {
  uint32_t x = rfl(seed);
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) { x = x * 3 + 1; x ^= x >> 3; x += 7; x ^= x << 5; }   // 8 dependent SALU ops
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = x; }
}
__global__ void k_readlane_chain(uint64_t* out, int iters, uint32_t seed) {
  uint32_t v = (threadIdx.x * 2654435761u + seed) & 63;
  uint32_t idx = rfl(seed & 63);
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) idx = (uint32_t)__builtin_amdgcn_readlane(v, idx);   // dependent v_readlane
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = idx; }
}
__global__ void k_lds_chain_store(uint64_t* out, uint8_t* dst, int iters, uint32_t seed) {
  uint32_t* t = (uint32_t*)smem;
  for (int i = threadIdx.x; i < 4096; i += 64) t[i] = (i * 2654435761u + seed) & 4095;
  __syncthreads();
  uint32_t idx = rfl(seed & 4095);
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) { idx = rfl(t[idx]); if (threadIdx.x == 0) dst[i] = (uint8_t)idx; }   // + one byte store per step
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = idx; }
}
__global__ void k_global_chain(uint64_t* out, const uint32_t* tab, int iters, uint32_t seed) {
  uint32_t idx = rfl(seed & 4095);
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) idx = rfl(tab[idx]);   // dependent global (L1/L2-resident) lookup
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = idx; }
}
int main() {
  uint64_t* d; hipMalloc(&d, 64); uint8_t* dst; hipMalloc(&dst, 1 << 20); uint32_t* tab; hipMalloc(&tab, 16384);
  uint32_t h[4096]; for (int i = 0; i < 4096; i++) h[i] = (i * 2654435761u + 12345) & 4095; hipMemcpy(tab, h, sizeof h, hipMemcpyHostToDevice);
  uint64_t r[2]; const int N = 100000;
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k_lds_chain, dim3(1), dim3(64), 16384, 0, d, N, 12345u); hipMemcpy(r, d, 16, hipMemcpyDeviceToHost);
    printf("lds chain (ds_read + readfirstlane): %.1f ticks/iter\n", (double)r[0] / N);
    hipLaunchKernelGGL(k_salu_chain, dim3(1), dim3(64), 0, 0, d, N, 12345u); hipMemcpy(r, d, 16, hipMemcpyDeviceToHost);
    printf("salu chain (8 dependent ops): %.1f ticks/iter\n", (double)r[0] / N);
    hipLaunchKernelGGL(k_readlane_chain, dim3(1), dim3(64), 0, 0, d, N, 12345u); hipMemcpy(r, d, 16, hipMemcpyDeviceToHost);
    printf("readlane chain: %.1f ticks/iter\n", (double)r[0] / N);
    hipLaunchKernelGGL(k_lds_chain_store, dim3(1), dim3(64), 16384, 0, d, dst, N, 12345u); hipMemcpy(r, d, 16, hipMemcpyDeviceToHost);
    printf("lds chain + byte store: %.1f ticks/iter\n", (double)r[0] / N);
    hipLaunchKernelGGL(k_global_chain, dim3(1), dim3(64), 0, 0, d, tab, N, 12345u); hipMemcpy(r, d, 16, hipMemcpyDeviceToHost);
    printf("global chain (cached): %.1f ticks/iter\n", (double)r[0] / N);
  }
  // tick rate
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k_salu_chain, dim3(1), dim3(64), 0, 0, d, 2000000, 1u); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(r, d, 16, hipMemcpyDeviceToHost);
  printf("s_memtime ticks per us: %.1f (kernel %.3f ms, %llu ticks)\n", r[0] / (ms * 1e3), ms, (unsigned long long)r[0]);
  return 0;
}
