// What a lone wave pays for branches, v_readlane chains and LDS round trips on gfx950 (s_memtime ticks, and the same in ns by HIP events).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane(v); }
#define R8(x) x x x x x x x x
__global__ void k_salu(uint64_t* out, int iters, uint32_t seed) {
  uint32_t x = rfl(seed); uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) asm volatile(R8("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 3\n") : "+s"(x) : : "scc");
  uint64_t t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = x; }
}
__global__ void k_branch_taken(uint64_t* out, int iters, uint32_t seed) {   // add + always-taken branch over one instruction
  uint32_t x = rfl(seed); uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) asm volatile(R8("s_add_u32 %0, %0, 1\n s_branch 1f\n s_add_u32 %0, %0, 100\n1:\n") : "+s"(x) : : "scc");
  uint64_t t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = x; }
}
__global__ void k_cbranch_not(uint64_t* out, int iters, uint32_t seed) {    // cmp + never-taken conditional branch
  uint32_t x = rfl(seed); uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) asm volatile(R8("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 2f\n") "s_add_u32 %0, %0, 1\n2:\n" : "+s"(x) : : "scc");
  uint64_t t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = x; }
}
__global__ void k_cbranch_taken(uint64_t* out, int iters, uint32_t seed) {  // cmp + always-taken conditional branch over one instruction
  uint32_t x = rfl(seed | 1u); uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) asm volatile(R8("s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1f\n s_mov_b32 %0, 1\n1:\n") : "+s"(x) : : "scc");
  uint64_t t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = x; }
}
__global__ void k_readlane(uint64_t* out, int iters, uint32_t seed) {       // dependent chain s -> v_readlane lane select -> s
  uint32_t x = rfl(seed & 63u), v = (threadIdx.x * 7u + 3u) & 63u; uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) asm volatile(R8("v_readlane_b32 %0, %1, %0\n s_and_b32 %0, %0, 63\n") : "+s"(x) : "v"(v) : "scc");
  uint64_t t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = x; }
}
__global__ void k_lds(uint64_t* out, int iters, uint32_t seed) {            // dependent chain s -> v_mov -> ds_read -> readfirstlane -> s
  __shared__ uint32_t tab[256];
  for (int i = threadIdx.x; i < 256; i += 64) tab[i] = ((i * 37 + 11) & 255) * 4;
  __syncthreads();
  uint32_t x = rfl((seed & 255u) * 4u); uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) asm volatile(R8("v_mov_b32 v10, %0\n ds_read_b32 v11, v10\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 %0, v11\n") : "+s"(x) : : "v10", "v11", "memory");
  uint64_t t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = x + tab[0]; }
}
__global__ void k_mix(uint64_t* out, int iters, uint32_t seed) {            // SALU, three independent of four
  uint32_t a = rfl(seed), b = rfl(seed + 1), c = rfl(seed + 2); uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) asm volatile(R8("s_add_u32 %0, %0, 1\n s_lshl_b32 %1, %1, 1\n s_xor_b32 %2, %2, %0\n s_sub_u32 %1, %1, %2\n") : "+s"(a), "+s"(b), "+s"(c) : : "scc");
  uint64_t t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = a + b + c; }
}
#define RUN(k, n_instr, label) do { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, N, 12345u); hipDeviceSynchronize(); \
  hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, N, 12345u); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(r, d, 16, hipMemcpyDeviceToHost); \
  printf("%-52s %.2f ticks, %.2f ns a unit of %d instructions\n", label, (double)r[0] / N / 8, ms * 1e6 / N / 8, n_instr); fflush(stdout); } while (0)
int main() {
  uint64_t* d; hipMalloc(&d, 4096); uint64_t r[2]; const int N = 20000;
  RUN(k_salu, 2, "2 dependent s_add");
  RUN(k_mix, 4, "4 SALU, mostly independent");
  RUN(k_branch_taken, 2, "s_add + taken s_branch");
  RUN(k_cbranch_not, 2, "s_cmp + s_cbranch never taken");
  RUN(k_cbranch_taken, 2, "s_cmp + s_cbranch always taken");
  RUN(k_readlane, 2, "v_readlane (lane from SGPR) + s_and, dependent");
  RUN(k_lds, 4, "v_mov + ds_read + waitcnt + readfirstlane, dependent");
  return 0;
}
