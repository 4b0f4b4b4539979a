// Issue-rate micro-benchmarks for one wave on gfx950: dependent vs independent SALU / VALU chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane(v); }
__global__ void k_salu_dep(uint64_t* out, int iters, uint32_t seed) {
  uint32_t x = rfl(seed);
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 5\n s_add_u32 %0, %0, 7\n"
                 "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 5\n s_add_u32 %0, %0, 7\n" : "+s"(x) : : "scc");
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = x; }
}
__global__ void k_salu_indep(uint64_t* out, int iters, uint32_t seed) {
  uint32_t a = rfl(seed), b = rfl(seed + 1), c = rfl(seed + 2), d = rfl(seed + 3);
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_add_u32 %3, %3, 7\n"
                 "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_add_u32 %3, %3, 7\n" : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : : "scc");
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = a + b + c + d; }
}
__global__ void k_valu_dep(uint64_t* out, int iters, uint32_t seed) {
  uint32_t x = seed + threadIdx.x;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 3\n v_add_u32 %0, %0, 5\n v_add_u32 %0, %0, 7\n"
                 "v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 3\n v_add_u32 %0, %0, 5\n v_add_u32 %0, %0, 7\n" : "+v"(x));
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; } out[2 + threadIdx.x] = x;
}
__global__ void k_valu_indep(uint64_t* out, int iters, uint32_t seed) {
  uint32_t a = seed + threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 3\n v_add_u32 %2, %2, 5\n v_add_u32 %3, %3, 7\n"
                 "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 3\n v_add_u32 %2, %2, 5\n v_add_u32 %3, %3, 7\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; } out[2 + threadIdx.x] = a + b + c + d;
}
__global__ void k_mixed(uint64_t* out, int iters, uint32_t seed) {   // alternate SALU / VALU, independent
  uint32_t a = rfl(seed), b = seed + threadIdx.x;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    asm volatile("s_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 3\n s_add_u32 %0, %0, 5\n v_add_u32 %1, %1, 7\n"
                 "s_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 3\n s_add_u32 %0, %0, 5\n v_add_u32 %1, %1, 7\n" : "+s"(a), "+v"(b) : : "scc");
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; } out[2 + threadIdx.x] = a + b;
}
#define RUN(k, n_instr, label) do { hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, N, 12345u); hipMemcpy(r, d, 16, hipMemcpyDeviceToHost); \
  printf("%-34s blocks %4d: %.2f ticks/instr\n", label, blocks, (double)r[0] / N / (n_instr)); fflush(stdout); } while (0)
int main() {
  uint64_t* d; hipMalloc(&d, 4096); uint64_t r[2]; const int N = 20000;
  for (int blocks : {1, 1024, 2048}) {
    RUN(k_salu_dep, 8, "SALU dependent");
    RUN(k_salu_indep, 8, "SALU 4 independent chains");
    RUN(k_valu_dep, 8, "VALU dependent");
    RUN(k_valu_indep, 8, "VALU 4 independent chains");
    RUN(k_mixed, 8, "SALU/VALU alternating");
  }
  return 0;
}
