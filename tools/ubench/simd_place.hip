// Where the waves of co-resident blocks land (HW_ID: SIMD, CU, SE, XCC) and what a scalar chain pays when the chains of a CU's four blocks
// run on the same SIMD or on different ones.  1024 blocks of 256 threads, 40 KB of LDS each (four a CU, as the record blocks are).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane(v); }
#define R8(x) x x x x x x x x
// mode 0: wave 0 of every block runs the chain; mode 1: the wave whose SIMD is (block / cus) & 3; mode 2: the wave whose SIMD is slot & 3, slot = a counter a CU
__global__ void __launch_bounds__(256) k(uint64_t* out, uint32_t* cu_ctr, int iters, int mode, int cus, int busy2) {
  extern __shared__ uint32_t lds[];
  uint32_t hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const uint32_t wave = threadIdx.x >> 6, simd = (hw >> 4) & 3u, cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
  const uint32_t cu_global = ((xcc & 15u) * 8u + se) * 32u + sh * 16u + cu;
  if (threadIdx.x == 0) lds[0] = 0;
  __syncthreads();
  if (mode == 2 && threadIdx.x == 0) lds[0] = atomicAdd(&cu_ctr[cu_global], 1u);
  __syncthreads();
  uint32_t want_simd = mode == 0 ? 0xffu : mode == 1 ? ((blockIdx.x / cus) & 3u) : (lds[0] & 3u);
  bool mine = mode == 0 ? wave == 0 : simd == want_simd;
  uint32_t x = rfl(threadIdx.x); uint64_t t0 = __builtin_amdgcn_s_memtime();
  if (mine) {
    for (int i = 0; i < iters; i++) asm volatile(R8("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 3\n") : "+s"(x) : : "scc");
  } else if (busy2 && ((wave + 2) & 3) == 0) {   // a second busy wave a block (VALU work, as the records' wave is)
    float f = threadIdx.x;
    for (int i = 0; i < iters * 4; i++) f = f * 1.0001f + 0.5f;
    if (f == 12345.f) out[7] = 1;
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { uint64_t* o = out + 8 + (blockIdx.x * 4 + wave) * 2; o[0] = mine ? t1 - t0 : 0; o[1] = (uint64_t)cu_global << 8 | simd << 4 | wave; }
  if (x == 77) out[6] = x;
}
int main() {
  const int B = 1024, N = 20000; uint64_t* d; uint32_t* c; hipMalloc(&d, (8 + B * 8) * 8); hipMalloc(&c, 4096 * 4); hipMemset(c, 0, 4096 * 4);
  std::vector<uint64_t> r(8 + B * 8);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
  for (int busy2 = 0; busy2 < 2; busy2++) for (int mode = 0; mode < 3; mode++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(B), dim3(256), 40960, 0, d, c, N, mode, 256, busy2); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(B), dim3(256), 40960, 0, d, c, N, mode, 256, busy2); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(r.data(), d, r.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; int n = 0; std::map<uint32_t, int> per_cu_simd; std::map<uint32_t, int> wave0_simd; std::map<uint32_t,int> blocks_per_cu;
    for (int b = 0; b < B; b++) for (int w = 0; w < 4; w++) { uint64_t t = r[8 + (b * 4 + w) * 2], id = r[8 + (b * 4 + w) * 2 + 1];
      if (w == 0) { wave0_simd[(id >> 4) & 3]++; blocks_per_cu[id >> 8]++; }
      if (t) { sum += t; n++; per_cu_simd[(uint32_t)(id >> 4)]++; } }
    int hist[8] = {0}; for (auto& p : per_cu_simd) hist[p.second < 7 ? p.second : 7]++;
    int bh[12] = {0}; for (auto& p : blocks_per_cu) bh[p.second < 11 ? p.second : 11]++;
    printf("busy2 %d mode %d: %.3f ms, %.2f ticks a 2-instruction unit (mean over %d chains); wave 0's SIMD: %d %d %d %d; chains on one (CU, SIMD): x1 %d x2 %d x3 %d x4 %d more %d; CUs %zu, blocks a CU hist 1:%d 2:%d 3:%d 4:%d 5:%d 6+:%d\n", busy2, mode, ms, sum / n / N / 8, n,
           wave0_simd[0], wave0_simd[1], wave0_simd[2], wave0_simd[3], hist[1], hist[2], hist[3], hist[4], hist[5] + hist[6] + hist[7], blocks_per_cu.size(), bh[1], bh[2], bh[3], bh[4], bh[5], bh[6]+bh[7]+bh[8]+bh[9]+bh[10]+bh[11]);
  }
  return 0;
}
