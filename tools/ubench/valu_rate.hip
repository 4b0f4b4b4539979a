// VALU issue rate per SIMD on gfx950 with 1, 2, 4 waves per SIMD (one block on one CU): integer ops, four independent chains
// a wave.  Prints clocks per wave-instruction per SIMD.  (Round 4: is the records phase of the path engine VALU-bound?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_valu(uint64_t* out, int iters, uint32_t seed) {
  uint32_t a = seed + threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
  __syncthreads();
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    asm volatile("v_add_u32 %0, %0, 1\n v_lshlrev_b32 %1, 1, %1\n v_and_b32 %2, 0x7f, %2\n v_bfe_u32 %3, %3, 1, 9\n"
                 "v_add_u32 %0, %0, 1\n v_xor_b32 %1, %1, %0\n v_alignbit_b32 %2, %2, %3, 3\n v_add_u32 %3, %3, 7\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  out[64 + threadIdx.x] = a + b + c + d;
}
__global__ void k_mix(uint64_t* out, int iters, uint32_t seed) {   // 4 VALU : 1 SALU, like the engine's loops
  uint32_t a = seed + threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
  uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane(seed);
  __syncthreads();
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    asm volatile("v_add_u32 %0, %0, 1\n v_lshlrev_b32 %1, 1, %1\n s_add_u32 %4, %4, 3\n v_and_b32 %2, 0x7f, %2\n v_bfe_u32 %3, %3, 1, 9\n"
                 "v_add_u32 %0, %0, 1\n v_xor_b32 %1, %1, %0\n s_lshl_b32 %4, %4, 1\n v_alignbit_b32 %2, %2, %3, 3\n v_add_u32 %3, %3, 7\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s) : : "scc");
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  out[64 + threadIdx.x] = a + b + c + d + s;
}
int main() {
  uint64_t* out; hipMalloc(&out, 8 * 2048);
  uint64_t h[16];
  const int iters = 20000;
  for (int which = 0; which < 2; which++)
  for (int waves = 4; waves <= 16; waves *= 2) {
    for (int rep = 0; rep < 2; rep++) {
      if (which == 0) hipLaunchKernelGGL(k_valu, dim3(1), dim3(64 * waves), 0, 0, out, iters, 1u);
      else hipLaunchKernelGGL(k_mix, dim3(1), dim3(64 * waves), 0, 0, out, iters, 1u);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, out, 8 * 16, hipMemcpyDeviceToHost);
    uint64_t mx = 0; for (int w = 0; w < waves; w++) mx = h[w] > mx ? h[w] : mx;
    const double instr_per_wave = (which == 0 ? 8.0 : 10.0) * iters, valu_per_wave = 8.0 * iters;
    printf("%s waves/SIMD %d: %llu ticks (s_memtime 100 MHz?) -> per wave %.2f ticks/instr; per SIMD %.3f ticks per VALU wave-instr\n", which ? "mix " : "valu", waves / 4,
           (unsigned long long)mx, (double)mx / instr_per_wave, (double)mx / (valu_per_wave * (waves / 4)));
  }
  // clock calibration: s_memtime vs wall
  return 0;
}
