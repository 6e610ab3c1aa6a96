// Does the matrix pipe of one CU slow down when more CUs are busy?  (Round-3 question: a conv6_1 workgroup alone on a CU runs
// a K32 tile in 0.655 us = the pipe time of its MFMAs; with 192 CUs busy the same workgroup needs 1.45 us.)  Pure
// v_mfma_f32_32x32x16_bf16 loops, no memory traffic: `blocks` workgroups of 4 waves (one wave per SIMD), 1 .. 3 per CU on
// 24 .. 256 CUs; four independent accumulators per wave.  Prints MFMAs per us per SIMD (nominal: 2400 MHz / 32 cycles = 75).
//   hipcc --offload-arch=gfx950 -O3 mfma_scaling.hip -o mfma_scaling && ./mfma_scaling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k(float* out, int iters, int rnd) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  // operands: pseudo-random bf16 bit patterns in [1, 2) x sign (data toggling like real operands), four sets per wave
  bf16x8 x[4], y[4];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + (unsigned)rnd;
  for (int q = 0; q < 4; q++)
    for (int i = 0; i < 8; i++) {
      h = h * 1664525u + 1013904223u;
      const unsigned short bx = (unsigned short)(0x3f80u | ((h >> 9) & 0x7fu) | ((h >> 3) & 0x8000u));
      h = h * 1664525u + 1013904223u;
      const unsigned short by = (unsigned short)(0x3f80u | ((h >> 9) & 0x7fu) | ((h >> 3) & 0x8000u));
      x[q][i] = __builtin_bit_cast(__bf16, rnd ? bx : (unsigned short)0x3f80u);
      y[q][i] = __builtin_bit_cast(__bf16, rnd ? by : (unsigned short)0x3f80u);
    }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[u & 3], y[(u + 1) & 3], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[(u + 1) & 3], y[(u + 2) & 3], a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[(u + 2) & 3], y[(u + 3) & 3], a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[(u + 3) & 3], y[u & 3], a3, 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; i++) s += a0[i] + a1[i] + a2[i] + a3[i];
  if (s == 12345.f) out[0] = s;
}

int main() {
  float* out; CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000;                      // 128 k MFMAs per wave: ~1.7 ms at the nominal rate
  printf("%8s %14s %22s\n", "blocks", "ms", "MFMA / us / SIMD-wave");
  for (int rnd = 0; rnd < 2; rnd++)
  for (int blocks : {24, 96, 192, 256, 512, 768}) {
    for (int i = 0; i < 2; i++) k<<<blocks, 256>>>(out, iters, rnd);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
      CK(hipEventRecord(e0)); k<<<blocks, 256>>>(out, iters, rnd); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double per_wave = 32.0 * iters / (best * 1e3);
    const double waves_per_simd = blocks <= 256 ? 1.0 : blocks / 256.0;
    printf("%s %8d %14.3f %12.1f per wave, %6.1f per SIMD (%4.1f %% of 75)\n", rnd ? "random operands  " : "constant operands", blocks, best, per_wave, per_wave * waves_per_simd,
           per_wave * waves_per_simd / 0.75);
  }
  return 0;
}
