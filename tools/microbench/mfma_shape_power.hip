// Which MFMA shape does more useful work under the board's power limit?  v_mfma_f32_32x32x16_bf16 moves 8 KB of accumulator registers per
// 32768 flop, v_mfma_f32_16x16x32_bf16 2 KB per 16384 — half the accumulator traffic per flop, twice the operand traffic.  Pure MFMA loops on
// pseudo-random operands (what real data looks like to the multiplier array; tools/microbench/mfma_scaling.hip), 2 waves per SIMD on all 256 CUs,
// SUSTAINED for ~3 s per shape (short bursts ride on banked power headroom).  Prints TFLOP/s; run under tools/power_trace.sh for clock / power.
//   hipcc --offload-arch=gfx950 -O3 mfma_shape_power.hip -o mfma_shape_power && ./mfma_shape_power <shape 0|1|2|3> [seconds] [blocks]
//   shape 0: 32x32x16 bf16   1: 16x16x32 bf16   2: 32x32x16 f16   3: 16x16x32 f16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void operands(u16x8 (&x)[4], u16x8 (&y)[4], bool f16) {
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int q = 0; q < 4; q++)
    for (int i = 0; i < 8; i++) {
      // random significand + sign, exponent of [1, 2): bf16 0x3f80 | 7 bits, fp16 0x3c00 | 10 bits
      h = h * 1664525u + 1013904223u;
      x[q][i] = f16 ? (unsigned short)(0x3c00u | ((h >> 9) & 0x3ffu) | ((h >> 3) & 0x8000u)) : (unsigned short)(0x3f80u | ((h >> 9) & 0x7fu) | ((h >> 3) & 0x8000u));
      h = h * 1664525u + 1013904223u;
      y[q][i] = f16 ? (unsigned short)(0x3c00u | ((h >> 9) & 0x3ffu) | ((h >> 3) & 0x8000u)) : (unsigned short)(0x3f80u | ((h >> 9) & 0x7fu) | ((h >> 3) & 0x8000u));
    }
}

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  u16x8 x[4], y[4];
  operands(x, y, SHAPE >= 2);
  float s = 0.f;
  if constexpr (SHAPE == 0 || SHAPE == 2) {
    f32x16 a[4] = {};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int u = 0; u < 8; u++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if constexpr (SHAPE == 0)
            a[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x[(u + q) & 3]), __builtin_bit_cast(bf16x8, y[(u + q + 1) & 3]), a[q], 0, 0, 0);
          else
            a[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[(u + q) & 3]), __builtin_bit_cast(f16x8, y[(u + q + 1) & 3]), a[q], 0, 0, 0);
        }
    }
    for (int q = 0; q < 4; q++) for (int i = 0; i < 16; i++) s += a[q][i];
  } else {
    f32x4 a[8] = {};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int u = 0; u < 8; u++)            // 64 instructions of half the flops = the same work per iteration as above
#pragma unroll
        for (int q = 0; q < 8; q++) {
          if constexpr (SHAPE == 1)
            a[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x[(u + q) & 3]), __builtin_bit_cast(bf16x8, y[(u + q + 1) & 3]), a[q], 0, 0, 0);
          else
            a[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x[(u + q) & 3]), __builtin_bit_cast(f16x8, y[(u + q + 1) & 3]), a[q], 0, 0, 0);
        }
    }
    for (int q = 0; q < 8; q++) for (int i = 0; i < 4; i++) s += a[q][i];
  }
  if (s == 12345.f) out[0] = s;
}

template <int SHAPE>
static void run(float* out, double seconds, const char* name, int blocks) {
  const int iters = 4000;                           // 2 waves per SIMD on 256 CUs; 32 x 32768 flop per wave and iteration
  const double flop_per_launch = (double)blocks * 4 * iters * 32 * 32768.0;
  for (int i = 0; i < 3; i++) k<SHAPE><<<blocks, 256>>>(out, iters);
  CK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  double el = 0;
  while (el < seconds) {
    for (int i = 0; i < 20; i++) k<SHAPE><<<blocks, 256>>>(out, iters);
    CK(hipDeviceSynchronize());
    launches += 20;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  printf("%-18s %6.2f s  %ld launches  %8.1f TFLOP/s  (%.1f %% of 2500 x blocks / 512)\n", name, el, launches, flop_per_launch * launches / el * 1e-12,
         flop_per_launch * launches / el * 1e-12 / 25.0 * 512.0 / blocks);
}

int main(int argc, char** argv) {
  const int shape = argc > 1 ? atoi(argv[1]) : 0;
  const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
  const int blocks = argc > 3 ? atoi(argv[3]) : 512;              // 512 = 2 waves per SIMD on all 256 CUs; 48 = 24 CUs (no power limit in sight)
  float* out; CK(hipMalloc(&out, 64));
  switch (shape) {
    case 0: run<0>(out, seconds, "32x32x16 bf16", blocks); break;
    case 1: run<1>(out, seconds, "16x16x32 bf16", blocks); break;
    case 2: run<2>(out, seconds, "32x32x16 f16", blocks); break;
    default: run<3>(out, seconds, "16x16x32 f16", blocks); break;
  }
  return 0;
}
