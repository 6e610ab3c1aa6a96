// Feasibility microbenchmark for "fp32-equivalent GEMM on the bf16 matrix cores" (DESIGN.md, next steps):
// a 3-way bf16 split of both operands and the 6 product terms hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid
// (dropped terms are <= 2^-23 relative) on v_mfma_f32_32x32x16_bf16, 128x128 block tile, 64x64 wave tiles, operands read
// from LDS with ds_read_b128.  Three modes:
//   0  MFMA + LDS reads only, bf16 planes already in LDS              (upper bound of a pre-split-operand design)
//   1  + per K-tile: every thread splits 32 fp32 values into 3 bf16 planes and stores them to LDS, 2 barriers
//      (the cost of splitting while staging fp32 activations/weights; global loads not modelled)
//   3  as 1 with the gfx950 hardware conversion v_cvt_pk_bf16_f32
//   2  fp32 v_mfma_f32_32x32x2_f32 loop of the shipped kernels for reference (same tile, LDS reads)
// Prints fp32-equivalent TFLOP/s = 2*M*N*K / time.     hipcc --offload-arch=gfx950 -O3 bf16x3_mfma_bench.hip -o bf16x3_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ u16 f2bf(float x) {   // round to nearest even
  unsigned u = __builtin_bit_cast(unsigned, x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ __forceinline__ float bf2f(u16 h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
// gfx950: packed hardware conversion, two fp32 -> two bf16 (round to nearest even) in one instruction
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

constexpr int BK = 32;                 // fp32 elements of K per tile
constexpr int PITCH = BK + 8;          // bf16 row pitch (80 bytes: 16-byte aligned, conflict-free b128 reads)
constexpr int PLANE = 128 * PITCH;     // one bf16 plane of a 128-row operand tile

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ src, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u16* As = reinterpret_cast<u16*>(smem_raw);            // [3 planes][128][PITCH]
  u16* Bs = As + 3 * PLANE;
  float* Af = reinterpret_cast<float*>(smem_raw);        // MODE 2: fp32 [128][36] x 2
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  // fill LDS with something non-trivial
  for (int i = tid; i < (MODE == 2 ? 2 * 128 * 36 : 6 * PLANE / 2); i += 256) {
    unsigned h = (unsigned)i * 2654435761u + (unsigned)blockIdx.x * 97u; h ^= h >> 13; h *= 0x5bd1e995u;
    if (MODE == 2) Af[i] = (float)(h & 0xffff) / 65536.f - 0.5f;
    else reinterpret_cast<unsigned*>(smem_raw)[i] = (h & 0x007f007fu) | 0x3f003f00u;   // two bf16 in [0.5, 1)
  }
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  const int wm = wid >> 1, wn = wid & 1;
  float xin[32];
  if (MODE == 1 || MODE == 3) for (int e = 0; e < 32; e++) xin[e] = src[(blockIdx.x * 256 + tid) * 32 + e];
  for (int it = 0; it < iters; it++) {
    if (MODE == 2) {
      const float* a = Af + (wm * 64 + l31) * 36 + 16 * lh;
      const float* b = Af + 128 * 36 + (wn * 64 + l31) * 36 + 16 * lh;
#pragma unroll
      for (int j4 = 0; j4 < 4; j4++) {
        float4 av[2], bv[2];
        for (int i = 0; i < 2; i++) av[i] = *reinterpret_cast<const float4*>(a + i * 32 * 36 + 4 * j4);
        for (int j = 0; j < 2; j++) bv[j] = *reinterpret_cast<const float4*>(b + j * 32 * 36 + 4 * j4);
#pragma unroll
        for (int e = 0; e < 4; e++)
          for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32((&av[i].x)[e], (&bv[j].x)[e], acc[i][j], 0, 0, 0);
      }
    } else {
      if (MODE == 3) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; q++) {
          unsigned pk[3][2];
#pragma unroll
          for (int e = 0; e < 2; e++) {
            const float x0 = xin[4 * q + 2 * e] + (float)it, x1 = xin[4 * q + 2 * e + 1] + (float)it;
            const unsigned h = cvt_pk_bf16(x0, x1);
            const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
            const unsigned m = cvt_pk_bf16(r0, r1);
            const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
            pk[0][e] = h; pk[1][e] = m; pk[2][e] = cvt_pk_bf16(s0, s1);
          }
          u16* base = (q < 4 ? As : Bs) + (tid >> 1) * PITCH + ((tid & 1) * 16 + (q & 3) * 4);
#pragma unroll
          for (int pl = 0; pl < 3; pl++) *reinterpret_cast<uint2*>(base + pl * PLANE) = make_uint2(pk[pl][0], pk[pl][1]);
        }
        __syncthreads();
      }
      if (MODE == 1) {
        // split 32 fp32 values (16 of the A tile, 16 of the B tile) into 3 bf16 planes and store them (8-byte stores)
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; q++) {
          u16 p[3][4];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float x = xin[4 * q + e] + (float)it;
            const u16 h = f2bf(x);
            const float r1 = x - bf2f(h);
            const u16 m = f2bf(r1);
            const float r2 = r1 - bf2f(m);
            p[0][e] = h; p[1][e] = m; p[2][e] = f2bf(r2);
          }
          u16* base = (q < 4 ? As : Bs) + (tid >> 1) * PITCH + ((tid & 1) * 16 + (q & 3) * 4);
#pragma unroll
          for (int pl = 0; pl < 3; pl++)
            *reinterpret_cast<uint2*>(base + pl * PLANE) =
                make_uint2((unsigned)p[pl][0] | ((unsigned)p[pl][1] << 16), (unsigned)p[pl][2] | ((unsigned)p[pl][3] << 16));
        }
        __syncthreads();
      }
      // K-tile of 32 = two K16 slabs; lane (row l31, half lh) reads 8 consecutive bf16 of its row: k = 16*slab + 8*lh
#pragma unroll
      for (int slab = 0; slab < 2; slab++) {
        bf16x8 a[3][2], b[3][2];
#pragma unroll
        for (int pl = 0; pl < 3; pl++)
#pragma unroll
          for (int i = 0; i < 2; i++) {
            a[pl][i] = *reinterpret_cast<const bf16x8*>(As + pl * PLANE + (wm * 64 + i * 32 + l31) * PITCH + 16 * slab + 8 * lh);
            b[pl][i] = *reinterpret_cast<const bf16x8*>(Bs + pl * PLANE + (wn * 64 + i * 32 + l31) * PITCH + 16 * slab + 8 * lh);
          }
        // smallest terms first
        const int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; t++)
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta[t]][i], b[tb[t]][j], acc[i][j], 0, 0, 0);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, int blocks_per_cu) {
  const int blocks = 256 * blocks_per_cu, iters = 2000;
  float *out, *src;
  CK(hipMalloc(&out, sizeof(float) * blocks * 256));
  CK(hipMalloc(&src, sizeof(float) * blocks * 256 * 32));
  CK(hipMemset(src, 0, sizeof(float) * blocks * 256 * 32));
  const size_t lds = MODE == 2 ? 2 * 128 * 36 * 4 : 6 * PLANE * 2;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<MODE><<<blocks, 256, lds>>>(out, src, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k<MODE><<<blocks, 256, lds>>>(out, src, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 2.0 * 128 * 128 * BK * (double)iters * blocks;     // fp32-equivalent
  printf("%-58s %d blocks/CU  LDS %5.1f KB  %8.3f ms  %7.1f TFLOP/s fp32-equivalent\n", name, blocks_per_cu, lds / 1024.0, ms,
         flop / ms / 1e9);
  CK(hipFree(out)); CK(hipFree(src));
}

int main() {
  run<2>("fp32 MFMA 32x32x2, operands from LDS (shipped inner loop)", 3);
  run<0>("bf16x3 (6 terms) MFMA 32x32x16, pre-split planes in LDS", 2);
  run<0>("bf16x3 (6 terms) MFMA 32x32x16, pre-split planes in LDS", 3);
  run<1>("bf16x3 + split-while-staging (VALU split, LDS stores, barriers)", 2);
  run<1>("bf16x3 + split-while-staging (VALU split, LDS stores, barriers)", 3);
  run<3>("bf16x3 + split-while-staging with v_cvt_pk_bf16_f32", 2);
  run<3>("bf16x3 + split-while-staging with v_cvt_pk_bf16_f32", 3);
  return 0;
}
