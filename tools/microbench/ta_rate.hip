// How fast can a CU pull 16-byte-per-lane buffer loads through its address / L1 path, as a function of HOW the 64 lanes of
// one instruction are spread over cache lines?  (Round-3 question: the plane gather kernels issue ~32 wave loads per us and
// CU at 45-58 % matrix-pipe utilisation with L2 at a quarter of its bandwidth — is the per-instruction cost of 16 half-lines
// the bound, and would 8 full lines be cheaper?)  Every wave loops over an L2-resident region issuing U independent
// buffer_load_dwordx4 (or buffer_load ... lds) per iteration:
//   pattern 0  1 KB contiguous                       (8 full 128-byte lines)
//   pattern 1  16 rows x 64 B, row stride 1024 B     (the gather kernels' K32 row pieces: 16 half lines)
//   pattern 2   8 rows x 128 B                       (8 full lines, 8 rows)
//   pattern 3   4 rows x 256 B                       (the filter-gradient DMA rows)
//   pattern 4  32 rows x 32 B                        (K16 row pieces)
//   pattern 5  16 rows x 64 B, row stride 192 B      (plane-interleaved rows: [row][plane][32 k], 3 planes adjacent)
// mfma = n: n v_mfma_f32_32x32x16_bf16 per load in the same wave (the gather loop has 4).  Prints wave loads per us per CU
// and the implied bytes/clk/CU at 2.1 GHz.
//   hipcc --offload-arch=gfx950 -O3 ta_rate.hip -o ta_rate && ./ta_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ int lane_off(int pattern, int lane) {
  switch (pattern) {
    case 0: return lane * 16;
    case 1: return (lane >> 2) * 1024 + (lane & 3) * 16;
    case 2: return (lane >> 3) * 1024 + (lane & 7) * 16;
    case 3: return (lane >> 4) * 1024 + (lane & 15) * 16;
    case 4: return (lane >> 1) * 1024 + (lane & 1) * 16;
    default: return (lane >> 2) * 192 + (lane & 3) * 16;
  }
}

template <int MFMA, bool DMA>
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ src, size_t region, float* out, int iters, int pattern) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 8 * 1024];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // every block works in a 1 MB window of a 2 MB region: resident in every XCD's 4 MB L2, far larger than a CU's 32 KB L1
  const size_t win = ((size_t)blockIdx.x * (256u << 10)) % region;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src + win), 0, 1 << 20, 0x00020000);
  const int lo = lane_off(pattern, lane) + wid * 64 * 1024;
  u32x4 acc = {0, 0, 0, 0};
  f32x16 c[4];
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) c[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
  constexpr int U = 8;
  for (int it = 0; it < iters; it++) {
    const int base = lo + ((it * 16384) & 0xffff);       // walks 64 KB per wave inside the block's window
    if constexpr (DMA) {
      const unsigned d0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)(lds + wid * 8192);
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int voff = base + u * 2048 * (pattern == 0 ? 1 : 0) + (pattern == 0 ? 0 : u * 64);
        unsigned keep;
        asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[r], 0 offen lds\n\ts_mov_b32 m0, %[keep]"
                     : [keep] "=&s"(keep) : [v] "v"(voff), [r] "s"(rs), [d] "s"(d0 + (unsigned)(u & 7) * 1024) : "memory");
        if constexpr (MFMA > 0) {
#pragma unroll
          for (int m = 0; m < MFMA; m++) c[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[m & 3], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int voff = base + u * 2048 * (pattern == 0 ? 1 : 0) + (pattern == 0 ? 0 : u * 64);
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
        if constexpr (MFMA > 0) {
#pragma unroll
          for (int m = 0; m < MFMA; m++) c[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[m & 3], 0, 0, 0);
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) acc ^= v[u];
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += c[i][r];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u || s == 1.2345f) out[0] = s + lds[lane];
}

template <int MFMA, bool DMA>
double run(const unsigned char* src, size_t region, float* out, int blocks, int pattern) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<MFMA, DMA><<<blocks, 256>>>(src, region, out, 50, pattern);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k<MFMA, DMA><<<blocks, 256>>>(src, region, out, iters, pattern);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double loads = (double)blocks * 4 * iters * 8;         // wave instructions
  return loads / (ms * 1e3) / 256.0;                             // per us per CU
}

int main() {
  const size_t region = 2u << 20;
  unsigned char* src; float* out;
  CK(hipMalloc(&src, region + (2u << 20))); CK(hipMalloc(&out, 64));
  CK(hipMemset(src, 1, region + (2u << 20)));
  const char* names[6] = {"1KB contiguous", "16 x 64B (stride 1K)", "8 x 128B", "4 x 256B", "32 x 32B", "16 x 64B (stride 192)"};
  printf("%-24s %6s | %-38s | %-38s\n", "pattern", "blk/CU", "register loads: /us/CU (B/clk/CU) mfma 0 / 2 / 4", "LDS-DMA: mfma 0 / 2 / 4");
  for (int p = 0; p < 6; p++)
    for (int bpc = 1; bpc <= 3; bpc++) {
      const int blocks = 256 * bpc;
      const double r0 = run<0, false>(src, region, out, blocks, p), r2 = run<2, false>(src, region, out, blocks, p), r4 = run<4, false>(src, region, out, blocks, p);
      const double d0 = run<0, true>(src, region, out, blocks, p), d2 = run<2, true>(src, region, out, blocks, p), d4 = run<4, true>(src, region, out, blocks, p);
      auto bpc_ = [](double r) { return r * 1024.0 / 2100.0; };
      printf("%-24s %6d | %6.1f (%4.1f) %6.1f (%4.1f) %6.1f (%4.1f) | %6.1f (%4.1f) %6.1f (%4.1f) %6.1f (%4.1f)\n", names[p], bpc, r0, bpc_(r0), r2,
             bpc_(r2), r4, bpc_(r4), d0, bpc_(d0), d2, bpc_(d2), d4, bpc_(d4));
    }
  return 0;
}
