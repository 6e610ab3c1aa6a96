// Shared helpers for the gfx950 kernels of libunflow_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/unflow_hip.h"

#define UNFLOW_API extern "C" __attribute__((visibility("default")))

// Every entry point converts its stream argument right before launching: also drop any stale error
// code a previous, unrelated runtime call of this thread left behind, so that launch_status() reports
// only this entry's launches.
static inline hipStream_t as_stream(unflow_stream_t s) {
  (void)hipGetLastError();
  return reinterpret_cast<hipStream_t>(s);
}

static inline int launch_status() {
  return hipGetLastError() == hipSuccess ? UNFLOW_OK : UNFLOW_ERR_LAUNCH;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: a process that drives several GPUs must
// set it on each of them before launching with more than 64 KB of dynamic LDS.  Grow-only book per call site (`set`: one
// slot per device ordinal, zero-initialised static of the caller); the runtime call happens once per (kernel, device, size
// increase), afterwards this is a hipGetDevice and a compare.  Returns the runtime's status: a kernel that has a fallback
// takes it when the set fails, the others report the failure through launch_status().  (A benign race between two host
// threads of one process: both set the same value.)
constexpr int UNFLOW_MAX_DEVICES = 64;
struct DynLdsBook {
  int set[UNFLOW_MAX_DEVICES];
};
static inline hipError_t ensure_dyn_lds(const void* kernel, int bytes, DynLdsBook& book) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= UNFLOW_MAX_DEVICES) return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (bytes <= book.set[dev]) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) book.set[dev] = bytes;
  return e;
}

// Grid size for HBM-bound grid-stride kernels: enough workgroups to fill 256 CUs x 8,
// capped (cdna guide, guideline 11).
static inline int stream_grid(long work_items, int block = 256) {
  long g = (work_items + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

// Workgroups go to the 8 XCDs round-robin by linear id (MI355X_MICROARCH.md), so a grid-stride pass over an image puts
// neighbouring row segments on different L2s, and every row that a gather / stencil kernel shares between two workgroups
// (the second tap row of a bilinear warp, the window rows of the census) is fetched from HBM by both.  This bijection of
// blockIdx.x gives each XCD a contiguous run of virtual block ids: rows are shared inside one L2 except at the 8 seams
// (the warps: 4.4 -> see profiles/r03_bench_ops_16x768x1024.jsonl).
__device__ __forceinline__ unsigned xcd_block() {
  const unsigned n = gridDim.x, b = blockIdx.x;
  if (n < 16) return b;
  const unsigned xcd = b & 7, idx = b >> 3, q = n >> 3, r = n & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ float leaky_relu(float v) { return fmaxf(0.1f * v, v); }
// tf.maximum(0.1*x, x): gradient 1 where x > 0 (output > 0), else 0.1 (ties go to the 0.1*x branch).
__device__ __forceinline__ float leaky_grad_from_out(float y) { return y > 0.f ? 1.f : 0.1f; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Block-wide sum of one float per thread; result valid in thread 0. `red` needs blockDim.x/64 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += red[i];
  __syncthreads();
  return t;
}

// ---- per-pixel kernels: 32-bit pixel decode (64-bit div/mod costs ~100 instructions per pixel)
struct Pix {
  int x, y, n;
};
__device__ __forceinline__ Pix decode_pix(unsigned i, unsigned W, unsigned H) {
  const unsigned r = i / W;
  Pix p;
  p.x = (int)(i - r * W);
  p.n = (int)(r / H);
  p.y = (int)(r - (unsigned)p.n * H);
  return p;
}

// Walk of an [N, H, W] image by 64 x 4 pixel tiles for the gather kernels (256-thread blocks): a wave owns 64 consecutive
// pixels of one row (its streamed loads / stores stay whole 256- / 768-byte runs), the four waves of a block four consecutive
// rows, and a block walks DOWN a 64-pixel column strip (tiles ordered row-fastest, a contiguous run of them per block) —
// the tap rows a bilinear footprint shares between neighbouring output rows then meet in the CU's L1, in the same pass or
// the next, instead of travelling from L2 twice (tools/microbench/hbm_stream.hip: the second tap row costs a row-major
// walk 25 % of its rate).  Loop: for (t = tile_first(T); t < tile_last(T); t++) with T = tile_count(W, H, N).
struct TilePix {
  int x, y, n;
  unsigned i;      // linear pixel index (n * H + y) * W + x
  bool ok;
};
constexpr unsigned TILE_WL = 6, TILE_HL = 8 - TILE_WL;       // log2 of the tile width / height (256 pixels)
__device__ __forceinline__ unsigned tile_count(unsigned W, unsigned H, unsigned N) {
  return ((W + (1u << TILE_WL) - 1) >> TILE_WL) * ((H + (1u << TILE_HL) - 1) >> TILE_HL) * N;
}
__device__ __forceinline__ unsigned tile_first(unsigned T) { return blockIdx.x * ((T + gridDim.x - 1) / gridDim.x); }
__device__ __forceinline__ unsigned tile_last(unsigned T) {
  const unsigned e = (blockIdx.x + 1) * ((T + gridDim.x - 1) / gridDim.x);
  return e < T ? e : T;
}
__device__ __forceinline__ TilePix tile_pix(unsigned t, unsigned W, unsigned H) {
  const unsigned c = threadIdx.x & ((1u << TILE_WL) - 1), r = threadIdx.x >> TILE_WL;
  const unsigned tx = (W + (1u << TILE_WL) - 1) >> TILE_WL, ty = (H + (1u << TILE_HL) - 1) >> TILE_HL;
  const unsigned q = t / ty, by = t - q * ty;
  const unsigned n = q / tx, bx = q - n * tx;
  TilePix p;
  p.x = (int)((bx << TILE_WL) + c); p.y = (int)((by << TILE_HL) + r); p.n = (int)n;
  p.ok = (unsigned)p.x < W && (unsigned)p.y < H;
  p.i = (n * H + (unsigned)p.y) * W + (unsigned)p.x;
  return p;
}

// ---- hardware transcendental forms (v_rsq_f32 / v_rcp_f32 / v_exp_f32 / v_log_f32, ~1 ulp): the IEEE sqrt / divide /
// powf expansions cost 10-100 instructions each and made the census and Charbonnier kernels VALU-bound.
__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_pow(float a, float e) { return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(a)); }  // a > 0
