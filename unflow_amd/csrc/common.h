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

// Grid size for HBM-bound grid-stride kernels: enough workgroups to fill 256 CUs x 8,
// capped (cdna guide, guideline 11).
static inline int stream_grid(long work_items, int block = 256) {
  long g = (work_items + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
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

// ---- hardware transcendental forms (v_rsq_f32 / v_rcp_f32 / v_exp_f32 / v_log_f32, ~1 ulp): the IEEE sqrt / divide /
// powf expansions cost 10-100 instructions each and made the census and Charbonnier kernels VALU-bound.
__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_pow(float a, float e) { return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(a)); }  // a > 0
