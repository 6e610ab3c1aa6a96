// FlowNetC cost volume for gfx950 — replaces Correlation()/CorrelationGrad()
// (ops/correlation_op.cu.cc:250-390) without the reference's padded NHWC copies
// (2 x cudaMemset + blob_rearrange_kernel2): zero padding is a bounds check.
//
// Channels-last kernels (what the training step calls) + NCHW wrappers (the op boundary).
//   generic kernels : any (kernel_size, max_displacement, pad, stride_1, stride_2)
//   MFMA kernels    : kernel_size == 1, stride_1 == 1 (the FlowNetC configuration, flownet.py:221-222)
//                     — see correlation_mfma.hip
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "options.h"
#include "correlation_geom.h"
#include "igemm_shared.h"   // split3

// ------------------------------------------------------------------ generic forward
// One 256-thread block per output pixel: the k*k*C patch of in0 is staged in LDS once, then the
// 4 waves split the displacement channels; each lane strides over the patch, wave-shuffle reduce.
__global__ __launch_bounds__(256) void corr_fwd_generic_kernel(const float* __restrict__ in0,
                                                               const float* __restrict__ in1, int ld_in, int shift,
                                                               float* __restrict__ out, int ld_out, int B, int C,
                                                               int H, int W, CorrGeom g) {
  extern __shared__ float patch[];
  const int ox = blockIdx.x, oy = blockIdx.y, n = blockIdx.z;
  const int n1 = (n + shift) % B;
  const int k = g.k;
  const int y1 = oy * g.s1 + g.md - g.pad, x1 = ox * g.s1 + g.md - g.pad;  // unpadded patch corner in in0
  const int pe = k * k * C;
  for (int e = threadIdx.x; e < pe; e += blockDim.x) {
    const int c = e % C, ji = e / C;
    const int j = ji / k, i = ji - j * k;
    const int y = y1 + j, x = x1 + i;
    patch[e] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                   ? in0[(((size_t)n * H + y) * W + x) * ld_in + c] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float denom = (float)(k * k * C);
  for (int ch = wid; ch < g.oc; ch += 4) {
    const int dx = (ch % g.gw - g.r) * g.s2, dy = (ch / g.gw - g.r) * g.s2;
    float s = 0.f;
    for (int e = lane; e < pe; e += 64) {
      const int c = e % C, ji = e / C;
      const int j = ji / k, i = ji - j * k;
      const int y = y1 + dy + j, x = x1 + dx + i;
      if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
        s += patch[e] * in1[(((size_t)n1 * H + y) * W + x) * ld_in + c];
    }
    s = wave_sum(s);
    if (lane == 0) out[(((size_t)n * g.oh + oy) * g.ow + ox) * ld_out + ch] = s / denom;
  }
}

// ------------------------------------------------------------------ generic backward
// One thread per input element (channel fastest).  Window of output positions whose patch covers
// the element: ceil((l-2kr-md-sx)/s1) .. floor((l-md-sx)/s1) in padded coords (ref :133-142, :211-216).
__device__ __forceinline__ int floordiv_i(int a, int b) {
  int q = a / b;
  return (a % b != 0 && a < 0) ? q - 1 : q;
}
__device__ __forceinline__ int ceildiv_i(int a, int b) { return -floordiv_i(-a, b); }

__global__ void corr_bwd_generic_kernel(const float* __restrict__ dout, int ld_dout, const float* __restrict__ in0,
                                        const float* __restrict__ in1, int ld_in, int shift, float* __restrict__ g0,
                                        float* __restrict__ g1, int ld_g, int fuse, int B, int C, int H, int W,
                                        CorrGeom g) {
  const size_t total = (size_t)B * H * W * C;
  const float denom = (float)((2 * g.kr + 1) * (2 * g.kr + 1) * C);
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const size_t pxl = e / C;
    const int x = (int)(pxl % W), y = (int)((pxl / W) % H), s = (int)(pxl / ((size_t)W * H));
    const int m = y + g.pad, l = x + g.pad;
    // ---- as FIRST input of pair (s, (s+shift)%B)
    float acc0 = 0.f;
    {
      const int n = s, n1 = (s + shift) % B;
      int x_lo = ceildiv_i(l - 2 * g.kr - g.md, g.s1), x_hi = floordiv_i(l - g.md, g.s1);
      int y_lo = ceildiv_i(m - 2 * g.kr - g.md, g.s1), y_hi = floordiv_i(m - g.md, g.s1);
      if (x_hi >= 0 && y_hi >= 0 && x_lo <= g.ow - 1 && y_lo <= g.oh - 1) {
        x_lo = max(x_lo, 0); x_hi = min(x_hi, g.ow - 1);
        y_lo = max(y_lo, 0); y_hi = min(y_hi, g.oh - 1);
        for (int p = -g.r; p <= g.r; p++)
          for (int o = -g.r; o <= g.r; o++) {
            const int yy1 = y + g.s2 * p, xx1 = x + g.s2 * o;
            if ((unsigned)yy1 >= (unsigned)H || (unsigned)xx1 >= (unsigned)W) continue;  // zero padding
            const float v1 = in1[(((size_t)n1 * H + yy1) * W + xx1) * ld_in + c];
            const int ch = (p + g.r) * g.gw + (o + g.r);
            for (int yy = y_lo; yy <= y_hi; yy++)
              for (int xx = x_lo; xx <= x_hi; xx++)
                acc0 += dout[(((size_t)n * g.oh + yy) * g.ow + xx) * ld_dout + ch] * v1;
          }
      }
    }
    // ---- as SECOND input of pair (n, s) with n = (s - shift) mod B
    float acc1 = 0.f;
    {
      const int n = ((s - shift) % B + B) % B;
      for (int p = -g.r; p <= g.r; p++)
        for (int o = -g.r; o <= g.r; o++) {
          const int sx = g.s2 * o, sy = g.s2 * p;
          int x_lo = ceildiv_i(l - 2 * g.kr - g.md - sx, g.s1), x_hi = floordiv_i(l - g.md - sx, g.s1);
          int y_lo = ceildiv_i(m - 2 * g.kr - g.md - sy, g.s1), y_hi = floordiv_i(m - g.md - sy, g.s1);
          if (!(x_hi >= 0 && y_hi >= 0 && x_lo <= g.ow - 1 && y_lo <= g.oh - 1)) continue;
          x_lo = max(x_lo, 0); x_hi = min(x_hi, g.ow - 1);
          y_lo = max(y_lo, 0); y_hi = min(y_hi, g.oh - 1);
          const int yy0 = y - sy, xx0 = x - sx;
          if ((unsigned)yy0 >= (unsigned)H || (unsigned)xx0 >= (unsigned)W) continue;
          const float v0 = in0[(((size_t)n * H + yy0) * W + xx0) * ld_in + c];
          const int ch = (p + g.r) * g.gw + (o + g.r);
          for (int yy = y_lo; yy <= y_hi; yy++)
            for (int xx = x_lo; xx <= x_hi; xx++)
              acc1 += dout[(((size_t)n * g.oh + yy) * g.ow + xx) * ld_dout + ch] * v0;
        }
    }
    if (fuse) {
      g0[pxl * ld_g + c] = acc0 / denom + acc1 / denom;
    } else {
      g0[pxl * ld_g + c] = acc0 / denom;
      g1[pxl * ld_g + c] = acc1 / denom;
    }
  }
}

// ------------------------------------------------------------------ layout transposes for the NCHW boundary
// [B, C, HW] <-> [B, HW, C] through a 32x33 LDS tile (coalesced on both sides).
__global__ void transpose_bcn_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int S) {
  // in: [B, R, S] -> out: [B, S, R]
  __shared__ float tile[32][33];
  const size_t b = blockIdx.z;
  const int r0 = blockIdx.y * 32, s0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, s = s0 + tx;
    tile[j][tx] = (r < R && s < S) ? in[(b * R + r) * S + s] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int s = s0 + j, r = r0 + tx;
    if (s < S && r < R) out[(b * S + s) * R + r] = tile[tx][j];
  }
}

// [B, C, HW] fp32 -> the tensor's three bf16 operand planes [plane][B * HW][C] (and, with `nhwc`, its channels-last fp32 copy
// in the same pass): 64 channels x 32 sites per workgroup through LDS, 128-byte rows on the way out.  C % 16 == 0.
__global__ __launch_bounds__(256) void transpose_bcn_planes_kernel(const float* __restrict__ in, unsigned short* __restrict__ pl,
                                                                    long plane_stride, float* __restrict__ nhwc, int C, int S) {
  __shared__ float tile[64][33];
  const size_t b = blockIdx.z;
  const int c0 = blockIdx.y * 64, s0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 64; j += 8) {
    const int c = c0 + j, sx = s0 + tx;
    tile[j][tx] = (c < C && sx < S) ? in[(b * C + c) * S + sx] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {                   // site s0 + j: lane tx writes channels c0 + 2 tx, + 1
    const int sx = s0 + j, c = c0 + 2 * tx;
    if (sx >= S || c >= C) continue;
    const float v0 = tile[2 * tx][j], v1 = tile[2 * tx + 1][j];
    unsigned short h0, m0, l0, h1, m1, l1;
    igemm::split3(v0, h0, m0, l0);
    igemm::split3(v1, h1, m1, l1);
    const size_t e = (b * S + sx) * (size_t)C + c;
    *reinterpret_cast<unsigned*>(pl + e) = h0 | ((unsigned)h1 << 16);
    *reinterpret_cast<unsigned*>(pl + plane_stride + e) = m0 | ((unsigned)m1 << 16);
    *reinterpret_cast<unsigned*>(pl + 2 * plane_stride + e) = l0 | ((unsigned)l1 << 16);
    if (nhwc) *reinterpret_cast<float2*>(nhwc + e) = make_float2(v0, v1);
  }
}

static int transpose_bcn(const float* in, float* out, int B, int R, int S, hipStream_t st) {
  dim3 grid(cdiv(S, 32), cdiv(R, 32), B);
  transpose_bcn_kernel<<<grid, 256, 0, st>>>(in, out, R, S);
  return launch_status();
}

// ------------------------------------------------------------------ dispatch
int corr_mfma_supported(const CorrGeom& g, int C, int ld_in);
int corr_mfma_fwd(const float* in0, const float* in1, int ld_in, int shift, float* out, int ld_out, int B, int C,
                  int H, int W, const CorrGeom& g, hipStream_t st);
int corr_mfma_bwd(const float* dout, int ld_dout, const float* in0, const float* in1, int ld_in, int shift, float* g0,
                  float* g1, int ld_g, int fuse, int B, int C, int H, int W, const CorrGeom& g, hipStream_t st);

// operand-plane forward (correlation_planes.hip)
int corr_pl_supported(const CorrGeom& g, int C, const unflow_planes* a, const unflow_planes* b);
int corr_pl_fwd(const unflow_planes* in0, const unflow_planes* in1, int shift, float* out, int ld_out, int B, int C, int H,
                int W, const CorrGeom& g, hipStream_t st);
int corr_pl_bwd(const float* dout, int ld_dout, const unflow_planes* in0, const unflow_planes* in1, int shift, float* g0, float* g1,
                int ld_g, int fuse, int B, int C, int H, int W, const CorrGeom& g, hipStream_t st);

// The plane kernels address a plane through one buffer descriptor with 32-bit byte offsets, and the output with int element
// offsets: larger tensors take the fp32 kernels (64-bit addressing).
static bool corr_pl_fits_32bit(const unflow_planes* pl, int B, int H, int W, int ld_out, const CorrGeom& g) {
  const unsigned long long plane_bytes = (unsigned long long)B * H * W * (unsigned long long)pl->ld * 2ull;
  const unsigned long long out_elems = (unsigned long long)B * g.oh * g.ow * (unsigned long long)ld_out;
  return plane_bytes < (1ull << 31) && out_elems < (1ull << 31);
}

static int corr_status(int H, int W, int k, int md, int pad, int s1, int s2, CorrGeom* g) {
  if (k <= 0 || s1 <= 0 || s2 <= 0 || md < 0 || pad < 0) return UNFLOW_ERR_SHAPE;
  if (k % 2 == 0) return UNFLOW_ERR_EVEN_KERNEL;
  *g = make_corr_geom(H, W, k, md, pad, s1, s2);
  if (g->ow <= 0 || g->oh <= 0) return UNFLOW_ERR_EMPTY_OUTPUT;
  return UNFLOW_OK;
}

UNFLOW_API int unflow_correlation_out_shape(int H, int W, int kernel_size, int max_displacement, int pad,
                                            int stride_1, int stride_2, int* out3) {
  if (!out3) return UNFLOW_ERR_NULL;
  CorrGeom g;
  const int st = corr_status(H, W, kernel_size, max_displacement, pad, stride_1, stride_2, &g);
  if (st == UNFLOW_ERR_EVEN_KERNEL || st == UNFLOW_ERR_SHAPE) return st;
  out3[0] = g.oc; out3[1] = g.oh; out3[2] = g.ow;
  return st;
}

UNFLOW_API int unflow_correlation_nhwc_fwd(const float* in0, const float* in1, int ld_in, int pair_shift, float* out,
                                           int ld_out, int B, int C, int H, int W, int kernel_size,
                                           int max_displacement, int pad, int stride_1, int stride_2,
                                           unflow_stream_t stream) {
  if (!in0 || !in1 || !out) return UNFLOW_ERR_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || ld_in < C) return UNFLOW_ERR_SHAPE;
  CorrGeom g;
  const int st = corr_status(H, W, kernel_size, max_displacement, pad, stride_1, stride_2, &g);
  if (st != UNFLOW_OK) return st;
  if (ld_out < g.oc) return UNFLOW_ERR_SHAPE;
  if (corr_mfma_supported(g, C, ld_in))
    return corr_mfma_fwd(in0, in1, ld_in, pair_shift, out, ld_out, B, C, H, W, g, as_stream(stream));
  const size_t smem = (size_t)g.k * g.k * C * sizeof(float);
  if (smem > 64 * 1024) return UNFLOW_ERR_UNSUPPORTED;
  dim3 grid(g.ow, g.oh, B);
  corr_fwd_generic_kernel<<<grid, 256, smem, as_stream(stream)>>>(in0, in1, ld_in, pair_shift, out, ld_out, B, C, H, W, g);
  return launch_status();
}

UNFLOW_API int unflow_correlation_nhwc_fwd_pl(const float* in0, const float* in1, int ld_in, const unflow_planes* in0_pl,
                                              const unflow_planes* in1_pl, int pair_shift, float* out, int ld_out, int B,
                                              int C, int H, int W, int kernel_size, int max_displacement, int pad,
                                              int stride_1, int stride_2, unflow_stream_t stream) {
  if (!out) return UNFLOW_ERR_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return UNFLOW_ERR_SHAPE;
  CorrGeom g;
  const int st = corr_status(H, W, kernel_size, max_displacement, pad, stride_1, stride_2, &g);
  if (st != UNFLOW_OK) return st;
  if (ld_out < g.oc) return UNFLOW_ERR_SHAPE;
  if (corr_pl_supported(g, C, in0_pl, in1_pl) && corr_pl_fits_32bit(in0_pl, B, H, W, ld_out, g))
    return corr_pl_fwd(in0_pl, in1_pl, pair_shift, out, ld_out, B, C, H, W, g, as_stream(stream));
  return unflow_correlation_nhwc_fwd(in0, in1, ld_in, pair_shift, out, ld_out, B, C, H, W, kernel_size, max_displacement,
                                     pad, stride_1, stride_2, stream);
}

UNFLOW_API int unflow_correlation_nhwc_bwd(const float* dout, int ld_dout, const float* in0, const float* in1,
                                           int ld_in, int pair_shift, float* grad0, float* grad1, int ld_grad,
                                           int accumulate_g1_into_g0, int B, int C, int H, int W, int kernel_size,
                                           int max_displacement, int pad, int stride_1, int stride_2,
                                           unflow_stream_t stream) {
  if (!dout || !in0 || !in1 || !grad0 || (!grad1 && !accumulate_g1_into_g0)) return UNFLOW_ERR_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || ld_in < C || ld_grad < C) return UNFLOW_ERR_SHAPE;
  CorrGeom g;
  const int st = corr_status(H, W, kernel_size, max_displacement, pad, stride_1, stride_2, &g);
  if (st != UNFLOW_OK) return st;
  if (ld_dout < g.oc) return UNFLOW_ERR_SHAPE;
  if (corr_mfma_supported(g, C, ld_in))
    return corr_mfma_bwd(dout, ld_dout, in0, in1, ld_in, pair_shift, grad0, grad1, ld_grad, accumulate_g1_into_g0, B,
                         C, H, W, g, as_stream(stream));
  corr_bwd_generic_kernel<<<stream_grid((long)B * H * W * C), 256, 0, as_stream(stream)>>>(
      dout, ld_dout, in0, in1, ld_in, pair_shift, grad0, grad1, ld_grad, accumulate_g1_into_g0, B, C, H, W, g);
  return launch_status();
}

UNFLOW_API int unflow_correlation_nhwc_bwd_pl(const float* dout, int ld_dout, const float* in0, const float* in1, int ld_in,
                                              const unflow_planes* in0_pl, const unflow_planes* in1_pl, int pair_shift,
                                              float* grad0, float* grad1, int ld_grad, int accumulate_g1_into_g0, int B, int C,
                                              int H, int W, int kernel_size, int max_displacement, int pad, int stride_1,
                                              int stride_2, unflow_stream_t stream) {
  if (!dout || !grad0 || (!grad1 && !accumulate_g1_into_g0)) return UNFLOW_ERR_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || ld_grad < C) return UNFLOW_ERR_SHAPE;
  CorrGeom g;
  const int st = corr_status(H, W, kernel_size, max_displacement, pad, stride_1, stride_2, &g);
  if (st != UNFLOW_OK) return st;
  if (ld_dout < g.oc) return UNFLOW_ERR_SHAPE;
  const unflow::Options& opt = unflow::options();
  const bool pl_on = opt.corr_bwd_planes && !opt.corr_math_fp32 && !opt.conv_math_fp32;
  if (pl_on && C % 64 == 0 && corr_pl_supported(g, C, in0_pl, in1_pl) && corr_pl_fits_32bit(in0_pl, B, H, W, ld_dout, g))
    return corr_pl_bwd(dout, ld_dout, in0_pl, in1_pl, pair_shift, grad0, grad1, ld_grad, accumulate_g1_into_g0, B, C, H, W, g,
                       as_stream(stream));
  return unflow_correlation_nhwc_bwd(dout, ld_dout, in0, in1, ld_in, pair_shift, grad0, grad1, ld_grad, accumulate_g1_into_g0, B,
                                     C, H, W, kernel_size, max_displacement, pad, stride_1, stride_2, stream);
}

// Operand planes of the two inputs inside the workspace of the reference-layout entry points: [input][plane][pixel][C] bf16,
// 256-byte aligned behind the fp32 part.  0 when the plane kernels do not take the shape.
static size_t corr_ws_plane_bytes(const CorrGeom& g, int B, int C, int H, int W) {
  if (g.k != 1 || g.s1 != 1 || g.md - g.pad > 0 || C % 16 != 0 || C > 1024) return 0;
  return 2 * 3 * (size_t)B * H * W * C * 2 + 256;
}
// lays them out behind the fp32 part; returns false when there is no room (or the shape is not theirs)
static bool corr_ws_planes(const CorrGeom& g, void* ws_end_fp32, size_t bytes_left, int B, int C, int H, int W, unflow_planes* pa,
                           unflow_planes* pb) {
  const size_t need = corr_ws_plane_bytes(g, B, C, H, W);
  if (!need || bytes_left < need) return false;
  const size_t npix = (size_t)B * H * W;
  unsigned char* base = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(ws_end_fp32) + 255) & ~(uintptr_t)255);
  *pa = unflow_planes{base, (long)(npix * C), C, 3, 0.f};
  *pb = unflow_planes{base + 3 * npix * C * 2, (long)(npix * C), C, 3, 0.f};
  return true;
}
// NCHW input -> planes (+ the channels-last fp32 copy when `nhwc` is given) in one pass
static int transpose_to_planes(const float* in, const unflow_planes& p, float* nhwc, int B, int C, int S, hipStream_t st) {
  dim3 grid(cdiv(S, 32), cdiv(C, 64), B);
  transpose_bcn_planes_kernel<<<grid, 256, 0, st>>>(in, reinterpret_cast<unsigned short*>(p.base), p.plane_stride, nhwc, C, S);
  return launch_status();
}

UNFLOW_API size_t unflow_correlation_workspace_bytes(int B, int C, int H, int W, int kernel_size,
                                                     int max_displacement, int pad, int stride_1, int stride_2) {
  CorrGeom g;
  if (corr_status(H, W, kernel_size, max_displacement, pad, stride_1, stride_2, &g) != UNFLOW_OK) return 0;
  const size_t in_e = (size_t)B * H * W * C, out_e = (size_t)B * g.oh * g.ow * g.oc;
  // bwd: in0,in1,g0,g1 + dout (fwd needs 2*in + out) as NHWC fp32, + the two inputs' bf16 x 3 operand planes when the matrix-core
  // kernels of correlation_planes.hip take the shape (the entry points below use them iff the workspace has room for them)
  return (4 * in_e + out_e) * sizeof(float) + corr_ws_plane_bytes(g, B, C, H, W);
}

UNFLOW_API int unflow_correlation_fwd(const float* in0, const float* in1, float* out, int B, int C, int H, int W,
                                      int kernel_size, int max_displacement, int pad, int stride_1, int stride_2,
                                      void* workspace, size_t workspace_bytes, unflow_stream_t stream) {
  if (!in0 || !in1 || !out) return UNFLOW_ERR_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return UNFLOW_ERR_SHAPE;
  CorrGeom g;
  const int st = corr_status(H, W, kernel_size, max_displacement, pad, stride_1, stride_2, &g);
  if (st != UNFLOW_OK) return st;
  const size_t in_e = (size_t)B * H * W * C, out_e = (size_t)B * g.oh * g.ow * g.oc;
  if (!workspace) return UNFLOW_ERR_NULL;
  if (workspace_bytes < (2 * in_e + out_e) * sizeof(float)) return UNFLOW_ERR_WORKSPACE;
  float* a = reinterpret_cast<float*>(workspace);
  float* b = a + in_e;
  float* o = b + in_e;
  hipStream_t s = as_stream(stream);
  int code;
  // with room for them in the workspace (unflow_correlation_workspace_bytes asks for it): operand planes straight from the NCHW
  // inputs and the matrix-core kernels of the training step; otherwise channels-last fp32 copies and the fp32 kernels
  unflow_planes pa, pb;
  const size_t used = (2 * in_e + out_e) * sizeof(float);
  if (corr_ws_planes(g, o + out_e, workspace_bytes - used, B, C, H, W, &pa, &pb) && corr_pl_supported(g, C, &pa, &pb) &&
      corr_pl_fits_32bit(&pa, B, H, W, g.oc, g)) {
    if ((code = transpose_to_planes(in0, pa, nullptr, B, C, H * W, s)) != UNFLOW_OK) return code;
    if ((code = transpose_to_planes(in1, pb, nullptr, B, C, H * W, s)) != UNFLOW_OK) return code;
    code = corr_pl_fwd(&pa, &pb, 0, o, g.oc, B, C, H, W, g, s);
  } else {
    if ((code = transpose_bcn(in0, a, B, C, H * W, s)) != UNFLOW_OK) return code;
    if ((code = transpose_bcn(in1, b, B, C, H * W, s)) != UNFLOW_OK) return code;
    code = unflow_correlation_nhwc_fwd(a, b, C, 0, o, g.oc, B, C, H, W, kernel_size, max_displacement, pad, stride_1, stride_2,
                                       stream);
  }
  if (code != UNFLOW_OK) return code;
  return transpose_bcn(o, out, B, g.oh * g.ow, g.oc, s);
}

UNFLOW_API int unflow_correlation_bwd(const float* dout, const float* in0, const float* in1, float* grad0,
                                      float* grad1, int B, int C, int H, int W, int kernel_size, int max_displacement,
                                      int pad, int stride_1, int stride_2, void* workspace, size_t workspace_bytes,
                                      unflow_stream_t stream) {
  if (!dout || !in0 || !in1 || !grad0 || !grad1) return UNFLOW_ERR_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return UNFLOW_ERR_SHAPE;
  CorrGeom g;
  const int st = corr_status(H, W, kernel_size, max_displacement, pad, stride_1, stride_2, &g);
  if (st != UNFLOW_OK) return st;
  const size_t in_e = (size_t)B * H * W * C, out_e = (size_t)B * g.oh * g.ow * g.oc;
  if (!workspace) return UNFLOW_ERR_NULL;
  if (workspace_bytes < (4 * in_e + out_e) * sizeof(float)) return UNFLOW_ERR_WORKSPACE;
  float* a = reinterpret_cast<float*>(workspace);
  float* b = a + in_e;
  float* ga = b + in_e;
  float* gb = ga + in_e;
  float* d = gb + in_e;
  hipStream_t s = as_stream(stream);
  int code;
  if ((code = transpose_bcn(dout, d, B, g.oc, g.oh * g.ow, s)) != UNFLOW_OK) return code;
  unflow_planes pa, pb;
  const size_t used = (4 * in_e + out_e) * sizeof(float);
  if (corr_ws_planes(g, d + out_e, workspace_bytes - used, B, C, H, W, &pa, &pb)) {
    // planes (and the fp32 copies the entry point falls back to when the planes kernel does not take the shape) in one pass
    if ((code = transpose_to_planes(in0, pa, a, B, C, H * W, s)) != UNFLOW_OK) return code;
    if ((code = transpose_to_planes(in1, pb, b, B, C, H * W, s)) != UNFLOW_OK) return code;
    code = unflow_correlation_nhwc_bwd_pl(d, g.oc, a, b, C, &pa, &pb, 0, ga, gb, C, 0, B, C, H, W, kernel_size, max_displacement, pad,
                                          stride_1, stride_2, stream);
  } else {
    if ((code = transpose_bcn(in0, a, B, C, H * W, s)) != UNFLOW_OK) return code;
    if ((code = transpose_bcn(in1, b, B, C, H * W, s)) != UNFLOW_OK) return code;
    code = unflow_correlation_nhwc_bwd(d, g.oc, a, b, C, 0, ga, gb, C, 0, B, C, H, W, kernel_size, max_displacement, pad, stride_1,
                                       stride_2, stream);
  }
  if (code != UNFLOW_OK) return code;
  if ((code = transpose_bcn(ga, grad0, B, H * W, C, s)) != UNFLOW_OK) return code;
  return transpose_bcn(gb, grad1, B, H * W, C, s);
}
