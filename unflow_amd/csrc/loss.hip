// Proxy-loss kernels of the unsupervised step (src/e2eflow/core/losses.py), fused per pyramid level.
// The reference builds each term from ~15-60 small TF ops (identity-filter conv2d to extract census
// patches, gathers, pads, pows ...); here each term is one stencil kernel forward and one gather-form
// kernel backward (no atomics on gradients, no [B,H,W,49] patch tensor).
//
// "Directed batch": N = 2B samples, [0,B) forward flow (first = im1, second = im2), [B,2B) backward;
// the second image of sample n is image (n + pair_shift) % N.
#include "common.h"

#define CHARB_ALPHA 0.45f
#define CHARB_EPS 0.001f

// ------------------------------------------------------------------ grayscale
// tf.image.rgb_to_grayscale weights, then * 255 (losses.py:94).
__device__ __forceinline__ float gray255(float r, float g, float b) {
  return ((r * 0.2989f + g * 0.5870f) + b * 0.1140f) * 255.0f;
}

__global__ void rgb_to_gray255_kernel(const float* __restrict__ im, int ld, float* __restrict__ gray, long npix) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x)
    gray[i] = gray255(im[i * ld], im[i * ld + 1], im[i * ld + 2]);
}

UNFLOW_API int unflow_rgb_to_gray255(const float* im, int ld_im, float* gray, long npix, unflow_stream_t stream) {
  if (!im || !gray) return UNFLOW_ERR_NULL;
  if (npix <= 0) return UNFLOW_OK;
  rgb_to_gray255_kernel<<<stream_grid(npix), 256, 0, as_stream(stream)>>>(im, ld_im, gray, npix);
  return launch_status();
}

// ------------------------------------------------------------------ image_warp (image_warp.py:4-76) + gray, fused
struct Taps {
  long ia, ib, ic, id;
  float xw, yw, wa, wb, wc, wd;
};

__device__ __forceinline__ Taps iw_taps(int px, int py, float u, float v, int H, int W) {
  Taps t;
  const float fu = floorf(u), fv = floorf(v);
  t.xw = u - fu;
  t.yw = v - fv;
  t.wa = (1.f - t.xw) * (1.f - t.yw);
  t.wb = (1.f - t.xw) * t.yw;
  t.wc = t.xw * (1.f - t.yw);
  t.wd = t.xw * t.yw;
  const int xi = px + (int)fu, yi = py + (int)fv;
  const int x0 = min(max(xi, 0), W - 1), x1 = min(max(xi + 1, 0), W - 1);
  const int y0 = min(max(yi, 0), H - 1), y1 = min(max(yi + 1, 0), H - 1);
  t.ia = (long)y0 * W + x0;
  t.ib = (long)y1 * W + x0;
  t.ic = (long)y0 * W + x1;
  t.id = (long)y1 * W + x1;
  return t;
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
__global__ void warp_gray_fwd_kernel(const float* __restrict__ im, int ld, const float* __restrict__ flow,
                                     float fscale, float* __restrict__ out, int shift, int N, int H, int W) {
  const unsigned T = tile_count(W, H, N);
  for (unsigned tile = tile_first(T), tile_end = tile_last(T); tile < tile_end; tile++) {
    const TilePix pp = tile_pix(tile, W, H);
    if (!pp.ok) continue;
    const unsigned i = pp.i;
    const int px = pp.x, py = pp.y, n = pp.n;
    const long sb = (long)((n + shift) % N) * H * W;
    const f32x2_t fv = __builtin_nontemporal_load(reinterpret_cast<const f32x2_t*>(flow) + i);   // streamed once
    const float2 f = make_float2(fv.x, fv.y);
    const Taps t = iw_taps(px, py, f.x * fscale, f.y * fscale, H, W);
    const float *pa = im + (sb + t.ia) * ld, *pb = im + (sb + t.ib) * ld, *pc = im + (sb + t.ic) * ld,
                *pd = im + (sb + t.id) * ld;
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = ((t.wa * pa[k] + t.wb * pb[k]) + t.wc * pc[k]) + t.wd * pd[k];
    __builtin_nontemporal_store(gray255(c[0], c[1], c[2]), out + i);
  }
}

// gray(im1) and gray(image_warp(im2, flow)) of a pyramid level in ONE launch (the two planes the census loss compares).
__device__ __forceinline__ void gray_pair_body(const float* __restrict__ im, int ld, const float* __restrict__ flow,
                                               float fscale, float* __restrict__ gray1, float* __restrict__ gray2w,
                                               int shift, int N, int H, int W, unsigned vb, unsigned vg) {
  const unsigned npx = (unsigned)N * H * W;
  for (unsigned i = vb * blockDim.x + threadIdx.x; i < npx; i += vg * blockDim.x) {
    const Pix pp = decode_pix(i, W, H);
    const int px = pp.x, py = pp.y, n = pp.n;
    const long sb = (long)((n + shift) % N) * H * W;
    const float* own = im + (size_t)i * ld;
    gray1[i] = gray255(own[0], own[1], own[2]);
    const float2 f = reinterpret_cast<const float2*>(flow)[i];
    const Taps t = iw_taps(px, py, f.x * fscale, f.y * fscale, H, W);
    const float *pa = im + (sb + t.ia) * ld, *pb = im + (sb + t.ib) * ld, *pc = im + (sb + t.ic) * ld,
                *pd = im + (sb + t.id) * ld;
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = ((t.wa * pa[k] + t.wb * pb[k]) + t.wc * pc[k]) + t.wd * pd[k];
    gray2w[i] = gray255(c[0], c[1], c[2]);
  }
}
__global__ void gray_pair_kernel(const float* __restrict__ im, int ld, const float* __restrict__ flow, float fscale,
                                 float* __restrict__ gray1, float* __restrict__ gray2w, int shift, int N, int H, int W) {
  gray_pair_body(im, ld, flow, fscale, gray1, gray2w, shift, N, H, W, blockIdx.x, gridDim.x);
}

// flow gradient of gray(image_warp(im, flow)) at one pixel, given d(loss)/d(gray) there (shared by warp_gray_bwd_kernel
// and the fused census backward)
__device__ __forceinline__ void warp_gray_bwd_pixel(float dg, const float* __restrict__ im, int ld,
                                                    const float* __restrict__ flow, float fscale,
                                                    float* __restrict__ dflow, int acc, int shift, int N, int H, int W,
                                                    unsigned i, int px, int py, int n) {
  const long sb = (long)((n + shift) % N) * H * W;
  const float2 f = reinterpret_cast<const float2*>(flow)[i];
  const Taps t = iw_taps(px, py, f.x * fscale, f.y * fscale, H, W);
  const float *pa = im + (sb + t.ia) * ld, *pb = im + (sb + t.ib) * ld, *pc = im + (sb + t.ic) * ld,
              *pd = im + (sb + t.id) * ld;
  const float g = dg * 255.0f;
  const float gw[3] = {g * 0.2989f, g * 0.5870f, g * 0.1140f};
  float ga = 0.f, gb = 0.f, gc = 0.f, gd = 0.f;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    ga += gw[k] * pa[k];
    gb += gw[k] * pb[k];
    gc += gw[k] * pc[k];
    gd += gw[k] * pd[k];
  }
  float du = ((gc - ga) * (1.f - t.yw) + (gd - gb) * t.yw) * fscale;
  float dv = ((gb - ga) * (1.f - t.xw) + (gd - gc) * t.xw) * fscale;
  float2* o = reinterpret_cast<float2*>(dflow) + i;
  if (acc) {
    const float2 e = *o;
    du += e.x;
    dv += e.y;
  }
  *o = make_float2(du, dv);
}

__global__ void warp_gray_bwd_kernel(const float* __restrict__ dgray, const float* __restrict__ im, int ld,
                                     const float* __restrict__ flow, float fscale, float* __restrict__ dflow, int acc,
                                     int shift, int N, int H, int W) {
  const unsigned T = tile_count(W, H, N);
  for (unsigned tile = tile_first(T), tile_end = tile_last(T); tile < tile_end; tile++) {
    const TilePix pp = tile_pix(tile, W, H);
    if (!pp.ok) continue;
    const unsigned i = pp.i;
    warp_gray_bwd_pixel(dgray[i], im, ld, flow, fscale, dflow, acc, shift, N, H, W, i, pp.x, pp.y, pp.n);
  }
}

UNFLOW_API int unflow_warp_gray_fwd(const float* im, int ld_im, const float* flow, float flow_scale, float* out_gray,
                                    int pair_shift, int N, int H, int W, unflow_stream_t stream) {
  if (!im || !flow || !out_gray) return UNFLOW_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || ld_im < 3) return UNFLOW_ERR_SHAPE;
  warp_gray_fwd_kernel<<<stream_grid((long)N * H * W), 256, 0, as_stream(stream)>>>(im, ld_im, flow, flow_scale,
                                                                                    out_gray, pair_shift, N, H, W);
  return launch_status();
}

UNFLOW_API int unflow_warp_gray_bwd(const float* d_gray, const float* im, int ld_im, const float* flow,
                                    float flow_scale, float* d_flow, int accumulate, int pair_shift, int N, int H,
                                    int W, unflow_stream_t stream) {
  if (!d_gray || !im || !flow || !d_flow) return UNFLOW_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || ld_im < 3) return UNFLOW_ERR_SHAPE;
  warp_gray_bwd_kernel<<<stream_grid((long)N * H * W), 256, 0, as_stream(stream)>>>(
      d_gray, im, ld_im, flow, flow_scale, d_flow, accumulate, pair_shift, N, H, W);
  return launch_status();
}

UNFLOW_API int unflow_gray_pair(const float* im, int ld_im, const float* flow, float flow_scale, float* gray1,
                                float* gray2w, int pair_shift, int N, int H, int W, unflow_stream_t stream) {
  if (!im || !flow || !gray1 || !gray2w) return UNFLOW_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || ld_im < 3) return UNFLOW_ERR_SHAPE;
  gray_pair_kernel<<<stream_grid((long)N * H * W), 256, 0, as_stream(stream)>>>(im, ld_im, flow, flow_scale, gray1,
                                                                                gray2w, pair_shift, N, H, W);
  return launch_status();
}

// ------------------------------------------------------------------ ternary / census loss (losses.py:90-122)
// t(z) = z / sqrt(0.81 + z^2);  soft Hamming term h(d) = d^2 / (0.1 + d^2), d = t1 - t2.
// 32x8 pixel tiles with a max_distance halo staged in LDS (each gray value is read (2D+1)^2 times); rsq / rcp
// hardware forms; the forward pass also leaves dL/d(dist) per pixel so the backward pass needs no pow.
__device__ __forceinline__ float census_t(float z) { return z * fast_rsqrt(0.81f + z * z); }

constexpr int CT_W = 32, CT_H = 8, CT_MAXD = 4;
constexpr int CT_LW = CT_W + 2 * CT_MAXD, CT_LH = CT_H + 2 * CT_MAXD;

// loads a (CT_W+2D)x(CT_H+2D) tile of a [H,W] plane into LDS (zero outside the image: SAME padding of losses.py:104)
__device__ __forceinline__ void load_tile(float* __restrict__ lds, const float* __restrict__ plane, int x0, int y0, int D,
                                          int H, int W) {
  const int tw = CT_W + 2 * D, th = CT_H + 2 * D;
  for (int e = threadIdx.x; e < tw * th; e += 256) {
    const int ly = e / tw, lx = e - ly * tw;
    const int gy = y0 - D + ly, gx = x0 - D + lx;
    lds[ly * CT_LW + lx] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? plane[(long)gy * W + gx] : 0.f;
  }
}

// DT: max_distance known at compile time (1..3: the reference's values, unsupervised.py:88; the tap loops unroll and their LDS
// offsets become immediates), 0: the run-time value D.  The LDS tiles belong to the caller (one set per kernel, not per DT).
template <int DT>
__device__ __forceinline__ void ternary_fwd_body_t(float* __restrict__ t1, float* __restrict__ t2, float* __restrict__ red,
                                                   const float* __restrict__ g1, const float* __restrict__ g2,
                                                   const float* __restrict__ mask, int n_mask,
                                                   float* __restrict__ wgt_out, float* __restrict__ loss_acc, float scale,
                                                   int D_rt, int N, int H, int W, unsigned vb, unsigned vg) {
  const int D = DT ? DT : D_rt;
  const int tiles_x = (W + CT_W - 1) / CT_W, tiles_y = (H + CT_H - 1) / CT_H;
  const int ntiles = tiles_x * tiles_y * N;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  float local = 0.f;
  // grid-stride over tiles: one loss atomic per BLOCK, not per tile (49k same-address atomics at 768x1024x16)
  for (int tile = (int)vb; tile < ntiles; tile += (int)vg) {
    const int n = tile / (tiles_x * tiles_y), tr = tile - n * tiles_x * tiles_y;
    const int x0 = (tr % tiles_x) * CT_W, y0 = (tr / tiles_x) * CT_H;
    __syncthreads();   // previous tile fully consumed
    load_tile(t1, g1 + (long)n * H * W, x0, y0, D, H, W);
    load_tile(t2, g2 + (long)n * H * W, x0, y0, D, H, W);
    __syncthreads();
    const int x = x0 + tx, y = y0 + ty;
    if (!(x < W && y < H)) continue;
    {
    const float c1 = t1[(ty + D) * CT_LW + tx + D], c2 = t2[(ty + D) * CT_LW + tx + D];
    float dist = 0.f;
#pragma unroll
    for (int dy = 0; dy <= 2 * (DT ? DT : CT_MAXD); dy++)
#pragma unroll
      for (int dx = 0; dx <= 2 * (DT ? DT : CT_MAXD); dx++) {
        if (!DT && (dy > 2 * D || dx > 2 * D)) continue;
        const float d = census_t(t1[(ty + dy) * CT_LW + tx + dx] - c1) - census_t(t2[(ty + dy) * CT_LW + tx + dx] - c2);
        const float d2 = d * d;
        dist += d2 * fast_rcp(0.1f + d2);
      }
    const bool interior = y >= D && y < H - D && x >= D && x < W - D;
    float wgt = 0.f;
    if (interior) {
      const float m = mask[(long)(n % n_mask) * H * W + (long)y * W + x];
      const float a = dist * dist + CHARB_EPS * CHARB_EPS;
      const float l2 = __builtin_amdgcn_logf(a);
      local += m * __builtin_amdgcn_exp2f(CHARB_ALPHA * l2);
      wgt = scale * m * CHARB_ALPHA * __builtin_amdgcn_exp2f((CHARB_ALPHA - 1.f) * l2) * 2.f * dist;
    }
    wgt_out[(long)n * H * W + (long)y * W + x] = wgt;
    }
  }
  const float t = block_sum(local, red);
  if (threadIdx.x == 0 && loss_acc) atomicAdd(loss_acc, t * scale);
}
__device__ __forceinline__ void ternary_fwd_body(const float* __restrict__ g1, const float* __restrict__ g2,
                                                 const float* __restrict__ mask, int n_mask,
                                                 float* __restrict__ wgt_out, float* __restrict__ loss_acc, float scale,
                                                 int D, int N, int H, int W, unsigned vb, unsigned vg) {
  __shared__ float t1[CT_LH * CT_LW], t2[CT_LH * CT_LW];
  __shared__ float red[4];
  switch (D) {
    case 1: ternary_fwd_body_t<1>(t1, t2, red, g1, g2, mask, n_mask, wgt_out, loss_acc, scale, D, N, H, W, vb, vg); break;
    case 2: ternary_fwd_body_t<2>(t1, t2, red, g1, g2, mask, n_mask, wgt_out, loss_acc, scale, D, N, H, W, vb, vg); break;
    case 3: ternary_fwd_body_t<3>(t1, t2, red, g1, g2, mask, n_mask, wgt_out, loss_acc, scale, D, N, H, W, vb, vg); break;
    default: ternary_fwd_body_t<0>(t1, t2, red, g1, g2, mask, n_mask, wgt_out, loss_acc, scale, D, N, H, W, vb, vg); break;
  }
}
__global__ __launch_bounds__(256) void ternary_fwd_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                          const float* __restrict__ mask, int n_mask,
                                                          float* __restrict__ wgt_out, float* __restrict__ loss_acc,
                                                          float scale, int D, int N, int H, int W) {
  ternary_fwd_body(g1, g2, mask, n_mask, wgt_out, loss_acc, scale, D, N, H, W, xcd_block(), gridDim.x);
}

// Gather form: d/dG2(q) = sum_e f(q, q+e) * (Wt(q+e) + Wt(q)), see DESIGN.md (census backward).
// With `im` non-null the pixel's d(loss)/d(gray2w) goes straight into the flow gradient (warp_gray_bwd_pixel) instead of
// (or in addition to) the dg2 plane: one launch and one [N,H,W] round trip less per pyramid level.
template <int DT>
__device__ __forceinline__ void ternary_bwd_body_t(float* __restrict__ t1, float* __restrict__ t2, float* __restrict__ tw,
                                                   const float* __restrict__ g1, const float* __restrict__ g2,
                                                   const float* __restrict__ wgt, float* __restrict__ dg2, int D_rt, int N,
                                                   int H, int W, const float* __restrict__ im, int ld,
                                                   const float* __restrict__ flow, float fscale, float* __restrict__ dflow,
                                                   int acc, int shift, unsigned tile) {
  const int D = DT ? DT : D_rt;
  const int tiles_x = (W + CT_W - 1) / CT_W, tiles_y = (H + CT_H - 1) / CT_H;
  const int n = (int)tile / (tiles_x * tiles_y), tr = (int)tile - n * tiles_x * tiles_y;
  const int x0 = (tr % tiles_x) * CT_W, y0 = (tr / tiles_x) * CT_H;
  load_tile(t1, g1 + (long)n * H * W, x0, y0, D, H, W);
  load_tile(t2, g2 + (long)n * H * W, x0, y0, D, H, W);
  load_tile(tw, wgt + (long)n * H * W, x0, y0, D, H, W);   // zero outside the image == "no such centre pixel"
  __syncthreads();
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int x = x0 + tx, y = y0 + ty;
  if (x >= W || y >= H) return;
  const float q1 = t1[(ty + D) * CT_LW + tx + D], q2 = t2[(ty + D) * CT_LW + tx + D], wq = tw[(ty + D) * CT_LW + tx + D];
  float grad = 0.f;
#pragma unroll
  for (int dy = 0; dy <= 2 * (DT ? DT : CT_MAXD); dy++)
#pragma unroll
    for (int dx = 0; dx <= 2 * (DT ? DT : CT_MAXD); dx++) {
      if (!DT && (dy > 2 * D || dx > 2 * D)) continue;
      const int yy = y + dy - D, xx = x + dx - D;
      if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;   // no centre pixel there
      const float ws = tw[(ty + dy) * CT_LW + tx + dx] + wq;
      // r = q+e as centre with neighbour q: z = G(q) - G(r)
      const float z1 = q1 - t1[(ty + dy) * CT_LW + tx + dx], z2 = q2 - t2[(ty + dy) * CT_LW + tx + dx];
      const float r2 = fast_rsqrt(0.81f + z2 * z2);
      const float d = census_t(z1) - z2 * r2;
      const float iden = fast_rcp(0.1f + d * d);
      const float dh = 2.f * d * 0.1f * iden * iden;        // dh/dd
      const float dt2 = 0.81f * r2 * r2 * r2;               // dt/dz at z2
      grad -= dh * dt2 * ws;                                // dd/dt2 = -1, dz2/dG2(q) = +1
    }
  const unsigned i = (unsigned)(((long)n * H + y) * W + x);
  if (dg2) dg2[i] = grad;
  if (im) warp_gray_bwd_pixel(grad, im, ld, flow, fscale, dflow, acc, shift, N, H, W, i, x, y, n);
}
__device__ __forceinline__ void ternary_bwd_body(const float* __restrict__ g1, const float* __restrict__ g2,
                                                 const float* __restrict__ wgt, float* __restrict__ dg2, int D, int N,
                                                 int H, int W, const float* __restrict__ im, int ld,
                                                 const float* __restrict__ flow, float fscale, float* __restrict__ dflow,
                                                 int acc, int shift, unsigned tile) {
  __shared__ float t1[CT_LH * CT_LW], t2[CT_LH * CT_LW], tw[CT_LH * CT_LW];
  switch (D) {
    case 1: ternary_bwd_body_t<1>(t1, t2, tw, g1, g2, wgt, dg2, D, N, H, W, im, ld, flow, fscale, dflow, acc, shift, tile); break;
    case 2: ternary_bwd_body_t<2>(t1, t2, tw, g1, g2, wgt, dg2, D, N, H, W, im, ld, flow, fscale, dflow, acc, shift, tile); break;
    case 3: ternary_bwd_body_t<3>(t1, t2, tw, g1, g2, wgt, dg2, D, N, H, W, im, ld, flow, fscale, dflow, acc, shift, tile); break;
    default: ternary_bwd_body_t<0>(t1, t2, tw, g1, g2, wgt, dg2, D, N, H, W, im, ld, flow, fscale, dflow, acc, shift, tile); break;
  }
}
__global__ __launch_bounds__(256) void ternary_bwd_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                          const float* __restrict__ wgt, float* __restrict__ dg2, int D,
                                                          int N, int H, int W, const float* __restrict__ im, int ld,
                                                          const float* __restrict__ flow, float fscale,
                                                          float* __restrict__ dflow, int acc, int shift) {
  ternary_bwd_body(g1, g2, wgt, dg2, D, N, H, W, im, ld, flow, fscale, dflow, acc, shift, xcd_block());
}

UNFLOW_API int unflow_ternary_fwd(const float* gray1, const float* gray2w, const float* mask, int n_mask,
                                  float* dist_out, float* loss_acc, float weight, float normalizer, int max_distance,
                                  int N, int H, int W, unflow_stream_t stream) {
  if (!gray1 || !gray2w || !mask || !dist_out) return UNFLOW_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || n_mask <= 0 || max_distance < 0) return UNFLOW_ERR_SHAPE;
  if (max_distance > CT_MAXD) return UNFLOW_ERR_UNSUPPORTED;   // the reference uses 1..3 (unsupervised.py:88)
  const long ntiles = (long)cdiv(W, CT_W) * cdiv(H, CT_H) * N;
  ternary_fwd_kernel<<<(int)min(ntiles, (long)2048), 256, 0, as_stream(stream)>>>(gray1, gray2w, mask, n_mask, dist_out, loss_acc,
                                                          weight / normalizer, max_distance, N, H, W);
  return launch_status();
}

UNFLOW_API int unflow_ternary_bwd(const float* gray1, const float* gray2w, const float* mask, int n_mask,
                                  const float* dist, float* d_gray2w, float weight, float normalizer,
                                  int max_distance, int N, int H, int W, unflow_stream_t stream) {
  if (!gray1 || !gray2w || !dist || !d_gray2w) return UNFLOW_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || max_distance < 0) return UNFLOW_ERR_SHAPE;
  if (max_distance > CT_MAXD) return UNFLOW_ERR_UNSUPPORTED;
  (void)mask; (void)n_mask; (void)weight; (void)normalizer;   // already folded into `dist` (= dL/d dist) by the forward pass
  const int grid = cdiv(W, CT_W) * cdiv(H, CT_H) * N;
  ternary_bwd_kernel<<<grid, 256, 0, as_stream(stream)>>>(gray1, gray2w, dist, d_gray2w, max_distance, N, H, W, nullptr, 0,
                                                          nullptr, 0.f, nullptr, 0, 0);
  return launch_status();
}

UNFLOW_API int unflow_ternary_warp_bwd(const float* gray1, const float* gray2w, const float* dist, const float* im,
                                       int ld_im, const float* flow, float flow_scale, float* d_flow, int accumulate,
                                       int pair_shift, int max_distance, int N, int H, int W, unflow_stream_t stream) {
  if (!gray1 || !gray2w || !dist || !im || !flow || !d_flow) return UNFLOW_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || max_distance < 0 || ld_im < 3) return UNFLOW_ERR_SHAPE;
  if (max_distance > CT_MAXD) return UNFLOW_ERR_UNSUPPORTED;
  const int grid = cdiv(W, CT_W) * cdiv(H, CT_H) * N;
  ternary_bwd_kernel<<<grid, 256, 0, as_stream(stream)>>>(gray1, gray2w, dist, nullptr, max_distance, N, H, W, im, ld_im,
                                                          flow, flow_scale, d_flow, accumulate, pair_shift);
  return launch_status();
}

// ------------------------------------------------------------------ second-order smoothness (losses.py:258-295)
// 4 stencils per flow channel: delta_k(p) = f(p+a_k) + f(p-a_k) - 2 f(p), a = (0,1),(1,0),(1,1),(1,-1) [dy,dx];
// masks (create_mask, :260-263) zero exactly the pixels whose stencil would leave the image.
// Charbonnier on each, normaliser N_dir*H*W*4 per channel (charbonnier_loss :311-312).
// ((x*beta)^2 + eps^2)^alpha and its derivative via exp2/log2 (the base is >= 1e-6 > 0)
__device__ __forceinline__ float charb(float x) { return fast_pow(x * x + CHARB_EPS * CHARB_EPS, CHARB_ALPHA); }
__device__ __forceinline__ float charb_grad(float x) {
  return CHARB_ALPHA * fast_pow(x * x + CHARB_EPS * CHARB_EPS, CHARB_ALPHA - 1.f) * 2.f * x;
}

__device__ __forceinline__ void second_order_body(const float* __restrict__ flow, float fs, float* __restrict__ loss_acc,
                                                  float* __restrict__ dflow, int acc, float scale, int N, int H, int W,
                                                  unsigned vb, unsigned vg) {
  __shared__ float red[4];
  const long npx = (long)N * H * W;
  const int ady[4] = {0, 1, 1, 1}, adx[4] = {1, 0, 1, -1};
  float local = 0.f;
  for (long i = vb * (long)blockDim.x + threadIdx.x; i < npx; i += (long)vg * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long n = i / ((long)W * H);
    const float2* f = reinterpret_cast<const float2*>(flow) + n * H * W;
    float gu = 0.f, gv = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int dy = ady[k], dx = adx[k];
      // delta at centre c = q + j*a for j = -1,0,1 is valid iff c +- a are inside the image
      float2 v[5];
#pragma unroll
      for (int j = -2; j <= 2; j++) {
        const int yy = y + j * dy, xx = x + j * dx;
        const bool inb = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        float2 t = inb ? f[(long)yy * W + xx] : make_float2(0.f, 0.f);
        v[j + 2] = make_float2(t.x * fs, t.y * fs);
      }
#pragma unroll
      for (int j = -1; j <= 1; j++) {
        const int cy = y + j * dy, cx = x + j * dx;
        const bool valid = (unsigned)(cy - dy) < (unsigned)H && (unsigned)(cy + dy) < (unsigned)H &&
                           (unsigned)(cx - dx) < (unsigned)W && (unsigned)(cx + dx) < (unsigned)W &&
                           (unsigned)cy < (unsigned)H && (unsigned)cx < (unsigned)W;
        if (!valid) continue;
        const float du = (v[j + 3].x + v[j + 1].x) - 2.f * v[j + 2].x;
        const float dv = (v[j + 3].y + v[j + 1].y) - 2.f * v[j + 2].y;
        const float coef = j == 0 ? -2.f : 1.f;
        gu += coef * charb_grad(du);
        gv += coef * charb_grad(dv);
        if (j == 0) local += charb(du) + charb(dv);
      }
    }
    if (dflow) {
      float2* o = reinterpret_cast<float2*>(dflow) + i;
      float a = gu * scale * fs, b = gv * scale * fs;
      if (acc) {
        const float2 e = *o;
        a += e.x;
        b += e.y;
      }
      *o = make_float2(a, b);
    }
  }
  const float t = block_sum(local, red);
  if (threadIdx.x == 0 && loss_acc) atomicAdd(loss_acc, t * scale);
}
__global__ __launch_bounds__(256) void second_order_kernel(const float* __restrict__ flow, float fs,
                                                           float* __restrict__ loss_acc, float* __restrict__ dflow,
                                                           int acc, float scale, int N, int H, int W) {
  second_order_body(flow, fs, loss_acc, dflow, acc, scale, N, H, W, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------ the default loss pyramid in four launches
// compute_losses with the default terms (ternary + second-order, border mask; config.ini [train]) over all pyramid
// levels: each of the four kernel types runs ONCE for all levels — a block finds its level from a prefix table and runs
// that level's body.  At 16k..48 pixels per sample the per-level launches were pure latency (~5 us each, 20 per step).
struct PyrLevelDev {
  const float* im; const float* flow; float* gray1; float* gray2w; const float* mask; float* dist; float* dflow;
  int H, W, n_mask, D;
  float fs, scale_tern, scale_smooth;
  int b_so, b_gp, b_tf, b_tb;      // first block of this level in each of the four launches
};
struct PyrArgs {
  PyrLevelDev lv[UNFLOW_MAX_PYR_LEVELS + 1];   // [n] holds the end offsets
  int n, N, shift;
  float* loss_acc;
};
#define PYR_FIND_LEVEL(FIELD)                                                        \
  int l = 0;                                                                         \
  while (l + 1 < a.n && (int)blockIdx.x >= a.lv[l + 1].FIELD) l++;                   \
  const PyrLevelDev& L = a.lv[l];                                                    \
  const unsigned vb = blockIdx.x - L.FIELD, vg = a.lv[l + 1].FIELD - L.FIELD;

__global__ __launch_bounds__(256) void pyr_second_order_kernel(const PyrArgs a) {
  PYR_FIND_LEVEL(b_so)
  second_order_body(L.flow, L.fs, a.loss_acc, L.dflow, 0, L.scale_smooth, a.N, L.H, L.W, vb, vg);
}
__global__ __launch_bounds__(256) void pyr_gray_pair_kernel(const PyrArgs a) {
  PYR_FIND_LEVEL(b_gp)
  gray_pair_body(L.im, 3, L.flow, L.fs, L.gray1, L.gray2w, a.shift, a.N, L.H, L.W, vb, vg);
}
__global__ __launch_bounds__(256) void pyr_ternary_fwd_kernel(const PyrArgs a) {
  PYR_FIND_LEVEL(b_tf)
  ternary_fwd_body(L.gray1, L.gray2w, L.mask, L.n_mask, L.dist, a.loss_acc, L.scale_tern, L.D, a.N, L.H, L.W, vb, vg);
}
__global__ __launch_bounds__(256) void pyr_ternary_bwd_kernel(const PyrArgs a) {
  PYR_FIND_LEVEL(b_tb)
  (void)vg;
  ternary_bwd_body(L.gray1, L.gray2w, L.dist, nullptr, L.D, a.N, L.H, L.W, L.im, 3, L.flow, L.fs, L.dflow, 1, a.shift, vb);
}

UNFLOW_API int unflow_sizeof_pyr_level(void) { return (int)sizeof(unflow_pyr_level); }

UNFLOW_API int unflow_loss_pyramid_default(const unflow_pyr_level* levels, int n_levels, int N, int pair_shift,
                                           float* loss_acc, int with_grad, unflow_stream_t stream) {
  if (!levels || !loss_acc) return UNFLOW_ERR_NULL;
  if (n_levels <= 0 || n_levels > UNFLOW_MAX_PYR_LEVELS || N <= 0) return UNFLOW_ERR_SHAPE;
  PyrArgs a{};
  a.n = n_levels; a.N = N; a.shift = pair_shift; a.loss_acc = loss_acc;
  int so = 0, gp = 0, tf = 0, tb = 0;
  for (int l = 0; l < n_levels; l++) {
    const unflow_pyr_level& s = levels[l];
    if (!s.im || !s.flow || !s.gray1 || !s.gray2w || !s.mask || !s.dist || (with_grad && !s.d_flow)) return UNFLOW_ERR_NULL;
    if (s.H <= 0 || s.W <= 0 || s.n_mask <= 0 || s.max_distance < 0) return UNFLOW_ERR_SHAPE;
    if (s.max_distance > CT_MAXD) return UNFLOW_ERR_UNSUPPORTED;
    PyrLevelDev& d = a.lv[l];
    d.im = s.im; d.flow = s.flow; d.gray1 = s.gray1; d.gray2w = s.gray2w; d.mask = s.mask; d.dist = s.dist;
    d.dflow = with_grad ? s.d_flow : nullptr;
    d.H = s.H; d.W = s.W; d.n_mask = s.n_mask; d.D = s.max_distance;
    d.fs = s.flow_scale; d.scale_tern = s.ternary_scale; d.scale_smooth = s.smooth_scale;
    d.b_so = so; d.b_gp = gp; d.b_tf = tf; d.b_tb = tb;
    const long npx = (long)N * s.H * s.W;
    const long ntiles = (long)cdiv(s.W, CT_W) * cdiv(s.H, CT_H) * N;
    so += stream_grid(npx);
    gp += stream_grid(npx);
    tf += (int)min(ntiles, (long)2048);
    tb += (int)ntiles;
  }
  PyrLevelDev& e = a.lv[n_levels];
  e.b_so = so; e.b_gp = gp; e.b_tf = tf; e.b_tb = tb;
  hipStream_t st = as_stream(stream);
  pyr_second_order_kernel<<<so, 256, 0, st>>>(a);
  pyr_gray_pair_kernel<<<gp, 256, 0, st>>>(a);
  pyr_ternary_fwd_kernel<<<tf, 256, 0, st>>>(a);
  if (with_grad) pyr_ternary_bwd_kernel<<<tb, 256, 0, st>>>(a);
  return launch_status();
}

UNFLOW_API int unflow_second_order_fwd_bwd(const float* flow, float flow_scale, float* loss_acc, float* d_flow,
                                           int accumulate, float weight, float normalizer, int N, int H, int W,
                                           unflow_stream_t stream) {
  if (!flow) return UNFLOW_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0) return UNFLOW_ERR_SHAPE;
  second_order_kernel<<<stream_grid((long)N * H * W), 256, 0, as_stream(stream)>>>(
      flow, flow_scale, loss_acc, d_flow, accumulate, weight / normalizer, N, H, W);
  return launch_status();
}

// ------------------------------------------------------------------ masks + fb / occ / sym terms (losses.py:25-73)
// One pass per level over the directed batch.  Sample n's partner is m = (n + shift) % N.
//   u      = flow[n]*fs                         (own flow, scaled)
//   wv     = warped[n]*fs                       (partner's flow warped by own flow: image_warp(flow_other, flow_own))
//   mask   = border mask (broadcast) or create_outgoing_mask(u) (losses.py:31-36,347-366)
//   fb_occ = |u+wv|^2 > 0.01(|u|^2+|wv|^2)+0.5  (losses.py:43-49)
//   disocc_other = forward_warp(flow_other*fs) < 0.8 at this pixel (losses.py:28-29)
//   mask_occlusion 'fb': mask *= 1-fb_occ; 'disocc': mask *= 1-disocc_other (losses.py:51-56)
//   losses: occ = charb(1-mask); sym = charb((1-mask) - disocc_other); fb = charb(u+wv, mask)   (losses.py:61-79)
// Only fb has a gradient (thresholds and masks are non-differentiable casts): d/du direct, d/dwv handed to
// image_warp's backward by the caller.
__global__ __launch_bounds__(256) void mask_terms_kernel(const float* __restrict__ flow, const float* __restrict__ warped,
                                                         const float* __restrict__ fwarp, const float* __restrict__ base,
                                                         int n_base, float fs, int occl_mode, float* __restrict__ mask_out,
                                                         float* __restrict__ loss_acc, float* __restrict__ gflow,
                                                         float* __restrict__ gwarped, int acc, float s_fb, float s_occ,
                                                         float s_sym, int shift, int N, int H, int W) {
  __shared__ float red[4];
  const unsigned npx = (unsigned)N * H * W;
  float local = 0.f;
  for (unsigned i = xcd_block() * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    const Pix pp = decode_pix(i, W, H);
    const int x = pp.x, y = pp.y, n = pp.n;
    const long pix = (long)y * W + x;
    const float2 f = reinterpret_cast<const float2*>(flow)[i];
    const float ux = f.x * fs, uy = f.y * fs;
    float mask;
    if (base) {
      mask = base[(long)(n % n_base) * H * W + pix];
    } else {
      const float px = (float)x + ux, py = (float)y + uy;
      mask = (px <= (float)(W - 1) && px >= 0.f && py <= (float)(H - 1) && py >= 0.f) ? 1.f : 0.f;
    }
    float dxx = 0.f, dyy = 0.f, fb_occ = 0.f;
    if (warped) {
      const float2 wv = reinterpret_cast<const float2*>(warped)[i];
      const float wx = wv.x * fs, wy = wv.y * fs;
      dxx = ux + wx;
      dyy = uy + wy;
      const float mag = (ux * ux + uy * uy) + (wx * wx + wy * wy);
      fb_occ = (dxx * dxx + dyy * dyy) > (0.01f * mag + 0.5f) ? 1.f : 0.f;
    }
    float dis = 0.f;
    if (fwarp) dis = fwarp[(long)((n + shift) % N) * H * W + pix] < 0.8f ? 1.f : 0.f;
    if (occl_mode == 1) mask *= (1.f - fb_occ);
    else if (occl_mode == 2) mask *= (1.f - dis);
    if (mask_out) mask_out[i] = mask;
    const float occ = 1.f - mask;
    if (s_occ != 0.f) local += s_occ * charb(occ);
    if (s_sym != 0.f) local += s_sym * charb(occ - dis);
    if (s_fb != 0.f) {
      local += s_fb * mask * (charb(dxx) + charb(dyy));
      if (gflow) {
        const float gx = s_fb * mask * charb_grad(dxx) * fs, gy = s_fb * mask * charb_grad(dyy) * fs;
        float2* o = reinterpret_cast<float2*>(gflow) + i;
        float2 e = acc ? *o : make_float2(0.f, 0.f);
        *o = make_float2(e.x + gx, e.y + gy);
        reinterpret_cast<float2*>(gwarped)[i] = make_float2(gx, gy);
      }
    }
  }
  const float t = block_sum(local, red);
  if (threadIdx.x == 0 && loss_acc) atomicAdd(loss_acc, t);
}

UNFLOW_API int unflow_mask_terms(const float* flow, const float* warped_other, const float* fwarp, const float* base_mask,
                                 int n_base, float flow_scale, int occlusion_mode, float* mask_out, float* loss_acc,
                                 float* d_flow, float* d_warped, int accumulate, float fb_weight, float occ_weight,
                                 float sym_weight, int batch_per_direction, int pair_shift, int N, int H, int W,
                                 unflow_stream_t stream) {
  if (!flow) return UNFLOW_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || batch_per_direction <= 0) return UNFLOW_ERR_SHAPE;
  if (fb_weight != 0.f && !warped_other) return UNFLOW_ERR_NULL;
  if ((occlusion_mode == 1 && !warped_other) || (occlusion_mode == 2 && !fwarp) || (sym_weight != 0.f && !fwarp)) return UNFLOW_ERR_NULL;
  if (d_flow && fb_weight != 0.f && !d_warped) return UNFLOW_ERR_NULL;
  const float n1 = (float)batch_per_direction * H * W;
  mask_terms_kernel<<<stream_grid((long)N * H * W), 256, 0, as_stream(stream)>>>(
      flow, warped_other, fwarp, base_mask, n_base > 0 ? n_base : 1, flow_scale, occlusion_mode, mask_out, loss_acc, d_flow,
      d_warped, accumulate, fb_weight / (n1 * 2.f), occ_weight / n1, sym_weight / n1, pair_shift, N, H, W);
  return launch_status();
}

// ------------------------------------------------------------------ photometric loss (losses.py:198-199), fused with the warp
// charbonnier(im1 - image_warp(im2, flow), mask, beta = 255); normaliser N_dir*H*W*3.  fwd + flow gradient in one pass.
__global__ __launch_bounds__(256) void photometric_kernel(const float* __restrict__ im, int ld, const float* __restrict__ flow,
                                                          float fs, const float* __restrict__ mask, int n_mask,
                                                          float* __restrict__ loss_acc, float* __restrict__ dflow, int acc,
                                                          float scale, int shift, int N, int H, int W) {
  __shared__ float red[4];
  const unsigned npx = (unsigned)N * H * W;
  float local = 0.f;
  for (unsigned i = xcd_block() * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    const Pix pp = decode_pix(i, W, H);
    const int px = pp.x, py = pp.y, n = pp.n;
    const long sb = (long)((n + shift) % N) * H * W;
    const float2 f = reinterpret_cast<const float2*>(flow)[i];
    const Taps t = iw_taps(px, py, f.x * fs, f.y * fs, H, W);
    const float *pa = im + (sb + t.ia) * ld, *pb = im + (sb + t.ib) * ld, *pc = im + (sb + t.ic) * ld,
                *pd = im + (sb + t.id) * ld;
    const float* p1 = im + i * ld;
    const float m = mask[(long)(n % n_mask) * H * W + (long)py * W + px];
    float ga = 0.f, gb = 0.f, gc = 0.f, gd = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float w = ((t.wa * pa[k] + t.wb * pb[k]) + t.wc * pc[k]) + t.wd * pd[k];
      const float d = (p1[k] - w) * 255.0f;
      const float l2 = __builtin_amdgcn_logf(d * d + CHARB_EPS * CHARB_EPS);
      local += m * __builtin_amdgcn_exp2f(CHARB_ALPHA * l2);
      // d/dw of ((x*beta)^2+eps^2)^alpha with x = im1 - w
      const float g = -m * CHARB_ALPHA * __builtin_amdgcn_exp2f((CHARB_ALPHA - 1.f) * l2) * 2.f * d * 255.0f;
      ga += g * pa[k]; gb += g * pb[k]; gc += g * pc[k]; gd += g * pd[k];
    }
    if (dflow) {
      float du = ((gc - ga) * (1.f - t.yw) + (gd - gb) * t.yw) * fs * scale;
      float dv = ((gb - ga) * (1.f - t.xw) + (gd - gc) * t.xw) * fs * scale;
      float2* o = reinterpret_cast<float2*>(dflow) + i;
      if (acc) { const float2 e = *o; du += e.x; dv += e.y; }
      *o = make_float2(du, dv);
    }
  }
  const float tt = block_sum(local, red);
  if (threadIdx.x == 0 && loss_acc) atomicAdd(loss_acc, tt * scale);
}

UNFLOW_API int unflow_photometric_fwd_bwd(const float* im, int ld_im, const float* flow, float flow_scale,
                                          const float* mask, int n_mask, float* loss_acc, float* d_flow, int accumulate,
                                          float weight, float normalizer, int pair_shift, int N, int H, int W,
                                          unflow_stream_t stream) {
  if (!im || !flow || !mask) return UNFLOW_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || ld_im < 3 || n_mask <= 0) return UNFLOW_ERR_SHAPE;
  photometric_kernel<<<stream_grid((long)N * H * W), 256, 0, as_stream(stream)>>>(
      im, ld_im, flow, flow_scale, mask, n_mask, loss_acc, d_flow, accumulate, weight / normalizer, pair_shift, N, H, W);
  return launch_status();
}

// ------------------------------------------------------------------ generic Charbonnier (losses.py:298-322) and length_sq (:12-13)
// charbonnier_loss(x, mask, truncate, alpha, beta, epsilon) = sum(min(mask * ((x*beta)^2 + eps^2)^alpha, truncate)) / (npix * C):
// the stand-alone form of the penalty every fused term above applies (there with the reference's defaults folded in).
// mask: [npix, mask_channels] with mask_channels 1 or C (or NULL); truncate < 0: none.  loss_acc[0] += weight * that.
__global__ __launch_bounds__(256) void charbonnier_kernel(const float* __restrict__ x, int C, const float* __restrict__ mask, int mc,
                                                          float truncate, float alpha, float beta, float eps2, float scale,
                                                          float* __restrict__ loss_acc, long n) {
  __shared__ float red[4];
  float local = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i] * beta;
    float e = fast_pow(v * v + eps2, alpha);
    if (mask) e *= mc == 1 ? mask[i / C] : mask[i];
    if (truncate >= 0.f) e = fminf(e, truncate);
    local += e;
  }
  const float t = block_sum(local, red);
  if (threadIdx.x == 0) atomicAdd(loss_acc, t * scale);
}

UNFLOW_API int unflow_charbonnier_loss(const float* x, const float* mask, int mask_channels, float truncate, float alpha, float beta,
                                       float epsilon, float* loss_acc, float weight, long npix, int C, unflow_stream_t stream) {
  if (!x || !loss_acc) return UNFLOW_ERR_NULL;
  if (npix <= 0 || C <= 0 || (mask && mask_channels != 1 && mask_channels != C)) return UNFLOW_ERR_SHAPE;
  const long n = npix * C;
  charbonnier_kernel<<<stream_grid(n), 256, 0, as_stream(stream)>>>(x, C, mask, mask_channels, truncate, alpha, beta,
                                                                    epsilon * epsilon, weight / (float)n, loss_acc, n);
  return launch_status();
}

__global__ void length_sq_kernel(const float* __restrict__ x, int C, float* __restrict__ out, long npix) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < C; c++) { const float v = x[i * C + c]; s += v * v; }
    out[i] = s;
  }
}

UNFLOW_API int unflow_length_sq(const float* x, float* out, long npix, int C, unflow_stream_t stream) {
  if (!x || !out) return UNFLOW_ERR_NULL;
  if (npix <= 0 || C <= 0) return UNFLOW_ERR_SHAPE;
  length_sq_kernel<<<stream_grid(npix), 256, 0, as_stream(stream)>>>(x, C, out, npix);
  return launch_status();
}

// ------------------------------------------------------------------ first-order smoothness (losses.py:206-255)
// delta_x(p) = f(p) - f(p+(0,1)) (masked on the last column), delta_y(p) = f(p) - f(p+(1,0)) (last row);
// Charbonnier on each, normaliser N_dir*H*W*2 per flow channel.
__global__ __launch_bounds__(256) void smooth_1st_kernel(const float* __restrict__ flow, float fs, float* __restrict__ loss_acc,
                                                         float* __restrict__ dflow, int acc, float scale, int N, int H, int W) {
  __shared__ float red[4];
  const unsigned npx = (unsigned)N * H * W;
  float local = 0.f;
  for (unsigned i = xcd_block() * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    const Pix pp = decode_pix(i, W, H);
    const int x = pp.x, y = pp.y;
    const float2* f = reinterpret_cast<const float2*>(flow) + (i - ((long)y * W + x));
    const float2 c = f[(long)y * W + x];
    float gu = 0.f, gv = 0.f;
    if (x + 1 < W) {  // own delta_x
      const float2 r = f[(long)y * W + x + 1];
      const float du = (c.x - r.x) * fs, dv = (c.y - r.y) * fs;
      local += charb(du) + charb(dv);
      gu += charb_grad(du); gv += charb_grad(dv);
    }
    if (x >= 1) {  // left neighbour's delta_x contains -f(p)
      const float2 l = f[(long)y * W + x - 1];
      gu -= charb_grad((l.x - c.x) * fs); gv -= charb_grad((l.y - c.y) * fs);
    }
    if (y + 1 < H) {
      const float2 d = f[(long)(y + 1) * W + x];
      const float du = (c.x - d.x) * fs, dv = (c.y - d.y) * fs;
      local += charb(du) + charb(dv);
      gu += charb_grad(du); gv += charb_grad(dv);
    }
    if (y >= 1) {
      const float2 u = f[(long)(y - 1) * W + x];
      gu -= charb_grad((u.x - c.x) * fs); gv -= charb_grad((u.y - c.y) * fs);
    }
    if (dflow) {
      float2* o = reinterpret_cast<float2*>(dflow) + i;
      float a = gu * scale * fs, b = gv * scale * fs;
      if (acc) { const float2 e = *o; a += e.x; b += e.y; }
      *o = make_float2(a, b);
    }
  }
  const float t = block_sum(local, red);
  if (threadIdx.x == 0 && loss_acc) atomicAdd(loss_acc, t * scale);
}

UNFLOW_API int unflow_smooth_1st_fwd_bwd(const float* flow, float flow_scale, float* loss_acc, float* d_flow,
                                         int accumulate, float weight, float normalizer, int N, int H, int W,
                                         unflow_stream_t stream) {
  if (!flow) return UNFLOW_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0) return UNFLOW_ERR_SHAPE;
  smooth_1st_kernel<<<stream_grid((long)N * H * W), 256, 0, as_stream(stream)>>>(flow, flow_scale, loss_acc, d_flow,
                                                                                 accumulate, weight / normalizer, N, H, W);
  return launch_status();
}

// ------------------------------------------------------------------ gradient (Sobel) constancy loss (losses.py:225-247)
// diff[c][x|y] = sobel(im1)[c] - sobel(im2_warped)[c] (SAME zero padding), Charbonnier, mask * gradient_mask
// (x-gradient channels zero on the first/last column, y-gradient channels on the first/last row); normaliser N_dir*H*W*6.
__device__ __forceinline__ float img_at(const float* __restrict__ im, int ld, int H, int W, int y, int x, int c) {
  return ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? im[((long)y * W + x) * ld + c] : 0.f;
}

__global__ __launch_bounds__(256) void gradient_loss_fwd_kernel(const float* __restrict__ im1, int ld1,
                                                                const float* __restrict__ im2w, const float* __restrict__ mask,
                                                                int n_mask, float* __restrict__ gdiff,
                                                                float* __restrict__ loss_acc, float scale, int N, int H, int W) {
  __shared__ float red[4];
  const unsigned npx = (unsigned)N * H * W;
  float local = 0.f;
  for (unsigned i = xcd_block() * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    const Pix pp = decode_pix(i, W, H);
    const int x = pp.x, y = pp.y;
    const long n = pp.n;
    const float* a = im1 + n * H * W * ld1;
    const float* b = im2w + n * H * W * 3;
    const float m = mask[(n % n_mask) * (long)H * W + (long)y * W + x];
    const float mx = (x >= 1 && x < W - 1) ? m : 0.f, my = (y >= 1 && y < H - 1) ? m : 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      float gx = 0.f, gy = 0.f;
#pragma unroll
      for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
          const float v = img_at(a, ld1, H, W, y + dy, x + dx, c) - img_at(b, 3, H, W, y + dy, x + dx, c);
          gx += (float)(dx * (dy == 0 ? 2 : 1)) * v;   // [[-1,0,1],[-2,0,2],[-1,0,1]]
          gy += (float)(dy * (dx == 0 ? 2 : 1)) * v;   // its transpose
        }
      local += mx * charb(gx) + my * charb(gy);
      gdiff[i * 6 + 2 * c] = scale * mx * charb_grad(gx);
      gdiff[i * 6 + 2 * c + 1] = scale * my * charb_grad(gy);
    }
  }
  const float t = block_sum(local, red);
  if (threadIdx.x == 0 && loss_acc) atomicAdd(loss_acc, t * scale);
}

// d/d(im2_warped)[q,c] = - sum_t coef(t) * gdiff[q - t]  (gather form)
__global__ void gradient_loss_bwd_kernel(const float* __restrict__ gdiff, float* __restrict__ d_im2w, int N, int H, int W) {
  const long npx = (long)N * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npx; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long n = i / ((long)W * H);
    const float* g = gdiff + n * H * W * 6;
    float out[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
      for (int dx = -1; dx <= 1; dx++) {
        const int py = y - dy, px = x - dx;  // pixel p whose stencil tap (dy,dx) lands on q
        if ((unsigned)py >= (unsigned)H || (unsigned)px >= (unsigned)W) continue;
        const float cx = (float)(dx * (dy == 0 ? 2 : 1)), cy = (float)(dy * (dx == 0 ? 2 : 1));
        const float* gp = g + ((long)py * W + px) * 6;
#pragma unroll
        for (int c = 0; c < 3; c++) out[c] -= cx * gp[2 * c] + cy * gp[2 * c + 1];
      }
    d_im2w[i * 3] = out[0];
    d_im2w[i * 3 + 1] = out[1];
    d_im2w[i * 3 + 2] = out[2];
  }
}

UNFLOW_API int unflow_gradient_loss_fwd(const float* im1, int ld_im1, const float* im2_warped, const float* mask, int n_mask,
                                        float* gdiff, float* loss_acc, float weight, float normalizer, int N, int H, int W,
                                        unflow_stream_t stream) {
  if (!im1 || !im2_warped || !mask || !gdiff) return UNFLOW_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || n_mask <= 0) return UNFLOW_ERR_SHAPE;
  gradient_loss_fwd_kernel<<<stream_grid((long)N * H * W), 256, 0, as_stream(stream)>>>(
      im1, ld_im1, im2_warped, mask, n_mask, gdiff, loss_acc, weight / normalizer, N, H, W);
  return launch_status();
}

UNFLOW_API int unflow_gradient_loss_bwd(const float* gdiff, float* d_im2_warped, int N, int H, int W, unflow_stream_t stream) {
  if (!gdiff || !d_im2_warped) return UNFLOW_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0) return UNFLOW_ERR_SHAPE;
  gradient_loss_bwd_kernel<<<stream_grid((long)N * H * W), 256, 0, as_stream(stream)>>>(gdiff, d_im2_warped, N, H, W);
  return launch_status();
}
