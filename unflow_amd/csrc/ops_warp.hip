// Warps and downsample for gfx950 (HBM-bound gathers; one thread per pixel or per output
// element, coalesced along the channels-last axis).
//
//   backward_warp  <- ops/backward_warp_op.cu.cc:14-138   (zero padding, floorf(float(x)+u))
//   image_warp     <- src/e2eflow/core/image_warp.py:4-76 (clamp, x + int(floor(u)), dual gradient)
//   forward_warp   <- ops/forward_warp_op.cu.cc:16-125    (gaussian splat, +-4 px)
//   downsample     <- ops/downsample_op.cu.cc:15-49       (box mean)
#include "common.h"

// ------------------------------------------------------------------ backward_warp
struct BwTaps {
  int x0, y0;
  float wl, wr, wt, wb;
};

__device__ __forceinline__ BwTaps bw_sample(int px, int py, float u, float v) {
  BwTaps t;
  const float sx = (float)px + u, sy = (float)py + v;  // fp32 sum, then floor (ref :27-31)
  t.x0 = (int)floorf(sx);
  t.y0 = (int)floorf(sy);
  t.wr = sx - (float)t.x0;
  t.wl = (float)(t.x0 + 1) - sx;
  t.wb = sy - (float)t.y0;
  t.wt = (float)(t.y0 + 1) - sy;
  return t;
}

// Streamed-once operands (the flow field in, the warped image / flow gradient out) bypass the cache hierarchy's retention
// (nt): the gathers of the warps are the only accesses with reuse, and they keep the L2 for themselves.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ float2 ld_nt2(const float* p) {
  const f32x2 v = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p));
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ void st_nt2(float* p, float a, float b) {
  f32x2 v; v.x = a; v.y = b;
  __builtin_nontemporal_store(v, reinterpret_cast<f32x2*>(p));
}
template <int CT>
__device__ __forceinline__ void st_nt_px(float* p, const float (&s)[CT ? CT : 1]) {
  if constexpr (CT == 3) {
    f32x3 v; v.x = s[0]; v.y = s[1]; v.z = s[2];
    __builtin_nontemporal_store(v, reinterpret_cast<f32x3*>(p));
  } else {
#pragma unroll
    for (int c = 0; c < CT; c++) __builtin_nontemporal_store(s[c], p + c);
  }
}

// One pixel's four taps: addresses, weights, in-range masks.
struct BwPix {
  const float *p_tl, *p_tr, *p_bl, *p_br;
  float w_tl, w_tr, w_bl, w_br;
  bool m_tl, m_tr, m_bl, m_br;
};
__device__ __forceinline__ BwPix bw_pix(const float* img, const TilePix& pp, float2 f, int H, int W, int CC) {
  const BwTaps t = bw_sample(pp.x, pp.y, f.x, f.y);
  const float* base = img + (long)pp.n * H * W * CC;
  const bool xl = t.x0 >= 0 && t.x0 < W, xr = t.x0 >= -1 && t.x0 < W - 1;
  const bool yt = t.y0 >= 0 && t.y0 < H, yb = t.y0 >= -1 && t.y0 < H - 1;
  // branch-free taps: addresses clamped into the image, out-of-range VALUES replaced by 0 (the reference
  // skips them, :44-67; adding w*0 is the same number and keeps the four-term order)
  const int xa = min(max(t.x0, 0), W - 1), xb = min(max(t.x0, -1), W - 2) + 1;
  const int ya = min(max(t.y0, 0), H - 1), yc = min(max(t.y0, -1), H - 2) + 1;
  BwPix q;
  q.p_tl = base + (size_t)(ya * W + xa) * CC;
  q.p_tr = base + (size_t)(ya * W + xb) * CC;
  q.p_bl = base + (size_t)(yc * W + xa) * CC;
  q.p_br = base + (size_t)(yc * W + xb) * CC;
  q.m_tl = xl && yt; q.m_tr = xr && yt; q.m_bl = xl && yb; q.m_br = xr && yb;
  q.w_tl = t.wl * t.wt; q.w_tr = t.wr * t.wt; q.w_bl = t.wl * t.wb; q.w_br = t.wr * t.wb;
  return q;
}
__device__ __forceinline__ float bw_mix(const BwPix& q, float v_tl, float v_tr, float v_bl, float v_br) {
  float s = 0.f;
  s += q.m_tl ? q.w_tl * v_tl : 0.f;
  s += q.m_tr ? q.w_tr * v_tr : 0.f;
  s += q.m_bl ? q.w_bl * v_bl : 0.f;
  s += q.m_br ? q.w_br * v_br : 0.f;
  return s;
}

template <int CT>
__global__ void backward_warp_fwd_kernel(const float* __restrict__ img, const float* __restrict__ flow,
                                         float* __restrict__ out, int B, int H, int W, int C) {
  const unsigned T = tile_count(W, H, B);
  for (unsigned tile = tile_first(T), tile_end = tile_last(T); tile < tile_end; tile++) {
    const TilePix tp = tile_pix(tile, W, H);
    if (!tp.ok) continue;
    const unsigned i = tp.i;
    const int CC = CT ? CT : C;
    const BwPix a = bw_pix(img, tp, ld_nt2(flow + 2 * (size_t)i), H, W, CC);
    if constexpr (CT != 0) {
      float r[CT ? CT : 1];
#pragma unroll
      for (int c = 0; c < CT; c++) r[c] = bw_mix(a, a.p_tl[c], a.p_tr[c], a.p_bl[c], a.p_br[c]);
      st_nt_px<CT>(out + (size_t)i * CT, r);
    } else {
      for (int c = 0; c < C; c++) out[(size_t)i * C + c] = bw_mix(a, a.p_tl[c], a.p_tr[c], a.p_bl[c], a.p_br[c]);
    }
  }
}

template <int CT>
__global__ void backward_warp_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ img,
                                         const float* __restrict__ flow, float* __restrict__ dflow, int B,
                                         int H, int W, int C) {
  const unsigned T = tile_count(W, H, B);
  for (unsigned tile = tile_first(T), tile_end = tile_last(T); tile < tile_end; tile++) {
    const TilePix pp = tile_pix(tile, W, H);
    if (!pp.ok) continue;
    const unsigned i = pp.i;
    const int px = pp.x, py = pp.y;
    const long b = pp.n;
    const float2 f = ld_nt2(flow + 2 * (size_t)i);
    const BwTaps t = bw_sample(px, py, f.x, f.y);
    const float* base = img + b * H * W * C;
    const bool xl = t.x0 >= 0 && t.x0 < W, xr = t.x0 >= -1 && t.x0 < W - 1;
    const bool yt = t.y0 >= 0 && t.y0 < H, yb = t.y0 >= -1 && t.y0 < H - 1;
    const int xa = min(max(t.x0, 0), W - 1), xb = min(max(t.x0, -1), W - 2) + 1;
    const int ya = min(max(t.y0, 0), H - 1), yc = min(max(t.y0, -1), H - 2) + 1;
    const int CC = CT ? CT : C;
    const float* p_tl = base + (size_t)(ya * W + xa) * CC;
    const float* p_tr = base + (size_t)(ya * W + xb) * CC;
    const float* p_bl = base + (size_t)(yc * W + xa) * CC;
    const float* p_br = base + (size_t)(yc * W + xb) * CC;
    const bool m_tl = xl && yt, m_tr = xr && yt, m_bl = xl && yb, m_br = xr && yb;
    float du = 0.f, dv = 0.f;
#pragma unroll 4
    for (int c = 0; c < CC; c++) {
      const float din = __builtin_nontemporal_load(dout + (size_t)i * CC + c);
      const float v_tl = p_tl[c], v_tr = p_tr[c], v_bl = p_bl[c], v_br = p_br[c];
      const float q_tl = m_tl ? v_tl * din : 0.f, q_tr = m_tr ? v_tr * din : 0.f;
      const float q_bl = m_bl ? v_bl * din : 0.f, q_br = m_br ? v_br * din : 0.f;
      du -= t.wt * q_tl; dv -= t.wl * q_tl;
      du += t.wt * q_tr; dv -= t.wr * q_tr;
      du -= t.wb * q_bl; dv += t.wl * q_bl;
      du += t.wb * q_br; dv += t.wr * q_br;
    }
    st_nt2(dflow + 2 * (size_t)i, du, dv);
  }
}

__global__ void backward_warp_indices_kernel(const float* __restrict__ flow, int* __restrict__ xy0, int B, int H,
                                             int W) {
  const long npx = (long)B * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npx; i += (long)gridDim.x * blockDim.x) {
    const float2 f = reinterpret_cast<const float2*>(flow)[i];
    const BwTaps t = bw_sample((int)(i % W), (int)((i / W) % H), f.x, f.y);
    reinterpret_cast<int2*>(xy0)[i] = make_int2(t.x0, t.y0);
  }
}

UNFLOW_API int unflow_backward_warp_fwd(const float* images, const float* flows, float* out, int B, int H, int W,
                                        int C, unflow_stream_t stream) {
  if (!images || !flows || !out) return UNFLOW_ERR_NULL;
  if (B < 0 || H < 0 || W < 0 || C < 0) return UNFLOW_ERR_SHAPE;
  const long npx = (long)B * H * W;
  if (npx == 0 || C == 0) return UNFLOW_OK;  // ref: `if (total_count == 0) return;`
  switch (C) {
    case 1: backward_warp_fwd_kernel<1><<<stream_grid(npx), 256, 0, as_stream(stream)>>>(images, flows, out, B, H, W, C); break;
    case 2: backward_warp_fwd_kernel<2><<<stream_grid(npx), 256, 0, as_stream(stream)>>>(images, flows, out, B, H, W, C); break;
    case 3: backward_warp_fwd_kernel<3><<<stream_grid(npx), 256, 0, as_stream(stream)>>>(images, flows, out, B, H, W, C); break;
    default: backward_warp_fwd_kernel<0><<<stream_grid(npx), 256, 0, as_stream(stream)>>>(images, flows, out, B, H, W, C); break;
  }
  return launch_status();
}

UNFLOW_API int unflow_backward_warp_bwd(const float* dout, const float* images, const float* flows, float* dflows,
                                        int B, int H, int W, int C, unflow_stream_t stream) {
  if (!dout || !images || !flows || !dflows) return UNFLOW_ERR_NULL;
  if (B < 0 || H < 0 || W < 0 || C < 0) return UNFLOW_ERR_SHAPE;
  const long npx = (long)B * H * W;
  if (npx == 0) return UNFLOW_OK;
  switch (C) {
    case 1: backward_warp_bwd_kernel<1><<<stream_grid(npx), 256, 0, as_stream(stream)>>>(dout, images, flows, dflows, B, H, W, C); break;
    case 2: backward_warp_bwd_kernel<2><<<stream_grid(npx), 256, 0, as_stream(stream)>>>(dout, images, flows, dflows, B, H, W, C); break;
    case 3: backward_warp_bwd_kernel<3><<<stream_grid(npx), 256, 0, as_stream(stream)>>>(dout, images, flows, dflows, B, H, W, C); break;
    default: backward_warp_bwd_kernel<0><<<stream_grid(npx), 256, 0, as_stream(stream)>>>(dout, images, flows, dflows, B, H, W, C); break;
  }
  return launch_status();
}

UNFLOW_API int unflow_backward_warp_indices(const float* flows, int* xy0, int B, int H, int W,
                                            unflow_stream_t stream) {
  if (!flows || !xy0) return UNFLOW_ERR_NULL;
  const long npx = (long)B * H * W;
  if (npx == 0) return UNFLOW_OK;
  backward_warp_indices_kernel<<<stream_grid(npx), 256, 0, as_stream(stream)>>>(flows, xy0, B, H, W);
  return launch_status();
}

// ------------------------------------------------------------------ image_warp
struct IwTaps {
  long ia, ib, ic, id;  // pixel indices within the batch (sample-relative, in pixels)
  float xw, yw, wa, wb, wc, wd;
};

__device__ __forceinline__ IwTaps iw_sample(int px, int py, float u, float v, int H, int W) {
  IwTaps t;
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

template <int CT>
__global__ void image_warp_fwd_kernel(const float* __restrict__ im, int ld_im, const float* __restrict__ flow,
                                      float fscale, float* __restrict__ out, int* __restrict__ idx4, int shift,
                                      int B, int H, int W, int C) {
  const unsigned T = tile_count(W, H, B);
  for (unsigned tile = tile_first(T), tile_end = tile_last(T); tile < tile_end; tile++) {
    const TilePix pp = tile_pix(tile, W, H);
    if (!pp.ok) continue;
    const unsigned i = pp.i;
    const int px = pp.x, py = pp.y, b = pp.n;
    const int bs = (b + shift) % B;
    const float2 f = ld_nt2(flow + 2 * (size_t)i);
    const IwTaps t = iw_sample(px, py, f.x * fscale, f.y * fscale, H, W);
    const long sbase = (long)bs * H * W;
    if (idx4) {
      reinterpret_cast<int4*>(idx4)[i] =
          make_int4((int)(sbase + t.ia), (int)(sbase + t.ib), (int)(sbase + t.ic), (int)(sbase + t.id));
    }
    const float* pa = im + (sbase + t.ia) * ld_im;
    const float* pb = im + (sbase + t.ib) * ld_im;
    const float* pc = im + (sbase + t.ic) * ld_im;
    const float* pd = im + (sbase + t.id) * ld_im;
    if constexpr (CT != 0) {
      float r[CT ? CT : 1];
#pragma unroll
      for (int c = 0; c < CT; c++) r[c] = ((t.wa * pa[c] + t.wb * pb[c]) + t.wc * pc[c]) + t.wd * pd[c];
      st_nt_px<CT>(out + (size_t)i * CT, r);
    } else {
#pragma unroll 4
      for (int c = 0; c < C; c++)
        out[(size_t)i * C + c] = ((t.wa * pa[c] + t.wb * pb[c]) + t.wc * pc[c]) + t.wd * pd[c];
    }
  }
}

template <int CT>
__global__ void image_warp_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ im, int ld_im,
                                      const float* __restrict__ flow, float fscale, float* __restrict__ d_im,
                                      float* __restrict__ d_flow, int acc_flow, int shift, int B, int H, int W,
                                      int C) {
  const unsigned T = tile_count(W, H, B);
  for (unsigned tile = tile_first(T), tile_end = tile_last(T); tile < tile_end; tile++) {
    const TilePix pp = tile_pix(tile, W, H);
    if (!pp.ok) continue;
    const unsigned i = pp.i;
    const int px = pp.x, py = pp.y, b = pp.n;
    const int bs = (b + shift) % B;
    const float2 f = reinterpret_cast<const float2*>(flow)[i];
    const IwTaps t = iw_sample(px, py, f.x * fscale, f.y * fscale, H, W);
    const long sbase = (long)bs * H * W;
    const long oa = (sbase + t.ia) * ld_im, ob = (sbase + t.ib) * ld_im, oc = (sbase + t.ic) * ld_im,
               od = (sbase + t.id) * ld_im;
    float ga = 0.f, gb = 0.f, gc = 0.f, gd = 0.f;
    const int CC = CT ? CT : C;
    if (CT) {  // all loads first (vectorised), then the scatter
      float g[CT ? CT : 1];
#pragma unroll
      for (int c = 0; c < CT; c++) g[c] = dout[(size_t)i * CT + c];
#pragma unroll
      for (int c = 0; c < CT; c++) {
        ga += g[c] * im[oa + c];
        gb += g[c] * im[ob + c];
        gc += g[c] * im[oc + c];
        gd += g[c] * im[od + c];
      }
      if (d_im) {  // gather-gradient = scatter-add (clamped duplicates accumulate)
#pragma unroll
        for (int c = 0; c < CT; c++) {
          atomicAdd(d_im + oa + c, t.wa * g[c]);
          atomicAdd(d_im + ob + c, t.wb * g[c]);
          atomicAdd(d_im + oc + c, t.wc * g[c]);
          atomicAdd(d_im + od + c, t.wd * g[c]);
        }
      }
    } else {
      for (int c = 0; c < CC; c++) {
        const float g = dout[(size_t)i * C + c];
        ga += g * im[oa + c];
        gb += g * im[ob + c];
        gc += g * im[oc + c];
        gd += g * im[od + c];
        if (d_im) {
          atomicAdd(d_im + oa + c, t.wa * g);
          atomicAdd(d_im + ob + c, t.wb * g);
          atomicAdd(d_im + oc + c, t.wc * g);
          atomicAdd(d_im + od + c, t.wd * g);
        }
      }
    }
    float du = ((gc - ga) * (1.f - t.yw) + (gd - gb) * t.yw) * fscale;
    float dv = ((gb - ga) * (1.f - t.xw) + (gd - gc) * t.xw) * fscale;
    float2* o = reinterpret_cast<float2*>(d_flow) + i;
    if (acc_flow) {
      const float2 e = *o;
      du += e.x;
      dv += e.y;
    }
    *o = make_float2(du, dv);
  }
}

UNFLOW_API int unflow_image_warp_fwd(const float* im, int ld_im, const float* flow, float flow_scale, float* out,
                                     int* idx4, int pair_shift, int B, int H, int W, int C,
                                     unflow_stream_t stream) {
  if (!im || !flow || !out) return UNFLOW_ERR_NULL;
  if (B < 0 || H < 0 || W < 0 || C < 0 || ld_im < C) return UNFLOW_ERR_SHAPE;
  const long npx = (long)B * H * W;
  if (npx == 0 || C == 0) return UNFLOW_OK;
  switch (C) {
    case 1: image_warp_fwd_kernel<1><<<stream_grid(npx), 256, 0, as_stream(stream)>>>(im, ld_im, flow, flow_scale, out, idx4, pair_shift, B, H, W, C); break;
    case 2: image_warp_fwd_kernel<2><<<stream_grid(npx), 256, 0, as_stream(stream)>>>(im, ld_im, flow, flow_scale, out, idx4, pair_shift, B, H, W, C); break;
    case 3: image_warp_fwd_kernel<3><<<stream_grid(npx), 256, 0, as_stream(stream)>>>(im, ld_im, flow, flow_scale, out, idx4, pair_shift, B, H, W, C); break;
    default: image_warp_fwd_kernel<0><<<stream_grid(npx), 256, 0, as_stream(stream)>>>(im, ld_im, flow, flow_scale, out, idx4, pair_shift, B, H, W, C); break;
  }
  return launch_status();
}

UNFLOW_API int unflow_image_warp_bwd(const float* dout, const float* im, int ld_im, const float* flow,
                                     float flow_scale, float* d_im, float* d_flow, int accumulate_d_flow,
                                     int pair_shift, int B, int H, int W, int C, unflow_stream_t stream) {
  if (!dout || !im || !flow || !d_flow) return UNFLOW_ERR_NULL;
  if (B < 0 || H < 0 || W < 0 || C < 0 || ld_im < C) return UNFLOW_ERR_SHAPE;
  const long npx = (long)B * H * W;
  if (npx == 0) return UNFLOW_OK;
  switch (C) {
    case 1: image_warp_bwd_kernel<1><<<stream_grid(npx), 256, 0, as_stream(stream)>>>(dout, im, ld_im, flow, flow_scale, d_im, d_flow, accumulate_d_flow, pair_shift, B, H, W, C); break;
    case 2: image_warp_bwd_kernel<2><<<stream_grid(npx), 256, 0, as_stream(stream)>>>(dout, im, ld_im, flow, flow_scale, d_im, d_flow, accumulate_d_flow, pair_shift, B, H, W, C); break;
    case 3: image_warp_bwd_kernel<3><<<stream_grid(npx), 256, 0, as_stream(stream)>>>(dout, im, ld_im, flow, flow_scale, d_im, d_flow, accumulate_d_flow, pair_shift, B, H, W, C); break;
    default: image_warp_bwd_kernel<0><<<stream_grid(npx), 256, 0, as_stream(stream)>>>(dout, im, ld_im, flow, flow_scale, d_im, d_flow, accumulate_d_flow, pair_shift, B, H, W, C); break;
  }
  return launch_status();
}

// ------------------------------------------------------------------ forward_warp
struct FwFoot {
  bool ok;
  int x_lo, x_hi, y_lo, y_hi;
  float tx, ty;
};

__device__ __forceinline__ FwFoot fw_footprint(int px, int py, float u, float v, int W, int H) {
  FwFoot f;
  const float k = 4.f;  // ceilf(2 + 2)
  f.tx = (float)px + u;
  f.ty = (float)py + v;
  f.ok = floorf(f.tx - k) < (float)W && floorf(f.tx + k) >= 0.f && floorf(f.ty - k) < (float)H &&
         floorf(f.ty + k) >= 0.f;
  f.x_lo = f.tx - k > 0.f ? (int)floorf(f.tx - k) : 0;
  f.y_lo = f.ty - k > 0.f ? (int)floorf(f.ty - k) : 0;
  f.x_hi = f.tx + k < (float)W ? (int)floorf(f.tx + k) : W - 1;
  f.y_hi = f.ty + k < (float)H ? (int)floorf(f.ty + k) : H - 1;
  return f;
}

__device__ __forceinline__ float splat_weight(float dx, float dy) {
  return expf(-(dx * dx + dy * dy) / 2.0f);  // gauss_divisor = 2*std^2 = 2 (ref :53-56)
}

// ---- tiled scatter --------------------------------------------------------------------------------------------------------
// The reference kernel (forward_warp_op.cu.cc:16-65) is one thread per source pixel and <= 81 float atomicAdds into the output.
// Its cost is the atomics: 81 device-scope read-modify-writes per pixel.  Here a workgroup owns a 64 x 16 tile of SOURCE pixels
// and a FW_WX x FW_WY window of TARGET pixels in LDS, centred on the mean target position of the tile's sources (a flow
// field moves a tile as a whole, so the window follows it; a few outliers do not drag it away); the <= 81 taps of a source are integer LDS atomics, and only the
// touched window entries go out to memory, once per tile: ~2.5 global atomics per source instead of 81 (memory-side atomics
// run at ~46 G/s chip-wide whether 4 or 8 bytes wide; a 64 x 64 tile needs only 1.6 per source but was slower, 1069 vs 685 us
// at 16 x 768 x 1024: two workgroups per CU instead of three — the LDS phase is latency-bound).
// Taps outside the window (a field that tears a tile apart) take the global path directly — always correct, only slower.
//   * Weights are separable: exp(-(dx^2 + dy^2) / 2) = exp(-dx^2 / 2) * exp(-dy^2 / 2): 18 expf per source instead of 81
//     (relative deviation from the reference's single expf <= ~4e-7, the parity tests allow 1e-5 absolute on sums of ~6.3).
//   * Sums are 2^-31 fixed point in 64-bit integers, in LDS and in memory: integer addition commutes, so the result does not
//     depend on the arrival order of waves, workgroups or XCDs — bit-reproducible — and cannot overflow (2^33 full-weight taps
//     per pixel).  The footprints are exactly the reference's (fw_footprint, shared with unflow_forward_warp_ranges).
// deterministic = 0 keeps the reference's contract (float atomicAdd into `out`, order-dependent last bits) but takes the same
// LDS path: the window entries are converted and added as floats, one atomic per touched entry.
constexpr int FW_TX = 64, FW_TY = 16;                 // source tile: 1024 pixels, 4 per thread
constexpr int FW_WX = 104, FW_WY = 56;                // target window: 46.6 KB of 64-bit sums (tile + 4-px splat rim + 16 px of spread): 3 workgroups per CU
constexpr int FW_SPT = FW_TX * FW_TY / 256;           // sources per thread
constexpr float FW_Q = 2147483648.f;                  // 2^31
constexpr double FW_QINV = 1.0 / 2147483648.0;

// FAR: a source whose footprint does not lie inside the window as a whole is not scattered here at all — its linear index goes
// to `far_list` (one global atomic per tile reserves the slots) and the binned gather below takes it; without FAR (no
// workspace for the bins) its out-of-window taps take the global atomics directly, as before.
constexpr int FB_T = 32;                              // target tile edge of the far path (>= 9: a footprint touches at most 2 x 2 tiles)
constexpr int FB_MAP = 16;                            // a source tile's far sources are counted in an LDS map of 16 x 16 target tiles first

// target tiles (<= 2 x 2) of a footprint -> callback(global tile index, index in the workgroup's map or -1)
template <typename F>
__device__ __forceinline__ void fb_tiles(const FwFoot& f, int b, int tt_x, int tt_y, int m0x, int m0y, F&& cb) {
  const int tx0 = f.x_lo / FB_T, tx1 = f.x_hi / FB_T, ty0 = f.y_lo / FB_T, ty1 = f.y_hi / FB_T;
  for (int ty = ty0; ty <= ty1; ty++)
    for (int tx = tx0; tx <= tx1; tx++) {
      const int rx = tx - m0x, ry = ty - m0y;
      const bool in = (unsigned)rx < (unsigned)FB_MAP && (unsigned)ry < (unsigned)FB_MAP;
      cb((b * tt_y + ty) * tt_x + tx, in ? ry * FB_MAP + rx : -1);
    }
}

template <bool DET, bool FAR>
__global__ __launch_bounds__(256) void forward_warp_tile_kernel(const float* __restrict__ flow, unsigned long long* __restrict__ acc,
                                                                float* __restrict__ outf, int B, int H, int W, int tiles_x,
                                                                int tiles_y, int* __restrict__ far_count, int* __restrict__ far_list,
                                                                int* __restrict__ bin_cnt, int4* __restrict__ tile_far, int tt_x, int tt_y) {
  __shared__ unsigned long long win[FW_WX * FW_WY];
  __shared__ int s_org[3];
  __shared__ int s_far[FAR ? FW_TX * FW_TY : 1];
  __shared__ int s_map[FAR ? FB_MAP * FB_MAP : 1];      // far sources per target tile around this source tile's window
  __shared__ int s_nfar, s_base;
  const int tid = threadIdx.x;
  const int ntiles = B * tiles_y * tiles_x;
  for (int tile = (int)xcd_block(); tile < ntiles; tile += gridDim.x) {
    int t = tile;
    const int txi = t % tiles_x; t /= tiles_x;
    const int tyi = t % tiles_y;
    const int b = t / tiles_y;
    if (tid < 3) s_org[tid] = 0;
    if (FAR && tid == 0) s_nfar = 0;
    if (FAR) s_map[tid] = 0;
    for (int e = tid; e < FW_WX * FW_WY; e += 256) win[e] = 0ull;
    __syncthreads();
    // the thread's sources: (x, y + 4 k)
    const int sx = txi * FW_TX + (tid & (FW_TX - 1));
    const int sy0 = tyi * FW_TY + (tid >> 6);
    const long img = (long)b * H * W;
    auto foot = [&](int k) {
      const int sy = sy0 + 4 * k;
      FwFoot f;
      f.ok = false;
      if (sx < W && sy < H) {
        const float2 fl = reinterpret_cast<const float2*>(flow)[img + (long)sy * W + sx];
        f = fw_footprint(sx, sy, fl.x, fl.y, W, H);
      }
      return f;
    };
    int mx = 0, my = 0, mc = 0;                 // integer sums: the window position is the same in every run
#pragma unroll 4
    for (int k = 0; k < FW_SPT; k++) {
      const FwFoot f = foot(k);
      if (f.ok) { mx += f.x_lo + f.x_hi; my += f.y_lo + f.y_hi; mc++; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mx += __shfl_xor(mx, off, 64);
      my += __shfl_xor(my, off, 64);
      mc += __shfl_xor(mc, off, 64);
    }
    if ((tid & 63) == 0) { atomicAdd(&s_org[0], mx); atomicAdd(&s_org[1], my); atomicAdd(&s_org[2], mc); }
    __syncthreads();
    const int cnt = s_org[2];
    if (cnt == 0) {                                    // no source of this tile reaches the image (uniform: nothing to flush)
      if (FAR && tid == 0) tile_far[tile] = make_int4(0, 0, 0, 0);
      __syncthreads();
      continue;
    }
    const int wx0 = s_org[0] / (2 * cnt) - FW_WX / 2, wy0 = s_org[1] / (2 * cnt) - FW_WY / 2;
    // origin of the 16 x 16 target-tile map: centred on the window (floor division: the window may start left of / above the image)
    const int m0x = ((wx0 + FW_WX / 2) >> 5) - FB_MAP / 2, m0y = ((wy0 + FW_WY / 2) >> 5) - FB_MAP / 2;
#pragma unroll 1
    for (int k = 0; k < FW_SPT; k++) {
      const FwFoot f = foot(k);                 // (the flow is read a second time: L2 / L1 hit)
      if (!f.ok) continue;
      const int lx0 = f.x_lo - wx0, ly0 = f.y_lo - wy0;
      if (FAR && !(lx0 >= 0 && f.x_hi - wx0 < FW_WX && ly0 >= 0 && f.y_hi - wy0 < FW_WY)) {
        s_far[atomicAdd(&s_nfar, 1)] = (int)(img + (long)(sy0 + 4 * k) * W + sx);      // the binned gather takes the whole source
        fb_tiles(f, b, tt_x, tt_y, m0x, m0y, [&](int gt, int mi) {
          if (mi >= 0) atomicAdd(&s_map[mi], 1);       // counted per workgroup first: ~30 global atomics per source tile, not ~1300
          else atomicAdd(bin_cnt + gt, 1);
        });
        continue;
      }
      float wxv[9];
#pragma unroll
      for (int j = 0; j < 9; j++) {
        const float dx = (float)(f.x_lo + j) - f.tx;
        wxv[j] = expf(-(dx * dx) / 2.0f);
      }
      if (f.x_hi - f.x_lo == 8 && f.y_hi - f.y_lo == 8 && lx0 >= 0 && lx0 + 8 < FW_WX && ly0 >= 0 && ly0 + 8 < FW_WY) {
        // the common case — a whole 9 x 9 footprint inside the window: no per-tap tests, constant LDS offsets
        unsigned long long* base = win + ly0 * FW_WX + lx0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
          const float dy = (float)(f.y_lo + i) - f.ty;
          const float wy = expf(-(dy * dy) / 2.0f);
#pragma unroll
          for (int j = 0; j < 9; j++)
            atomicAdd(base + i * FW_WX + j, (unsigned long long)(unsigned)(wy * wxv[j] * FW_Q + 0.5f));
        }
        continue;
      }
#pragma unroll
      for (int i = 0; i < 9; i++) {
        const int ny = f.y_lo + i;
        if (ny > f.y_hi) continue;
        const float dy = (float)ny - f.ty;
        const float wy = expf(-(dy * dy) / 2.0f);
        const int ly = ny - wy0;
#pragma unroll
        for (int j = 0; j < 9; j++) {
          const int nx = f.x_lo + j;
          if (nx > f.x_hi) continue;
          const float w = wy * wxv[j];
          const unsigned long long q = (unsigned long long)(unsigned)(w * FW_Q + 0.5f);
          const int lx = nx - wx0;
          if ((unsigned)lx < (unsigned)FW_WX && (unsigned)ly < (unsigned)FW_WY) {
            atomicAdd(&win[ly * FW_WX + lx], q);
          } else if (DET) {
            atomicAdd(acc + img + (long)ny * W + nx, q);
          } else {
            atomicAdd(outf + img + (long)ny * W + nx, w);
          }
        }
      }
    }
    __syncthreads();
    if (FAR) {
      const int nf = s_nfar;
      if (nf > 0) {
        if (tid == 0) s_base = atomicAdd(far_count, nf);
        const int c = s_map[tid];
        if (c > 0) {
          const int ty = m0y + tid / FB_MAP, tx = m0x + tid % FB_MAP;      // (inside the image: footprints are clipped to it)
          atomicAdd(bin_cnt + (b * tt_y + ty) * tt_x + tx, c);
        }
        __syncthreads();
        for (int e = tid; e < nf; e += 256) far_list[s_base + e] = s_far[e];
        if (tid == 0) tile_far[tile] = make_int4(s_base, nf, m0x, m0y);
      } else if (tid == 0) {
        tile_far[tile] = make_int4(0, 0, 0, 0);
      }
    }
    for (int e = tid; e < FW_WX * FW_WY; e += 256) {
      const unsigned long long v = win[e];
      if (v == 0ull) continue;
      const int ly = e / FW_WX, lx = e - ly * FW_WX;
      const long g = img + (long)(wy0 + ly) * W + (wx0 + lx);      // inside the image: every tap was clipped to it
      if (DET) atomicAdd(acc + g, v);
      else atomicAdd(outf + g, (float)((double)v * FW_QINV));
    }
    __syncthreads();
  }
}

// ---- far sources: counting sort by TARGET tile, then a gather without global atomics ----------------------------------------
// A field that tears a source tile apart (|flow| of tens of pixels, uncorrelated) leaves most footprints outside the tile's
// window; as global atomics those taps cost 81 memory-side read-modify-writes per source (22 ms at 16 x 768 x 1024, U(-50, 50)).
// Instead: (1) the tile kernel lists such sources and (2) counts them into the 32 x 32-pixel TARGET tiles their footprints touch
// (<= 2 x 2: a footprint is 9 pixels wide) — per workgroup in an LDS map of 16 x 16 tiles first, (3) an exclusive scan turns the
// counts into bin offsets, (4) the sources are written into their bins, (5) one workgroup per target tile sums the taps of its bin's sources that fall inside the
// tile in LDS (64-bit fixed point, as above) and adds the tile to the output with plain stores — every pixel has one owner, and
// integer sums do not depend on the order of a bin's entries: bit-reproducible like the window path.  Everything is sized and
// looped from device-side counts: no host synchronisation, graph-capturable.
__device__ __forceinline__ FwFoot fw_foot_of(const float* __restrict__ flow, int src, int H, int W) {
  const int x = src % W, y = (src / W) % H;
  const float2 fl = reinterpret_cast<const float2*>(flow)[src];
  return fw_footprint(x, y, fl.x, fl.y, W, H);
}

// (4) one workgroup per SOURCE tile writes its far sources into their bins: the (source, target tile) pairs are counted in the
// LDS map again, each touched bin's range is reserved with ONE global atomic on its cursor, and the pairs take consecutive
// slots of that range.  The order of a bin's entries varies run to run; the integer sums of the gather do not.
__global__ __launch_bounds__(256) void forward_warp_fill_kernel(const float* __restrict__ flow, const int* __restrict__ far_list,
                                                                const int4* __restrict__ tile_far, int* __restrict__ cursor,
                                                                const int* __restrict__ off, int* __restrict__ entries, int H, int W,
                                                                int tt_x, int tt_y, int ntiles, int tiles_per_image) {
  __shared__ int s_map[FB_MAP * FB_MAP], s_res[FB_MAP * FB_MAP];
  for (int tile = (int)xcd_block(); tile < ntiles; tile += gridDim.x) {
    const int4 tf = tile_far[tile];
    const int base = tf.x, nf = tf.y, m0x = tf.z, m0y = tf.w;
    if (nf == 0) continue;                              // (uniform)
    const int b = tile / tiles_per_image;
    s_map[threadIdx.x] = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < nf; e += 256) {
      const FwFoot f = fw_foot_of(flow, far_list[base + e], H, W);
      fb_tiles(f, b, tt_x, tt_y, m0x, m0y, [&](int gt, int mi) { if (mi >= 0) atomicAdd(&s_map[mi], 1); });
    }
    __syncthreads();
    {
      const int c = s_map[threadIdx.x];
      if (c > 0) {
        const int gt = (b * tt_y + m0y + threadIdx.x / FB_MAP) * tt_x + m0x + threadIdx.x % FB_MAP;
        s_res[threadIdx.x] = off[gt] + atomicAdd(cursor + gt, c);
      }
      s_map[threadIdx.x] = 0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nf; e += 256) {
      const int src = far_list[base + e];
      const FwFoot f = fw_foot_of(flow, src, H, W);
      fb_tiles(f, b, tt_x, tt_y, m0x, m0y, [&](int gt, int mi) {
        const int slot = mi >= 0 ? s_res[mi] + atomicAdd(&s_map[mi], 1) : off[gt] + atomicAdd(cursor + gt, 1);
        entries[slot] = src;
      });
    }
    __syncthreads();
  }
}

// exclusive scan of the bin counts (one workgroup); leaves the counts zeroed for their second life as cursors
__global__ __launch_bounds__(1024) void forward_warp_scan_kernel(int* __restrict__ cnt, int* __restrict__ off, int ntt) {
  __shared__ int part[1024];
  const int per = (ntt + 1023) / 1024;
  const int lo = threadIdx.x * per, hi = min(ntt, lo + per);
  int s = 0;
  for (int i = lo; i < hi; i++) s += cnt[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - s;
  for (int i = lo; i < hi; i++) {
    const int c = cnt[i];
    off[i] = run;
    run += c;
    cnt[i] = 0;
  }
  if (threadIdx.x == 1023) off[ntt] = part[1023];
}

template <bool DET>
__global__ __launch_bounds__(256) void forward_warp_gather_kernel(const float* __restrict__ flow, const int* __restrict__ off,
                                                                  const int* __restrict__ entries, unsigned long long* __restrict__ acc,
                                                                  float* __restrict__ outf, int H, int W, int tt_x, int tt_y, int ntt) {
  __shared__ unsigned long long win[FB_T * FB_T];
  for (int t = (int)xcd_block(); t < ntt; t += gridDim.x) {
    const int e0 = off[t], e1 = off[t + 1];
    if (e0 == e1) continue;                            // (uniform)
    for (int e = threadIdx.x; e < FB_T * FB_T; e += 256) win[e] = 0ull;
    __syncthreads();
    const int tx = t % tt_x, ty = (t / tt_x) % tt_y, b = t / (tt_x * tt_y);
    const int X0 = tx * FB_T, Y0 = ty * FB_T;
    for (int e = e0 + threadIdx.x; e < e1; e += 256) {
      const FwFoot f = fw_foot_of(flow, entries[e], H, W);
      float wxv[9];                                    // separable weights, the same two expf per tap as the window path: same bits
#pragma unroll
      for (int j = 0; j < 9; j++) {
        const float dx = (float)(f.x_lo + j) - f.tx;
        wxv[j] = expf(-(dx * dx) / 2.0f);
      }
      const int lx0 = f.x_lo - X0;
#pragma unroll
      for (int i = 0; i < 9; i++) {
        const int ny = f.y_lo + i;
        if (ny > f.y_hi || ny < Y0 || ny >= Y0 + FB_T) continue;
        const float dy = (float)ny - f.ty;
        const float wy = expf(-(dy * dy) / 2.0f);
        unsigned long long* row = win + (ny - Y0) * FB_T + lx0;
#pragma unroll
        for (int j = 0; j < 9; j++) {
          const int nx = f.x_lo + j;
          if (nx > f.x_hi || nx < X0 || nx >= X0 + FB_T) continue;
          atomicAdd(row + j, (unsigned long long)(unsigned)(wy * wxv[j] * FW_Q + 0.5f));
        }
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < FB_T * FB_T; e += 256) {
      const unsigned long long v = win[e];
      const int y = Y0 + e / FB_T, x = X0 + e % FB_T;
      if (v == 0ull || y >= H || x >= W) continue;
      const long gidx = ((long)b * H + y) * W + x;
      if (DET) acc[gidx] += v;                         // this pixel's only writer in this launch; the tile kernel's atomics are done
      else outf[gidx] += (float)((double)v * FW_QINV);
    }
    __syncthreads();
  }
}

__global__ void forward_warp_fixed_to_float_kernel(const unsigned long long* __restrict__ acc,
                                                   float* __restrict__ out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = (float)((double)acc[i] * FW_QINV);
}

// Gradient (forward_warp_op.cu.cc:67-125): a gather over the source's own footprint — no atomics in the reference either.
// Same separable weights; dout is read through L1 (neighbouring sources share most of their 9 x 9 windows).
__global__ void forward_warp_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ flow,
                                        float* __restrict__ dflow, int B, int H, int W) {
  const unsigned T = tile_count((unsigned)W, (unsigned)H, (unsigned)B);
  for (unsigned t = tile_first(T); t < tile_last(T); t++) {
    const TilePix px = tile_pix(t, (unsigned)W, (unsigned)H);
    if (!px.ok) continue;
    const float2 fl = reinterpret_cast<const float2*>(flow)[px.i];
    const FwFoot f = fw_footprint(px.x, px.y, fl.x, fl.y, W, H);
    float du = 0.f, dv = 0.f;
    if (f.ok) {
      const float* img = dout + (long)px.n * H * W;
      float wxv[9], dxv[9];
#pragma unroll
      for (int j = 0; j < 9; j++) {
        dxv[j] = (float)(f.x_lo + j) - f.tx;
        wxv[j] = expf(-(dxv[j] * dxv[j]) / 2.0f);
      }
#pragma unroll
      for (int i = 0; i < 9; i++) {
        const int ny = f.y_lo + i;
        if (ny > f.y_hi) continue;
        const float dy = (float)ny - f.ty;
        const float wy = expf(-(dy * dy) / 2.0f);
#pragma unroll
        for (int j = 0; j < 9; j++) {
          const int nx = f.x_lo + j;
          if (nx > f.x_hi) continue;
          const float factor = img[(long)ny * W + nx] * (wy * wxv[j]);      // 2 * din * weight / gauss_divisor, gauss_divisor = 2
          du += factor * dxv[j];
          dv += factor * dy;
        }
      }
    }
    reinterpret_cast<float2*>(dflow)[px.i] = make_float2(du, dv);
  }
}

// The same with a footprint row fetched as 16 + 16 + 4 bytes (dword-aligned buffer loads, zeros past the end of the tensor)
// instead of nine dword loads: a third of the load instructions — what the gather costs on torn fields, where every lane of a
// load touches its own cache line (i.i.d. +-50 px at 16 x 768 x 1024: 4.0 ms with dword loads).  Tensors below 2 GiB.
__global__ void forward_warp_bwd_rows_kernel(const float* __restrict__ dout, const float* __restrict__ flow,
                                             float* __restrict__ dflow, int B, int H, int W) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dout), 0, (int)((long)B * H * W * 4), 0x00020000);
  const unsigned T = tile_count((unsigned)W, (unsigned)H, (unsigned)B);
  for (unsigned t = tile_first(T); t < tile_last(T); t++) {
    const TilePix px = tile_pix(t, (unsigned)W, (unsigned)H);
    if (!px.ok) continue;
    const float2 fl = reinterpret_cast<const float2*>(flow)[px.i];
    const FwFoot f = fw_footprint(px.x, px.y, fl.x, fl.y, W, H);
    float du = 0.f, dv = 0.f;
    if (f.ok) {
      float wxv[9], dxv[9];
#pragma unroll
      for (int j = 0; j < 9; j++) {
        dxv[j] = (float)(f.x_lo + j) - f.tx;
        // columns past the footprint (clipped at the image border) get weight 0: whatever the wide loads brought is dropped
        wxv[j] = f.x_lo + j <= f.x_hi ? expf(-(dxv[j] * dxv[j]) / 2.0f) : 0.f;
      }
      const int base = ((px.n * H + f.y_lo) * W + f.x_lo) * 4;
#pragma unroll
      for (int i = 0; i < 9; i++) {
        const int ny = f.y_lo + i;
        const bool row = ny <= f.y_hi;
        const int off = row ? base + i * W * 4 : 0x7fffff00;            // out of range: zeros, no traffic
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        const u4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0);
        const unsigned c = __builtin_amdgcn_raw_buffer_load_b32(rs, off + 32, 0, 0);
        const float dy = (float)ny - f.ty;
        const float wy = row ? expf(-(dy * dy) / 2.0f) : 0.f;
        const float v[9] = {__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(a[2]), __uint_as_float(a[3]),
                            __uint_as_float(b[0]), __uint_as_float(b[1]), __uint_as_float(b[2]), __uint_as_float(b[3]), __uint_as_float(c)};
#pragma unroll
        for (int j = 0; j < 9; j++) {
          // (a dropped column may hold a NaN of the neighbouring row: select, do not multiply by zero)
          const float factor = wxv[j] != 0.f && row ? v[j] * (wy * wxv[j]) : 0.f;
          du += factor * dxv[j];
          dv += factor * dy;
        }
      }
    }
    reinterpret_cast<float2*>(dflow)[px.i] = make_float2(du, dv);
  }
}

__global__ void forward_warp_ranges_kernel(const float* __restrict__ flow, int* __restrict__ ranges, int B, int H,
                                           int W) {
  const long npx = (long)B * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npx; i += (long)gridDim.x * blockDim.x) {
    const float2 fl = reinterpret_cast<const float2*>(flow)[i];
    const FwFoot f = fw_footprint((int)(i % W), (int)((i / W) % H), fl.x, fl.y, W, H);
    reinterpret_cast<int4*>(ranges)[i] = f.ok ? make_int4(f.x_lo, f.x_hi, f.y_lo, f.y_hi) : make_int4(-1, -1, -1, -1);
  }
}

// workspace layout: [64-bit sums, deterministic only][far count (64 B)][bin counts / cursors][bin offsets + 1][per source tile: list
// range + map origin][far list][bin entries]
struct FwWs {
  size_t acc, count, cnt, off, tfar, list, entries, total;
  int tt_x, tt_y, ntt;
};
static FwWs fw_ws_layout(int B, int H, int W, int deterministic) {
  FwWs w{};
  const size_t npx = (size_t)B * H * W;
  w.tt_x = cdiv(W, FB_T); w.tt_y = cdiv(H, FB_T); w.ntt = B * w.tt_x * w.tt_y;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t o = 0;
  w.acc = o; o += deterministic ? al(sizeof(unsigned long long) * npx) : 0;
  w.count = o; o += 256;
  w.cnt = o; o += al(sizeof(int) * (size_t)w.ntt);
  w.off = o; o += al(sizeof(int) * ((size_t)w.ntt + 1));
  w.tfar = o; o += al(sizeof(int4) * (size_t)B * cdiv(H, FW_TY) * cdiv(W, FW_TX));
  w.list = o; o += al(sizeof(int) * npx);
  w.entries = o; o += al(sizeof(int) * 4 * npx);       // a footprint touches at most 2 x 2 target tiles
  w.total = o;
  return w;
}

UNFLOW_API size_t unflow_forward_warp_workspace_bytes(int B, int H, int W, int deterministic) {
  if (B <= 0 || H <= 0 || W <= 0 || (long)B * H * W >= (1L << 31)) return 0;
  return fw_ws_layout(B, H, W, deterministic).total;
}

UNFLOW_API int unflow_forward_warp_fwd(const float* flows, float* out, int B, int H, int W, int deterministic,
                                       void* workspace, size_t workspace_bytes, unflow_stream_t stream) {
  if (!flows || !out) return UNFLOW_ERR_NULL;
  if (B < 0 || H < 0 || W < 0) return UNFLOW_ERR_SHAPE;
  const long npx = (long)B * H * W;
  if (npx == 0) return UNFLOW_OK;
  if (npx >= (1L << 31)) return UNFLOW_ERR_UNSUPPORTED;
  const int tiles_x = cdiv(W, FW_TX), tiles_y = cdiv(H, FW_TY);
  const int ntiles = B * tiles_y * tiles_x;
  const int grid = ntiles < 2048 ? ntiles : 2048;
  hipStream_t st = as_stream(stream);
  if (deterministic) {
    if (!workspace) return UNFLOW_ERR_NULL;
    if (workspace_bytes < sizeof(unsigned long long) * (size_t)npx) return UNFLOW_ERR_WORKSPACE;
  }
  // With the full workspace (unflow_forward_warp_workspace_bytes) far sources go through the binned gather; with the minimum
  // (8 bytes per pixel, deterministic; none otherwise) their taps are global atomics — same result, slower on torn fields.
  const FwWs w = fw_ws_layout(B, H, W, deterministic);
  const bool far = workspace && workspace_bytes >= w.total;
  char* base = reinterpret_cast<char*>(workspace);
  unsigned long long* acc = deterministic ? reinterpret_cast<unsigned long long*>(base + w.acc) : nullptr;
  int* far_count = far ? reinterpret_cast<int*>(base + w.count) : nullptr;
  int* cnt = far ? reinterpret_cast<int*>(base + w.cnt) : nullptr;
  int* off = far ? reinterpret_cast<int*>(base + w.off) : nullptr;
  int* far_list = far ? reinterpret_cast<int*>(base + w.list) : nullptr;
  int4* tile_far = far ? reinterpret_cast<int4*>(base + w.tfar) : nullptr;
  int* entries = far ? reinterpret_cast<int*>(base + w.entries) : nullptr;
  if (deterministic) {
    if (hipMemsetAsync(acc, 0, sizeof(unsigned long long) * npx, st) != hipSuccess) return UNFLOW_ERR_LAUNCH;
  } else {
    if (hipMemsetAsync(out, 0, sizeof(float) * npx, st) != hipSuccess) return UNFLOW_ERR_LAUNCH;
  }
  if (far && hipMemsetAsync(base + w.count, 0, (w.off - w.count), st) != hipSuccess) return UNFLOW_ERR_LAUNCH;      // far count + bin counts
  if (deterministic) {
    if (far) forward_warp_tile_kernel<true, true><<<grid, 256, 0, st>>>(flows, acc, nullptr, B, H, W, tiles_x, tiles_y, far_count, far_list, cnt, tile_far, w.tt_x, w.tt_y);
    else forward_warp_tile_kernel<true, false><<<grid, 256, 0, st>>>(flows, acc, nullptr, B, H, W, tiles_x, tiles_y, nullptr, nullptr, nullptr, nullptr, 0, 0);
  } else {
    if (far) forward_warp_tile_kernel<false, true><<<grid, 256, 0, st>>>(flows, nullptr, out, B, H, W, tiles_x, tiles_y, far_count, far_list, cnt, tile_far, w.tt_x, w.tt_y);
    else forward_warp_tile_kernel<false, false><<<grid, 256, 0, st>>>(flows, nullptr, out, B, H, W, tiles_x, tiles_y, nullptr, nullptr, nullptr, nullptr, 0, 0);
  }
  if (far) {
    forward_warp_scan_kernel<<<1, 1024, 0, st>>>(cnt, off, w.ntt);
    forward_warp_fill_kernel<<<grid, 256, 0, st>>>(flows, far_list, tile_far, cnt, off, entries, H, W, w.tt_x, w.tt_y, ntiles, tiles_y * tiles_x);
    const int ggrid = w.ntt < 4096 ? w.ntt : 4096;
    if (deterministic) forward_warp_gather_kernel<true><<<ggrid, 256, 0, st>>>(flows, off, entries, acc, nullptr, H, W, w.tt_x, w.tt_y, w.ntt);
    else forward_warp_gather_kernel<false><<<ggrid, 256, 0, st>>>(flows, off, entries, nullptr, out, H, W, w.tt_x, w.tt_y, w.ntt);
  }
  if (deterministic) forward_warp_fixed_to_float_kernel<<<stream_grid(npx), 256, 0, st>>>(acc, out, npx);
  return launch_status();
}

UNFLOW_API int unflow_forward_warp_bwd(const float* dout, const float* flows, float* dflows, int B, int H, int W,
                                       unflow_stream_t stream) {
  if (!dout || !flows || !dflows) return UNFLOW_ERR_NULL;
  const long npx = (long)B * H * W;
  if (npx == 0) return UNFLOW_OK;
  if (npx * 4 < (1l << 31) - 1024)
    forward_warp_bwd_rows_kernel<<<stream_grid(npx), 256, 0, as_stream(stream)>>>(dout, flows, dflows, B, H, W);
  else
    forward_warp_bwd_kernel<<<stream_grid(npx), 256, 0, as_stream(stream)>>>(dout, flows, dflows, B, H, W);
  return launch_status();
}

UNFLOW_API int unflow_forward_warp_ranges(const float* flows, int* ranges, int B, int H, int W,
                                          unflow_stream_t stream) {
  if (!flows || !ranges) return UNFLOW_ERR_NULL;
  const long npx = (long)B * H * W;
  if (npx == 0) return UNFLOW_OK;
  forward_warp_ranges_kernel<<<stream_grid(npx), 256, 0, as_stream(stream)>>>(flows, ranges, B, H, W);
  return launch_status();
}

// ------------------------------------------------------------------ downsample
// One thread per output element; consecutive threads = consecutive channels then x (coalesced).
__global__ void downsample_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int H, int W, int C,
                                  int scale) {
  const int oh = H / scale, ow = W / scale;
  const long n = (long)B * oh * ow * C;
  const float inv = (float)(scale * scale);
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int ox = (int)((e / C) % ow), oy = (int)((e / C / ow) % oh);
    const long b = e / ((long)C * ow * oh);
    float s = 0.f;
    for (int yy = oy * scale; yy < (oy + 1) * scale; yy++)
      for (int xx = ox * scale; xx < (ox + 1) * scale; xx++) s += img[((b * H + yy) * W + xx) * C + c];
    out[e] = s / inv;
  }
}

// C = 3, scale 2 / 4 (the loss pyramid's own calls): one thread per OUTPUT PIXEL, each input row of its box as 3 S contiguous
// floats in dword-aligned 16 / 8-byte buffer loads (4 or 12 load instructions per pixel instead of 12 or 48), the same
// summation order per channel (rows outer, columns inner, one division) — bit-identical to downsample_kernel.  < 2 GiB.
template <int S>
__global__ __launch_bounds__(256) void downsample3_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int H, int W) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, (int)((long)B * H * W * 12), 0x00020000);
  const int oh = H / S, ow = W / S;
  const long n = (long)B * oh * ow;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(e % ow), oy = (int)((e / ow) % oh);
    const int b = (int)(e / ((long)ow * oh));
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int yy = 0; yy < S; yy++) {
      const int off = (((b * H + oy * S + yy) * W) + ox * S) * 12;
      float v[3 * S];
      if constexpr (S == 2) {
        const u4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        const u2 c = __builtin_amdgcn_raw_buffer_load_b64(rs, off + 16, 0, 0);
        v[0] = __uint_as_float(a[0]); v[1] = __uint_as_float(a[1]); v[2] = __uint_as_float(a[2]); v[3] = __uint_as_float(a[3]);
        v[4] = __uint_as_float(c[0]); v[5] = __uint_as_float(c[1]);
      } else {
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const u4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16 * k, 0, 0);
          v[4 * k] = __uint_as_float(a[0]); v[4 * k + 1] = __uint_as_float(a[1]);
          v[4 * k + 2] = __uint_as_float(a[2]); v[4 * k + 3] = __uint_as_float(a[3]);
        }
      }
#pragma unroll
      for (int xx = 0; xx < S; xx++) { s0 += v[3 * xx]; s1 += v[3 * xx + 1]; s2 += v[3 * xx + 2]; }
    }
    const float inv = (float)(S * S);
    float* o = out + e * 3;
    o[0] = s0 / inv; o[1] = s1 / inv; o[2] = s2 / inv;
  }
}

// The image pyramid of the loss (unsupervised.py:99-100,145-146: downsample(im, 4), then four times downsample(., 2)) in ONE
// launch: a workgroup owns a 64 x 64 tile of the full-resolution image = 16 x 16 / 8 x 8 / 4 x 4 / 2 x 2 / 1 pixels of the five
// levels; every level is the box mean of the previous level's VALUES with downsample_kernel's own summation order (rows outer,
// columns inner, then one division by scale^2), handed down through LDS — so each level is bit-identical to the chain of five
// launches it replaces (each 5-6 us of latency on the step's critical path).  H, W multiples of 64, C = 3.
__global__ __launch_bounds__(256) void image_pyramid5_kernel(const float* __restrict__ img, float* __restrict__ l0, float* __restrict__ l1,
                                                              float* __restrict__ l2, float* __restrict__ l3, float* __restrict__ l4,
                                                              int H, int W) {
  __shared__ float lv[2][16 * 16 * 3];
  const int tiles_x = W >> 6, tiles_y = H >> 6;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int n = t / tiles_y;
  const int tid = threadIdx.x;
  {   // level 0: 16 x 16 pixels per tile, each the mean of 4 x 4 full-resolution pixels
    const int ox = tid & 15, oy = tid >> 4;
    const float* src = img + (((long)n * H + (ty * 64 + oy * 4)) * W + tx * 64 + ox * 4) * 3;
    float s[3] = {0.f, 0.f, 0.f};
    for (int yy = 0; yy < 4; yy++)
      for (int xx = 0; xx < 4; xx++)
#pragma unroll
        for (int c = 0; c < 3; c++) s[c] += src[((long)yy * W + xx) * 3 + c];
    const int h0 = H >> 2, w0 = W >> 2;
    float* dst = l0 + (((long)n * h0 + ty * 16 + oy) * w0 + tx * 16 + ox) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float v = s[c] / 16.0f;
      dst[c] = v;
      lv[0][(oy * 16 + ox) * 3 + c] = v;
    }
  }
  float* outs[4] = {l1, l2, l3, l4};
  int side = 16;                                  // pixels per tile side of the level held in lv[cur]
#pragma unroll
  for (int k = 0; k < 4; k++) {
    __syncthreads();
    const int cur = k & 1, ns = side >> 1;
    if (tid < ns * ns) {
      const int ox = tid % ns, oy = tid / ns;
      float s[3] = {0.f, 0.f, 0.f};
      for (int yy = 0; yy < 2; yy++)
        for (int xx = 0; xx < 2; xx++)
#pragma unroll
          for (int c = 0; c < 3; c++) s[c] += lv[cur][((oy * 2 + yy) * side + ox * 2 + xx) * 3 + c];
      const int hk = H >> (3 + k), wk = W >> (3 + k);
      float* dst = outs[k] + (((long)n * hk + ty * ns + oy) * wk + tx * ns + ox) * 3;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float v = s[c] / 4.0f;
        dst[c] = v;
        lv[cur ^ 1][(oy * ns + ox) * 3 + c] = v;
      }
    }
    side = ns;
  }
}

// levels[0..4]: [N, H/4, W/4, 3], [N, H/8, W/8, 3], ... [N, H/64, W/64, 3]
UNFLOW_API int unflow_image_pyramid5(const float* images, float* const* levels, int N, int H, int W, unflow_stream_t stream) {
  if (!images || !levels) return UNFLOW_ERR_NULL;
  for (int k = 0; k < 5; k++)
    if (!levels[k]) return UNFLOW_ERR_NULL;
  if (N < 0 || H <= 0 || W <= 0) return UNFLOW_ERR_SHAPE;
  if (H % 64 != 0 || W % 64 != 0) return UNFLOW_ERR_NOT_DIVISIBLE;
  if (N == 0) return UNFLOW_OK;
  image_pyramid5_kernel<<<N * (H >> 6) * (W >> 6), 256, 0, as_stream(stream)>>>(images, levels[0], levels[1], levels[2], levels[3],
                                                                                levels[4], H, W);
  return launch_status();
}

// ------------------------------------------------------------------ resize_area (core/util.py:12-14, 26)
// tf.image.resize_area (TF's ResizeAreaOp): output pixel (oy, ox) is the area-weighted mean of the source rectangle
// [oy*sy, (oy+1)*sy) x [ox*sx, (ox+1)*sx), sy = H / oh, sx = W / ow: a source pixel counts with the covered fraction of its
// unit square; indices clamped to the image like TF's BOUND.  One thread per output element (channels fastest); the
// reference calls it only for odd-sized tensors, once per pyramid level, off the training step.
__global__ void resize_area_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int H, int W, int C, int oh, int ow) {
  const long n = (long)B * oh * ow * C;
  // source intervals in fp64 (TF's kernel forms x * scale in fp32: at x ~ 1200 that is 1e-4 of a pixel of coverage, 5e-5 of the
  // result; the exact intervals are the definition, and this runs once per odd-sized level, off the step)
  const double sy = (double)H / (double)oh, sx = (double)W / (double)ow;
  const float inv = (float)(1.0 / (sy * sx));
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int ox = (int)((e / C) % ow), oy = (int)((e / C / ow) % oh);
    const long b = e / ((long)C * ow * oh);
    const double y0 = oy * sy, y1 = (oy + 1) * sy, x0 = ox * sx, x1 = (ox + 1) * sx;
    float s = 0.f;
    for (int j = (int)floor(y0); j < (int)ceil(y1); j++) {
      const float wy = (float)(fmin(y1, (double)(j + 1)) - fmax(y0, (double)j));
      const int jj = j < 0 ? 0 : (j > H - 1 ? H - 1 : j);
      float row = 0.f;
      for (int i = (int)floor(x0); i < (int)ceil(x1); i++) {
        const float wx = (float)(fmin(x1, (double)(i + 1)) - fmax(x0, (double)i));
        const int ii = i < 0 ? 0 : (i > W - 1 ? W - 1 : i);
        row += wx * img[((b * H + jj) * W + ii) * C + c];
      }
      s += wy * row;
    }
    out[e] = s * inv;
  }
}

UNFLOW_API int unflow_resize_area(const float* images, float* out, int B, int H, int W, int C, int out_h, int out_w,
                                  unflow_stream_t stream) {
  if (!images || !out) return UNFLOW_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || out_h <= 0 || out_w <= 0) return UNFLOW_ERR_SHAPE;
  const long n = (long)B * out_h * out_w * C;
  resize_area_kernel<<<stream_grid(n), 256, 0, as_stream(stream)>>>(images, out, B, H, W, C, out_h, out_w);
  return launch_status();
}

UNFLOW_API int unflow_downsample_fwd(const float* images, float* out, int B, int H, int W, int C, int scale,
                                     unflow_stream_t stream) {
  if (!images || !out) return UNFLOW_ERR_NULL;
  if (B < 0 || H < 0 || W < 0 || C < 0) return UNFLOW_ERR_SHAPE;
  if (scale <= 0 || H % scale != 0 || W % scale != 0) return UNFLOW_ERR_NOT_DIVISIBLE;
  const long n = (long)B * (H / scale) * (W / scale) * C;
  if (n == 0) return UNFLOW_OK;
  if (C == 3 && (scale == 2 || scale == 4) && (long)B * H * W * 12 < (1l << 31) - 64) {
    if (scale == 2) downsample3_kernel<2><<<stream_grid(n / 3), 256, 0, as_stream(stream)>>>(images, out, B, H, W);
    else downsample3_kernel<4><<<stream_grid(n / 3), 256, 0, as_stream(stream)>>>(images, out, B, H, W);
    return launch_status();
  }
  downsample_kernel<<<stream_grid(n), 256, 0, as_stream(stream)>>>(images, out, B, H, W, C, scale);
  return launch_status();
}
