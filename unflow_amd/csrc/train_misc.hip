// Step plumbing around the network: input normalisation (unsupervised.py:29-32,69-70), TF1
// resize_bilinear (unsupervised.py:103-104), L2 regulariser (flownet.py:176 via tf.nn.l2_loss),
// fused TF-form Adam (train.py:151-152) and the EPE sums (flow_util.py:98-103).  All HBM-bound
// streaming kernels: 16-byte accesses, grid-stride, >= 2048 workgroups when the data allows.
#include "common.h"
#include "igemm_shared.h"

UNFLOW_API const char* unflow_status_string(int status) {
  switch (status) {
    case UNFLOW_OK: return "ok";
    case UNFLOW_ERR_NULL: return "null pointer argument";
    case UNFLOW_ERR_EMPTY_OUTPUT: return "Invalid correlation settings";
    case UNFLOW_ERR_EVEN_KERNEL: return "kernel_size must be odd";
    case UNFLOW_ERR_NOT_DIVISIBLE: return "Input height and width must be divisible by scale";
    case UNFLOW_ERR_SHAPE: return "Input shapes have to be the same";
    case UNFLOW_ERR_UNSUPPORTED: return "unsupported configuration";
    case UNFLOW_ERR_LAUNCH: return "HIP launch failed";
    case UNFLOW_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown status";
  }
}

UNFLOW_API int unflow_version(void) { return 100; }

__global__ void prepare_images_kernel(const float* __restrict__ im, float* __restrict__ net4,
                                      float* __restrict__ out01, float m0, float m1, float m2, long npix) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const float r = im[3 * i] / 255.0f, g = im[3 * i + 1] / 255.0f, b = im[3 * i + 2] / 255.0f;
    reinterpret_cast<float4*>(net4)[i] = make_float4(r - m0, g - m1, b - m2, 0.f);
    if (out01) {
      out01[3 * i] = r;
      out01[3 * i + 1] = g;
      out01[3 * i + 2] = b;
    }
  }
}

UNFLOW_API int unflow_prepare_images(const float* im_u8range, float* net_in4, float* out01, const float* mean3,
                                     long npix, unflow_stream_t stream) {
  if (!im_u8range || !net_in4 || !mean3) return UNFLOW_ERR_NULL;
  if (npix <= 0) return UNFLOW_OK;
  // mean3 is a HOST pointer to the three channel means in [0,255] (core/input.py:45); mean/255 in fp32
  const float m0 = mean3[0] / 255.0f, m1 = mean3[1] / 255.0f, m2 = mean3[2] / 255.0f;
  prepare_images_kernel<<<stream_grid(npix), 256, 0, as_stream(stream)>>>(im_u8range, net_in4, out01, m0, m1, m2, npix);
  return launch_status();
}

// Both frames of the minibatch in one pass (im1 = samples [0, B), im2 = [B, 2B) of the directed batch), optionally with the
// 16-bit operand planes of the network input (csrc/conv_planes.hip: conv1 reads planes) — the step's input preparation is
// this one launch.
__global__ void prepare_image_pair_kernel(const float* __restrict__ im1, const float* __restrict__ im2, long npix_each,
                                          float* __restrict__ net4, float* __restrict__ out01, float m0, float m1, float m2,
                                          igemm::PlaneOut pl) {
  const long npix = 2 * npix_each;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const float* im = i < npix_each ? im1 + 3 * i : im2 + 3 * (i - npix_each);
    const float r = im[0] / 255.0f, g = im[1] / 255.0f, b = im[2] / 255.0f;
    const float4 v = make_float4(r - m0, g - m1, b - m2, 0.f);
    reinterpret_cast<float4*>(net4)[i] = v;
    if (out01) {
      out01[3 * i] = r;
      out01[3 * i + 1] = g;
      out01[3 * i + 2] = b;
    }
    igemm::store_planes4(pl, (size_t)i, 0, v);
  }
}

UNFLOW_API int unflow_prepare_image_pair(const float* im1, const float* im2, long npix_each, float* net_in4, float* out01,
                                         const float* mean3, const unflow_planes* net_pl, unflow_stream_t stream) {
  if (!im1 || !im2 || !net_in4 || !mean3) return UNFLOW_ERR_NULL;
  if (npix_each <= 0) return UNFLOW_OK;
  igemm::PlaneOut pl{};
  if (net_pl && net_pl->base) {
    if ((net_pl->n_planes != 1 && net_pl->n_planes != 3) || net_pl->ld < 4 || net_pl->ld % 4 != 0 ||
        (reinterpret_cast<uintptr_t>(net_pl->base) & 7) != 0)
      return UNFLOW_ERR_UNSUPPORTED;
    pl.base = reinterpret_cast<unsigned short*>(net_pl->base);
    pl.plane_stride = net_pl->plane_stride;
    pl.ld = net_pl->ld; pl.lo = 0; pl.hi = 4; pl.n_planes = net_pl->n_planes;
  }
  const float m0 = mean3[0] / 255.0f, m1 = mean3[1] / 255.0f, m2 = mean3[2] / 255.0f;
  prepare_image_pair_kernel<<<stream_grid(2 * npix_each), 256, 0, as_stream(stream)>>>(im1, im2, npix_each, net_in4, out01, m0,
                                                                                     m1, m2, pl);
  return launch_status();
}

// ---- the few elementwise helpers the step driver needs between kernels (buffers stay caller-owned torch tensors, but no
// torch compute kernel runs inside a step): y = s * x;  y *= x;  y += x;  zero-fill / copy as stream-ordered memset / memcpy
__global__ void ew_kernel(const float* __restrict__ x, float* __restrict__ y, float s, long n, int op) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float a = x[i];
    y[i] = op == 0 ? s * a : op == 1 ? y[i] * a : y[i] + a;
  }
}
static int ew_launch(const float* x, float* y, float s, long n, int op, unflow_stream_t stream) {
  if (!x || !y) return UNFLOW_ERR_NULL;
  if (n <= 0) return UNFLOW_OK;
  ew_kernel<<<stream_grid(n), 256, 0, as_stream(stream)>>>(x, y, s, n, op);
  return launch_status();
}
UNFLOW_API int unflow_scale(const float* x, float s, float* y, long n, unflow_stream_t stream) { return ew_launch(x, y, s, n, 0, stream); }
UNFLOW_API int unflow_mul_inplace(float* y, const float* x, long n, unflow_stream_t stream) { return ew_launch(x, y, 1.f, n, 1, stream); }
UNFLOW_API int unflow_add_inplace(float* y, const float* x, long n, unflow_stream_t stream) { return ew_launch(x, y, 1.f, n, 2, stream); }
UNFLOW_API int unflow_zero(void* p, size_t bytes, unflow_stream_t stream) {
  if (!p) return UNFLOW_ERR_NULL;
  if (bytes == 0) return UNFLOW_OK;
  return hipMemsetAsync(p, 0, bytes, as_stream(stream)) == hipSuccess ? UNFLOW_OK : UNFLOW_ERR_LAUNCH;
}
UNFLOW_API int unflow_copy(void* dst, const void* src, size_t bytes, unflow_stream_t stream) {
  if (!dst || !src) return UNFLOW_ERR_NULL;
  if (bytes == 0) return UNFLOW_OK;
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)) == hipSuccess ? UNFLOW_OK : UNFLOW_ERR_LAUNCH;
}

// TF1 legacy bilinear: src = dst * (in/out); lo = floor(src); hi = min(lo+1, in-1); lerp x then y.
__global__ void resize_bilinear_tf1_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                                           int C, int OH, int OW, float sy, float sx, float scale) {
  const long n = (long)B * OH * OW * C;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int ox = (int)((e / C) % OW), oy = (int)((e / C / OW) % OH);
    const long b = e / ((long)C * OW * OH);
    const float fy = (float)oy * sy, fx = (float)ox * sx;
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* base = in + b * H * W * C + c;
    const float tl = base[((long)y0 * W + x0) * C], tr = base[((long)y0 * W + x1) * C];
    const float bl = base[((long)y1 * W + x0) * C], br = base[((long)y1 * W + x1) * C];
    const float top = tl + (tr - tl) * lx, bot = bl + (br - bl) * lx;
    out[e] = (top + (bot - top) * ly) * scale;
  }
}

UNFLOW_API int unflow_resize_bilinear_tf1(const float* in, float* out, int B, int H, int W, int C, int out_h,
                                          int out_w, float scale, unflow_stream_t stream) {
  if (!in || !out) return UNFLOW_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || out_h <= 0 || out_w <= 0) return UNFLOW_ERR_SHAPE;
  const long n = (long)B * out_h * out_w * C;
  resize_bilinear_tf1_kernel<<<stream_grid(n), 256, 0, as_stream(stream)>>>(
      in, out, B, H, W, C, out_h, out_w, (float)H / (float)out_h, (float)W / (float)out_w, scale);
  return launch_status();
}

typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* q) {
  const f32x4_nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(q));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void nt_store4(float4* q, float4 a) {
  f32x4_nt t; t.x = a.x; t.y = a.y; t.z = a.z; t.w = a.w;
  __builtin_nontemporal_store(t, reinterpret_cast<f32x4_nt*>(q));
}

// Fused Adam (TF form, epsilon outside the bias correction) + L2-regulariser gradient.
template <bool REG_LOSS>
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, long n_reg, float gscale, float l2, float lr_t, float b1,
                            float b2, float eps, float* __restrict__ loss_acc) {
  __shared__ float red[4];
  float sq = 0.f;   // REG_LOSS: sum of squares of the PRE-update regularised parameters (this step's L2 loss term)
  const long n4 = n >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    // gradient and moments are touched once per step (155 MB each): non-temporal, they leave the caches to the parameters,
    // which the weight-plane kernel of the next step reads again (tools/microbench/hbm_stream.hip: 4 read : 3 written
    // streams reach 5.2 TB/s plain, 5.7 TB/s with the hints)
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = nt_load4(reinterpret_cast<const float4*>(g) + i);
    float4 mm = nt_load4(reinterpret_cast<const float4*>(m) + i), vv = nt_load4(reinterpret_cast<const float4*>(v) + i);
    float* pa = &pp.x;
    const float* ga = &gg.x;
    float* ma = &mm.x;
    float* va = &vv.x;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const long idx = 4 * i + j;
      float gr = ga[j] * gscale;
      if (idx < n_reg) {
        gr += l2 * pa[j];
        if (REG_LOSS) sq += pa[j] * pa[j];
      }
      ma[j] = b1 * ma[j] + (1.f - b1) * gr;
      va[j] = b2 * va[j] + (1.f - b2) * (gr * gr);
      pa[j] = pa[j] - lr_t * ma[j] / (sqrtf(va[j]) + eps);
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    nt_store4(reinterpret_cast<float4*>(m) + i, mm);
    nt_store4(reinterpret_cast<float4*>(v) + i, vv);
  }
  // tail
  const long t0 = n4 << 2;
  for (long idx = t0 + blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    float gr = g[idx] * gscale;
    if (idx < n_reg) {
      gr += l2 * p[idx];
      if (REG_LOSS) sq += p[idx] * p[idx];
    }
    const float mn = b1 * m[idx] + (1.f - b1) * gr;
    const float vn = b2 * v[idx] + (1.f - b2) * (gr * gr);
    m[idx] = mn;
    v[idx] = vn;
    p[idx] = p[idx] - lr_t * mn / (sqrtf(vn) + eps);
  }
  if (REG_LOSS) {
    const float t = block_sum(sq, red);
    if (threadIdx.x == 0) atomicAdd(loss_acc, t * 0.5f * l2);
  }
}

UNFLOW_API int unflow_adam_step(float* p, const float* grad, float* m, float* v, long n, long n_regularized,
                                float grad_scale, float l2_scale, float lr_t, float beta1, float beta2, float eps,
                                unflow_stream_t stream) {
  if (!p || !grad || !m || !v) return UNFLOW_ERR_NULL;
  if (n <= 0) return UNFLOW_OK;
  adam_kernel<false><<<stream_grid(n / 4 + 1), 256, 0, as_stream(stream)>>>(p, grad, m, v, n, n_regularized, grad_scale,
                                                                             l2_scale, lr_t, beta1, beta2, eps, nullptr);
  return launch_status();
}

UNFLOW_API int unflow_adam_step_regloss(float* p, const float* grad, float* m, float* v, long n, long n_regularized,
                                        float grad_scale, float l2_scale, float lr_t, float beta1, float beta2,
                                        float eps, float* loss_acc, unflow_stream_t stream) {
  if (!p || !grad || !m || !v || !loss_acc) return UNFLOW_ERR_NULL;
  if (n <= 0) return UNFLOW_OK;
  adam_kernel<true><<<stream_grid(n / 4 + 1), 256, 0, as_stream(stream)>>>(p, grad, m, v, n, n_regularized, grad_scale,
                                                                            l2_scale, lr_t, beta1, beta2, eps, loss_acc);
  return launch_status();
}

// loss_acc[0] += scale * 0.5 * sum(p^2).  Single-block-ordered: per-block partials are combined by
// one atomicAdd per block (order varies run to run at the last bits; deterministic variant = 1 block).
__global__ void l2_loss_kernel(const float* __restrict__ p, long n, float scale, float* __restrict__ acc) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += p[i] * p[i];
  const float t = block_sum(s, red);
  if (threadIdx.x == 0) atomicAdd(acc, t * 0.5f * scale);
}

UNFLOW_API int unflow_l2_loss(const float* p, long n, float scale, float* loss_acc, unflow_stream_t stream) {
  if (!p || !loss_acc) return UNFLOW_ERR_NULL;
  if (n <= 0) return UNFLOW_OK;
  l2_loss_kernel<<<stream_grid(n), 256, 0, as_stream(stream)>>>(p, n, scale, loss_acc);
  return launch_status();
}

// Backward of stack_input_kernel wrt the coarse flow (train_all, flownet.py:51-54 without the stop_gradient): the flow
// reaches the stage input three times — as channels 6..7, through the warp (8..10) and through |warp - first| (11..13).
// Per full-resolution pixel the gradient of the upsampled flow is formed (image_warp's flow gradient, image_warp.py:26-73
// by the chain rule; d|x| = sign(x), 0 at 0) and scattered to the <= 4 coarse pixels of the TF1 bilinear upsample.
// d_prev is accumulated with float atomics (caller zeroes it): this optional path is not bit-reproducible.
__global__ void stack_input_bwd_kernel(const float* __restrict__ dout, int ldo, const float* __restrict__ im,
                                       const float* __restrict__ prev, float* __restrict__ d_prev, int shift, int N,
                                       int H, int W, int h, int w, float fscale) {
  const long npx = (long)N * H * W;
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npx; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const int n = (int)(i / ((long)W * H));
    const long sb = (long)((n + shift) % N) * H * W;
    const float4 a = reinterpret_cast<const float4*>(im)[i];
    const float fy = (float)y * sy, fx = (float)x * sx;
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float2* pf = reinterpret_cast<const float2*>(prev) + (long)n * h * w;
    const float2 tl = pf[(long)y0 * w + x0], tr = pf[(long)y0 * w + x1], bl = pf[(long)y1 * w + x0], br = pf[(long)y1 * w + x1];
    const float tu = tl.x + (tr.x - tl.x) * lx, bu = bl.x + (br.x - bl.x) * lx;
    const float tv = tl.y + (tr.y - tl.y) * lx, bv = bl.y + (br.y - bl.y) * lx;
    const float u = (tu + (bu - tu) * ly) * fscale, v = (tv + (bv - tv) * ly) * fscale;
    const float fu = floorf(u), fv = floorf(v);
    const float xw = u - fu, yw = v - fv;
    const float wa = (1.f - xw) * (1.f - yw), wb = (1.f - xw) * yw, wc = xw * (1.f - yw), wd = xw * yw;
    const int xi = x + (int)fu, yi = y + (int)fv;
    const int xa = min(max(xi, 0), W - 1), xb = min(max(xi + 1, 0), W - 1);
    const int ya = min(max(yi, 0), H - 1), yb = min(max(yi + 1, 0), H - 1);
    const float4 Ia = reinterpret_cast<const float4*>(im)[sb + (long)ya * W + xa];
    const float4 Ib = reinterpret_cast<const float4*>(im)[sb + (long)yb * W + xa];
    const float4 Ic = reinterpret_cast<const float4*>(im)[sb + (long)ya * W + xb];
    const float4 Id = reinterpret_cast<const float4*>(im)[sb + (long)yb * W + xb];
    const float* g = dout + i * ldo;
    const float ia[3] = {Ia.x, Ia.y, Ia.z}, ib[3] = {Ib.x, Ib.y, Ib.z}, ic[3] = {Ic.x, Ic.y, Ic.z}, id[3] = {Id.x, Id.y, Id.z};
    const float first[3] = {a.x, a.y, a.z};
    float du = g[6], dv = g[7];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float wv = ((wa * ia[c] + wb * ib[c]) + wc * ic[c]) + wd * id[c];
      const float df = wv - first[c];
      const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
      const float gc = g[8 + c] + sg * g[11 + c];
      du += gc * ((ic[c] - ia[c]) * (1.f - yw) + (id[c] - ib[c]) * yw);
      dv += gc * ((ib[c] - ia[c]) * (1.f - xw) + (id[c] - ic[c]) * xw);
    }
    du *= fscale;
    dv *= fscale;
    float* dp = d_prev + (long)n * h * w * 2;
    const float wtl = (1.f - lx) * (1.f - ly), wtr = lx * (1.f - ly), wbl = (1.f - lx) * ly, wbr = lx * ly;
    atomicAdd(dp + ((long)y0 * w + x0) * 2, du * wtl); atomicAdd(dp + ((long)y0 * w + x0) * 2 + 1, dv * wtl);
    atomicAdd(dp + ((long)y0 * w + x1) * 2, du * wtr); atomicAdd(dp + ((long)y0 * w + x1) * 2 + 1, dv * wtr);
    atomicAdd(dp + ((long)y1 * w + x0) * 2, du * wbl); atomicAdd(dp + ((long)y1 * w + x0) * 2 + 1, dv * wbl);
    atomicAdd(dp + ((long)y1 * w + x1) * 2, du * wbr); atomicAdd(dp + ((long)y1 * w + x1) * 2 + 1, dv * wbr);
  }
}

UNFLOW_API int unflow_stack_input_bwd(const float* d_out, int ld_out, const float* net_in4, const float* prev_flow2,
                                      float* d_prev_flow2, int pair_shift, int N, int H, int W, int h, int w,
                                      float flow_scale, unflow_stream_t stream) {
  if (!d_out || !net_in4 || !prev_flow2 || !d_prev_flow2) return UNFLOW_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0 || ld_out < 14) return UNFLOW_ERR_SHAPE;
  stack_input_bwd_kernel<<<stream_grid((long)N * H * W), 256, 0, as_stream(stream)>>>(
      d_out, ld_out, net_in4, prev_flow2, d_prev_flow2, pair_shift, N, H, W, h, w, flow_scale);
  return launch_status();
}

__global__ void flow_error_sums_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                       const float* __restrict__ mask, float* __restrict__ out2, long npix) {
  __shared__ float red[4];
  float s = 0.f, ms = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const float2 a = reinterpret_cast<const float2*>(f1)[i], b = reinterpret_cast<const float2*>(f2)[i];
    const float mk = mask ? mask[i] : 1.f;
    const float dx = a.x - b.x, dy = a.y - b.y;
    s += sqrtf(dx * dx + dy * dy) * mk;
    ms += mk;
  }
  const float t = block_sum(s, red);
  const float tm = block_sum(ms, red);
  if (threadIdx.x == 0) {
    atomicAdd(out2, t);
    atomicAdd(out2 + 1, tm);
  }
}

UNFLOW_API int unflow_flow_error_sums(const float* f1, const float* f2, const float* mask, float* out2, long npix,
                                      unflow_stream_t stream) {
  if (!f1 || !f2 || !out2) return UNFLOW_ERR_NULL;
  if (hipMemsetAsync(out2, 0, 2 * sizeof(float), as_stream(stream)) != hipSuccess) return UNFLOW_ERR_LAUNCH;
  if (npix <= 0) return UNFLOW_OK;
  flow_error_sums_kernel<<<stream_grid(npix), 256, 0, as_stream(stream)>>>(f1, f2, mask, out2, npix);
  return launch_status();
}

// Input of a FlowNetS stage (flownet.py:46-59): out[n] = [first(3), second(3)] for a first-stage S, or
// [first, second, flow(2), warp(3), |warp - first|(3)] for a refinement stage, where
// flow = resize_bilinear_tf1(prev_flow2[n], [H,W]) * 4 * FLOW_SCALE, warp = image_warp(second, flow).
// first = im[n], second = im[(n + shift) % N] (directed batch); im is the [N,H,W,4] mean-subtracted network input.
__global__ void stack_input_kernel(const float* __restrict__ im, const float* __restrict__ prev, float* __restrict__ out,
                                   int ldo, int shift, int N, int H, int W, int h, int w, float fscale) {
  const long npx = (long)N * H * W;
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npx; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const int n = (int)(i / ((long)W * H));
    const long sb = (long)((n + shift) % N) * H * W;
    const float4 a = reinterpret_cast<const float4*>(im)[i];
    const float4 b = reinterpret_cast<const float4*>(im)[sb + (long)y * W + x];
    float* o = out + i * ldo;
    o[0] = a.x; o[1] = a.y; o[2] = a.z;
    o[3] = b.x; o[4] = b.y; o[5] = b.z;
    if (!prev) continue;
    // TF1 legacy bilinear upsample of the coarse flow
    const float fy = (float)y * sy, fx = (float)x * sx;
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float2* pf = reinterpret_cast<const float2*>(prev) + (long)n * h * w;
    const float2 tl = pf[(long)y0 * w + x0], tr = pf[(long)y0 * w + x1], bl = pf[(long)y1 * w + x0], br = pf[(long)y1 * w + x1];
    const float tu = tl.x + (tr.x - tl.x) * lx, bu = bl.x + (br.x - bl.x) * lx;
    const float tv = tl.y + (tr.y - tl.y) * lx, bv = bl.y + (br.y - bl.y) * lx;
    const float u = (tu + (bu - tu) * ly) * fscale, v = (tv + (bv - tv) * ly) * fscale;
    o[6] = u; o[7] = v;
    // image_warp(second, flow): x + int(floor(u)), clamp, bilinear (image_warp.py:26-73)
    const float fu = floorf(u), fv = floorf(v);
    const float xw = u - fu, yw = v - fv;
    const float wa = (1.f - xw) * (1.f - yw), wb = (1.f - xw) * yw, wc = xw * (1.f - yw), wd = xw * yw;
    const int xi = x + (int)fu, yi = y + (int)fv;
    const int xa = min(max(xi, 0), W - 1), xb = min(max(xi + 1, 0), W - 1);
    const int ya = min(max(yi, 0), H - 1), yb = min(max(yi + 1, 0), H - 1);
    const float4 Ia = reinterpret_cast<const float4*>(im)[sb + (long)ya * W + xa];
    const float4 Ib = reinterpret_cast<const float4*>(im)[sb + (long)yb * W + xa];
    const float4 Ic = reinterpret_cast<const float4*>(im)[sb + (long)ya * W + xb];
    const float4 Id = reinterpret_cast<const float4*>(im)[sb + (long)yb * W + xb];
    const float w0 = ((wa * Ia.x + wb * Ib.x) + wc * Ic.x) + wd * Id.x;
    const float w1 = ((wa * Ia.y + wb * Ib.y) + wc * Ic.y) + wd * Id.y;
    const float w2 = ((wa * Ia.z + wb * Ib.z) + wc * Ic.z) + wd * Id.z;
    o[8] = w0; o[9] = w1; o[10] = w2;
    o[11] = fabsf(w0 - a.x); o[12] = fabsf(w1 - a.y); o[13] = fabsf(w2 - a.z);
  }
}

UNFLOW_API int unflow_stack_input(const float* net_in4, const float* prev_flow2, float* out, int ld_out, int pair_shift,
                                  int N, int H, int W, int h, int w, float flow_scale, unflow_stream_t stream) {
  if (!net_in4 || !out) return UNFLOW_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || ld_out < (prev_flow2 ? 14 : 6)) return UNFLOW_ERR_SHAPE;
  if (prev_flow2 && (h <= 0 || w <= 0)) return UNFLOW_ERR_SHAPE;
  stack_input_kernel<<<stream_grid((long)N * H * W), 256, 0, as_stream(stream)>>>(net_in4, prev_flow2, out, ld_out,
                                                                                  pair_shift, N, H, W, h, w, flow_scale);
  return launch_status();
}
