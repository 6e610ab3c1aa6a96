// GPU augmentation for gfx950: the spatial-transformer resampling behind random_affine and the
// per-sample photometric transform.  HBM-bound gathers / streams, one thread per output pixel.
//
//   stn_affine      <- src/e2eflow/core/spatial_transformer.py:56-175 (as called by core/augment.py:50-55)
//   photometric     <- src/e2eflow/core/augment.py:78-108, then the mean subtraction of core/unsupervised.py:67-68
//
// The random draws themselves (augment.py:23-29,37-39,78-91) are host-side in unflow_amd/core/augment.py: TF's
// RNG streams cannot be reproduced, the deterministic transforms given the draws are what parity covers.
#include "common.h"

template <int CT>
__global__ void stn_affine_kernel(const float* __restrict__ U, int n_u, int ld_u, const float* __restrict__ theta,
                                  int n_theta, float* __restrict__ out, int ld_out, int B, int H, int W, int C,
                                  int Ho, int Wo) {
  const unsigned npx = (unsigned)B * Ho * Wo;
  // tf.linspace(-1, 1, n): start + i * ((stop - start) / (n - 1))   (spatial_transformer.py:129-133)
  const float step_x = (1.0f - -1.0f) / (float)(Wo - 1), step_y = (1.0f - -1.0f) / (float)(Ho - 1);
  const float Wf = (float)W, Hf = (float)H;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    const Pix pp = decode_pix(i, Wo, Ho);
    const float* t = theta + (size_t)(pp.n % n_theta) * 6;
    const float xt = -1.0f + (float)pp.x * step_x, yt = -1.0f + (float)pp.y * step_y;
    const float xs = t[0] * xt + t[1] * yt + t[2];          // theta @ (x_t, y_t, 1)  (:160-163)
    const float ys = t[3] * xt + t[4] * yt + t[5];
    const float x = (xs + 1.0f) * Wf / 2.0f, y = (ys + 1.0f) * Hf / 2.0f;   // (:75-76)
    const float fx = fminf(fmaxf(floorf(x), -4.0f), Wf + 4.0f), fy = fminf(fmaxf(floorf(y), -4.0f), Hf + 4.0f);
    // indices are clipped BEFORE the weights are formed (:84-87,113-120): border samples are not convex combinations
    const int x0 = min(max((int)fx, 0), W - 1), x1 = min(max((int)fx + 1, 0), W - 1);
    const int y0 = min(max((int)fy, 0), H - 1), y1 = min(max((int)fy + 1, 0), H - 1);
    const float x0f = (float)x0, x1f = (float)x1, y0f = (float)y0, y1f = (float)y1;
    const float wa = (x1f - x) * (y1f - y), wb = (x1f - x) * (y - y0f);
    const float wc = (x - x0f) * (y1f - y), wd = (x - x0f) * (y - y0f);
    const float* base = U + (size_t)(pp.n % n_u) * H * W * ld_u;
    const float* pa = base + (size_t)(y0 * W + x0) * ld_u;
    const float* pb = base + (size_t)(y1 * W + x0) * ld_u;
    const float* pc = base + (size_t)(y0 * W + x1) * ld_u;
    const float* pd = base + (size_t)(y1 * W + x1) * ld_u;
    float* o = out + (size_t)i * ld_out;
    const int CC = CT ? CT : C;
#pragma unroll 4
    for (int c = 0; c < CC; c++) o[c] = ((wa * pa[c] + wb * pb[c]) + wc * pc[c]) + wd * pd[c];   // add_n order (:121)
  }
}

UNFLOW_API int unflow_stn_affine_fwd(const float* U, int n_u, int ld_u, const float* theta, int n_theta, float* out,
                                     int ld_out, int B, int H, int W, int C, int out_h, int out_w,
                                     unflow_stream_t stream) {
  if (!U || !theta || !out) return UNFLOW_ERR_NULL;
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || out_h <= 1 || out_w <= 1 || n_u <= 0 || n_theta <= 0 || ld_u < C ||
      ld_out < C)
    return UNFLOW_ERR_SHAPE;
  const long npx = (long)B * out_h * out_w;
  if (npx == 0) return UNFLOW_OK;
  if (npx > 0x7fffffffL) return UNFLOW_ERR_UNSUPPORTED;
  const int grid = stream_grid(npx);
  if (C == 1 && ld_u == 1)
    stn_affine_kernel<1><<<grid, 256, 0, as_stream(stream)>>>(U, n_u, 1, theta, n_theta, out, ld_out, B, H, W, C, out_h, out_w);
  else if (C == 3 && ld_u == 3)
    stn_affine_kernel<3><<<grid, 256, 0, as_stream(stream)>>>(U, n_u, 3, theta, n_theta, out, ld_out, B, H, W, C, out_h, out_w);
  else
    stn_affine_kernel<0><<<grid, 256, 0, as_stream(stream)>>>(U, n_u, ld_u, theta, n_theta, out, ld_out, B, H, W, C, out_h, out_w);
  return launch_status();
}

__global__ void photometric_augment_kernel(const float* __restrict__ im, int ld_in, float* __restrict__ out,
                                           int ld_out, const float* __restrict__ contrast,
                                           const float* __restrict__ brightness, const float* __restrict__ colour,
                                           const float* __restrict__ gamma, const float* __restrict__ noise, int n_par,
                                           float m0, float m1, float m2, int N, int HW) {
  const unsigned npx = (unsigned)N * HW;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    const int s = (int)(i / (unsigned)HW) % n_par;
    const float c1 = contrast[s] + 1.0f, br = brightness[s], ginv = 1.0f / gamma[s], nz = noise[s];
    const float* p = im + (size_t)i * ld_in;
    float* o = out + (size_t)i * ld_out;
    const float mean[3] = {m0, m1, m2};
#pragma unroll
    for (int c = 0; c < 3; c++) {
      float v = (p[c] * c1 + br) * colour[3 * s + c];     // augment.py:97
      v = fmaxf(0.0f, fminf(1.0f, v));                    // :98
      v = powf(v, ginv);                                  // :99 (accurate pow: one pass per step, not hot)
      o[c] = (v + nz) - mean[c];                          // :101, unsupervised.py:67-68
    }
    for (int c = 3; c < ld_out; c++) o[c] = 0.f;
  }
}

UNFLOW_API int unflow_photometric_augment(const float* im, int ld_in, float* out, int ld_out, const float* contrast,
                                          const float* brightness, const float* colour3, const float* gamma,
                                          const float* noise, int n_par, const float* mean3, int N, int H, int W,
                                          unflow_stream_t stream) {
  if (!im || !out || !contrast || !brightness || !colour3 || !gamma || !noise) return UNFLOW_ERR_NULL;
  if (N < 0 || H < 0 || W < 0 || ld_in < 3 || ld_out < 3 || n_par <= 0) return UNFLOW_ERR_SHAPE;
  const long npx = (long)N * H * W;
  if (npx == 0) return UNFLOW_OK;
  if (npx > 0x7fffffffL) return UNFLOW_ERR_UNSUPPORTED;
  // mean3: HOST pointer to channel means in [0,255] (core/input.py:45) or NULL for "no mean subtraction"
  const float m0 = mean3 ? mean3[0] / 255.0f : 0.f, m1 = mean3 ? mean3[1] / 255.0f : 0.f,
              m2 = mean3 ? mean3[2] / 255.0f : 0.f;
  photometric_augment_kernel<<<stream_grid(npx), 256, 0, as_stream(stream)>>>(
      im, ld_in, out, ld_out, contrast, brightness, colour3, gamma, noise, n_par, m0, m1, m2, N, H * W);
  return launch_status();
}
