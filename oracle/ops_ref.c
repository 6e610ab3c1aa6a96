/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path
 * (unflow_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this.
 *
 * CPU restatement, in plain C, of the arithmetic of the reference's custom ops.
 * The reference registers GPU kernels only (REGISTER_KERNEL_BUILDER(...DEVICE_GPU)
 * in ops/correlation_op.cc:189-194, backward_warp_op.cc:93-98,
 * forward_warp_op.cc:102-107, downsample_op.cc:83-87) and needs nvcc plus the
 * TensorFlow headers, neither of which exists here, so it cannot be compiled;
 * this file states the same math per output element, keeping the reference's
 * fp32 operation ORDER wherever the order changes the rounding.  Every function
 * cites the lines whose behaviour it restates.
 *
 * Pinned by the reference's own known-answer tests (tests/golden/ref_kats.json,
 * from src/e2eflow/test/ops/{correlation,backward_warp,downsample}.py and
 * src/e2eflow/test/test_image_warp.py).  forward_warp VALUES are unpinned by
 * the reference (its test is a Jacobian check only).
 *
 * Build: oracle/Makefile  (gcc -O2 -fopenmp -shared -> oracle/_build/liboracle.so)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* floor(a/b), ceil(a/b) for b>0 and any sign of a.  The reference gets these
 * with a "+50000*stride" offset trick (correlation_op.cu.cc:19,133-142); for
 * every |a| < 50000*b the results are identical. */
static inline int floordiv(int a, int b) { int q = a / b; return (a % b != 0 && a < 0) ? q - 1 : q; }
static inline int ceildiv(int a, int b) { return -floordiv(-a, b); }

/* ------------------------------------------------------------------ */
/* Correlation geometry — ops/correlation_op.h:36-51                    */
/* ------------------------------------------------------------------ */
typedef struct {
  int k, kr, md, pad, s1, s2;
  int r, gw;          /* displacement grid radius / width            */
  int ph, pw;         /* padded input size                            */
  int oh, ow, oc;     /* output size                                  */
} corr_geom;

static corr_geom make_geom(int H, int W, int k, int md, int pad, int s1, int s2) {
  corr_geom g;
  g.k = k; g.kr = (k - 1) / 2; g.md = md; g.pad = pad; g.s1 = s1; g.s2 = s2;
  g.ph = H + 2 * pad; g.pw = W + 2 * pad;
  int border = md + g.kr;
  g.r = md / s2; g.gw = 2 * g.r + 1;
  g.ow = (int)ceilf((float)(g.pw - 2 * border) / (float)s1);
  g.oh = (int)ceilf((float)(g.ph - 2 * border) / (float)s1);
  g.oc = g.gw * g.gw;
  return g;
}

/* Status codes shared with include/unflow_hip.h */
#define ST_OK 0
#define ST_EMPTY_OUTPUT (-2)  /* correlation_op.cc:60-61 "Invalid correlation settings" */
#define ST_EVEN_KERNEL (-3)   /* correlation_op.h:16-17  "kernel_size must be odd"      */
#define ST_NOT_DIVISIBLE (-4) /* downsample_op.cc:37-40                                   */

int ref_correlation_out_shape(int H, int W, int k, int md, int pad, int s1, int s2, int* out3) {
  if (k % 2 == 0) return ST_EVEN_KERNEL;
  corr_geom g = make_geom(H, W, k, md, pad, s1, s2);
  out3[0] = g.oc; out3[1] = g.oh; out3[2] = g.ow;
  /* the reference tests the PRODUCT (correlation_op.cc:60), which lets two negative sizes through to a failing
   * TF allocation; both sizes must be positive here */
  return (g.ow > 0 && g.oh > 0) ? ST_OK : ST_EMPTY_OUTPUT;
}

/* Zero-padded channels-last copy of an NCHW tensor: what the two memsets and
 * blob_rearrange_kernel2 produce (correlation_op.cu.cc:30-49,282-293). */
static float* padded_nhwc(const float* nchw, int B, int C, int H, int W, int pad) {
  int ph = H + 2 * pad, pw = W + 2 * pad;
  size_t n = (size_t)B * ph * pw * C;
  float* p = (float*)calloc(n ? n : 1, sizeof(float));
  for (int b = 0; b < B; b++)
    for (int c = 0; c < C; c++)
      for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
          p[(((size_t)b * ph + y + pad) * pw + x + pad) * C + c] =
              nchw[(((size_t)b * C + c) * H + y) * W + x];
  return p;
}

/* Forward — CorrelateData, correlation_op.cu.cc:51-117.
 * Rounding order kept: 32 lanes each own the channels {lane, lane+32, ...}
 * and run over (j, i, channel) accumulating in fp32 (:93-101); lane 0 then
 * adds the 32 partials in lane order (:107-111) and divides by k*k*C (:112-114).
 * Output channel index = (p+r)*gw + (o+r), o = x-displacement (:87-88). */
int ref_correlation_fwd(const float* in0, const float* in1, float* out, int B, int C,
                        int H, int W, int k, int md, int pad, int s1, int s2) {
  if (k % 2 == 0) return ST_EVEN_KERNEL;
  corr_geom g = make_geom(H, W, k, md, pad, s1, s2);
  if (g.ow <= 0 || g.oh <= 0) return ST_EMPTY_OUTPUT;
  float* P0 = padded_nhwc(in0, B, C, H, W, pad);
  float* P1 = padded_nhwc(in1, B, C, H, W, pad);
  const float denom = (float)(k * k * C);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; b++)
    for (int oy = 0; oy < g.oh; oy++)
      for (int ox = 0; ox < g.ow; ox++) {
        const int ay = oy * s1 + md, ax = ox * s1 + md; /* patch corner in P0 */
        for (int ch = 0; ch < g.oc; ch++) {
          const int dx = (ch % g.gw - g.r) * s2;
          const int dy = (ch / g.gw - g.r) * s2;
          float lane_sum[32];
          for (int lane = 0; lane < 32; lane++) {
            float s = 0.f;
            for (int j = 0; j < k; j++)
              for (int i = 0; i < k; i++) {
                const float* a = P0 + (((size_t)b * g.ph + ay + j) * g.pw + ax + i) * C;
                const float* c1 = P1 + (((size_t)b * g.ph + ay + dy + j) * g.pw + ax + dx + i) * C;
                for (int c = lane; c < C; c += 32) s += a[c] * c1[c];
              }
            lane_sum[lane] = s;
          }
          float total = 0.f;
          for (int lane = 0; lane < 32; lane++) total += lane_sum[lane];
          out[(((size_t)b * g.oc + ch) * g.oh + oy) * g.ow + ox] = total / denom;
        }
      }
  free(P0); free(P1);
  return ST_OK;
}

/* Backward — CorrelateDataBackward0 / Backward1, correlation_op.cu.cc:119-248.
 * For input position (y,x) [padded coords m=y+pad, l=x+pad] the output
 * positions whose k x k patch covers it are
 *     ceil((l - 2kr - md - sx)/s1) .. floor((l - md - sx)/s1)      (:133-142, :211-216)
 * with sx = 0 for grad0 and sx = +s2*o for grad1 (clipped to the output).
 * grad0 multiplies dOut by P1 at (m+s2p, l+s2o) (:154-155); grad1 by P0 at
 * (m-s2p, l-s2o) (:227-228).  Summation order: p, o, then y, x (:147-168).
 * grad0 skips everything when its (displacement-independent) window is empty
 * (:145); grad1 tests per displacement (:218). Result / (k*k*C) (:175-178). */
int ref_correlation_bwd(const float* dout, const float* in0, const float* in1,
                        float* g0, float* g1, int B, int C, int H, int W, int k,
                        int md, int pad, int s1, int s2) {
  if (k % 2 == 0) return ST_EVEN_KERNEL;
  corr_geom g = make_geom(H, W, k, md, pad, s1, s2);
  if (g.ow <= 0 || g.oh <= 0) return ST_EMPTY_OUTPUT;
  float* P0 = padded_nhwc(in0, B, C, H, W, pad);
  float* P1 = padded_nhwc(in1, B, C, H, W, pad);
  const float denom = (float)((2 * g.kr + 1) * (2 * g.kr + 1) * C);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; b++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        const int m = y + pad, l = x + pad;
        for (int c = 0; c < C; c++) {
          /* ---------------- grad wrt first input ---------------- */
          float acc0 = 0.f;
          {
            int x_lo = ceildiv(l - 2 * g.kr - md, s1), x_hi = floordiv(l - md, s1);
            int y_lo = ceildiv(m - 2 * g.kr - md, s1), y_hi = floordiv(m - md, s1);
            if (x_hi >= 0 && y_hi >= 0 && x_lo <= g.ow - 1 && y_lo <= g.oh - 1) {
              x_lo = x_lo < 0 ? 0 : x_lo; x_hi = x_hi > g.ow - 1 ? g.ow - 1 : x_hi;
              y_lo = y_lo < 0 ? 0 : y_lo; y_hi = y_hi > g.oh - 1 ? g.oh - 1 : y_hi;
              for (int p = -g.r; p <= g.r; p++)
                for (int o = -g.r; o <= g.r; o++) {
                  float v1 = P1[(((size_t)b * g.ph + m + s2 * p) * g.pw + l + s2 * o) * C + c];
                  size_t chan = (size_t)b * g.oc + (p + g.r) * g.gw + (o + g.r);
                  for (int yy = y_lo; yy <= y_hi; yy++)
                    for (int xx = x_lo; xx <= x_hi; xx++)
                      acc0 += dout[(chan * g.oh + yy) * g.ow + xx] * v1;
                }
            }
          }
          g0[(((size_t)b * C + c) * H + y) * W + x] = acc0 / denom;
          /* ---------------- grad wrt second input ---------------- */
          float acc1 = 0.f;
          for (int p = -g.r; p <= g.r; p++)
            for (int o = -g.r; o <= g.r; o++) {
              const int sx = s2 * o, sy = s2 * p;
              int x_lo = ceildiv(l - 2 * g.kr - md - sx, s1), x_hi = floordiv(l - md - sx, s1);
              int y_lo = ceildiv(m - 2 * g.kr - md - sy, s1), y_hi = floordiv(m - md - sy, s1);
              if (x_hi >= 0 && y_hi >= 0 && x_lo <= g.ow - 1 && y_lo <= g.oh - 1) {
                x_lo = x_lo < 0 ? 0 : x_lo; x_hi = x_hi > g.ow - 1 ? g.ow - 1 : x_hi;
                y_lo = y_lo < 0 ? 0 : y_lo; y_hi = y_hi > g.oh - 1 ? g.oh - 1 : y_hi;
                float v0 = P0[(((size_t)b * g.ph + m - sy) * g.pw + l - sx) * C + c];
                size_t chan = (size_t)b * g.oc + (p + g.r) * g.gw + (o + g.r);
                for (int yy = y_lo; yy <= y_hi; yy++)
                  for (int xx = x_lo; xx <= x_hi; xx++)
                    acc1 += dout[(chan * g.oh + yy) * g.ow + xx] * v0;
              }
            }
          g1[(((size_t)b * C + c) * H + y) * W + x] = acc1 / denom;
        }
      }
  free(P0); free(P1);
  return ST_OK;
}

/* ------------------------------------------------------------------ */
/* backward_warp — ops/backward_warp_op.cu.cc:14-68 (fwd), :70-138 (grad)
 * NHWC image, flow [B,H,W,2] with u (x) first (:26-28).  The sample position
 * is formed in fp32 as float(x)+u and THEN floored (:27-31) — unlike
 * image_warp.  Out-of-image taps are dropped (zero padding, :46-63).
 * Tap order TL, TR, BL, BR; weights (w_left*w_top) etc. as products of the
 * 1-D weights (:33-36).                                                 */
/* ------------------------------------------------------------------ */
typedef struct { int x0, y0; float wl, wr, wt, wb; } bw_taps;

static inline bw_taps bw_sample(int px, int py, float u, float v) {
  bw_taps t;
  const float sx = px + u, sy = py + v;
  t.x0 = (int)floorf(sx); t.y0 = (int)floorf(sy);
  t.wr = sx - t.x0; t.wl = (t.x0 + 1) - sx;
  t.wb = sy - t.y0; t.wt = (t.y0 + 1) - sy;
  return t;
}
static inline int inside(int x, int y, int W, int H) { return x >= 0 && x < W && y >= 0 && y < H; }

void ref_backward_warp_fwd(const float* img, const float* flow, float* out, int B, int H, int W, int C) {
  const int npx = B * H * W;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < npx; i++) {
    const int px = i % W, py = (i / W) % H, b = i / (W * H);
    const bw_taps t = bw_sample(px, py, flow[2 * i], flow[2 * i + 1]);
    const float* base = img + (size_t)b * H * W * C;
    for (int c = 0; c < C; c++) {
      float s = 0.f;
      if (inside(t.x0, t.y0, W, H)) s += t.wl * t.wt * base[((size_t)t.y0 * W + t.x0) * C + c];
      if (inside(t.x0 + 1, t.y0, W, H)) s += t.wr * t.wt * base[((size_t)t.y0 * W + t.x0 + 1) * C + c];
      if (inside(t.x0, t.y0 + 1, W, H)) s += t.wl * t.wb * base[((size_t)(t.y0 + 1) * W + t.x0) * C + c];
      if (inside(t.x0 + 1, t.y0 + 1, W, H)) s += t.wr * t.wb * base[((size_t)(t.y0 + 1) * W + t.x0 + 1) * C + c];
      out[(size_t)i * C + c] = s;
    }
  }
}

/* (x0,y0) of every pixel: the integer part that must match bit-exactly. */
void ref_backward_warp_indices(const float* flow, int* xy0, int B, int H, int W) {
  const int npx = B * H * W;
  for (int i = 0; i < npx; i++) {
    const bw_taps t = bw_sample(i % W, (i / W) % H, flow[2 * i], flow[2 * i + 1]);
    xy0[2 * i] = t.x0; xy0[2 * i + 1] = t.y0;
  }
}

/* Gradient wrt flow only (ops.py:80-84 returns [None, grad]).  Per channel and
 * per in-bounds tap, px = I*din, then du -/+= w_y*px and dv -/+= w_x*px in the
 * order TL,TR,BL,BR (:107-132). */
void ref_backward_warp_bwd(const float* dout, const float* img, const float* flow,
                           float* dflow, int B, int H, int W, int C) {
  const int npx = B * H * W;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < npx; i++) {
    const int px = i % W, py = (i / W) % H, b = i / (W * H);
    const bw_taps t = bw_sample(px, py, flow[2 * i], flow[2 * i + 1]);
    const float* base = img + (size_t)b * H * W * C;
    float du = 0.f, dv = 0.f;
    for (int c = 0; c < C; c++) {
      const float din = dout[(size_t)i * C + c];
      float q;
      if (inside(t.x0, t.y0, W, H)) { q = base[((size_t)t.y0 * W + t.x0) * C + c] * din; du -= t.wt * q; dv -= t.wl * q; }
      if (inside(t.x0 + 1, t.y0, W, H)) { q = base[((size_t)t.y0 * W + t.x0 + 1) * C + c] * din; du += t.wt * q; dv -= t.wr * q; }
      if (inside(t.x0, t.y0 + 1, W, H)) { q = base[((size_t)(t.y0 + 1) * W + t.x0) * C + c] * din; du -= t.wb * q; dv += t.wl * q; }
      if (inside(t.x0 + 1, t.y0 + 1, W, H)) { q = base[((size_t)(t.y0 + 1) * W + t.x0 + 1) * C + c] * din; du += t.wb * q; dv += t.wr * q; }
    }
    dflow[2 * i] = du; dflow[2 * i + 1] = dv;
  }
}

/* ------------------------------------------------------------------ */
/* forward_warp — ops/forward_warp_op.cu.cc:16-65 (fwd), :67-125 (grad)
 * Each pixel splats exp(-(dx^2+dy^2)/2) onto the integer sites within +-4
 * of its target (k = ceil(2+2) = 4, gauss_divisor = 2*1^2, :38-40,53).
 * Footprint: reject when entirely outside (:46-47); else
 *   min = t-k > 0 ? floor(t-k) : 0 ; max = t+k < size ? floor(t+k) : size-1 (:48-51)
 * The CUDA kernel adds with float atomics (:59) -> summation order there is
 * nondeterministic; here: pixel order, x outer / y inner like the kernel. */
/* ------------------------------------------------------------------ */
typedef struct { int ok, x_lo, x_hi, y_lo, y_hi; float tx, ty; } fw_foot;

static inline fw_foot fw_footprint(int px, int py, float u, float v, int W, int H) {
  fw_foot f;
  const int k = (int)ceilf(2.0f + 2);
  f.tx = px + u; f.ty = py + v;
  f.ok = floorf(f.tx - k) < W && floorf(f.tx + k) >= 0 && floorf(f.ty - k) < H && floorf(f.ty + k) >= 0;
  f.x_lo = f.tx - k > 0 ? (int)floorf(f.tx - k) : 0;
  f.y_lo = f.ty - k > 0 ? (int)floorf(f.ty - k) : 0;
  f.x_hi = f.tx + k < W ? (int)floorf(f.tx + k) : W - 1;
  f.y_hi = f.ty + k < H ? (int)floorf(f.ty + k) : H - 1;
  return f;
}

void ref_forward_warp_fwd(const float* flow, float* out, int B, int H, int W) {
  const int npx = B * H * W;
  memset(out, 0, sizeof(float) * (size_t)npx);
  const float gd = 2 * powf(1.0f, 2);
  for (int i = 0; i < npx; i++) {
    const int b = i / (W * H);
    const fw_foot f = fw_footprint(i % W, (i / W) % H, flow[2 * i], flow[2 * i + 1], W, H);
    if (!f.ok) continue;
    for (int nx = f.x_lo; nx <= f.x_hi; nx++)
      for (int ny = f.y_lo; ny <= f.y_hi; ny++) {
        const float dx = nx - f.tx, dy = ny - f.ty;
        out[((size_t)b * H + ny) * W + nx] += expf(-(powf(dx, 2) + powf(dy, 2)) / gd);
      }
  }
}

/* ranges[4i..] = {x_lo,x_hi,y_lo,y_hi} or all -1 when rejected. */
void ref_forward_warp_ranges(const float* flow, int* ranges, int B, int H, int W) {
  const int npx = B * H * W;
  for (int i = 0; i < npx; i++) {
    const fw_foot f = fw_footprint(i % W, (i / W) % H, flow[2 * i], flow[2 * i + 1], W, H);
    int* r = ranges + 4 * (size_t)i;
    if (f.ok) { r[0] = f.x_lo; r[1] = f.x_hi; r[2] = f.y_lo; r[3] = f.y_hi; }
    else r[0] = r[1] = r[2] = r[3] = -1;
  }
}

/* Gather-form gradient: factor = 2*din*w/gauss_divisor; du += factor*dx (:108-119). */
void ref_forward_warp_bwd(const float* dout, const float* flow, float* dflow, int B, int H, int W) {
  const int npx = B * H * W;
  const float gd = 2 * powf(1.0f, 2);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < npx; i++) {
    const int b = i / (W * H);
    const fw_foot f = fw_footprint(i % W, (i / W) % H, flow[2 * i], flow[2 * i + 1], W, H);
    float du = 0.f, dv = 0.f;
    if (f.ok)
      for (int nx = f.x_lo; nx <= f.x_hi; nx++)
        for (int ny = f.y_lo; ny <= f.y_hi; ny++) {
          const float dx = nx - f.tx, dy = ny - f.ty;
          const float w = expf(-(powf(dx, 2) + powf(dy, 2)) / gd);
          const float factor = 2 * dout[((size_t)b * H + ny) * W + nx] * w / gd;
          du += factor * dx; dv += factor * dy;
        }
    dflow[2 * i] = du; dflow[2 * i + 1] = dv;
  }
}

/* ------------------------------------------------------------------ */
/* downsample — ops/downsample_op.cu.cc:15-49: box mean, rows outer /
 * columns inner (:40-44), one division by scale^2 (:46); divisibility
 * check from downsample_op.cc:37-40.                                   */
/* ------------------------------------------------------------------ */
int ref_downsample(const float* img, float* out, int B, int H, int W, int C, int scale) {
  if (scale <= 0 || H % scale || W % scale) return ST_NOT_DIVISIBLE;
  const int oh = H / scale, ow = W / scale;
  const long n = (long)B * oh * ow * C;
#pragma omp parallel for schedule(static)
  for (long e = 0; e < n; e++) {
    const int c = (int)(e % C);
    const int ox = (int)((e / C) % ow), oy = (int)((e / C / ow) % oh), b = (int)(e / C / ow / oh);
    float s = 0.f;
    for (int yy = oy * scale; yy < (oy + 1) * scale; yy++)
      for (int xx = ox * scale; xx < (ox + 1) * scale; xx++)
        s += img[(((size_t)b * H + yy) * W + xx) * C + c];
    out[e] = s / (float)(scale * scale);
  }
  return ST_OK;
}

/* ------------------------------------------------------------------ */
/* image_warp — src/e2eflow/core/image_warp.py:4-76 (pure-TF graph).
 *   q = floor(flow) as int, added to the integer grid (:26,46-49);
 *   fractional part -> weights wa=(1-xw)(1-yw), wb=(1-xw)yw, wc=xw(1-yw),
 *   wd=xw*yw (:27,41-44); indices clamped to the image (:51-54);
 *   a=(y0,x0) b=(y1,x0) c=(y0,x1) d=(y1,x1) (:61-66); add_n a,b,c,d (:73). */
/* ------------------------------------------------------------------ */
typedef struct { size_t ia, ib, ic, id; float xw, yw, wa, wb, wc, wd; } iw_taps;

static inline iw_taps iw_sample(int i, const float* flow, int H, int W) {
  iw_taps t;
  const int px = i % W, py = (i / W) % H, b = i / (W * H);
  const float u = flow[2 * i], v = flow[2 * i + 1];
  const float fu = floorf(u), fv = floorf(v);
  t.xw = u - fu; t.yw = v - fv;
  t.wa = (1 - t.xw) * (1 - t.yw); t.wb = (1 - t.xw) * t.yw;
  t.wc = t.xw * (1 - t.yw);       t.wd = t.xw * t.yw;
  const int x0 = clampi(px + (int)fu, 0, W - 1), x1 = clampi(px + (int)fu + 1, 0, W - 1);
  const int y0 = clampi(py + (int)fv, 0, H - 1), y1 = clampi(py + (int)fv + 1, 0, H - 1);
  const size_t base = (size_t)b * W * H;
  t.ia = base + (size_t)y0 * W + x0; t.ib = base + (size_t)y1 * W + x0;
  t.ic = base + (size_t)y0 * W + x1; t.id = base + (size_t)y1 * W + x1;
  return t;
}

/* idx4 (may be NULL) receives the 4 flat gather indices a,b,c,d per pixel. */
void ref_image_warp_fwd(const float* im, const float* flow, float* warped, int* idx4,
                        int B, int H, int W, int C) {
  const int npx = B * H * W;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < npx; i++) {
    const iw_taps t = iw_sample(i, flow, H, W);
    if (idx4) { idx4[4 * (size_t)i] = (int)t.ia; idx4[4 * (size_t)i + 1] = (int)t.ib;
                idx4[4 * (size_t)i + 2] = (int)t.ic; idx4[4 * (size_t)i + 3] = (int)t.id; }
    for (int c = 0; c < C; c++)
      warped[(size_t)i * C + c] = ((t.wa * im[t.ia * C + c] + t.wb * im[t.ib * C + c]) +
                                   t.wc * im[t.ic * C + c]) + t.wd * im[t.id * C + c];
  }
}

/* What TF autodiff produces for that graph: the gathers' gradient is a
 * scatter-add into im (clamped duplicates accumulate) and the flow gradient
 * comes through the four weights; floor() contributes none.  d_im may be NULL. */
void ref_image_warp_bwd(const float* dwarp, const float* im, const float* flow,
                        float* d_im, float* d_flow, int B, int H, int W, int C) {
  const int npx = B * H * W;
  if (d_im) memset(d_im, 0, sizeof(float) * (size_t)npx * C);
  for (int i = 0; i < npx; i++) {
    const iw_taps t = iw_sample(i, flow, H, W);
    float ga = 0, gb = 0, gc = 0, gdd = 0;
    for (int c = 0; c < C; c++) {
      const float g = dwarp[(size_t)i * C + c];
      ga += g * im[t.ia * C + c]; gb += g * im[t.ib * C + c];
      gc += g * im[t.ic * C + c]; gdd += g * im[t.id * C + c];
      if (d_im) {
        d_im[t.ia * C + c] += t.wa * g; d_im[t.ib * C + c] += t.wb * g;
        d_im[t.ic * C + c] += t.wc * g; d_im[t.id * C + c] += t.wd * g;
      }
    }
    d_flow[2 * i] = (gc - ga) * (1 - t.yw) + (gdd - gb) * t.yw;
    d_flow[2 * i + 1] = (gb - ga) * (1 - t.xw) + (gdd - gc) * t.xw;
  }
}
