// FlowNetC cost volume on the fp32 matrix cores (kernel_size == 1, stride_1 == 1 — flownet.py:221-222).
//
// For one output row oy, displacement row p and x-residue class q (x' = x + stride_2*o stays in the
// class of x), the 21 correlations of every pixel are a BAND of the 32x32 Gram matrix
//     G[i][j] = sum_c f0[y, x_i, c] * f1[y + s2*p, x_j, c],       out[x_i, (p,o)] = G[i][i+o] / C.
// forward : one wave owns (sample, row, class, 32-site tile); its f0 fragment (C/2 fp32 per lane) stays
//           in registers for all its displacement rows; f1 fragments stream from L2/HBM as 16-byte
//           loads (the K order inside the MFMA is permuted so that a lane's float4 = 4 k-steps);
//           128 x v_mfma_f32_32x32x2_f32 per Gram, band extracted straight from the accumulator.
// backward: out[site][c] += Band(dOut)[site][site'] * Src[site'][c] — the same banded product with the
//           band as the A operand (gathered from dOut, 16 dwords per lane) and the feature row as the
//           B operand (coalesced along c).  Both roles of a sample (first / second input) accumulate
//           into the same accumulators when the two inputs are one tensor (the training step), so
//           the gradient is written once, without atomics.
// No LDS, no barriers; zero padding is a bounds check (the reference materialises two padded copies).
// Work decomposition: sample index fastest in the block id, so block b runs on XCD b % 8 = its
// sample when B == 8 and the rows of one sample stay in one XCD's L2.
#include "igemm_shared.h"
#include "options.h"
#include "correlation_geom.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

struct CorrMfmaParams {
  const float* in0;
  const float* in1;
  float* out;
  const float* dout;
  float* g0;
  float* g1;
  int ld_in, ld_out, ld_dout, ld_g;
  int shift, fuse;
  int B, C, H, W;
  int oh, ow, r, gw, s2;
  int off;  // input coordinate = output coordinate + off (= max_displacement - pad)
  int nA;   // 32-site tiles per residue class
  int T;    // neighbour tiles on each side that the band can reach: ceil(r / 32)
};

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int C>
__global__ __launch_bounds__(256) void corr_fwd_mfma_kernel(const CorrMfmaParams p) {
  const int lane = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  int b = blockIdx.x;
  const int n = b % p.B; b /= p.B;
  const int ia = b % p.nA; b /= p.nA;
  const int q = b % p.s2; b /= p.s2;
  const int oy = b;
  const int i0 = ia * 32;
  const int n1 = (n + p.shift) % p.B;
  const int y0 = oy + p.off;

  // f0 fragment source (clamped; masked by a select): it is RE-READ from L1/L2 in NCH chunks per Gram instead of
  // living in C/2 registers for the whole kernel — 244 VGPRs allowed only 2 waves per SIMD, i.e. 1.5 rounds for the
  // 3072 waves of the FlowNetC shape.
  // C == 0: channel count at run time (any multiple of 32), chunks of 32 channels
  constexpr int NCH_C = C >= 128 ? 2 : 1;      // channel chunks per Gram
  constexpr int CQ = C == 0 ? 4 : C / 8 / NCH_C;   // float4 loads per chunk
  const int NCH = C == 0 ? p.C / 32 : NCH_C;
  const float* asrc;
  bool aok;
  {
    const int ox = q + p.s2 * (i0 + l31), x0 = ox + p.off;
    aok = ox < p.ow && (unsigned)x0 < (unsigned)p.W && (unsigned)y0 < (unsigned)p.H;
    asrc = p.in0 + (((size_t)n * p.H + (aok ? y0 : 0)) * p.W + (aok ? x0 : 0)) * p.ld_in + 4 * h;
  }
  const float cf = (float)(C == 0 ? p.C : C);
  const int per = (p.gw + 3) >> 2;
  // the short last share (gw = 21: 6,6,6,3) rotates over the waves — wave w of every block sits on SIMD w, a fixed
  // assignment would leave one SIMD of each CU with half the work
  const int share = (pg + (int)blockIdx.x) & 3;
  const int pa = share * per, pb = min(p.gw, pa + per);
  for (int pi = pa; pi < pb; pi++) {
    const int y2 = y0 + p.s2 * (pi - p.r);
    const bool rowok = (unsigned)y2 < (unsigned)p.H;
    for (int t = -p.T; t <= p.T; t++) {
      const int j0 = i0 + 32 * t;
      const int xb = q + p.off + p.s2 * (j0 + l31);
      const bool bok = rowok && (unsigned)xb < (unsigned)p.W;
      f32x16 acc;
#pragma unroll
      for (int r_ = 0; r_ < 16; r_++) acc[r_] = 0.f;
      if (__any(bok)) {
        const float* src = p.in1 + (((size_t)n1 * p.H + (bok ? y2 : 0)) * p.W + (bok ? xb : 0)) * p.ld_in + 4 * h;
#pragma unroll 1
        for (int ch = 0; ch < NCH; ch++) {
          float4 a[CQ];
#pragma unroll
          for (int q8 = 0; q8 < CQ; q8++) {
            a[q8] = ldg4(asrc + 8 * (ch * CQ + q8));
            a[q8].x = aok ? a[q8].x : 0.f; a[q8].y = aok ? a[q8].y : 0.f;
            a[q8].z = aok ? a[q8].z : 0.f; a[q8].w = aok ? a[q8].w : 0.f;
          }
#pragma unroll
          for (int q8 = 0; q8 < CQ; q8++) {
            float4 v = ldg4(src + 8 * (ch * CQ + q8));      // clamped address: unconditional load, masked by selects
            v.x = bok ? v.x : 0.f; v.y = bok ? v.y : 0.f; v.z = bok ? v.z : 0.f; v.w = bok ? v.w : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q8].x, v.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q8].y, v.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q8].z, v.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q8].w, v.w, acc, 0, 0, 0);
          }
        }
      }
      // band extraction: acc[r_] of lane (col j = l31, half h) is G[(r_&3) + 8*(r_>>2) + 4*h][j]
#pragma unroll
      for (int r_ = 0; r_ < 16; r_++) {
        const int i = i0 + (r_ & 3) + 8 * (r_ >> 2) + 4 * h;
        const int o = (j0 + l31) - i;
        const int ox = q + p.s2 * i;
        if (o >= -p.r && o <= p.r && ox < p.ow)
          p.out[(((size_t)n * p.oh + oy) * p.ow + ox) * p.ld_out + pi * p.gw + o + p.r] = acc[r_] / cf;
      }
    }
  }
}

// One wave: 32 sites x (CT*32) channels of the gradient of sample s, row y, class q, tile ia.
template <int CT>
__global__ __launch_bounds__(256) void corr_bwd_mfma_kernel(const CorrMfmaParams p) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int ncg = p.C / (32 * CT);
  long job = (long)blockIdx.x * 4 + wid;
  const int cg = (int)(job % ncg); job /= ncg;
  const int s = (int)(job % p.B); job /= p.B;
  const int ia = (int)(job % p.nA); job /= p.nA;
  const int q = (int)(job % p.s2); job /= p.s2;
  const int y = (int)job;
  if (y >= p.H) return;
  const int i0 = ia * 32, c0 = cg * 32 * CT;
  const int role_lo = p.fuse ? 0 : (int)blockIdx.y, role_hi = p.fuse ? 1 : (int)blockIdx.y;

  f32x16 acc[CT];
#pragma unroll
  for (int c = 0; c < CT; c++)
#pragma unroll
    for (int r_ = 0; r_ < 16; r_++) acc[c][r_] = 0.f;

  // Iteration space: (role, displacement row, neighbour tile).  All 16 + 16*CT operand values of an iteration
  // are loaded first — unconditional loads from clamped addresses, masked by selects: a load under `ok ? :` becomes a
  // branch per load and serialises load -> wait -> MFMA (that form ran at 405 us) — then the 16*CT MFMAs are issued.
  auto load_operands = [&](int role, int pi, int t, float (&av)[16], float (&bvv)[16][CT]) -> bool {
    // role 0: s is the FIRST input of pair (s, s+shift): band rows = own sites, source = in1[(s+shift)%B] at y + s2*p
    // role 1: s is the SECOND input of pair (s-shift, s): band cols = own sites, source = in0[(s-shift)%B] at y - s2*p
    const int nd = role == 0 ? s : ((s - p.shift) % p.B + p.B) % p.B;  // sample whose dOut is read
    const int ns = role == 0 ? (s + p.shift) % p.B : nd;              // sample whose features are the B operand
    const float* srcbase = role == 0 ? p.in1 : p.in0;
    const int dyp = p.s2 * (pi - p.r);
    const int ysrc = role == 0 ? y + dyp : y - dyp;            // feature row multiplied in
    const int oy = (role == 0 ? y : y - dyp) - p.off;          // output row whose dOut is used
    if ((unsigned)ysrc >= (unsigned)p.H || (unsigned)oy >= (unsigned)p.oh) return false;
    const int k0 = i0 + 32 * t;  // first contracted site index
    const int xk_lo = q + p.off + p.s2 * k0, xk_hi = q + p.off + p.s2 * (k0 + 31);
    if (xk_hi < 0 || xk_lo >= p.W) return false;               // no contracted site inside the image (wave-uniform)
    const float* drow = p.dout + ((size_t)nd * p.oh + oy) * p.ow * p.ld_dout + pi * p.gw + p.r;
    const float* srow = srcbase + ((size_t)ns * p.H + ysrc) * p.W * p.ld_in + c0 + l31;
#pragma unroll
    for (int st = 0; st < 16; st++) {
      const int k = k0 + 2 * st + h;      // contracted site
      const int own = i0 + l31;           // this lane's own site (row of the MFMA tile)
      // role 0: band element dOut[ox(own), o = k - own]; role 1: dOut[ox(k), o = own - k]
      const int site = role == 0 ? own : k;
      const int o = role == 0 ? k - own : own - k;
      const int ox = q + p.s2 * site;
      const bool ok = o >= -p.r && o <= p.r && (unsigned)ox < (unsigned)p.ow;
      const float v = drow[(size_t)(ok ? ox : 0) * p.ld_dout + (ok ? o : 0)];
      av[st] = ok ? v : 0.f;
    }
#pragma unroll
    for (int st = 0; st < 16; st++) {
      const int xs = q + p.off + p.s2 * (k0 + 2 * st + h);
      const bool ok = (unsigned)xs < (unsigned)p.W;
      const float* sp = srow + (size_t)(ok ? xs : 0) * p.ld_in;
#pragma unroll
      for (int c = 0; c < CT; c++) {
        const float tv = sp[32 * c];
        bvv[st][c] = ok ? tv : 0.f;
      }
    }
    return true;
  };

  // (Loading the operands of iteration it+1 before the MFMAs of iteration it was measured SLOWER — 305 vs 243 us at
  // the FlowNetC shape: the second operand set costs a wave of occupancy.  Three resident waves per SIMD overlap instead.)
  for (int role = role_lo; role <= role_hi; role++)
    for (int pi = 0; pi < p.gw; pi++)
      for (int t = -p.T; t <= p.T; t++) {
        float av[16], bvv[16][CT];
        if (!load_operands(role, pi, t, av, bvv)) continue;
#pragma unroll
        for (int st = 0; st < 16; st++)
#pragma unroll
          for (int c = 0; c < CT; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st], bvv[st][c], acc[c], 0, 0, 0);
      }
  float* gout = (p.fuse || blockIdx.y == 0) ? p.g0 : p.g1;
  const float cf = (float)p.C;
#pragma unroll
  for (int r_ = 0; r_ < 16; r_++) {
    const int i = i0 + (r_ & 3) + 8 * (r_ >> 2) + 4 * h;
    const int x = q + p.off + p.s2 * i;
    if ((unsigned)x >= (unsigned)p.W) continue;
    float* d = gout + (((size_t)s * p.H + y) * p.W + x) * p.ld_g + c0 + l31;
#pragma unroll
    for (int c = 0; c < CT; c++) d[32 * c] = acc[c][r_] / cf;
  }
}

// eight fp32 values (consecutive k of one MFMA operand row) -> their three bf16 planes (x = hi + mid + lo exactly, igemm_shared.h)
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float a = v[2 * i], b = v[2 * i + 1];
    const unsigned hh = igemm::cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(hh << 16), rb = b - __uint_as_float(hh & 0xffff0000u);
    const unsigned mm = igemm::cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(mm << 16), sb = rb - __uint_as_float(mm & 0xffff0000u);
    hi[i] = hh; mid[i] = mm; lo[i] = igemm::cvt_pk_bf16(sa, sb);
  }
}

// The backward banded product on the bf16 matrix cores: the same decomposition, loads and masks as corr_bwd_mfma_kernel,
// but a lane holds 8 CONSECUTIVE contracted sites (the K layout of v_mfma_f32_32x32x16_bf16) and every operand value is
// split into its three bf16 planes in registers (v_cvt_pk_bf16_f32); six product terms per K16 slab, fp32 accumulation —
// the arithmetic class of the conv kernels (§4.1b of DESIGN.md).  32 fp32 MFMAs of 64 cycles per iteration become 24 bf16
// MFMAs of 32 cycles; the ~220 split instructions run on the vector ALU beside them.
template <int CT>
__global__ __launch_bounds__(256) void corr_bwd_b3_kernel(const CorrMfmaParams p) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int ncg = p.C / (32 * CT);
  long job = (long)blockIdx.x * 4 + wid;
  const int cg = (int)(job % ncg); job /= ncg;
  const int s = (int)(job % p.B); job /= p.B;
  const int ia = (int)(job % p.nA); job /= p.nA;
  const int q = (int)(job % p.s2); job /= p.s2;
  const int y = (int)job;
  if (y >= p.H) return;
  const int i0 = ia * 32, c0 = cg * 32 * CT;
  const int role_lo = p.fuse ? 0 : (int)blockIdx.y, role_hi = p.fuse ? 1 : (int)blockIdx.y;

  f32x16 acc[CT];
#pragma unroll
  for (int c = 0; c < CT; c++)
#pragma unroll
    for (int r_ = 0; r_ < 16; r_++) acc[c][r_] = 0.f;

  // operands of one iteration: av[slab][e], bvv[slab][e][c] for contracted site k = k0 + 16*slab + 8*h + e
  auto load_operands = [&](int role, int pi, int t, float (&av)[2][8], float (&bvv)[2][8][CT]) -> bool {
    const int nd = role == 0 ? s : ((s - p.shift) % p.B + p.B) % p.B;  // sample whose dOut is read
    const int ns = role == 0 ? (s + p.shift) % p.B : nd;              // sample whose features are the B operand
    const float* srcbase = role == 0 ? p.in1 : p.in0;
    const int dyp = p.s2 * (pi - p.r);
    const int ysrc = role == 0 ? y + dyp : y - dyp;
    const int oy = (role == 0 ? y : y - dyp) - p.off;
    if ((unsigned)ysrc >= (unsigned)p.H || (unsigned)oy >= (unsigned)p.oh) return false;
    const int k0 = i0 + 32 * t;
    const int xk_lo = q + p.off + p.s2 * k0, xk_hi = q + p.off + p.s2 * (k0 + 31);
    if (xk_hi < 0 || xk_lo >= p.W) return false;
    const float* drow = p.dout + ((size_t)nd * p.oh + oy) * p.ow * p.ld_dout + pi * p.gw + p.r;
    const float* srow = srcbase + ((size_t)ns * p.H + ysrc) * p.W * p.ld_in + c0 + l31;
    const int own = i0 + l31;
#pragma unroll
    for (int sl = 0; sl < 2; sl++)
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int k = k0 + 16 * sl + 8 * h + e;
        const int site = role == 0 ? own : k;
        const int o = role == 0 ? k - own : own - k;
        const int ox = q + p.s2 * site;
        const bool ok = o >= -p.r && o <= p.r && (unsigned)ox < (unsigned)p.ow;
        const float v = drow[(size_t)(ok ? ox : 0) * p.ld_dout + (ok ? o : 0)];
        av[sl][e] = ok ? v : 0.f;
      }
#pragma unroll
    for (int sl = 0; sl < 2; sl++)
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int xs = q + p.off + p.s2 * (k0 + 16 * sl + 8 * h + e);
        const bool ok = (unsigned)xs < (unsigned)p.W;
        const float* sp = srow + (size_t)(ok ? xs : 0) * p.ld_in;
#pragma unroll
        for (int c = 0; c < CT; c++) {
          const float tv = sp[32 * c];
          bvv[sl][e][c] = ok ? tv : 0.f;
        }
      }
    return true;
  };

  constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first (conv_planes.hip mfma_terms)
  for (int role = role_lo; role <= role_hi; role++)
    for (int pi = 0; pi < p.gw; pi++)
      for (int t = -p.T; t <= p.T; t++) {
        float av[2][8], bvv[2][8][CT];
        if (!load_operands(role, pi, t, av, bvv)) continue;
#pragma unroll
        for (int sl = 0; sl < 2; sl++) {
          u32x4 ap[3], bp[CT][3];
          split8(av[sl], ap[0], ap[1], ap[2]);
#pragma unroll
          for (int c = 0; c < CT; c++) {
            float col[8];
#pragma unroll
            for (int e = 0; e < 8; e++) col[e] = bvv[sl][e][c];
            split8(col, bp[c][0], bp[c][1], bp[c][2]);
          }
#pragma unroll
          for (int tt = 0; tt < 6; tt++)
#pragma unroll
            for (int c = 0; c < CT; c++)
              acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ap[ta[tt]]), __builtin_bit_cast(bf16x8, bp[c][tb[tt]]),
                                                               acc[c], 0, 0, 0);
        }
      }
  float* gout = (p.fuse || blockIdx.y == 0) ? p.g0 : p.g1;
  const float cf = (float)p.C;
#pragma unroll
  for (int r_ = 0; r_ < 16; r_++) {
    const int i = i0 + (r_ & 3) + 8 * (r_ >> 2) + 4 * h;
    const int x = q + p.off + p.s2 * i;
    if ((unsigned)x >= (unsigned)p.W) continue;
    float* d = gout + (((size_t)s * p.H + y) * p.W + x) * p.ld_g + c0 + l31;
#pragma unroll
    for (int c = 0; c < CT; c++) d[32 * c] = acc[c][r_] / cf;
  }
}

CorrMfmaParams base_params(int B, int C, int H, int W, const CorrGeom& g) {
  CorrMfmaParams p{};
  p.B = B; p.C = C; p.H = H; p.W = W;
  p.oh = g.oh; p.ow = g.ow; p.r = g.r; p.gw = g.gw; p.s2 = g.s2;
  p.off = g.md - g.pad;
  // sites of a class cover output columns [0,ow) AND input columns [0,W): x = q + off + s2*i
  const int span = max(g.ow, W - p.off);
  const int nq = (span + g.s2 - 1) / g.s2;
  p.nA = (nq + 31) / 32;
  p.T = (g.r + 31) / 32;
  return p;
}

}  // namespace

int corr_mfma_supported(const CorrGeom& g, int C, int ld_in) {
  if (g.k != 1 || g.s1 != 1) return 0;
  if (C % 32 != 0 || C < 32 || C > 512) return 0;     // 3/8-width FlowNetC: C = 96 (flownet.py:22-23)
  if (ld_in % 4 != 0) return 0;
  if (g.md - g.pad > 0) return 0;  // never valid for the reference geometry (output would be empty) — keep it simple
  return 1;
}

int corr_mfma_fwd(const float* in0, const float* in1, int ld_in, int shift, float* out, int ld_out, int B, int C,
                  int H, int W, const CorrGeom& g, hipStream_t st) {
  CorrMfmaParams p = base_params(B, C, H, W, g);
  p.in0 = in0; p.in1 = in1; p.out = out; p.ld_in = ld_in; p.ld_out = ld_out; p.shift = shift;
  const int blocks = B * p.nA * g.s2 * g.oh;
  switch (C) {
    case 256: corr_fwd_mfma_kernel<256><<<blocks, 256, 0, st>>>(p); break;
    case 128: corr_fwd_mfma_kernel<128><<<blocks, 256, 0, st>>>(p); break;
    case 64: corr_fwd_mfma_kernel<64><<<blocks, 256, 0, st>>>(p); break;
    case 96: corr_fwd_mfma_kernel<96><<<blocks, 256, 0, st>>>(p); break;
    case 32: corr_fwd_mfma_kernel<32><<<blocks, 256, 0, st>>>(p); break;
    default: corr_fwd_mfma_kernel<0><<<blocks, 256, 0, st>>>(p); break;     // any other multiple of 32: runtime channel loop
  }
  return launch_status();
}

int corr_mfma_bwd(const float* dout, int ld_dout, const float* in0, const float* in1, int ld_in, int shift, float* g0,
                  float* g1, int ld_g, int fuse, int B, int C, int H, int W, const CorrGeom& g, hipStream_t st) {
  CorrMfmaParams p = base_params(B, C, H, W, g);
  p.in0 = in0; p.in1 = in1; p.dout = dout; p.g0 = g0; p.g1 = g1;
  p.ld_in = ld_in; p.ld_dout = ld_dout; p.ld_g = ld_g; p.shift = shift; p.fuse = fuse;
  const int CT = C % 64 == 0 ? 2 : 1;
  const long jobs = (long)(C / (32 * CT)) * B * p.nA * g.s2 * H;
  dim3 grid((unsigned)((jobs + 3) / 4), fuse ? 1 : 2);
  // default: fp32-equivalent products on the bf16 matrix cores (options corr_math_fp32 / conv_math_fp32: v_mfma_f32_32x32x2_f32)
  const bool b3 = !(unflow::options().corr_math_fp32 || unflow::options().conv_math_fp32);
  if (b3) {
    if (CT == 2) corr_bwd_b3_kernel<2><<<grid, 256, 0, st>>>(p);
    else corr_bwd_b3_kernel<1><<<grid, 256, 0, st>>>(p);
    return launch_status();
  }
  if (CT == 2) corr_bwd_mfma_kernel<2><<<grid, 256, 0, st>>>(p);
  else corr_bwd_mfma_kernel<1><<<grid, 256, 0, st>>>(p);
  return launch_status();
}
