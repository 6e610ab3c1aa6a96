// FlowNetC cost volume (kernel_size 1, stride_1 1 — flownet.py:221-222; CorrelateData of ops/correlation_op.cu.cc:51-117)
// on the bf16 matrix cores from the features' operand planes (csrc/conv_planes.hip: x = hi + mid + lo, six product terms,
// fp32 accumulation — the arithmetic class of correlation_mfma.hip's v_mfma_f32_32x32x2_f32 at 6/16 of its matrix-core
// time).
//
// For output row oy, displacement row p and x-residue class q the 2r+1 correlations of a pixel are a band of the 32x32 Gram
// matrix G[i][j] = sum_c f0[y, x_i, c] * f1[y + s2 p, x_j, c] (correlation_mfma.hip).  Here a block owns (sample, row,
// class, 32-site tile): its f0 tile — 32 pixels x C channels x 3 planes — is loaded ONCE into LDS with coalesced
// 16-byte loads and serves all 2r+1 displacement rows of the block's four waves (padded rows: conflict-free b128 fragment
// reads); the f1 fragments stream from L2 as 16-byte loads (8 consecutive channels of one pixel and plane = one MFMA operand
// granule), four K16 slabs in flight per wave.  The forward f0 traffic drops from (2r+1) fetches per row to one.
#include "igemm_shared.h"
#include "correlation_geom.h"

namespace {
using namespace igemm;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct CorrPlParams {
  const unsigned short* f0;   // planes of in0, channel 0 of the feature slice; [pixel][ld] per plane
  const unsigned short* f1;
  long ps;                    // plane stride (elements), common to both
  int ld;
  float* out;
  int ld_out, shift;
  int B, C, H, W;
  int oh, ow, r, gw, s2;
  int off;  // input coordinate = output coordinate + off (= max_displacement - pad)
  int nA;   // 32-site tiles per residue class
  int T;    // neighbour tiles on each side that the band can reach: ceil(r / 32)
};

__global__ __launch_bounds__(256) void corr_fwd_pl_kernel(const CorrPlParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const int tid = threadIdx.x, lane = tid & 63, pg = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  int b = blockIdx.x;
  const int n = b % p.B; b /= p.B;
  const int ia = b % p.nA; b /= p.nA;
  const int q = b % p.s2; b /= p.s2;
  const int oy = b;
  const int i0 = ia * 32;
  const int n1 = (n + p.shift) % p.B;
  const int y0 = oy + p.off;
  const int C = p.C, Cg = C >> 3;
  const int APITCH = C + 8;                // elements: (2C + 16)-byte rows -> the 16 lanes of a b128 group hit 16 bank slots
  const int APLANE = 32 * APITCH;
  const int ld2 = p.ld * 2;
  const size_t recs = (((size_t)p.B * p.H * p.W - 1) * (size_t)p.ld + (size_t)C) * 2;
  __amdgpu_buffer_rsrc_t f0_rs[3], f1_rs[3];
#pragma unroll
  for (int pl = 0; pl < 3; pl++) {
    f0_rs[pl] = make_rsrc(p.f0 + pl * p.ps, recs);
    f1_rs[pl] = make_rsrc(p.f1 + pl * p.ps, recs);
  }
  // ---- the block's f0 tile -> LDS (zero rows for sites outside the output / the image)
  for (int it = tid; it < 32 * Cg; it += 256) {
    const int row = it / Cg, g = it - row * Cg;
    const int ox = q + p.s2 * (i0 + row), x0 = ox + p.off;
    const bool ok = ox < p.ow && (unsigned)x0 < (unsigned)p.W && (unsigned)y0 < (unsigned)p.H;
    const int voff = ok ? ((n * p.H + y0) * p.W + x0) * ld2 + g * 16 : OOB_MARK;
#pragma unroll
    for (int pl = 0; pl < 3; pl++)
      *reinterpret_cast<u32x4*>(lds + pl * APLANE + row * APITCH + g * 8) = buf_ld16(f0_rs[pl], voff);
  }
  __syncthreads();
  const unsigned short* a_rd = lds + l31 * APITCH + h * 8;
  const float cf = (float)C;
  const int per = (p.gw + 3) >> 2;
  // the short last share (gw = 21: 6,6,6,3) rotates over the waves (wave w of every block sits on SIMD w)
  const int share = (pg + (int)blockIdx.x) & 3;
  const int pa = share * per, pb = min(p.gw, pa + per);
  const int NS = C >> 4;                   // K16 slabs
  const int NG = (NS + 3) >> 2;            // groups of four slabs
  // The (displacement row, neighbour tile) items of this wave, software-pipelined: the f1 fragments of the NEXT group of
  // four slabs — of this Gram or the first group of the next one — are requested before the MFMAs of the current group
  // (two register sets; with one set every group exposed an L2 round trip: 98 us for 20 us of matrix-core work).
  int pi = pa, t = -p.T - 1, b_off = OOB_MARK, j0 = 0;
  auto next_item = [&]() -> bool {         // advance (pi, t) to the next item with a site inside the image
    for (;;) {
      if (++t > p.T) { t = -p.T; pi++; }
      if (pi >= pb) return false;
      const int y2 = y0 + p.s2 * (pi - p.r);
      j0 = i0 + 32 * t;
      const int xb = q + p.off + p.s2 * (j0 + l31);
      const bool bok = (unsigned)y2 < (unsigned)p.H && (unsigned)xb < (unsigned)p.W;
      if (!__any(bok)) continue;
      b_off = bok ? ((n1 * p.H + y2) * p.W + xb) * ld2 + h * 16 : OOB_MARK;
      return true;
    }
  };
  u32x4 b0[4][3], b1[4][3];
  auto load_group = [&](u32x4 (&bq)[4][3], int grp, bool live) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int pl = 0; pl < 3; pl++)
        bq[u][pl] = buf_ld16(f1_rs[pl], (live && 4 * grp + u < NS) ? b_off + (4 * grp + u) * 32 : OOB_MARK);
  };
  f32x16 acc;
  auto mfma_group = [&](const u32x4 (&bq)[4][3], int grp) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (4 * grp + u >= NS) break;
      s16x8 av[3];
#pragma unroll
      for (int pl = 0; pl < 3; pl++) av[pl] = *reinterpret_cast<const s16x8*>(a_rd + pl * APLANE + (4 * grp + u) * 16);
      constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
#pragma unroll
      for (int tt = 0; tt < 6; tt++)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[ta[tt]]),
                                                      __builtin_bit_cast(bf16x8, bq[u][tb[tt]]), acc, 0, 0, 0);
    }
  };
  bool have = next_item();
  if (have) load_group(b0, 0, true);
  while (have) {
    const int pi_c = pi, j0_c = j0;        // the item being multiplied (next_item() below moves on)
#pragma unroll
    for (int r_ = 0; r_ < 16; r_++) acc[r_] = 0.f;
    bool more = true;
    for (int g2 = 0; g2 < NG; g2 += 2) {
      load_group(b1, g2 + 1, true);        // (an absent odd group loads zeros and multiplies nothing)
      mfma_group(b0, g2);
      if (g2 + 2 < NG) {
        load_group(b0, g2 + 2, true);
      } else {
        more = next_item();
        load_group(b0, 0, more);
      }
      mfma_group(b1, g2 + 1);
    }
    // band extraction: acc[r_] of lane (col j = l31, half h) is G[(r_&3) + 8*(r_>>2) + 4*h][j]
#pragma unroll
    for (int r_ = 0; r_ < 16; r_++) {
      const int i = i0 + (r_ & 3) + 8 * (r_ >> 2) + 4 * h;
      const int o = (j0_c + l31) - i;
      const int ox = q + p.s2 * i;
      if (o >= -p.r && o <= p.r && ox < p.ow)
        p.out[(((size_t)n * p.oh + oy) * p.ow + ox) * p.ld_out + pi_c * p.gw + o + p.r] = acc[r_] / cf;
    }
    have = more;
  }
}

}  // namespace

int corr_pl_supported(const CorrGeom& g, int C, const unflow_planes* a, const unflow_planes* b) {
  if (g.k != 1 || g.s1 != 1 || g.md - g.pad > 0) return 0;
  if (C % 16 != 0 || C > 1024) return 0;
  if (!a || !b || !a->base || !b->base || a->n_planes != 3 || b->n_planes != 3) return 0;
  if (a->ld != b->ld || a->plane_stride != b->plane_stride || a->ld % 4 != 0 || a->ld < C) return 0;
  if ((reinterpret_cast<uintptr_t>(a->base) | reinterpret_cast<uintptr_t>(b->base)) & 7) return 0;
  return 1;
}

int corr_pl_fwd(const unflow_planes* in0, const unflow_planes* in1, int shift, float* out, int ld_out, int B, int C, int H,
                int W, const CorrGeom& g, hipStream_t st) {
  CorrPlParams p{};
  p.f0 = reinterpret_cast<const unsigned short*>(in0->base);
  p.f1 = reinterpret_cast<const unsigned short*>(in1->base);
  p.ps = in0->plane_stride; p.ld = in0->ld;
  p.out = out; p.ld_out = ld_out; p.shift = shift;
  p.B = B; p.C = C; p.H = H; p.W = W;
  p.oh = g.oh; p.ow = g.ow; p.r = g.r; p.gw = g.gw; p.s2 = g.s2;
  p.off = g.md - g.pad;
  const int span = max(g.ow, W - p.off);
  const int nq = (span + g.s2 - 1) / g.s2;
  p.nA = (nq + 31) / 32;
  p.T = (g.r + 31) / 32;
  const int smem = 3 * 32 * (C + 8) * 2;
  static int smem_set = 0;   // grow-only: benign race, idempotent
  if (smem > smem_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_fwd_pl_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              smem);
    smem_set = smem;
  }
  const int blocks = B * p.nA * g.s2 * g.oh;
  corr_fwd_pl_kernel<<<blocks, 256, smem, st>>>(p);
  return launch_status();
}
