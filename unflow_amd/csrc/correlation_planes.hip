// FlowNetC cost volume (kernel_size 1, stride_1 1 — flownet.py:221-222; CorrelateData of ops/correlation_op.cu.cc:51-117)
// on the bf16 matrix cores from the features' operand planes (csrc/conv_planes.hip: x = hi + mid + lo, six product terms,
// fp32 accumulation — the arithmetic class of correlation_mfma.hip's v_mfma_f32_32x32x2_f32 at 6/16 of its matrix-core
// time).  For output row oy, displacement row p and x-residue class q the 2r+1 correlations of a pixel are a band of the
// 32x32 Gram matrix G[i][j] = sum_c f0[y, x_i, c] * f1[y + s2 p, x_j, c] (correlation_mfma.hip).
//
// Forward kernels (corr_pl_fwd picks one):
//   corr_fwd_wb_kernel  wide band (r > 6, or one site tile per row), C % 64 == 0, C <= 256 (the step's kernel of rounds 2-3):
//                       K split over the waves, f1 tiles by LDS-DMA in whole cache lines, two output rows per block.
//   corr_fwd_rw_kernel  wide band, C = 128 or 256: a wave pair per output row, four rows per workgroup sharing whole f1 tiles
//                       (the step's kernel since round 4; corr_fwd_wb_kernel stays for C = 64 / 192 and as its A/B).
//   corr_fwd_nb_kernel  narrow band (r <= 6 over several site tiles: the +-4 / 81-channel cost volume), same limits on C:
//                       tiles own 32 - 2r sites, one Gram per displacement row.
//   corr_fwd_pl_kernel  every other C % 16 == 0: the first planes kernel.  A block owns (sample, row, class, 32-site tile): its
//                       f0 tile — 32 pixels x C channels x 3 planes — is loaded ONCE into LDS with coalesced 16-byte loads and
//                       serves all 2r+1 displacement rows of the block's four waves (padded rows: conflict-free b128 fragment
//                       reads); the f1 fragments stream from L2 as 16-byte loads (8 consecutive channels of one pixel and
//                       plane = one MFMA operand granule), four K16 slabs in flight per wave — 32 cache lines per load
//                       instruction, which is what bounds it.
// Backward: corr_bwd_pl_kernel (C % 64 == 0), feature operand from the planes by LDS-DMA + transposing reads.
#include <cstdlib>
#include <type_traits>
#include "igemm_shared.h"
#include "options.h"
#include "correlation_geom.h"

namespace {
using namespace igemm;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// Phase trace of corr_fwd_rw_kernel (diagnostic builds only: -DUNFLOW_CORR_TRACE=<workgroup>, tools/debug/corr_phase_trace.py):
// every wave of that workgroup sums shader cycles per phase of a step in registers; never compiled into the shipped library.
#ifdef UNFLOW_CORR_TRACE
__device__ unsigned long long g_corr_trace[8 * 8];
#define CT_DECL unsigned long long ct_prev = __builtin_readcyclecounter(), ct_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define CT_STAMP(slot) do { const unsigned long long ct_now = __builtin_readcyclecounter(); ct_acc[slot] += ct_now - ct_prev; ct_prev = ct_now; } while (0)
#define CT_COUNT(slot) do { ct_acc[slot]++; } while (0)
#define CT_FLUSH do { if (blockIdx.x == UNFLOW_CORR_TRACE && (threadIdx.x & 63) == 0) for (int ct_i = 0; ct_i < 8; ct_i++) g_corr_trace[(threadIdx.x >> 6) * 8 + ct_i] = ct_acc[ct_i]; } while (0)
#else
#define CT_DECL do { } while (0)
#define CT_STAMP(slot) do { } while (0)
#define CT_COUNT(slot) do { } while (0)
#define CT_FLUSH do { } while (0)
#endif

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
  int vr;   // valid rows of a 32-row tile: 32, or 32 - 2r in narrow-band mode (then T = 0 and the column tile starts at i0 - r)
  int joff; // column-tile offset: 0, or -r in narrow-band mode
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
  const int i0 = ia * p.vr;
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
  // band extraction: acc[r_] of lane (col j = l31, half h) is G[(r_&3) + 8*(r_>>2) + 4*h][j]
  auto store_band = [&](int pi_c, int j0_c, const f32x16* g) {
#pragma unroll
    for (int r_ = 0; r_ < 16; r_++) {
      const int i = i0 + (r_ & 3) + 8 * (r_ >> 2) + 4 * h;
      const int o = (j0_c + l31) - i;
      const int ox = q + p.s2 * i;
      if (o >= -p.r && o <= p.r && ox < p.ow && i - i0 < p.vr)
        p.out[(((size_t)n * p.oh + oy) * p.ow + ox) * p.ld_out + pi_c * p.gw + o + p.r] = g ? (*g)[r_] / cf : 0.f;
    }
  };
  auto next_item = [&]() -> bool {         // advance (pi, t) to the next item with a site inside the image
    for (;;) {
      if (++t > p.T) { t = -p.T; pi++; }
      if (pi >= pb) return false;
      const int y2 = y0 + p.s2 * (pi - p.r);
      j0 = i0 + 32 * t + p.joff;
      const int xb = q + p.off + p.s2 * (j0 + l31);
      const bool bok = (unsigned)y2 < (unsigned)p.H && (unsigned)xb < (unsigned)p.W;
      if (!__any(bok)) {                   // no site of this Gram lies in the image: its band entries are zero
        store_band(pi, j0, nullptr);
        continue;
      }
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
    store_band(pi_c, j0_c, &acc);
    have = more;
  }
}


// ------------------------------------------------------------------------------------------- narrow-band forward
// The +-4 cost volume of the full-resolution networks (r <= 6, several site tiles per row: corr_pl_tiles).  Streaming the
// f1 fragments from L2 into registers (corr_fwd_pl_kernel) touches 32 cache lines per load instruction — one per site, 32
// bytes of each — and with one Gram per displacement row there is no reuse to pay for it: the address path, not the
// matrix cores or HBM, bounds that kernel here (481 us for 16 x 96 x 128 x 256; the matrix-core and HBM floors are ~125 us).
// This kernel moves whole lines instead.  A block owns (sample, row, class, tile of 32 - 2r sites); wave w owns channels
// 64 w .. 64 w + 63 (K is split over the waves) and keeps its f0 fragments in registers for all 2r+1 displacement rows.
// The f1 rows go, one after the other, HBM/L2 -> a wave-private 12 KB LDS tile by LDS-DMA (8 full 128-byte lines per
// instruction, source-side XOR swizzle so the b128 fragment reads are conflict-free); the next row's tile is requested as
// soon as this row's fragments are in registers and lands under the 24 MFMAs.  The useful band of each partial Gram
// ((32 - 2r) x (2r+1) of 1024 entries) goes to a wave-private LDS buffer; after a barrier the block sums the waves'
// partials in a fixed order and writes each site's (2r+1)^2 contiguous outputs.
//
// Measured alternatives (16 x 96 x 128 x 256, r = 4): two output rows per block sharing each f1 tile,
// with the band partials of two waves ADDED into one LDS buffer (ds_add_f32; commutative, so still reproducible) to keep
// two blocks per CU: 505 us against 331-406 us — the 16 masked LDS atomics per Gram cost ~20 cycles each (190 us of the
// 505; plain stores 58 us); one block per CU (larger LDS footprint): +37 %.  Ablations of this kernel: no MFMAs -27 %, no
// f1 DMA -21 %, no band stores -10 %, none of the three: 160 us — the phases of a wave add up (two waves per SIMD, limited by
// the 80 KB of LDS per block), which is what bounds it, not the matrix cores (floor ~125 us) or HBM (~130 us).
__global__ __launch_bounds__(256, 2) void corr_fwd_nb_kernel(const CorrPlParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  constexpr int TILE = 3 * 32 * 64;                 // elements per wave: 3 planes x 32 sites x 64 channels
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw = blockDim.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  // work order: XCD-contiguous, site tiles fastest, then rows — the ~64 blocks resident together on an XCD are a few
  // consecutive rows of one sample: neighbouring tiles share 2r f1 columns, neighbouring rows 2r of their 2r+1 f1 rows
  int b = xcd_remap(blockIdx.x, gridDim.x, 1);
  const int ia = b % p.nA; b /= p.nA;
  const int oy = b % p.oh; b /= p.oh;
  const int q = b % p.s2; b /= p.s2;
  const int n = b;
  const int i0 = ia * p.vr, c0 = wid * 64;
  const int n1 = (n + p.shift) % p.B;
  const int y0 = oy + p.off;
  const int ld2 = p.ld * 2;
  const size_t recs = (((size_t)p.B * p.H * p.W - 1) * (size_t)p.ld + (size_t)p.C) * 2;
  u32x4 f1_rs[3];
  __amdgpu_buffer_rsrc_t f0_rs[3];
#pragma unroll
  for (int pl = 0; pl < 3; pl++) {
    f0_rs[pl] = make_rsrc(p.f0 + pl * p.ps, recs);
    f1_rs[pl] = raw_rsrc(p.f1 + pl * p.ps, recs);
  }
  unsigned short* tile = lds + wid * TILE;
  const unsigned tile_addr = lds_addr(tile);
  const int bsz = p.gw * p.vr * p.gw;               // band partials of one wave: [displacement row][site][offset]
  float* band = reinterpret_cast<float*>(lds + nw * TILE);
  float* myband = band + wid * bsz;

  // f0 fragments, straight to registers (once per block; in flight together with the first f1 tile): slab u of lane
  // (site l31, half h) = channels c0 + 16u + 8h .. + 7
  s16x8 af[4][3];
  {
    const int xs = q + p.off + p.s2 * (i0 + l31);
    // (the Gram rows past the owned sites are never read out: zeros, no traffic)
    const bool ok = (unsigned)xs < (unsigned)p.W && (unsigned)y0 < (unsigned)p.H && l31 < p.vr;
    const int voff = ok ? ((n * p.H + y0) * p.W + xs) * ld2 + (c0 + h * 8) * 2 : OOB_MARK;
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int pl = 0; pl < 3; pl++) af[u][pl] = __builtin_bit_cast(s16x8, buf_ld16(f0_rs[pl], voff + u * 32));
  }
  // DMA lane mapping: instruction j covers sites 8j .. 8j+7 (one 128-byte line each); LDS slot lane % 8 of site 8j + lane/8
  // receives the granule slot ^ ((site >> 1) & 7)
  const int d_site = lane >> 3, d_slot = lane & 7;
  const int k0 = i0 + p.joff;
  auto issue_tile = [&](int yy) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int site = 8 * j + d_site;
      const int g = d_slot ^ ((site >> 1) & 7);
      const int xs = q + p.off + p.s2 * (k0 + site);
      const bool ok = (unsigned)xs < (unsigned)p.W && (unsigned)yy < (unsigned)p.H;
      const int voff = ok ? ((n1 * p.H + yy) * p.W + xs) * ld2 + (c0 + g * 8) * 2 : OOB_MARK;
      const unsigned d = tile_addr + (unsigned)(j * 1024);
      dma3(voff, f1_rs[0], f1_rs[1], f1_rs[2], d, d + 32 * 64 * 2, d + 2 * 32 * 64 * 2);
    }
  };
  // The displacement rows are visited in a rotated order: step t multiplies the f1 row R with R / s2 = t (mod 2r+1), so the
  // 2r+1 blocks (neighbouring output rows) that need a given f1 row ask for it in the same step — one of them brings it
  // into the XCD's L2, the others hit (measured: 2-3 %).
  const int yq = (y0 - (p.s2 - 1) * (y0 < 0)) / p.s2;                       // floor(y0 / s2)
  const int rot = ((p.r - yq) % p.gw + p.gw) % p.gw;
  issue_tile(y0 + p.s2 * (rot - p.r));
  constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
  for (int t = 0; t < p.gw; t++) {
    const int pi = t + rot < p.gw ? t + rot : t + rot - p.gw;
    const int pn = pi + 1 < p.gw ? pi + 1 : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this row's tile has landed
    __builtin_amdgcn_sched_barrier(0);
    s16x8 bf[4][3];
    // fragment of slab u: lane (site l31, half h) holds granule 2u + h of its site
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int pl = 0; pl < 3; pl++)
        bf[u][pl] = *reinterpret_cast<const s16x8*>(tile + pl * (32 * 64) + l31 * 64 + (((2 * u + h) ^ ((l31 >> 1) & 7)) << 3));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // fragments in registers: the tile may be overwritten
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < p.gw) issue_tile(y0 + p.s2 * (pn - p.r));
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int tt = 0; tt < 6; tt++)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[u][ta[tt]]),
                                                      __builtin_bit_cast(bf16x8, bf[u][tb[tt]]), acc, 0, 0, 0);
    // acc[e] of lane (column l31, half h) is G[li][l31], li = (e&3) + 8*(e>>2) + 4h; column j is site i0 - r + j, so the
    // band offset index (o + r) of entry (li, j) is j - li
    float* brow = myband + pi * p.vr * p.gw;
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int li = (e & 3) + 8 * (e >> 2) + 4 * h;
      const int oi = l31 - li;
      if (li < p.vr && oi >= 0 && oi < p.gw) brow[li * p.gw + oi] = acc[e];
    }
  }
  __syncthreads();
  const float cf = (float)p.C;
  const int g2 = p.gw * p.gw;
  const float inv_g2 = 1.0f / (float)g2, inv_gw = 1.0f / (float)p.gw;
  float* orow = p.out + ((size_t)n * p.oh + oy) * p.ow * p.ld_out;
  for (int idx = threadIdx.x; idx < p.vr * g2; idx += blockDim.x) {
    // idx = (li, pi, oi); the quotients are exact in fp32 at these sizes (the fractions stay >= 0.5 / 169 from an integer)
    const int li = (int)(((float)idx + 0.5f) * inv_g2), rem = idx - li * g2;
    const int pi = (int)(((float)rem + 0.5f) * inv_gw), oi = rem - pi * p.gw;
    const int ox = q + p.s2 * (i0 + li);
    if (ox >= p.ow) break;
    const float* src = band + (pi * p.vr + li) * p.gw + oi;
    float sum = src[0];
    for (int w = 1; w < nw; w++) sum += src[w * bsz];
    orow[(size_t)ox * p.ld_out + rem] = sum / cf;
  }
}


// ------------------------------------------------------------------------------- narrow-band forward, rows shared (round 5)
// corr_fwd_nb_kernel fetches every f1 row once per OUTPUT row that needs it: 2r+1 = 9 times for the +-4 cost volume — 4.5 GB
// through L2 -> LDS for 0.6 GB of planes at 16 x 96 x 128 x 256, and that stream, not the matrix cores, paces it (339 us).
// Here a workgroup (4 waves) owns RS_R = 4 consecutive output rows of one 24-site tile, one row per wave, and walks the
// channels in chunks of 32: per chunk the RS_R + 2r f1 rows the four output rows share are brought in ONCE (LDS-DMA, 64
// contiguous bytes per site and plane, source-side XOR swizzle for conflict-free b128 fragment reads) and every wave multiplies
// its own f0 fragments (straight from L2 to registers: one row, used by all 2r+1 displacement rows) against its 2r+1 rows of
// them.  The 2r+1 Grams of a wave stay in its accumulators over all chunks (9 x 16 registers), so nothing is exchanged between
// waves and no partial bands travel through LDS; f1 traffic drops 9 -> 3 fetches per row.  Two workgroups per CU (72 KB of
// LDS each): one multiplies while the other waits for its chunk.  stride_2 = 1, r <= 4, C % 32 == 0.
constexpr int RS_R = 4, RS_MAXG = 9;
constexpr int RS_ROWS = RS_R + RS_MAXG - 1;             // f1 rows of a chunk (r = 4: 12)
constexpr int RS_PLANE = RS_ROWS * 32 * 64;             // bytes per plane: [row][32 sites][64 B]
constexpr int RS_SMEM = 3 * RS_PLANE;                   // 73,728 B; the epilogue's band staging (4 x 24 x 81 floats) reuses it

template <int GW>                                      // 2r + 1: compile-time, so that the step loop has no branches (a conditional
                                                       // fragment read makes hipcc wait for ALL outstanding LDS reads at the join)
__global__ __launch_bounds__(256, 2) void corr_fwd_rs_kernel(const CorrPlParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  unsigned char* f1s = reinterpret_cast<unsigned char*>(lds);
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int l31 = lane & 31, h = lane >> 5;
  // work order: XCD-contiguous, site tiles fastest, then row groups, then samples
  int b = xcd_remap(blockIdx.x, gridDim.x, 1);
  const int ia = b % p.nA; b /= p.nA;
  const int ngr = (p.oh + RS_R - 1) / RS_R;
  const int gy = b % ngr;
  const int n = b / ngr;
  const int i0 = ia * p.vr;                             // first owned site (output x) of the tile
  const int oy0 = gy * RS_R, oy = oy0 + wid;            // this wave's output row
  const int n1 = (n + p.shift) % p.B;
  const int ld2 = p.ld * 2;
  constexpr int nrows = RS_R + GW - 1;                    // f1 rows of a chunk
  const size_t recs = (((size_t)p.B * p.H * p.W - 1) * (size_t)p.ld + (size_t)p.C) * 2;
  u32x4 f1_rs[3];
  __amdgpu_buffer_rsrc_t f0_rs[3];
#pragma unroll
  for (int pl = 0; pl < 3; pl++) {
    f0_rs[pl] = make_rsrc(p.f0 + pl * p.ps, recs);
    f1_rs[pl] = raw_rsrc(p.f1 + pl * p.ps, recs);
  }
  const unsigned f1_addr = lds_addr(f1s);
  // f0 fragment offset of the lane: site i0 + l31 (Gram rows past the owned sites are never read out: zeros, no traffic), granule h
  int a_off;
  {
    const int xs = p.off + i0 + l31, ys = oy + p.off;
    const bool ok = l31 < p.vr && (unsigned)xs < (unsigned)p.W && (unsigned)ys < (unsigned)p.H && oy < p.oh;
    a_off = ok ? ((n * p.H + ys) * p.W + xs) * ld2 + h * 16 : OOB_MARK;
  }
  // DMA units of a chunk: (f1 row j, half hh of its 32 sites) -> 16 sites x 64 B = 1 KB, lane-linear in LDS: lane i fills slot
  // i & 3 of site 16 hh + (i >> 2) with granule slot ^ ((site >> 2) & 3).  Wave w takes units w, w + 4, ...
  const int d_site = lane >> 2, d_slot = lane & 3;
  int d_off[2 * RS_ROWS / 4];
#pragma unroll
  for (int k = 0; k < 2 * RS_ROWS / 4; k++) {
    const int unit = wid + 4 * k, j = unit >> 1, hh = unit & 1;
    const int site = 16 * hh + d_site;
    const int g = d_slot ^ ((site >> 2) & 3);
    const int xs = p.off + i0 - p.r + site, yy = oy0 + p.off - p.r + j;
    const bool ok = j < nrows && (unsigned)xs < (unsigned)p.W && (unsigned)yy < (unsigned)p.H;
    d_off[k] = ok ? ((n1 * p.H + yy) * p.W + xs) * ld2 + g * 16 : OOB_MARK;
  }
  f32x16 acc[GW];
#pragma unroll
  for (int j = 0; j < GW; j++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[j][e] = 0.f;
  constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
  // B fragment of (row jj, slab u): lane (site l31, half h) holds granule 2u + h of its site
  const unsigned char* b_rd = f1s + (wid * 32 + l31) * 64;
  const int bsw = (l31 >> 2) & 3;
  const int nchunk = p.C >> 5;
  for (int c = 0; c < nchunk; c++) {
    __syncthreads();                                   // every wave is done with the previous chunk's rows
#pragma unroll
    for (int k = 0; k < 2 * RS_ROWS / 4; k++) {
      const int unit = wid + 4 * k;
      const unsigned d = f1_addr + (unsigned)(unit * 1024);
      dma3(d_off[k] + c * 64, f1_rs[0], f1_rs[1], f1_rs[2], d, d + RS_PLANE, d + 2 * RS_PLANE);
    }
    s16x8 af[2][3];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int pl = 0; pl < 3; pl++) af[u][pl] = __builtin_bit_cast(s16x8, buf_ld16(f0_rs[pl], a_off + c * 64 + u * 32));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the rows has landed (and its f0 fragments)
    __syncthreads();                                   // ... and everybody else's
    s16x8 bf[2][2][3];
    auto rd = [&](int j, s16x8 (&f)[2][3]) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int pl = 0; pl < 3; pl++)
          f[u][pl] = *reinterpret_cast<const s16x8*>(b_rd + pl * RS_PLANE + j * (32 * 64) + (((2 * u + h) ^ bsw) << 4));
    };
    rd(0, bf[0]);
#pragma unroll
    for (int j = 0; j < GW; j++) {
      if (j + 1 < GW) rd(j + 1, bf[(j + 1) & 1]);
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int tt = 0; tt < 6; tt++)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[u][ta[tt]]),
                                                          __builtin_bit_cast(bf16x8, bf[j & 1][u][tb[tt]]), acc[j], 0, 0, 0);
    }
  }
  __syncthreads();                                     // the f1 rows are dead: their LDS becomes the band staging
  // acc[j][e] of lane (column l31, half h) is G[li][l31], li = (e & 3) + 8 (e >> 2) + 4 h; column l31 is site i0 - r + l31, so
  // the band offset index of entry (li, l31) is l31 - li.  Staging [site][displacement row][offset], one area per wave.
  constexpr int g2 = GW * GW;
  float* stg = reinterpret_cast<float*>(lds) + wid * (p.vr * g2);
#pragma unroll
  for (int j = 0; j < GW; j++) {
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int li = (e & 3) + 8 * (e >> 2) + 4 * h;
      const int oi = l31 - li;
      if (li < p.vr && oi >= 0 && oi < GW) stg[li * g2 + j * GW + oi] = acc[j][e];
    }
  }
  // (wave-private: the same wave reads it back — LDS operations of one wave complete in order)
  if (oy < p.oh) {
    const float cf = (float)p.C;
    const float inv_g2 = 1.0f / (float)g2;
    float* orow = p.out + ((size_t)n * p.oh + oy) * p.ow * p.ld_out;
    for (int idx = lane; idx < p.vr * g2; idx += 64) {
      // idx = (li, rem); the quotient is exact in fp32 at these sizes
      const int li = (int)(((float)idx + 0.5f) * inv_g2), rem = idx - li * g2;
      const int ox = i0 + li;
      if (ox >= p.ow) break;
      orow[(size_t)ox * p.ld_out + rem] = stg[idx] / cf;
    }
  }
}

// ------------------------------------------------------------------------------- +-4 forward, f1 rows as a stream (round 5)
// The row-shared kernel above alternates fetch and multiply per chunk (two workgroups per CU hide each other's fetch at best
// half of the time) and still reads every f1 row three times; per-dispatch counters put it at 1.0 GB from HBM and 1.5 GB
// through L2 for 0.6 GB of planes, 4.3 TB/s with the matrix cores idle 55 % of the time.  Here ONE workgroup per CU (8 waves,
// one output row each: RG_R = 8 rows of a 24-site tile, 16 f1 rows per chunk: two fetches per row) treats LDS as a ring of
// RG_NS = 24 row slots (6 KB each: [plane][32 sites][64 B]) that the (chunk, row) stream runs through in order: row (c, r) sits
// in slot (16 c + r) mod 24, is first needed at step max(0, r - 7) of its chunk (step j: wave w multiplies its f0 fragments
// with row w + j) and is dead after step min(r, 8) — so the slots of chunk c free up while chunk c is multiplied and take
// chunk c + 1 with a lead of five or more steps, the whole time:
//   step 0: [wait rows 0..11 of c] barrier, request rows 0..7 of c + 1 (slots of rows 8..15 of c - 1) and the f0 fragments of c + 1
//   step 4: [wait rows 12..15 of c] barrier, request rows 8..11 of c + 1 (slots of rows 0..3 of c)
//   step 8:                         barrier, request rows 12..15 of c + 1 (slots of rows 4..7 of c)
// Every wave issues the same number of loads at every point (2 + 1 + 1 three-plane DMA units of 1 KB per plane, then 6 fragment
// loads), so the waits are immediates: vmcnt(3) at step 0 (the step-8 request may still be out), vmcnt(12) at step 4.  All
// loads are inline asm — the compiler's own bookkeeping would wait for everything at the first use of a fragment — and the
// fragment registers are tied through the wait statement.  Past the last chunk the requests are out of range (zeros, no traffic).
// Products: v_mfma_f32_16x16x32_bf16 — a chunk's 32 channels are ONE instruction deep, and the band (9 offsets of 24 owned sites)
// wastes less of 16 x 16 blocks than of a 32 x 32 Gram: per f1 row three products (f0 sites 0..15 x staged sites 0..15 and
// 16..31, f0 sites 16..23 x staged sites 16..31), 3 x 6 terms x 16 cycles = 288 instead of 2 x 6 x 32 = 384, and the three
// accumulators alternate (a wait or a read between two MFMAs on the SAME accumulator costs ~43 cycles on gfx950, between
// different ones ~6).  Lane l of an operand holds granule l >> 4 (8 channels) of site l & 15; staged granule q of site s sits in
// slot q ^ swz(s) of the site's 64 bytes, which makes every 16-lane group of the b128 fragment reads hit 16 distinct bank slots.
// (Measured and dropped, profiles/r05_corr_ring.txt: one PERSISTENT workgroup per CU walking 4-5 items with the ring running
// across item boundaries and the band staged in the slot of the wave's last row — the band stores and the next item's first
// chunk overlap the products, but the workgroups of an XCD drift apart, neighbours stop meeting in L2 (1.17 GB from HBM instead
// of 0.90) and the kernel is slower, 218 against 206 us; odd row groups run bottom-up so that vertical neighbours ask for their
// shared rows at the same step: no fewer bytes.)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int RG_R = 8, RG_GW = 9, RG_NS = 24;       // 16 = RG_R + RG_GW - 1 f1 rows per chunk
constexpr int RG_SLOT = 3 * 32 * 64;                    // bytes per slot
constexpr int RG_SMEM = RG_NS * RG_SLOT;                // 147,456 B; the epilogue's band staging (8 x 24 x 81 floats) reuses it

__device__ __forceinline__ u32x4 asm_ld16(u32x4 rs, int voff) {
  u32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rs) : "memory");
  return v;
}

__global__ __launch_bounds__(512, 1) void corr_fwd_ring_kernel(const CorrPlParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  unsigned char* f1s = reinterpret_cast<unsigned char*>(lds);
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int l15 = lane & 15, q = lane >> 4;
  int b = xcd_remap(blockIdx.x, gridDim.x, 1);          // XCD-contiguous: site tiles fastest, then row groups, then samples
  const int ia = b % p.nA; b /= p.nA;
  const int ngr = (p.oh + RG_R - 1) / RG_R;
  const int gy = b % ngr;
  const int n = b / ngr;
  const int i0 = ia * 24;
  const int oy0 = gy * RG_R, oy = oy0 + wid;
  const int n1 = (n + p.shift) % p.B;
  const int ld2 = p.ld * 2;
  const int nchunk = p.C >> 5;
  const size_t recs = (((size_t)p.B * p.H * p.W - 1) * (size_t)p.ld + (size_t)p.C) * 2;
  u32x4 f0_rs[3], f1_rs[3];
#pragma unroll
  for (int pl = 0; pl < 3; pl++) {
    f0_rs[pl] = raw_rsrc(p.f0 + pl * p.ps, recs);
    f1_rs[pl] = raw_rsrc(p.f1 + pl * p.ps, recs);
  }
  const unsigned f1_addr = lds_addr(f1s);
  auto swz = [](int site) { return (0x72 >> (((site >> 2) & 3) * 2)) & 3; };   // {0, 2, 3, 1}[(site >> 2) & 3]
  // f0 fragments: row tile rt holds sites 16 rt + l15 (owned: < 24), granule q
  int a_off[2];
#pragma unroll
  for (int rt = 0; rt < 2; rt++) {
    const int s0 = 16 * rt + l15;
    const int xs = p.off + i0 + s0, ys = oy + p.off;
    const bool ok = s0 < 24 && (unsigned)xs < (unsigned)p.W && (unsigned)ys < (unsigned)p.H && oy < p.oh;
    a_off[rt] = ok ? ((n * p.H + ys) * p.W + xs) * ld2 + q * 16 : OOB_MARK;
  }
  // this wave's DMA units: rows (wid >> 1) + 4 k, k = 0..3, half wid & 1 of the row's 32 sites (16 sites x 64 B per plane):
  // lane i fills slot i & 3 of site 16 hh + (i >> 2) with source granule (i & 3) ^ swz(site)
  const int hh = wid & 1;
  int d_off[4];
  {
    const int site = 16 * hh + (lane >> 2);
    const int g = (lane & 3) ^ swz(site);
    const int xs = p.off + i0 - p.r + site;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int yy = oy0 + p.off - p.r + (wid >> 1) + 4 * k;
      const bool ok = (unsigned)xs < (unsigned)p.W && (unsigned)yy < (unsigned)p.H;
      d_off[k] = ok ? ((n1 * p.H + yy) * p.W + xs) * ld2 + g * 16 : OOB_MARK;
    }
  }
  // request row (wid >> 1) + 4 k of chunk cc (cbx = (16 cc) mod 24: slot of its row 0)
  auto request = [&](int k, int cc, int cbx) __attribute__((always_inline)) {
    int sl = cbx + (wid >> 1) + 4 * k;
    sl = sl >= RG_NS ? sl - RG_NS : sl;
    const unsigned d = f1_addr + (unsigned)(sl * RG_SLOT + hh * 1024);
    dma3(cc < nchunk ? d_off[k] + cc * 64 : OOB_MARK, f1_rs[0], f1_rs[1], f1_rs[2], d, d + 2048, d + 4096);
  };
  // f0 fragments: TWO register sets in turn (chunk c multiplies out of set c & 1 while the loads of chunk c + 1 land in the
  // other).  Round 5 carried ONE in-flight set across the loop back-edge: the compiler's phi copy (`v_mov_b64 v[108:131] <-
  // v[132:155]` at the END of the body) read the registers before any wait that covered the inline-asm loads — VGPR reads are
  // not interlocked with VMEM returns, the compiler knows nothing about loads issued by inline asm.  With
  // two named sets there is no loop-carried copy at all; each set still enters its chunk only through a wait tied to it.
  // tools/isa_load_hazard.py scans the assembly for any touch of a register with a load in flight.
  u32x4 fa[2][3], fb[2][3];
  auto request_a = [&](u32x4 (&dst)[2][3], int cc) __attribute__((always_inline)) {
#pragma unroll
    for (int rt = 0; rt < 2; rt++) {
      const int off = cc < nchunk ? a_off[rt] + cc * 64 : OOB_MARK;
#pragma unroll
      for (int pl = 0; pl < 3; pl++) dst[rt][pl] = asm_ld16(f0_rs[pl], off);
    }
  };
  f32x4 acc[RG_GW][3];
#pragma unroll
  for (int j = 0; j < RG_GW; j++)
#pragma unroll
    for (int pr = 0; pr < 3; pr++) acc[j][pr] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
  constexpr int prt[3] = {0, 0, 1}, pct[3] = {0, 1, 1};                   // product -> (f0 row tile, f1 column tile)
  // fragment of column tile ct: site 16 ct + l15 (the swizzle repeats every 16 sites: tile 1 = tile 0 + 1 KB)
  const unsigned char* b_rd = f1s + l15 * 64 + ((q ^ swz(l15)) << 4);

  // chunk 0 in the steady state's order: rows 0..7, fragments, rows 8..11, rows 12..15
  request(0, 0, 0); request(1, 0, 0);
  request_a(fa, 0);
  request(2, 0, 0);
  request(3, 0, 0);
  int cb = 0;                                          // (16 c) mod 24
  // one chunk: multiply out of `cur` (requested a chunk ago), request chunk c + 1's fragments into `nxt`
  auto chunk = [&](int c, u32x4 (&cur)[2][3], u32x4 (&nxt)[2][3]) __attribute__((always_inline)) {
    int cbn = cb + 16;
    cbn = cbn >= RG_NS ? cbn - RG_NS : cbn;
    // ---- step 0: rows 0..11 of this chunk and its fragments have landed (the step-8 request of the last chunk may still be out)
    asm volatile("s_waitcnt vmcnt(3)"
                 : "+v"(cur[0][0]), "+v"(cur[0][1]), "+v"(cur[0][2]), "+v"(cur[1][0]), "+v"(cur[1][1]), "+v"(cur[1][2])::"memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    s16x8 af[2][3];
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
      for (int pl = 0; pl < 3; pl++) af[rt][pl] = __builtin_bit_cast(s16x8, cur[rt][pl]);
    request(0, c + 1, cbn); request(1, c + 1, cbn);
    request_a(nxt, c + 1);
    s16x8 bf[2][2][3];                                 // [step parity][column tile][plane]
    auto rd = [&](int j, s16x8 (&f)[2][3]) __attribute__((always_inline)) {
      int sl = cb + wid + j;
      sl = sl >= RG_NS ? sl - RG_NS : sl;
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int pl = 0; pl < 3; pl++) f[ct][pl] = *reinterpret_cast<const s16x8*>(b_rd + sl * RG_SLOT + ct * 1024 + pl * 2048);
    };
    rd(0, bf[0]);
#pragma unroll
    for (int j = 0; j < RG_GW; j++) {
      if (j == 4 || j == 8) {
        if (j == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        request(j == 4 ? 2 : 3, c + 1, cbn);
      }
      if (j + 1 < RG_GW) rd(j + 1, bf[(j + 1) & 1]);
#pragma unroll
      for (int tt = 0; tt < 6; tt++)
#pragma unroll
        for (int pr = 0; pr < 3; pr++)
          acc[j][pr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[prt[pr]][ta[tt]]),
                                                              __builtin_bit_cast(bf16x8, bf[j & 1][pct[pr]][tb[tt]]), acc[j][pr], 0, 0, 0);
    }
    cb = cbn;
  };
  int c = 0;
#pragma unroll 1
  for (; c + 1 < nchunk; c += 2) {
    chunk(c, fa, fb);
    chunk(c + 1, fb, fa);
  }
  if (c < nchunk) chunk(c, fa, fb);
  // The requests past the last chunk are out of range (zeros): let them land — the ring's rows before its LDS is reused, and
  // the fragment loads before their registers are: a destination the compiler considers dead would be handed to the next
  // value while the load is still in flight (it did: the scanner found the last chunk's B fragments in those registers), so
  // both sets stay live up to this wait.
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fb[0][0]),
                 "+v"(fb[0][1]), "+v"(fb[0][2]), "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[1][2])::"memory");
  __syncthreads();                                     // the ring is dead: its LDS becomes the band staging
  // acc[j][pr][e] of lane (column l15, quarter q) is the product of f0 site s0 = 16 rt + 4 q + e and staged site s1 = 16 ct + l15
  // (x = i0 - 4 + s1): band offset index s1 - s0.  Staging [site][displacement row][offset], one area per wave.
  constexpr int g2 = RG_GW * RG_GW;
  float* stg = reinterpret_cast<float*>(lds) + wid * (24 * g2);
#pragma unroll
  for (int pr = 0; pr < 3; pr++)
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int s0 = 16 * prt[pr] + 4 * q + e;
      const int oi = 16 * pct[pr] + l15 - s0;
      if (s0 < 24 && oi >= 0 && oi < RG_GW) {            // (one predicate per (product, e): the displacement rows inside)
        float* w = stg + s0 * g2 + oi;
#pragma unroll
        for (int j = 0; j < RG_GW; j++) w[j * RG_GW] = acc[j][pr][e];
      }
    }
  // (wave-private: the same wave reads it back — LDS operations of one wave complete in order)
  if (oy < p.oh) {
    const float cf = (float)p.C;
    float* orow = p.out + ((size_t)n * p.oh + oy) * p.ow * p.ld_out;
    const int cnt = min(24, p.ow - i0) * g2;             // the tile's band entries of this row
    float* dst = orow + (size_t)i0 * p.ld_out;
    if (p.ld_out == g2 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
      // dense output rows: the tile's entries are one contiguous run — 16 bytes per lane (the stores are issue-bound)
      const f32x4* src4 = reinterpret_cast<const f32x4*>(stg);
      f32x4* dst4 = reinterpret_cast<f32x4*>(dst);
      for (int v = lane; v < (cnt >> 2); v += 64) {
        f32x4 x = src4[v];
        x[0] /= cf; x[1] /= cf; x[2] /= cf; x[3] /= cf;
        dst4[v] = x;
      }
      for (int t = (cnt & ~3) + lane; t < cnt; t += 64) dst[t] = stg[t] / cf;
    } else {
      const float inv_g2 = 1.0f / (float)g2;
      for (int idx = lane; idx < cnt; idx += 64) {
        const int li = (int)(((float)idx + 0.5f) * inv_g2), rem = idx - li * g2;   // exact at these sizes
        dst[(size_t)li * p.ld_out + rem] = stg[idx] / cf;
      }
    }
  }
}

// --------------------------------------------------------------------------------------- wide-band forward by DMA
// FlowNetC's own cost volume (r = 10: the band of a 32-site tile covers most of its Gram and reaches into the neighbour
// tiles).  corr_fwd_pl_kernel streams the f1 fragments from L2 at 32 cache lines per load instruction, ~64 cycles of the
// CU's address path each — 3 blocks x 1008 such loads per CU are ~90 of its ~120 us at the step's shape.  Same remedy as the
// narrow-band kernel above: K split over the waves (wave w: channels 64 w .. + 63, f0 fragments in registers), f1 tiles by
// LDS-DMA in whole lines into a wave-private 12 KB tile, the next tile requested as soon as this one's fragments are in
// registers.  A block owns a PAIR of output rows (oy, oy + s2): f1 row m is displacement row m of the first and m - 1 of the
// second, so every tile feeds two Grams (48 MFMAs per DMA round trip).  The band is too large to keep per wave for the
// whole block (2r+1 Grams x 2.7 KB), so each step's partial Grams are exchanged at once: every wave stores its 32 x 32
// partials (LDS, 4 KB per Gram), barrier, the block sums the waves' partials in a fixed order and writes the band entries
// of this (displacement row, column tile), barrier.  Column tiles / f1 rows outside the image are not multiplied; their
// band entries are zero-filled up front, while the first tile is in flight.
// (Measured and dropped: three rows per block with double-buffered tiles, so that the step's shape is exactly one block per
// CU instead of 384 pair blocks on 256 CUs: 117 us against 108; issuing the band stores before the next step's MFMAs instead
// of right before its s_waitcnt; one batch of LDS reads per row before the stores: both neutral to worse.  Ablation of the
// pair kernel, 113 us on that box: no band exchange 74, no MFMAs 91, neither 55, nothing but fragments and barriers 50 — the
// per-block prologue and the serial chain of a step, not a throughput limit.)
constexpr int WB_ROWS = 2;                             // output rows per block
__global__ __launch_bounds__(256, 2) void corr_fwd_wb_kernel(const CorrPlParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  constexpr int TILE = 3 * 32 * 64;                 // elements per wave: 3 planes x 32 sites x 64 channels
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw = blockDim.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  // work order: XCD-contiguous; row groups fastest, samples slowest (the step's 8 samples: one per XCD)
  int b = xcd_remap(blockIdx.x, gridDim.x, 1);
  const int npr = ((p.oh + p.s2 - 1) / p.s2 + WB_ROWS - 1) / WB_ROWS;
  const int pr = b % npr; b /= npr;
  const int ry = b % p.s2; b /= p.s2;
  const int ia = b % p.nA; b /= p.nA;
  const int q = b % p.s2; b /= p.s2;
  const int n = b;
  const int oy = ry + p.s2 * WB_ROWS * pr;
  if (oy >= p.oh) return;
  const int nv = min(WB_ROWS, (p.oh - oy + p.s2 - 1) / p.s2);     // rows of the group that exist
  const int i0 = ia * 32, c0 = wid * 64;
  const int n1 = (n + p.shift) % p.B;
  const int y0 = oy + p.off;
  const int ld2 = p.ld * 2;
  const size_t recs = (((size_t)p.B * p.H * p.W - 1) * (size_t)p.ld + (size_t)p.C) * 2;
  u32x4 f1_rs[3];
  __amdgpu_buffer_rsrc_t f0_rs[3];
#pragma unroll
  for (int pl = 0; pl < 3; pl++) {
    f0_rs[pl] = make_rsrc(p.f0 + pl * p.ps, recs);
    f1_rs[pl] = raw_rsrc(p.f1 + pl * p.ps, recs);
  }
  unsigned short* tile = lds + wid * TILE;
  const unsigned tile_addr = lds_addr(tile);
  float* part = reinterpret_cast<float*>(lds + nw * TILE);     // [wave][output row][32 x 32]
  float* mypart = part + wid * (WB_ROWS * 1024);

  // f0 fragments of the rows, straight to registers (in flight together with the first f1 tile)
  s16x8 af[WB_ROWS][4][3];
  {
    const int xs = q + p.off + p.s2 * (i0 + l31);
    const bool okx = (unsigned)xs < (unsigned)p.W;
#pragma unroll
    for (int rw = 0; rw < WB_ROWS; rw++) {
      const int yy = y0 + rw * p.s2;
      const bool ok = okx && (unsigned)yy < (unsigned)p.H && rw < nv;
      const int voff = ok ? ((n * p.H + yy) * p.W + xs) * ld2 + (c0 + h * 8) * 2 : OOB_MARK;
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int pl = 0; pl < 3; pl++) af[rw][u][pl] = __builtin_bit_cast(s16x8, buf_ld16(f0_rs[pl], voff + u * 32));
    }
  }
  const int d_site = lane >> 3, d_slot = lane & 7;  // DMA lane mapping: see corr_fwd_nb_kernel
  auto issue_tile = [&](int m, int t) {
    const int yy = y0 + p.s2 * (m - p.r), k0 = i0 + 32 * t;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int site = 8 * j + d_site;
      const int g = d_slot ^ ((site >> 1) & 7);
      const int xs = q + p.off + p.s2 * (k0 + site);
      const bool ok = (unsigned)xs < (unsigned)p.W;
      const int voff = ok ? ((n1 * p.H + yy) * p.W + xs) * ld2 + (c0 + g * 8) * 2 : OOB_MARK;
      const unsigned d = tile_addr + (unsigned)(j * 1024);
      dma3(voff, f1_rs[0], f1_rs[1], f1_rs[2], d, d + 32 * 64 * 2, d + 2 * 32 * 64 * 2);
    }
  };
  const int M = p.gw + nv - 1;                      // f1 rows y0 + s2 (m - r), m = 0 .. M-1: row m is displacement row m - rw of row rw
  auto live = [&](int m, int t) -> bool {           // has f1 row m, column tile t a site inside the image?
    const int yy = y0 + p.s2 * (m - p.r), k0 = i0 + 32 * t;
    return (unsigned)yy < (unsigned)p.H && q + p.off + p.s2 * k0 < p.W && q + p.off + p.s2 * (k0 + 31) >= 0;
  };
  auto next_live = [&](int& m, int& t) -> bool {    // advance (m, t) to the next live tile
    for (;;) {
      if (++t > p.T) { t = -p.T; m++; }
      if (m >= M) return false;
      if (live(m, t)) return true;
    }
  };
  const float cf = (float)p.C, rcf = 1.0f / cf;
  const bool pow2 = (p.C & (p.C - 1)) == 0;         // then x * (1/C) == x / C exactly
  const float inv_gw = 1.0f / (float)p.gw;
  // The band entries of f1 row m: those whose column lies in tile t are summed from the waves' partial Grams (t = tsum);
  // those of DEAD column tiles (no site inside the image: nothing is multiplied) are zero-filled when `zeros` is set — by the
  // first live tile of the row, or, for a row with no live tile at all, by the pass below (tsum = an impossible tile).
  // a thread's band entries idx = tid + k * blockDim (li, oi) are the same in every step: partial-Gram index, column tile
  // and output offset once (NE covers 32 x 41 entries at 256 threads; the host checks it)
  constexpr int NE = 6;
  int e_src[NE], e_dst[NE], e_tj[NE];
#pragma unroll
  for (int k = 0; k < NE; k++) {
    const int idx = threadIdx.x + k * blockDim.x;
    const int li = (int)(((float)idx + 0.5f) * inv_gw), oi = idx - li * p.gw;     // exact in fp32 at these sizes
    const int jabs = li + oi - p.r;                                                // column relative to the tile's first site
    const int ox = q + p.s2 * (i0 + li);
    const bool ok = idx < 32 * p.gw && ox < p.ow;
    e_src[k] = ok ? li * 32 + (jabs & 31) : 0;
    e_tj[k] = ok ? (jabs >> 5) : 0x7ffe;                                           // (matches no tile)
    e_dst[k] = ox * p.ld_out + oi;
  }
  // (Code size matters here: a block runs each instruction only ~2r+3 times, so the first pass through the kernel is
  // instruction-fetch bound — measured floor with every phase switched off: 41 us at 27 KB of code.  write_band is therefore
  // instantiated ONCE, its row loop stays rolled, and the zero-fill of rows outside the image is a plain rolled loop.)
  auto write_band = [&](int m, int tsum, bool zeros) {
#pragma unroll 1
    for (int rw = 0; rw < WB_ROWS; rw++) {
      const int pi = m - rw;
      if (pi < 0 || pi >= p.gw || rw >= nv) continue;
      float* orow = p.out + ((size_t)n * p.oh + oy + rw * p.s2) * p.ow * p.ld_out + pi * p.gw;
#pragma unroll
      for (int k = 0; k < NE; k++) {
        if (k * (int)blockDim.x >= 32 * p.gw) break;
        const int tj = e_tj[k];
        float sum = 0.f;
        if (tj == tsum) {
          const float* src = part + rw * 1024 + e_src[k];
          sum = src[0];
          for (int w = 1; w < nw; w++) sum += src[w * (WB_ROWS * 1024)];
          sum = pow2 ? sum * rcf : sum / cf;
        } else if (tj == 0x7ffe || !zeros || live(m, tj)) {
          continue;
        }
        orow[e_dst[k]] = sum;
      }
    }
  };

  // the tile pipeline: `cur` is multiplied in this step, `nxt` is in flight
  int m = 0, t = -p.T - 1;
  bool cur_ok = next_live(m, t), nxt_ok = false;
  int cur_m = m, cur_t = t, nxt_m = 0, nxt_t = 0;
  if (cur_ok) issue_tile(cur_m, cur_t);
#pragma unroll 1
  for (int dm = 0; dm < M; dm++) {                  // f1 rows outside the image: all their entries are zero
    bool any = false;
    for (int dt = -p.T; dt <= p.T; dt++) any = any || live(dm, dt);
    if (any) continue;
#pragma unroll 1
    for (int rw = 0; rw < nv; rw++) {
      const int pi = dm - rw;
      if (pi < 0 || pi >= p.gw) continue;
      float* orow = p.out + ((size_t)n * p.oh + oy + rw * p.s2) * p.ow * p.ld_out + pi * p.gw;
#pragma unroll 1
      for (int idx = threadIdx.x; idx < 32 * p.gw; idx += blockDim.x) {
        const int li = (int)(((float)idx + 0.5f) * inv_gw), oi = idx - li * p.gw;
        const int ox = q + p.s2 * (i0 + li);
        if (ox < p.ow) orow[(size_t)ox * p.ld_out + oi] = 0.f;
      }
    }
  }
  int prv_m = -1, prv_t = 0, pp_m = -1;             // the step whose partial Grams are in LDS, and the f1 row before it
  constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
  // A step: wait for its tile - fragments -> registers - the PREVIOUS step's band (LDS partials -> global stores) - request
  // a tile - barrier - 24 MFMAs per row, partial Grams -> LDS - barrier; one more pass writes the last band.
#pragma unroll 1
  for (;;) {
    const int mc = cur_m, tc = cur_t;
    s16x8 bf[4][3];
    if (cur_ok) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this step's tile has landed
      __builtin_amdgcn_sched_barrier(0);
      const unsigned short* tl = tile;
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int pl = 0; pl < 3; pl++)
          bf[u][pl] = *reinterpret_cast<const s16x8*>(tl + pl * (32 * 64) + l31 * 64 + (((2 * u + h) ^ ((l31 >> 1) & 7)) << 3));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // fragments in registers: the tile may be overwritten
      __builtin_amdgcn_sched_barrier(0);
    }
    if (prv_m >= 0) write_band(prv_m, prv_t, prv_m != pp_m);
    if (!cur_ok) break;
    __builtin_amdgcn_sched_barrier(0);
    nxt_ok = next_live(m, t);
    nxt_m = m; nxt_t = t;
    if (nxt_ok) issue_tile(nxt_m, nxt_t);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // (the previous partials have been read)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                    // raw barriers: a __syncthreads() would also wait for the tiles in flight
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rw = 0; rw < WB_ROWS; rw++) {
      const int pi = mc - rw;
      if (pi < 0 || pi >= p.gw || rw >= nv) continue;              // (block-uniform)
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; e++) acc[e] = 0.f;
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int tt = 0; tt < 6; tt++)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[rw][u][ta[tt]]),
                                                        __builtin_bit_cast(bf16x8, bf[u][tb[tt]]), acc, 0, 0, 0);
      // acc[e] of lane (column l31, half h) is G[li][l31], li = (e&3) + 8*(e>>2) + 4h
#pragma unroll
      for (int e = 0; e < 16; e++) mypart[rw * 1024 + ((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + l31] = acc[e];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    pp_m = prv_m; prv_m = mc; prv_t = tc;
    cur_ok = nxt_ok; cur_m = nxt_m; cur_t = nxt_t;
  }
}


// --------------------------------------------------------------------------------- wide-band forward, a wave pair per output row
// corr_fwd_wb_kernel splits K over its four waves, so every 32 x 32 Gram exists as four partial Grams that have to meet in LDS
// in a DIFFERENT element order than the accumulators hold them (store 4 KB per wave and row, barrier, scalar band sums,
// barrier): 44 of its 108 us (profiles/r03_corr_ablation.txt).  Here the work is cut by OUTPUT ROW instead: a workgroup owns
// RW_ROWS rows of one row class, oy + s2 k, and streams the f1 rows m = 0 .. gw + RW_ROWS - 2 they share ONCE for all of them
// (row m is displacement row m - k of output row k: 6 tile loads per output row at the step's shape, 10.5 before) — whole tiles
// of 32 sites x C channels x 3 planes, double-buffered, LDS-DMA in whole lines, every wave moving its share.  A row belongs to
// a PAIR of waves, each holding the f0 fragments of half the channels in registers (96 VGPRs at C = 256).  A Gram is then two
// partials in the SAME register layout: one wave of the pair parks its 16 accumulators in a private 4 KB slot (four 16-byte
// stores per lane), and after the step's single barrier — the one that also hands the tile buffers over — the other adds them
// with four 16-byte reads.  The finishing role alternates inside the pair by step, so both waves carry the same load and a slot
// is rewritten only two barriers after it was read; a + b == b + a, so the alternation does not show in the bits.
// The band leaves the registers BAND-major: for accumulator e (Gram rows li = lc(e) + 4 h) lane l31 of a half-wave takes
// displacement o = l31, i.e. column li + l31 - r, from its neighbour by ds_bpermute and the half-wave writes the 2r+1 entries of
// one output pixel as ONE run — including the zeros of columns that fall into a column tile outside the image (written by the
// row's first live tile) and the all-zero bands of f1 rows outside the image (a step without tile and products), so there is
// no separate zero-fill pass.  Which accumulators of a lane belong to which column tile depends on the lane only: three 16-bit
// masks, computed once.  The previous step's band is finished INSIDE the straight-line block of this step's products (stores
// through a bounds-checked buffer descriptor instead of branches), so its address arithmetic and stores run under the MFMAs.
constexpr int RW_ROWS = 4;
template <int CH>                                      // 64-channel chunks per K half: C = 128 CH
__global__ __launch_bounds__(512) void corr_fwd_rw_kernel(const CorrPlParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  CT_DECL;
  constexpr int CHUNK = 3 * 32 * 64;                   // elements of a 64-channel chunk of a tile: 3 planes x 32 sites x 64 channels
  constexpr int NCH = 2 * CH;                          // chunks of a tile
  constexpr int TILE = NCH * CHUNK;
  constexpr int NU = 4 * CH;                           // K16 steps of a wave
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int rw = wid >> 1, kh = wid & 1;               // output row of the group, K half
  const int l31 = lane & 31, h = lane >> 5;
  int b = xcd_remap(blockIdx.x, gridDim.x, 1);         // row groups fastest, samples slowest (the step's 8 samples: one per XCD)
  const int npr = ((p.oh + p.s2 - 1) / p.s2 + RW_ROWS - 1) / RW_ROWS;
  const int pr = b % npr; b /= npr;
  const int ry = b % p.s2; b /= p.s2;
  const int ia = b % p.nA; b /= p.nA;
  const int q = b % p.s2; b /= p.s2;
  const int n = b;
  const int oy = ry + p.s2 * RW_ROWS * pr;
  if (oy >= p.oh) return;
  const int nv = min(RW_ROWS, (p.oh - oy + p.s2 - 1) / p.s2);      // rows of the group that exist
  const int i0 = ia * 32;
  const int n1 = (n + p.shift) % p.B;
  const int y0 = oy + p.off;
  const int ld2 = p.ld * 2;
  const size_t recs = (((size_t)p.B * p.H * p.W - 1) * (size_t)p.ld + (size_t)p.C) * 2;
  u32x4 f1_rs[3];
#pragma unroll
  for (int pl = 0; pl < 3; pl++) f1_rs[pl] = raw_rsrc(p.f1 + pl * p.ps, recs);
  const __amdgpu_buffer_rsrc_t out_rs = make_rsrc(p.out, (((size_t)p.B * p.oh * p.ow - 1) * (size_t)p.ld_out + (size_t)p.gw * p.gw) * 4);
  const unsigned tiles_addr = lds_addr(lds);
  float* slots = reinterpret_cast<float*>(lds + 2 * TILE);         // [wave][4][64 lanes][4]: a wave's accumulators as they lie
  float* myslot = slots + wid * 1024 + lane * 4;
  const float* peer = slots + (wid ^ 1) * 1024 + lane * 4;

  // the wave's f0 fragments: row rw, channels 64 CH kh + 16 u + 8 h ..+7, straight to registers (in flight together with the
  // first f1 tile).  (Measured: the same rows by whole-line DMA into the two tile buffers, two rows at a time, fragments then
  // read from LDS — 8 x fewer cache-line look-ups — is no faster: the prologue is the cold first fetch of 37 MB by all
  // workgroups at once, and the two barrier-separated rounds cost what the look-ups saved: 62.2 against 61.0 us.)
  s16x8 af[NU][3];
  {
    const int xs = q + p.off + p.s2 * (i0 + l31);
    const int yy = y0 + rw * p.s2;
    const bool ok = (unsigned)xs < (unsigned)p.W && (unsigned)yy < (unsigned)p.H && rw < nv;
    const int voff = ok ? ((n * p.H + yy) * p.W + xs) * ld2 + (kh * 64 * CH + h * 8) * 2 : OOB_MARK;
    const __amdgpu_buffer_rsrc_t f0_ld[3] = {make_rsrc(p.f0, recs), make_rsrc(p.f0 + p.ps, recs), make_rsrc(p.f0 + 2 * p.ps, recs)};
#pragma unroll
    for (int u = 0; u < NU; u++)
#pragma unroll
      for (int pl = 0; pl < 3; pl++) af[u][pl] = __builtin_bit_cast(s16x8, buf_ld16(f0_ld[pl], voff + u * 32));
  }
  // a tile = NCH chunks x 4 groups of 8 sites x 3 planes of 1 KB DMA instructions; wave w moves NCH / 2 (chunk, group) units.
  const int d_site = lane >> 3, d_slot = lane & 7;     // DMA lane mapping: see corr_fwd_nb_kernel
  int d_xs[NCH / 2], d_px[NCH / 2];
#pragma unroll
  for (int i = 0; i < NCH / 2; i++) {
    const int un = wid * (NCH / 2) + i, ch = un >> 2, j = un & 3;
    const int site = 8 * j + d_site;
    d_px[i] = (ch * 64 + (d_slot ^ ((site >> 1) & 7)) * 8) * 2;
    d_xs[i] = q + p.off + p.s2 * (i0 + site);
  }
  auto tile_voff = [&](int m, int t, bool real, int (&v)[NCH / 2]) __attribute__((always_inline)) {       // a lane's source offsets of f1 row m, column tile t
    const int row = ((n1 * p.H + y0 + p.s2 * (m - p.r)) * p.W) * ld2;
#pragma unroll
    for (int i = 0; i < NCH / 2; i++) {
      const int xs = d_xs[i] + t * (32 * p.s2);
      const int in = row + xs * ld2 + d_px[i];
      const bool ok = (int)real & (int)((unsigned)xs < (unsigned)p.W);
      v[i] = ok ? in : OOB_MARK;
    }
  };
  auto unit_lds = [&](int i) __attribute__((always_inline)) -> unsigned {             // LDS byte offset of unit i inside a tile buffer
    const int un = wid * (NCH / 2) + i;
    return (unsigned)((un >> 2) * CHUNK * 2 + (un & 3) * 1024);
  };
  const int M = p.gw + nv - 1;                         // f1 rows y0 + s2 (m - r): row m is displacement row m - k of output row k
  auto col_live = [&](int t) __attribute__((always_inline)) -> bool {                 // has column tile t a site inside the image?
    const int k0 = i0 + 32 * t;
    return q + p.off + p.s2 * k0 < p.W && q + p.off + p.s2 * (k0 + 31) >= 0;
  };
  // column tiles -1, 0, 1 (the host admits r <= 15: a band reaches one tile to either side); the live ones in order: lt[0 .. nl)
  const int dead_cols = (p.T >= 1 && col_live(-1) ? 0 : 1) | (col_live(0) ? 0 : 2) | (p.T >= 1 && col_live(1) ? 0 : 4);
  const int nl = 3 - __builtin_popcount(dead_cols);
  const int lt0 = !(dead_cols & 1) ? -1 : !(dead_cols & 2) ? 0 : 1;
  const int lt1 = !(dead_cols & 1) ? (!(dead_cols & 2) ? 0 : 1) : 1;
  // (the selects below are written over plain values and bitwise conditions on purpose: a `?:` or `&&` with work in its arms
  // reaches the back end as a branch, and a branch inside a step's block ends the region the MFMAs can be scheduled across)
  auto row_live = [&](int m) __attribute__((always_inline)) -> int { return (int)(nl > 0) & (int)((unsigned)(y0 + p.s2 * (m - p.r)) < (unsigned)p.H); };
  // Steps: one per (live f1 row, live column tile); ONE for an f1 row that is multiplied with nothing (t = 2: matches no tile).
  // (m, j) -> the step after it, branch-free: it is computed inside the previous step's block, under its MFMAs.
  auto step_after = [&](int m, int j, int& m2, int& j2, int& t2, bool& ok2) __attribute__((always_inline)) {
    const int cnt = row_live(m) ? nl : 1;
    const int wrap = (int)(j + 1 >= cnt);
    m2 = m + wrap;
    j2 = (j + 1) & (wrap - 1);
    ok2 = m2 < M;
    const int tlive = (-(int)(j2 == 0) & lt0) | (-(int)(j2 == 1) & lt1) | (-(int)(j2 >= 2) & 1);
    t2 = row_live(m2) ? tlive : 2;
  };

  // Accumulator e of a lane finishes band entry (row li = lc(e) + 4 h, displacement l31): column li + l31 - r, tile (that) >> 5.
  // msk_lo / mid / hi (tile -1 / 0 / 1) bit e: the entry exists (pixel inside the output, l31 < gw) and its column lies in that tile.
  int msk_lo = 0, msk_mid = 0, msk_hi = 0;             // (three scalars, not an array: a select between array elements becomes a scratch load)
#pragma unroll
  for (int e = 0; e < 16; e++) {
    const int li = (e & 3) + 8 * (e >> 2) + 4 * h;
    const int tj = (li + l31 - p.r) >> 5;
    const int bit = l31 < p.gw && q + p.s2 * (i0 + li) < p.ow ? 1 << e : 0;
    msk_lo |= tj == -1 ? bit : 0;
    msk_mid |= tj == 0 ? bit : 0;
    msk_hi |= tj == 1 ? bit : 0;
  }
  const int msk_all = msk_lo | msk_mid | msk_hi;
  const int msk_dead = (dead_cols & 1 ? msk_lo : 0) | (dead_cols & 2 ? msk_mid : 0) | (dead_cols & 4 ? msk_hi : 0);
  int perm[16];                                        // ds_bpermute byte address = 4 x the source lane: column li + l31 - r (mod 32) of the lane's own half
#pragma unroll
  for (int e = 0; e < 16; e++) perm[e] = ((4 * ((e & 3) + 8 * (e >> 2) + 4 * h + l31 - p.r)) & 124) | (h << 7);
  const int lane_out = ((q + p.s2 * (i0 + 4 * h)) * p.ld_out + l31) * 4;      // byte offset of (row 4 h, displacement l31) in a band
  const int lc_bytes = p.s2 * p.ld_out * 4;                                   // one Gram row further = one output pixel of the class
  const int row_out = (((n * p.oh + oy + rw * p.s2) * p.ow) * p.ld_out - rw * p.gw) * 4;   // + 4 gw m: band (row rw, displacement row m - rw)
  const float rcf = 1.0f / (float)p.C;                  // C is a power of two here: x * (1/C) == x / C exactly

  // ---- the step pipeline's registers: cur is multiplied in this step, nxt's tile is requested during it
  int cur_m = 0, cur_t = row_live(0) ? lt0 : 2, cur_buf = 0, nbuf = 0;
  bool cur_ok = M > 0;
  if (cur_ok && cur_t != 2) {
    int v[NCH / 2];
    tile_voff(cur_m, cur_t, true, v);
#pragma unroll
    for (int i = 0; i < NCH / 2; i++) {
      const unsigned d = tiles_addr + unit_lds(i);
      dma3(v[i], f1_rs[0], f1_rs[1], f1_rs[2], d, d + 32 * 64 * 2, d + 2 * 32 * 64 * 2);
    }
    nbuf = 1;
  }
  int nxt_m, nxt_j, nxt_t, nxt_buf;
  bool nxt_ok;
  step_after(cur_m, 0, nxt_m, nxt_j, nxt_t, nxt_ok);
  int dvoff[NCH / 2];
  tile_voff(nxt_m, nxt_t, nxt_ok && nxt_t != 2, dvoff);
  unsigned dbase = tiles_addr + (unsigned)(nbuf * TILE * 2);
  nxt_buf = nbuf;
  nbuf ^= nxt_ok && nxt_t != 2 ? 1 : 0;
  int take = 0, any = 0, out_soff = 0;                 // of the Gram being finished: set by the step that computed it

  constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; e++) acc[e] = 0.f;
  // One step's straight-line block.  FIN: finish the previous step's Gram held in acc (add the partner's half, band-major
  // stores; take / any: the lane's accumulators that carry a value / are stored) — in the first K16 steps, so that the stores
  // have drained when the step ends.  MF: this step's products into a fresh acc, fragments one K16 step ahead.  Always: the
  // wave's share of the NEXT tile, one 1 KB DMA instruction per K16 step (in one burst right after the barrier the 48
  // instructions of a workgroup queue up in front of the address unit), and the bookkeeping of the step after next — which
  // step, where its tile comes from, which accumulators the band of THIS step's Gram takes — about 150 scalar and vector
  // instructions that cost 1200-1700 cycles per step when they sat between the barrier and the first MFMA
  // (tools/debug/corr_phase_trace.py): inside the block they issue between the MFMAs.  dvoff = the out-of-range mark when
  // there is no next tile: the DMA then only zero-fills a buffer nobody reads.
  int nn_m, nn_j, nn_t, nn_buf, dvoff2[NCH / 2], take2, any2, out_soff2;
  bool nn_ok;
  unsigned dbase2;
  auto block = [&](auto mf_tag, auto fin_tag) __attribute__((always_inline)) {
    constexpr bool MF = decltype(mf_tag)::value, FIN = decltype(fin_tag)::value;
    const unsigned short* tl = lds + cur_buf * TILE + kh * CH * CHUNK + l31 * 64;
    s16x8 bf[2][3];
    auto rd = [&](int u, s16x8 (&o)[3]) __attribute__((always_inline)) {
#pragma unroll
      for (int pl = 0; pl < 3; pl++)
        o[pl] = *reinterpret_cast<const s16x8*>(tl + (u >> 2) * CHUNK + pl * (32 * 64) + (((2 * (u & 3) + h) ^ ((l31 >> 1) & 7)) << 3));
    };
    if (MF) rd(0, bf[0]);
    f32x16 prev;
    if (FIN) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float4 o = *reinterpret_cast<const float4*>(peer + j * 256);
        prev[4 * j] = acc[4 * j] + o.x; prev[4 * j + 1] = acc[4 * j + 1] + o.y;
        prev[4 * j + 2] = acc[4 * j + 2] + o.z; prev[4 * j + 3] = acc[4 * j + 3] + o.w;
      }
    }
    if (MF) {
#pragma unroll
      for (int e = 0; e < 16; e++) acc[e] = 0.f;
    }
    // the step after next and its tile; the band masks of this step's Gram
    step_after(nxt_m, nxt_j, nn_m, nn_j, nn_t, nn_ok);
    const bool nn_real = (int)nn_ok & (int)(nn_t != 2);
    tile_voff(nn_m, nn_t, nn_real, dvoff2);
    dbase2 = tiles_addr + (unsigned)(nbuf * TILE * 2);
    nn_buf = nbuf;
    nbuf ^= nn_real ? 1 : 0;
    // (masks instead of `c ? a : b`: over two captured variables that is an lvalue — a select between ADDRESSES, which keeps
    // every local of the kernel in scratch)
    take2 = (-(int)(cur_t == -1) & msk_lo) | (-(int)(cur_t == 0) & msk_mid) | (-(int)(cur_t == 1) & msk_hi);
    any2 = take2 | (-(int)(cur_t == lt0) & msk_dead) | (-(int)(cur_t == 2) & msk_all);
    out_soff2 = row_out + cur_m * (p.gw * 4);
#pragma unroll
    for (int u = 0; u < NU; u++) {
      if (MF && u + 1 < NU) rd(u + 1, bf[(u + 1) & 1]);
      if (u >= 1 && u <= 3 * (NCH / 2)) {
        const int i = (u - 1) / 3, pl = (u - 1) % 3;
        dma1(dvoff[i], f1_rs[pl], dbase + unit_lds(i) + pl * (32 * 64 * 2));
      }
      if (MF) {
#pragma unroll
        for (int tt = 0; tt < 6; tt++)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[u][ta[tt]]),
                                                        __builtin_bit_cast(bf16x8, bf[u & 1][tb[tt]]), acc, 0, 0, 0);
      }
      if (FIN && u < 4) {
#pragma unroll
        for (int ee = 0; ee < 4; ee++) {
          const int e = u * 4 + ee, lc = (e & 3) + 8 * (e >> 2);
          const float v = __int_as_float(__builtin_amdgcn_ds_bpermute(perm[e], __float_as_int(prev[e])));
          const int tm = (take << (31 - e)) >> 31, am = (any << (31 - e)) >> 31;          // bit e as 0 / -1: bfe, and, bfe, bfi
          const unsigned val = __float_as_uint(v * rcf) & (unsigned)tm;
          const int voff = (lane_out & am) | (OOB_MARK & ~am);
          __builtin_amdgcn_raw_buffer_store_b32(val, out_rs, voff, out_soff + lc * lc_bytes, 0);
        }
      }
    }
  };
  using T_ = std::true_type;
  using F_ = std::false_type;

  bool fin_prev = false;                               // this wave holds the finishing half of the previous step's Gram
  int step = 0;
#pragma unroll 1
  for (;;) {
    CT_STAMP(step == 0 ? 0 : 4);                       // 0: prologue; 4: tail of the previous step
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the current tile has landed
    __builtin_amdgcn_sched_barrier(0);
    CT_STAMP(1);                                       // 1: waiting for the tile
    __builtin_amdgcn_s_barrier();                      // ... everybody's; the other buffer and the slots written last step are free / ready
    __builtin_amdgcn_sched_barrier(0);
    CT_STAMP(2);                                       // 2: barrier
    const int pi = cur_m - rw;
    const bool mine = cur_ok && pi >= 0 && pi < p.gw && rw < nv;     // (uniform in the pair)
    const bool mf = mine && cur_t != 2;
    __builtin_amdgcn_sched_barrier(0);
    CT_STAMP(3);                                       // 3: dispatch
    if (mf) CT_COUNT(7);
    if (mf) {
      if (fin_prev) block(T_{}, T_{});
      else block(T_{}, F_{});
    } else {
      if (fin_prev) block(F_{}, T_{});
      else block(F_{}, F_{});
    }
    __builtin_amdgcn_sched_barrier(0);
    CT_STAMP(mf ? (fin_prev ? 5 : 6) : 4);             // 5: products + finish, 6: products only
    if (!cur_ok) break;
    if (mine && !mf) {
#pragma unroll
      for (int e = 0; e < 16; e++) acc[e] = 0.f;       // an f1 row outside the image: the band is zero
    }
    fin_prev = mine && ((step + rw) & 1) == kh;
    if (mine && !fin_prev) {
#pragma unroll
      for (int j = 0; j < 4; j++)
        *reinterpret_cast<float4*>(myslot + j * 256) = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // tile reads and slot stores done before the next barrier
    __builtin_amdgcn_sched_barrier(0);
    take = take2; any = any2; out_soff = out_soff2;
    cur_ok = nxt_ok; cur_m = nxt_m; cur_t = nxt_t; cur_buf = nxt_buf;
    nxt_ok = nn_ok; nxt_m = nn_m; nxt_j = nn_j; nxt_t = nn_t; nxt_buf = nn_buf;
#pragma unroll
    for (int i = 0; i < NCH / 2; i++) dvoff[i] = dvoff2[i];
    dbase = dbase2;
    step++;
  }
  CT_FLUSH;
}


// ------------------------------------------------------------------------------------------------ backward from planes
// out[site][c] += Band(dOut)[site][k] * F[k][c] (correlation_mfma.hip: corr_bwd_b3_kernel) with the FEATURE operand taken from
// its bf16 planes instead of being split in registers: per iteration a wave's 32 contracted sites x 64 channels x 3 planes
// (12 KB) go HBM/L2 -> a wave-private LDS tile by LDS-DMA (no staging registers, no conversion work), and the MFMA B
// fragments — 8 consecutive SITES of one channel — come out through ds_read_b64_tr_b16, exactly like the filter-gradient
// kernels of conv_planes.hip.  The band operand (scattered dOut values) is still gathered as fp32 and split in registers.
// Wave-private tiles: no barriers; the tile of the next iteration is requested as soon as this iteration's fragments are in
// registers, and lands under its MFMAs.
struct CorrBwdPlParams {
  const unsigned short* f0;
  const unsigned short* f1;
  long ps;
  int ld;
  const float* dout;
  float* g0;
  float* g1;
  int ld_dout, ld_g, shift, fuse;
  int B, C, H, W;
  int oh, ow, r, gw, s2;
  int off, nA, T;
  int vr, joff;   // narrow-band mode: see CorrPlParams
  unsigned dout_bytes;   // bytes of dout up to its last band entry (0: larger than 1 GiB, dword gathers only)
  size_t dout_total;     // the same, always (shared band loads: single-dword buffer loads, clamped to 1 GiB by make_rsrc)
  int rot;               // rotated displacement-row order
};

__device__ __forceinline__ int corr_tr_swz64(int k, int granule) {      // conv_planes.hip tr_swz<64>
  const int pair = granule >> 1;
  const int sw = ((k >> 1) & 1) << 1;
  return k * 64 + (((pair ^ sw) << 1) | (granule & 1)) * 8;
}

__device__ __forceinline__ void corr_split8(const float (&v)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float a = v[2 * i], b = v[2 * i + 1];
    const unsigned hh = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(hh << 16), rb = b - __uint_as_float(hh & 0xffff0000u);
    const unsigned mm = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(mm << 16), sb = rb - __uint_as_float(mm & 0xffff0000u);
    hi[i] = hh; mid[i] = mm; lo[i] = cvt_pk_bf16(sa, sb);
  }
}

// SHARE (C a multiple of 256: the four waves of a workgroup are the four 64-channel groups of ONE (sample, class, site tile,
// row) and walk the same items): the band operand of an item — 32 x 32 values of dout, the same for every channel group — is
// fetched ONCE per workgroup, four values per thread, and handed to the waves through LDS (two barriers per item), instead
// of by each wave for itself with 32-line gathers.  Round 3 ablation (profiles/r03_corr_ablation.txt): the per-wave band
// loads were 61 of the kernel's 172 us at the step's shape, 246 of 756 us at the 81-channel one.
constexpr int BAND_PITCH = 36;                       // floats per band row in LDS (144 B: conflict-free 16-byte reads of 32 rows)
template <bool SHARE>
__global__ __launch_bounds__(256, 2) void corr_bwd_pl_kernel(const CorrBwdPlParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  constexpr int TILE = 3 * 32 * 64;                 // elements per wave: 3 planes x 32 sites x 64 channels
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int l31 = lane & 31, h = lane >> 5;
  const int ncg = p.C >> 6;
  // work order: XCD-contiguous (xcd_remap), rows fastest, samples slowest: the blocks resident together on an XCD are
  // consecutive rows of one (sample, site tile) and share the feature rows their displacement rows read (the step's shape:
  // one sample per XCD, as before; the 81-channel point: FETCH_SIZE 2.30 -> see DESIGN.md §4.2)
  long job = (long)xcd_remap(blockIdx.x, gridDim.x, 1) * 4 + wid;
  const int cg = (int)(job % ncg); job /= ncg;
  const int y = (int)(job % p.H); job /= p.H;
  const int ia = (int)(job % p.nA); job /= p.nA;
  const int q = (int)(job % p.s2); job /= p.s2;
  const int s = (int)job;
  if (s >= p.B) return;
  const int i0 = ia * p.vr, c0 = cg * 64;
  const int role_lo = p.fuse ? 0 : (int)blockIdx.y, role_hi = p.fuse ? 1 : (int)blockIdx.y;
  unsigned short* tile = lds + wid * TILE;
  const unsigned tile_addr = lds_addr(tile);
  float* band = reinterpret_cast<float*>(lds + 4 * TILE);      // SHARE: [32][BAND_PITCH] floats behind the four tiles

  const size_t recs = (((size_t)p.B * p.H * p.W - 1) * (size_t)p.ld + (size_t)p.C) * 2;
  u32x4 f0_rs[3], f1_rs[3];
#pragma unroll
  for (int pl = 0; pl < 3; pl++) {
    f0_rs[pl] = raw_rsrc(p.f0 + pl * p.ps, recs);
    f1_rs[pl] = raw_rsrc(p.f1 + pl * p.ps, recs);
  }
  const int ld2 = p.ld * 2;

  f32x16 acc[2];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[c][e] = 0.f;

  // fragment addresses (transposing reads): lane -> 4 consecutive channels (i16 & 3) of site row (i16 >> 2) of its half
  const int i16 = lane & 15, grp = lane >> 4;
  const int krow = 8 * (grp >> 1) + (i16 >> 2);
  int b_rd[2];
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const int ch = 32 * c + 16 * (grp & 1) + 4 * (i16 & 3);
    b_rd[c] = corr_tr_swz64(krow, ch >> 3) + (ch & 7);
  }
  // DMA lane mapping: instruction j covers sites 8j .. 8j+7, lane -> (site 8j + lane/8, slot lane%8 -> granule)
  const int d_site = lane >> 3, d_slot = lane & 7;

  // iteration state
  // The displacement rows are visited in a rotated order (corr_fwd_nb_kernel): step k of a role takes the row whose feature
  // row R has R / s2 = k (mod 2r+1), so the blocks of neighbouring rows y ask for a feature row in the same step.
  const int yq = (y - (p.s2 - 1) * (y < 0)) / p.s2;
  const int rot0 = p.rot ? ((p.r - yq) % p.gw + p.gw) % p.gw : 0, rot1 = p.rot ? (yq + p.r) % p.gw : 0;
  // Items: (role, displacement row, column tile).  Only live ones are visited: the column tiles with a site inside the image
  // are a range [t_lo, t_hi], and so are the displacement rows of a role whose feature row and dout row exist — at the step's
  // shape (T = 1, one tile wide) the old walk tried three tiles per item, two of them dead, in a 250-instruction scalar loop.
  int t_lo = p.T + 1, t_hi = -p.T - 1;
  for (int t = -p.T; t <= p.T; t++) {
    const int kk = i0 + 32 * t + p.joff;
    if (q + p.off + p.s2 * (kk + 31) >= 0 && q + p.off + p.s2 * kk < p.W) { t_lo = min(t_lo, t); t_hi = max(t_hi, t); }
  }
  const int nd_r1 = ((s - p.shift) % p.B + p.B) % p.B, ns_r0 = (s + p.shift) % p.B;
  auto row_ok = [&](int role, int pi) -> bool {
    const int dyp = p.s2 * (pi - p.r);
    const int ys = role == 0 ? y + dyp : y - dyp;
    const int o = (role == 0 ? y : ys) - p.off;
    return (unsigned)ys < (unsigned)p.H && (unsigned)o < (unsigned)p.oh;
  };
  int it_role = role_lo, it_k = -1, it_pi = 0, it_t = t_hi;      // advanced before use
  int nd = 0, ns = 0, ysrc = 0, oy = 0, k0 = 0;
  auto next_item = [&]() -> bool {
    if (t_lo > t_hi) return false;
    if (++it_t > t_hi) {
      it_t = t_lo;
      for (;;) {                                                  // the next live displacement row
        if (++it_k >= p.gw) { it_k = 0; if (++it_role > role_hi) return false; }
        it_pi = !p.rot ? it_k : it_role == 0 ? (it_k + rot0) % p.gw : ((rot1 - it_k) % p.gw + p.gw) % p.gw;
        if (row_ok(it_role, it_pi)) break;
      }
      const int dyp = p.s2 * (it_pi - p.r);
      nd = it_role == 0 ? s : nd_r1;
      ns = it_role == 0 ? ns_r0 : nd_r1;
      ysrc = it_role == 0 ? y + dyp : y - dyp;
      oy = (it_role == 0 ? y : ysrc) - p.off;
    }
    k0 = i0 + 32 * it_t + p.joff;
    return true;
  };
  // the feature tile of the current item -> LDS (role 0 multiplies in1, role 1 in0); a lane's column and channel offsets once
  int t_xs[4], t_off[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int site = 8 * j + d_site;
    const int g = ((((d_slot >> 1) ^ (((site >> 1) & 1) << 1)) << 1) | (d_slot & 1));
    t_xs[j] = q + p.off + p.s2 * site;
    t_off[j] = t_xs[j] * ld2 + (c0 + g * 8) * 2;
  }
  auto issue_tile = [&]() {
    const int xk = p.s2 * k0;
    const int rowb = ((ns * p.H + ysrc) * p.W + xk) * ld2;
    int voff[4];
#pragma unroll
    for (int j = 0; j < 4; j++) voff[j] = (unsigned)(t_xs[j] + xk) < (unsigned)p.W ? rowb + t_off[j] : OOB_MARK;
    if (it_role == 0) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const unsigned d = tile_addr + (unsigned)(j * 1024);
        dma3(voff[j], f1_rs[0], f1_rs[1], f1_rs[2], d, d + 32 * 64 * 2, d + 2 * 32 * 64 * 2);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const unsigned d = tile_addr + (unsigned)(j * 1024);
        dma3(voff[j], f0_rs[0], f0_rs[1], f0_rs[2], d, d + 32 * 64 * 2, d + 2 * 32 * 64 * 2);
      }
    }
  };
  // the band operand of the current item: av[slab][e] for contracted site k0 + 16 slab + 8 h + e.
  // Role 1 (site = k, offset own - k): for a fixed e the lanes own = 0 .. 31 read consecutive addresses — 16 coalesced dword
  // loads.  Role 0 (site = own, offset k - own): for a fixed e the lanes are ld_dout - 1 floats apart, 32 cache lines per
  // instruction; but the 8 values of a lane's slab half are CONSECUTIVE in memory, so they come as two 16-byte loads (dword
  // aligned; entries outside the band masked afterwards) — 4 instructions per item instead of 16 on the address path that
  // bounds this kernel.  A run that would start before / end after the buffer (first / last site only) takes the dword path.
  const __amdgpu_buffer_rsrc_t dout_rs = make_rsrc(p.dout, p.dout_bytes);
  const __amdgpu_buffer_rsrc_t dout_rs_all = make_rsrc(p.dout, p.dout_total);
  auto load_band = [&](float (&av)[2][8]) {
    const int role = it_role;
    const size_t srow = ((size_t)nd * p.oh + oy) * p.ow;
    const int own = i0 + l31;
    if (role == 0 && p.dout_bytes != 0) {
      const int ox = q + p.s2 * own;
      const bool site_ok = (unsigned)ox < (unsigned)p.ow;
      const long base = (long)(srow + (site_ok ? ox : 0)) * p.ld_dout + it_pi * p.gw;      // float index of offset -r
      int ois[2];
      bool run_ok[2], edge = false;
#pragma unroll
      for (int sl = 0; sl < 2; sl++) {
        ois[sl] = k0 + 16 * sl + 8 * h - own + p.r;
        run_ok[sl] = site_ok && ois[sl] + 7 >= 0 && ois[sl] < p.gw;
        const long idx = base + ois[sl];
        edge = edge || (run_ok[sl] && (idx < 0 || (idx + 8) * 4 > (long)p.dout_bytes));
      }
      if (!__any(edge)) {
#pragma unroll
        for (int sl = 0; sl < 2; sl++) {
          const int voff = run_ok[sl] ? (int)((base + ois[sl]) * 4) : OOB_MARK;
          const u32x4 lo = buf_ld16(dout_rs, voff), hi = buf_ld16(dout_rs, voff + 16);
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const float v = __uint_as_float(e < 4 ? lo[e] : hi[e - 4]);
            av[sl][e] = (unsigned)(ois[sl] + e) < (unsigned)p.gw ? v : 0.f;
          }
        }
        return;
      }
    }
    const float* drow = p.dout + srow * p.ld_dout + it_pi * p.gw + p.r;
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
  };

  constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
  // SHARE: thread t fetches Band[a][b .. b+3] with (a, b) = (t / 8, 4 (t % 8)); role 0: a = own site row, b = contracted site
  // (four consecutive offsets of one site: consecutive floats); role 1: a = contracted site, b = own site (consecutive
  // offsets again).  Values outside the band / the image are zero.
  const int sh_a = (int)(threadIdx.x >> 3), sh_b = (int)(threadIdx.x & 7) << 2;
  float breg[4];
  int bmask = 0;
  // (32-bit arithmetic: SHARE requires dout_total < 2^30)
  const int sh_c = sh_b - sh_a;
  auto load_band_shared = [&]() {
    const int role = it_role;
    const int srow = (nd * p.oh + oy) * p.ow;
    const int site = (role == 0 ? i0 : k0) + sh_a;            // the site whose dout row is read
    const int ox = q + p.s2 * site;
    const bool site_ok = (unsigned)ox < (unsigned)p.ow;
    // offsets of the four values: role 0: o = (k0 + b + e) - (i0 + a); role 1: o = (i0 + b + e) - (k0 + a)
    const int dk = k0 - i0;
    const int o0 = (role == 0 ? dk : -dk) + sh_c;
    const int base = (srow + (site_ok ? ox : 0)) * p.ld_dout + it_pi * p.gw + p.r + o0;   // float index of value 0
    // the band entries among the four: -r <= o0 + e <= r
    const int e_lo = max(0, -p.r - o0), e_hi = min(3, p.r - o0);
    bmask = site_ok && e_hi >= e_lo ? ((2 << e_hi) - 1) & ~((1 << e_lo) - 1) : 0;
    // the four values are consecutive floats: one dword-aligned 16-byte load unless the run would start before / end after
    // the tensor (first / last site only), masked afterwards
    const bool inb = base >= 0 && (size_t)(base + 4) * 4 <= p.dout_total;
    if (__all(bmask == 0 || inb)) {
      const u32x4 v = buf_ld16(dout_rs_all, bmask ? base * 4 : OOB_MARK);
#pragma unroll
      for (int e = 0; e < 4; e++) breg[e] = __uint_as_float(v[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const bool ok = (bmask >> e) & 1;
        breg[e] = buf_ld1(dout_rs_all, ok ? (base + e) * 4 : OOB_MARK, 0);
      }
    }
  };
  auto store_band_shared = [&]() {
    if (it_role == 0) {
      float4 v = make_float4((bmask & 1) ? breg[0] : 0.f, (bmask & 2) ? breg[1] : 0.f, (bmask & 4) ? breg[2] : 0.f,
                             (bmask & 8) ? breg[3] : 0.f);
      *reinterpret_cast<float4*>(band + sh_a * BAND_PITCH + sh_b) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; e++) band[(sh_b + e) * BAND_PITCH + sh_a] = ((bmask >> e) & 1) ? breg[e] : 0.f;
    }
  };
  float av0[2][8], av1[2][8];
  bool have = next_item();
  if (have) {
    issue_tile();
    if constexpr (SHARE) load_band_shared(); else load_band(av0);
  }
  int par = 0;
  while (have) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this item's tile (and band values) have landed
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SHARE) {
      __builtin_amdgcn_s_barrier();                            // every wave has read the previous item's band
      __builtin_amdgcn_sched_barrier(0);
      store_band_shared();
    }
    s16x8 bv[2][2][3];
#pragma unroll
    for (int sl = 0; sl < 2; sl++)
#pragma unroll
      for (int pl = 0; pl < 3; pl++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
          const unsigned short* b0 = tile + pl * (32 * 64) + b_rd[c] + sl * 16 * 64;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(b0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(b0 + 4 * 64));
          bv[sl][c][pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // fragments in registers: the tile may be overwritten
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SHARE) {
      __builtin_amdgcn_s_barrier();                            // the item's band is complete in LDS
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int sl = 0; sl < 2; sl++) {
        const float4 lo = *reinterpret_cast<const float4*>(band + l31 * BAND_PITCH + 16 * sl + 8 * h);
        const float4 hi = *reinterpret_cast<const float4*>(band + l31 * BAND_PITCH + 16 * sl + 8 * h + 4);
        av0[sl][0] = lo.x; av0[sl][1] = lo.y; av0[sl][2] = lo.z; av0[sl][3] = lo.w;
        av0[sl][4] = hi.x; av0[sl][5] = hi.y; av0[sl][6] = hi.z; av0[sl][7] = hi.w;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    // request the next item (tile by DMA, band values by plain loads into the other register set)
    const bool more = next_item();
    if (more) {
      issue_tile();
      if constexpr (SHARE) load_band_shared();
      else if (par == 0) load_band(av1); else load_band(av0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sl = 0; sl < 2; sl++) {
      u32x4 ap[3];
      if (SHARE || par == 0) corr_split8(av0[sl], ap[0], ap[1], ap[2]); else corr_split8(av1[sl], ap[0], ap[1], ap[2]);
#pragma unroll
      for (int tt = 0; tt < 6; tt++)
#pragma unroll
        for (int c = 0; c < 2; c++)
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ap[ta[tt]]),
                                                           __builtin_bit_cast(bf16x8, bv[sl][c][tb[tt]]), acc[c], 0, 0, 0);
    }
    par ^= 1;
    have = more;
  }
  float* gout = (p.fuse || blockIdx.y == 0) ? p.g0 : p.g1;
  const float cf = (float)p.C;
#pragma unroll
  for (int e = 0; e < 16; e++) {
    const int i = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
    const int x = q + p.off + p.s2 * i;
    if ((unsigned)x >= (unsigned)p.W || i - i0 >= p.vr) continue;
    float* d = gout + (((size_t)s * p.H + y) * p.W + x) * p.ld_g + c0 + l31;
#pragma unroll
    for (int c = 0; c < 2; c++) d[32 * c] = acc[c][e] / cf;
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

// Tiles of 32 sites per residue class.  Wide band (FlowNetC: r = 10, and the image is one tile wide): the band of a tile reaches
// into its neighbour tiles, T = ceil(r / 32) Gram tiles on each side.  Narrow band over several tiles (the +-4 cost volume of
// the north star: r = 4): a tile owns 32 - 2r sites and ONE Gram tile whose columns start r sites to the left covers their
// whole band — 24 x 9 useful products of 1024 instead of 32 x 9 of 3072.
// options corr_nb / corr_wb = 0 keep the streaming kernel in narrow- / wide-band mode; corr_bwd_b128 = 0 keeps the dword
// gathers of the band operand (A/B switches, options.h)
static bool corr_nb_enabled() { return unflow::options().corr_nb != 0; }
static bool corr_wb_enabled() { return unflow::options().corr_wb != 0; }
static bool corr_bwd_b128_enabled() { return unflow::options().corr_bwd_b128 != 0; }

static void corr_pl_tiles(int nq, int r, int* nA, int* T, int* vr, int* joff) {
  if (nq > 32 && r <= 6) {
    *vr = 32 - 2 * r; *joff = -r; *T = 0;
    *nA = (nq + *vr - 1) / *vr;
  } else {
    *vr = 32; *joff = 0; *T = (r + 31) / 32;
    *nA = (nq + 31) / 32;
  }
}

#ifdef UNFLOW_CORR_TRACE
extern "C" __attribute__((visibility("default"))) int unflow_debug_corr_trace(unsigned long long* host_out) {      // [8 waves][8]
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_corr_trace), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

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
  corr_pl_tiles(nq, g.r, &p.nA, &p.T, &p.vr, &p.joff);
  const bool al16 = in0->ld % 8 == 0 && ((reinterpret_cast<uintptr_t>(in0->base) | reinterpret_cast<uintptr_t>(in1->base)) & 15) == 0;
  if (p.joff != 0 && g.s2 == 1 && g.r == 4 && p.vr == 24 && C % 32 == 0 && al16 && unflow::options().corr_rs >= 2) {
    // the +-4 cost volume: f1 rows streamed through an LDS ring, one 8-row workgroup per CU
    static DynLdsBook rg_book{};
    if (ensure_dyn_lds(reinterpret_cast<const void*>(&corr_fwd_ring_kernel), RG_SMEM, rg_book) == hipSuccess) {
      corr_fwd_ring_kernel<<<B * p.nA * ((g.oh + RG_R - 1) / RG_R), 512, RG_SMEM, st>>>(p);
      return launch_status();
    }
    (void)hipGetLastError();       // 147 KB of LDS refused on this device: the row-shared kernel below
  }
  if (p.joff != 0 && g.s2 == 1 && g.r >= 1 && g.r <= 4 && p.vr == 32 - 2 * g.r && C % 32 == 0 && al16 && unflow::options().corr_rs) {
    // rows shared by a workgroup (4 output rows, double-buffered chunks of 32 channels): a third of the narrow-band kernel's f1 traffic
    static DynLdsBook rs_book[4]{};
    const void* rs_fn[4] = {reinterpret_cast<const void*>(&corr_fwd_rs_kernel<3>), reinterpret_cast<const void*>(&corr_fwd_rs_kernel<5>),
                            reinterpret_cast<const void*>(&corr_fwd_rs_kernel<7>), reinterpret_cast<const void*>(&corr_fwd_rs_kernel<9>)};
    (void)ensure_dyn_lds(rs_fn[g.r - 1], RS_SMEM, rs_book[g.r - 1]);
    const int rs_grid = B * p.nA * ((g.oh + RS_R - 1) / RS_R);
    switch (g.gw) {
      case 3: corr_fwd_rs_kernel<3><<<rs_grid, 256, RS_SMEM, st>>>(p); break;
      case 5: corr_fwd_rs_kernel<5><<<rs_grid, 256, RS_SMEM, st>>>(p); break;
      case 7: corr_fwd_rs_kernel<7><<<rs_grid, 256, RS_SMEM, st>>>(p); break;
      default: corr_fwd_rs_kernel<9><<<rs_grid, 256, RS_SMEM, st>>>(p); break;
    }
    return launch_status();
  }
  if (p.joff != 0 && C % 64 == 0 && C <= 256 && al16 && corr_nb_enabled()) {
    // (16-byte granules; with ld % 64 == 0 and a 128-byte-aligned base — the step's feature buffers — every DMA
    // instruction moves 8 whole cache lines)
    const int nw = C / 64;
    const int nb_smem = nw * (3 * 32 * 64 * 2 + g.gw * p.vr * g.gw * 4);
    static DynLdsBook nb_book{};
    (void)ensure_dyn_lds(reinterpret_cast<const void*>(&corr_fwd_nb_kernel), nb_smem, nb_book);
    corr_fwd_nb_kernel<<<B * p.nA * g.s2 * g.oh, 64 * nw, nb_smem, st>>>(p);
    return launch_status();
  }
  const size_t out_bytes = (((size_t)B * g.oh * g.ow - 1) * (size_t)ld_out + (size_t)g.gw * g.gw) * 4;
  if (p.joff == 0 && (C == 128 || C == 256) && al16 && g.r <= 15 && out_bytes <= 0x3fffffffu && unflow::options().corr_rw) {
    const int npr = ((g.oh + g.s2 - 1) / g.s2 + RW_ROWS - 1) / RW_ROWS;      // groups of RW_ROWS rows per row class
    const int smem = 2 * (C / 64) * (3 * 32 * 64 * 2) + 8 * 4096;             // two tiles + the waves' accumulator slots
    static DynLdsBook rw_book[2]{};
    (void)ensure_dyn_lds(C == 128 ? reinterpret_cast<const void*>(&corr_fwd_rw_kernel<1>) : reinterpret_cast<const void*>(&corr_fwd_rw_kernel<2>),
                         smem, rw_book[C == 128 ? 0 : 1]);
    const int blocks = B * g.s2 * p.nA * g.s2 * npr;
    if (C == 128) corr_fwd_rw_kernel<1><<<blocks, 512, smem, st>>>(p);
    else corr_fwd_rw_kernel<2><<<blocks, 512, smem, st>>>(p);
    return launch_status();
  }
  if (p.joff == 0 && C % 64 == 0 && C <= 256 && al16 && 32 * g.gw <= 6 * 64 * (C / 64) && corr_wb_enabled()) {
    const int nw = C / 64;
    const int npr = ((g.oh + g.s2 - 1) / g.s2 + 1) / 2;       // row pairs (oy, oy + s2) per row class
    constexpr int per_wave = 3 * 32 * 64 * 2 + 2 * 32 * 32 * 4;
    static DynLdsBook attr_book{};
    (void)ensure_dyn_lds(reinterpret_cast<const void*>(&corr_fwd_wb_kernel), 4 * per_wave, attr_book);
    corr_fwd_wb_kernel<<<B * g.s2 * p.nA * g.s2 * npr, 64 * nw, nw * per_wave, st>>>(p);
    return launch_status();
  }
  const int smem = 3 * 32 * (C + 8) * 2;
  static DynLdsBook pl_book{};
  (void)ensure_dyn_lds(reinterpret_cast<const void*>(&corr_fwd_pl_kernel), smem, pl_book);
  const int blocks = B * p.nA * g.s2 * g.oh;
  corr_fwd_pl_kernel<<<blocks, 256, smem, st>>>(p);
  return launch_status();
}

int corr_pl_bwd(const float* dout, int ld_dout, const unflow_planes* in0, const unflow_planes* in1, int shift, float* g0, float* g1,
                int ld_g, int fuse, int B, int C, int H, int W, const CorrGeom& g, hipStream_t st) {
  CorrBwdPlParams p{};
  p.f0 = reinterpret_cast<const unsigned short*>(in0->base);
  p.f1 = reinterpret_cast<const unsigned short*>(in1->base);
  p.ps = in0->plane_stride; p.ld = in0->ld;
  p.dout = dout; p.g0 = g0; p.g1 = g1; p.ld_dout = ld_dout; p.ld_g = ld_g; p.shift = shift; p.fuse = fuse;
  p.B = B; p.C = C; p.H = H; p.W = W;
  p.oh = g.oh; p.ow = g.ow; p.r = g.r; p.gw = g.gw; p.s2 = g.s2;
  p.off = g.md - g.pad;
  const int span = max(g.ow, W - p.off);
  const int nq = (span + g.s2 - 1) / g.s2;
  corr_pl_tiles(nq, g.r, &p.nA, &p.T, &p.vr, &p.joff);
  // rotated row order: HBM reads 1.4x instead of 2.3x / 3.4x algorithmic at both measured shapes, but only the narrow-band
  // shape got faster with it (758 -> 740 us); the step's wide-band shape lost 7-16 % standalone (consumers of a row in
  // lock-step), so it keeps the natural order.  Option corr_bwd_rot = 0 / 1 forces it (read per call).
  { const int e = unflow::options().corr_bwd_rot; p.rot = e >= 0 ? (e != 0) : (p.joff != 0); }
  const size_t dbytes = (((size_t)B * g.oh * g.ow - 1) * (size_t)ld_dout + (size_t)g.gw * g.gw) * 4;
  p.dout_bytes = dbytes < ((size_t)1 << 30) && corr_bwd_b128_enabled() ? (unsigned)dbytes : 0u;
  p.dout_total = dbytes;
  const bool share = (C / 64) % 4 == 0 && dbytes < ((size_t)1 << 30) && unflow::options().corr_bwd_share;
  const int smem = 4 * 3 * 32 * 64 * 2 + (share ? 32 * BAND_PITCH * 4 : 0);
  static DynLdsBook bwd_book[2]{};
  (void)ensure_dyn_lds(share ? reinterpret_cast<const void*>(&corr_bwd_pl_kernel<true>) : reinterpret_cast<const void*>(&corr_bwd_pl_kernel<false>),
                       smem, bwd_book[share ? 1 : 0]);
  const long jobs = (long)(C / 64) * B * p.nA * g.s2 * H;
  dim3 grid((unsigned)((jobs + 3) / 4), fuse ? 1 : 2);
  if (share) corr_bwd_pl_kernel<true><<<grid, 256, smem, st>>>(p);
  else corr_bwd_pl_kernel<false><<<grid, 256, smem, st>>>(p);
  return launch_status();
}
