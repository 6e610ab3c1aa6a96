// The halo gather kernel of the plane-format implicit GEMM (conv_planes.hip) as a header: conv_planes.hip instantiates the
// production tiles, conv_halo_wide.hip the fp16 128 x 256 tile, which needs its own compiler flags (build.py).
#pragma once
#include "planes_shared.h"

namespace {
using namespace igemm;

// Phase trace (diagnostic builds only: -DUNFLOW_PHASE_TRACE, tools/phase_trace.py): the waves of workgroup 0 stamp
// s_memtime at the phase boundaries of their first 64 K tiles; never compiled into the shipped library.
#ifdef UNFLOW_PHASE_TRACE
__device__ unsigned long long g_phase_trace[8 * 8];          // [wave][phase 0..5 cycle sums, 6 = tiles]
#define PHASE_DECL unsigned long long ph_prev = 0, ph_acc[7] = {0, 0, 0, 0, 0, 0, 0}
// cycles since the previous stamp go to phase `slot` (slot 6: the start of a tile — counts it and takes what is left of the
// loop tail into phase 5); sums stay in registers (a store per stamp would sit in every s_waitcnt vmcnt that follows)
#define PHASE_STAMP(slot)                                                  \
  do {                                                                     \
    const unsigned long long ph_now = __builtin_readcyclecounter();        \
    if ((slot) == 6) { if (ph_prev) ph_acc[5] += ph_now - ph_prev; ph_acc[6]++; } \
    else ph_acc[slot] += ph_now - ph_prev;                                 \
    ph_prev = ph_now;                                                      \
  } while (0)
#define PHASE_FLUSH                                                                                            \
  do {                                                                                                         \
    if (blockIdx.x == UNFLOW_PHASE_TRACE && (threadIdx.x & 63) == 0)                                           \
      for (int ph_i = 0; ph_i < 7; ph_i++) g_phase_trace[(threadIdx.x >> 6) * 8 + ph_i] = ph_acc[ph_i];        \
  } while (0)
#else
#define PHASE_DECL do { } while (0)
#define PHASE_STAMP(slot) do { } while (0)
#define PHASE_FLUSH do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------ halo gather kernel
// Source stride 1 (3x3 / 1x1 stride-1 convs, every conv data gradient incl. the 4 parity classes of stride 2, conv_transpose
// forward): the taps of a site are its neighbours, so a spatially compact tile re-reads almost the same pixels for every
// tap.  igemm_pl_gather_kernel fetches them once PER TAP (its 128-site tiles are image rows; the re-reads miss L1 and,
// with ~100 resident tiles per XCD, L2 too: conv2's data gradient moved 1.9 GB for 0.6 GB of operands and ran at the
// Infinity-Cache rate, 64 TFLOP/s).  Here a block owns a 4 x 32-site tile and walks K chunk-major: for every 32-channel
// chunk the (4 + nty - 1) x (32 + ntx - 1) halo of source pixels is loaded ONCE into LDS ([pixel][32 ch], 80-byte pitch:
// conflict-free b128 reads of consecutive pixels) and serves all taps through a per-tap address offset; only the weight
// tile streams per tap.  A-operand loads drop by taps * 128 / halo (6.4x for 3x3), all loads by ~1.7x.
// Tile = 4 rows x 32 sites: an MFMA sub-tile (32 lanes) is 32 CONSECUTIVE halo pixels, which with the 80-byte pitch makes
// every 16-lane group of a ds_read_b128 hit 16 distinct 16-byte bank slots (an 8 x 16 tile puts two image rows into one
// sub-tile: SQ_LDS_BANK_CONFLICT was 85 % of the LDS-active cycles).

//
// DB (fp16, one plane; round 6): the weight tile is DOUBLE-BUFFERED in LDS.  With 8 MFMAs per wave and K tile the single-buffer loop
// above spends most of a tile outside the MFMA phase (phase trace, profiles/r06_f16_phase_trace.txt: MFMA phase 28 %, waiting for the
// next tile's loads 14 %, LDS stores 12 %, the two barriers 15 %, scalar bookkeeping 31 %): the loads have only the 8 MFMAs to land.
// Here tile kk+1's loads are issued at the END of iteration kk-1 (right after tile kk's registers went to the other buffer), land
// under the whole of iteration kk, and go to LDS after its MFMAs; one barrier per tile, two only where a chunk's halo is replaced.
//
// BM = 256 (fp16, conv_halo_tall.hip): an 8 x 32-site tile, four waves of 128 x 64 — a third of the operand bytes per MFMA (the weight
// tile serves twice the sites), which is what bounds the fp16 form (knock-outs, profiles/r06_f16_knockouts.txt).
template <int BN, int WM, int WN, int NPL, bool F16, bool DB = false, int BM = 128>
__global__ __launch_bounds__(256, 2) void igemm_pl_halo_kernel(const PlGatherParams p, int HPmax) {
  static_assert(!DB || (F16 && NPL == 1), "double-buffered weights: the fp16 one-plane form");
  static_assert(BM == 128 || BM == 256, "4 x 32 or 8 x 32 sites");
  constexpr int TH = BM / 32;                    // (shadows planes_shared.h's TH = 4 for the tall tile)
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves");
  constexpr int B_PLANE = BN * LDH;
  constexpr int NB = BN / 64;
  constexpr int NH = BM == 128 ? 4 : 6;          // halo granules per thread and plane (covers 256 / 384 pixels)
  constexpr int NT = mfma_nt(NPL, F16);
  constexpr int KCH = F16 ? NPL : 1;            // 32-channel chunks per K tile (fp16: the planes ARE consecutive chunks)

  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  const int H_PLANE = HPmax * HPITCH;
  unsigned short* Hh = smem16;
  unsigned short* Bh = Hh + NPL * H_PLANE;
  int* pix = reinterpret_cast<int*>(reinterpret_cast<char*>(smem16) + pl_halo_main_bytes(BN, WN, NPL, HPmax, DB ? 2 : 1));

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  int t, ntile, cls_id, split;
  work_decode(xcd_remap(blockIdx.x, gridDim.x, p.xcd), p.mt, p.nt, p.acc ? 1 : p.ncls, p.nsplit, p.order, p.mgroup, t, ntile, cls_id,
              split);
  if (t < 0) return;
  // Tap classes of this block.  Output-parity classes (data gradient of a stride-2 conv, conv_transpose forward) write
  // different pixels: one class per block (cls_id).  ACCUMULATING classes (p.acc: forward of a stride-2 conv, data gradient
  // of a conv_transpose — source stride 2) all add into the same output tile: on the four parity sub-lattices of the source
  // (pixel pitch p.sp = 2) a stride-2 k x k conv is the sum of four stride-1 convs with ceil / floor (k/2)^2 taps, each
  // served by its own halo; the block walks class after class (class-major, then chunk, then tap).
  const int c_first = p.acc ? 0 : cls_id, c_last = p.acc ? p.ncls : cls_id + 1;
  const TapClass tc = p.cls[c_first];           // (geometry of the pixel table; the first class to load / multiply)
  const int n0 = ntile * BN;
  const int Cg = p.Cs >> 3;
  const int nchunk = (((Cg + 3) >> 2) + KCH - 1) / KCH;        // chunks of 32 KCH channels
  const int ch_per = (nchunk + p.nsplit - 1) / p.nsplit;
  const int ch0 = split * ch_per, ch1 = min(nchunk, (split + 1) * ch_per);
  int sumtaps = 0, mtx = 0;
  for (int c = c_first; c < c_last; c++) {
    sumtaps += p.cls[c].nty * p.cls[c].ntx;
    mtx = max(mtx, p.cls[c].ntx);
  }
  const int T = max(ch1 - ch0, 0) * sumtaps;    // K tiles of this block

  // tile -> (image, tile row, tile column)
  const int txi = t % p.tiles_x; t /= p.tiles_x;
  const int tyi = t % p.tiles_y;
  const int b = t / p.tiles_y;
  const int y0 = tyi * TH, x0 = txi * TW;
  // halo image in LDS: HC pixels per row for every class of the block (the widest class's); halo pixel (hy, hx) = source
  // pixel ((y0 + hy) * sp + dmin_y, (x0 + hx) * sp + dmin_x) of the class being loaded
  const int HC = TW + mtx - 1;

  __amdgpu_buffer_rsrc_t src_rs[NPL], w_rs[NPL];
#pragma unroll
  for (int pl = 0; pl < NPL; pl++) {
    // (fp16 chunk planes: plane pl starts 64 pl bytes into the tensor — the range shrinks by as much)
    src_rs[pl] = make_rsrc(p.src + pl * p.src_ps, (((size_t)p.B * p.Hs * p.Ws - 1) * (size_t)p.lds + (size_t)p.Cs) * 2 - (KCH > 1 ? pl * 64 : 0));
    w_rs[pl] = make_rsrc(p.w + pl * p.w_ps, (size_t)p.wtaps * p.N * p.Cs * 2 - (KCH > 1 ? pl * 64 : 0));
  }
  const int kq = tid & 3;
  const int lds2 = p.lds * 2;
  // this thread's halo granules: pixel hp = (tid >> 2) + 64 j, granule kq of the chunk — for the class being LOADED
  int h_off[NH];
  TapClass ltc = tc;                            // load-side class
  auto set_load_class = [&]() {
    const int dmy = p.dstep > 0 ? ltc.dy0 : ltc.dy0 - (ltc.nty - 1);     // source offset of halo pixel (0, 0)
    const int dmx = p.dstep > 0 ? ltc.dx0 : ltc.dx0 - (ltc.ntx - 1);
    const int HRc = TH + ltc.nty - 1, HCc = TW + ltc.ntx - 1;
#pragma unroll
    for (int j = 0; j < NH; j++) {
      const int hp = (tid >> 2) + 64 * j;
      const int hy = hp / HC, hx = hp - hy * HC;
      const int y = (y0 + hy) * p.sp + dmy, x = (x0 + hx) * p.sp + dmx;
      const bool ok = hy < HRc && hx < HCc && (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;
      h_off[j] = ok ? ((b * p.Hs + y) * p.Ws + x) * lds2 + kq * 16 : OOB_MARK;
    }
  };
  set_load_class();
  if (tid < BM) {
    const int yg = y0 + (tid >> TWL), xg = x0 + (tid & (TW - 1));
    pix[tid] = (yg < p.Hg && xg < p.Wg) ? (b * p.Hd + yg * p.so + tc.py) * p.Wd + xg * p.so + tc.px : -1;
  }
  int b_row[NB];
#pragma unroll
  for (int i = 0; i < NB; i++) {
    const int n = n0 + (tid >> 2) + 64 * i;
    b_row[i] = n < p.N ? n * p.Cs * 2 + kq * 16 : OOB_MARK;
  }

  u32x4 rh[NH][NPL], rb[NB][NPL];
  // loads of the NEXT K tile (chunk ld_chunk, tap (ld_ty, ld_tx); advanced once per tile, no divisions in the loop):
  // weights always, the halo when the tile opens a chunk
  int ld_c = c_first, ld_chunk = ch0, ld_ty = 0, ld_tx = 0;
  bool ld_live = T > 0;
  auto load_b = [&](int i) {
    const int widx = (ltc.ky0 + ld_ty * p.kstep) * p.KW + ltc.kx0 + ld_tx * p.kstep;
#if defined(UNFLOW_DIAG_B_SAME)       // knock-out diagnostics (never in the shipped library): every weight tile = the first one (L1 hits)
    const int voff = b_row[i] + 0 * widx;
#else
    const int voff = b_row[i] + widx * p.N * p.Cs * 2 + ld_chunk * (64 * KCH);
#endif
#pragma unroll
    for (int pl = 0; pl < NPL; pl++) {
      const bool ok = ld_live && (ld_chunk * KCH + (KCH > 1 ? pl : 0)) * 4 + kq < Cg;
#if defined(UNFLOW_DIAG_NO_B)         // ... or no weight loads at all (behind a branch that is never taken: T >= 0)
      if (T < 0)
#endif
      rb[i][pl] = buf_ld16(w_rs[pl], ok ? voff : OOB_MARK);
    }
  };
  auto load_h = [&](int j) {
    const int voff = h_off[j] + ld_chunk * (64 * KCH);
#pragma unroll
    for (int pl = 0; pl < NPL; pl++) {
      const bool ok = ld_live && (ld_chunk * KCH + (KCH > 1 ? pl : 0)) * 4 + kq < Cg;     // (h_off may be OOB_MARK itself: stays out of range)
      rh[j][pl] = buf_ld16(src_rs[pl], ok ? voff : OOB_MARK);
    }
  };
  auto ld_advance = [&](int kk_next) {     // the loads now target tile kk_next + 1... called after tile kk_next's loads
    ld_tx++;
    if (ld_tx == ltc.ntx) { ld_tx = 0; ld_ty++; }
    if (ld_ty == ltc.nty) { ld_ty = 0; ld_chunk++; }
    if (ld_chunk == ch1 && ld_c + 1 < c_last) {          // next class (accumulating classes only)
      ld_chunk = ch0;
      ld_c++;
      ltc = p.cls[ld_c];
      set_load_class();
    }
    ld_live = kk_next + 1 < T;
  };
  auto swz = [](int row, int g) { return row * LDH + 8 * (g ^ ((row >> 2) & 3)); };
  auto store_b = [&]() {
#pragma unroll
    for (int i = 0; i < NB; i++)
#pragma unroll
      for (int pl = 0; pl < NPL; pl++)
        *reinterpret_cast<u32x4*>(Bh + pl * B_PLANE + swz((tid >> 2) + 64 * i, kq)) = rb[i][pl];
  };
  auto store_h = [&]() {
#pragma unroll
    for (int j = 0; j < NH; j++) {
      const int hp = (tid >> 2) + 64 * j;
      if (hp < HPmax) {
#pragma unroll
        for (int pl = 0; pl < NPL; pl++)
          *reinterpret_cast<u32x4*>(Hh + pl * H_PLANE + hp * HPITCH + kq * 8) = rh[j][pl];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  // A fragment of sub-tile i: site s = wm*WM + i*32 + l31 -> halo pixel (s / TW) * HC + s % TW (+ the tap's offset)
  int a_rd[TM];
#pragma unroll
  for (int i = 0; i < TM; i++) {
    const int sidx = wm * WM + i * 32 + l31;
    a_rd[i] = ((sidx >> TWL) * HC + (sidx & (TW - 1))) * HPITCH + lh * 8;
  }
  const unsigned short* bh_rd = Bh + (wn * WN + l31) * LDH;
  const int gsw = lh ^ ((l31 >> 2) & 3);

#pragma unroll
  for (int j = 0; j < NH; j++) load_h(j);
#pragma unroll
  for (int i = 0; i < NB; i++) load_b(i);
  ld_advance(0);
  store_h();
  store_b();
  __syncthreads();
  constexpr int NGROUP = 2 * TM * NT;
  constexpr int NPIECE = NB + NH;
  constexpr int PPG = (NPIECE + NGROUP - 1) / NGROUP;
  int ty = 0, tx = 0;                            // tap of the tile being multiplied, and the halo position of its class's tap (0, 0)
  int c_hy0 = p.dstep > 0 ? 0 : tc.nty - 1, c_hx0 = p.dstep > 0 ? 0 : tc.ntx - 1;
  PHASE_DECL;
  if constexpr (DB) {
    auto store_b2 = [&](int buf) {
#pragma unroll
      for (int i = 0; i < NB; i++) *reinterpret_cast<u32x4*>(Bh + buf * B_PLANE + swz((tid >> 2) + 64 * i, kq)) = rb[i][0];
    };
    auto issue_next = [&]() {        // the loads of the tile the cursor points at (weights; the halo when it opens a chunk)
#pragma unroll
      for (int i = 0; i < NB; i++) load_b(i);
      if (ld_ty == 0 && ld_tx == 0) {
#pragma unroll
        for (int j = 0; j < NH; j++) load_h(j);
      }
    };
    issue_next();                    // tile 1 (the prologue above left the cursor there; dead loads stay out of range)
    int buf = 0;
    for (int kk = 0; kk < T; kk++) {
      const int hyi = c_hy0 + ty * p.dstep, hxi = c_hx0 + tx * p.dstep;
      const int tapoff = (hyi * HC + hxi) * HPITCH;
      const bool new_chunk = ld_ty == 0 && ld_tx == 0;   // the next tile opens a chunk: its halo is in flight beside its weights
      const unsigned short* bcur = bh_rd + buf * B_PLANE;
      if constexpr (BN >= 128) {
        s16x8 bv[2][TN][1], av[2][1];
        auto read_b = [&](int slab) {
#pragma unroll
          for (int j = 0; j < TN; j++) bv[slab & 1][j][0] = *reinterpret_cast<const s16x8*>(bcur + j * 32 * LDH + 8 * (gsw ^ (2 * slab)));
        };
        auto read_a = [&](int step) {
          const int slab = step / TM, i = step % TM;
          av[step & 1][0] = *reinterpret_cast<const s16x8*>(Hh + a_rd[i] + tapoff + 16 * slab);
        };
        read_b(0);
        read_a(0);
#pragma unroll
        for (int step = 0; step < 2 * TM; step++) {
          const int slab = step / TM, i = step % TM;
          if (step + 1 < 2 * TM) {
            if ((step + 1) / TM != slab) read_b(slab + 1);
            read_a(step + 1);
          }
#pragma unroll
          for (int j = 0; j < TN; j++) mfma_terms<1, true>(av[step & 1], bv[slab & 1][j], acc[i][j], 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int slab = 0; slab < 2; slab++) {
          s16x8 bv[TN][1];
#pragma unroll
          for (int j = 0; j < TN; j++) bv[j][0] = *reinterpret_cast<const s16x8*>(bcur + j * 32 * LDH + 8 * (gsw ^ (2 * slab)));
#pragma unroll
          for (int i = 0; i < TM; i++) {
            s16x8 av[1];
            av[0] = *reinterpret_cast<const s16x8*>(Hh + a_rd[i] + tapoff + 16 * slab);
#pragma unroll
            for (int j = 0; j < TN; j++) mfma_terms<1, true>(av, bv[j], acc[i][j], 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      store_b2(buf ^ 1);             // tile kk+1's weights (in flight since the end of the previous iteration) -> the other buffer
      if (new_chunk) {
        __syncthreads();             // every wave is done with the old halo
        store_h();
      }
      ty = ld_ty; tx = ld_tx;
      c_hy0 = p.dstep > 0 ? 0 : ltc.nty - 1;
      c_hx0 = p.dstep > 0 ? 0 : ltc.ntx - 1;
      ld_advance(kk + 1);
      issue_next();                  // tile kk+2
      __syncthreads();
      buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the dead loads of the last iterations)
    pl_gather_epilogue<WM, WN>(p, acc, pix, smem16, wm, wn, wid, lane, n0, split);
    return;
  }
  for (int kk = 0; kk < T; kk++) {
    PHASE_STAMP(6);
    const int hyi = c_hy0 + ty * p.dstep, hxi = c_hx0 + tx * p.dstep;
    const int tapoff = (hyi * HC + hxi) * HPITCH;
    const bool new_chunk = ld_ty == 0 && ld_tx == 0;   // the next tile opens a chunk: its halo is loaded during this tile
    auto piece = [&](int step) {
      if (step < NB) load_b(step);
      else if (step < NB + NH && new_chunk) load_h(step - NB);
    };
    if constexpr (BN >= 128) {
      // software-pipelined over the 2 x TM (slab, sub-tile) steps: the fragments of step s+1 are read from LDS while the
      // MFMAs of step s run (two waves per SIMD are not enough to hide a ds_read round trip in front of every step)
      s16x8 bv[2][TN][NPL], av[2][NPL];
      auto read_b = [&](int slab) {
  #pragma unroll
        for (int pl = 0; pl < NPL; pl++)
  #pragma unroll
          for (int j = 0; j < TN; j++)
            bv[slab & 1][j][pl] = *reinterpret_cast<const s16x8*>(bh_rd + pl * B_PLANE + j * 32 * LDH + 8 * (gsw ^ (2 * slab)));
      };
      auto read_a = [&](int step) {
        const int slab = step / TM, i = step % TM;
  #pragma unroll
        for (int pl = 0; pl < NPL; pl++)
          av[step & 1][pl] = *reinterpret_cast<const s16x8*>(Hh + pl * H_PLANE + a_rd[i] + tapoff + 16 * slab);
      };
      read_b(0);
      read_a(0);
  #pragma unroll
      for (int step = 0; step < 2 * TM; step++) {
        const int slab = step / TM, i = step % TM;
        if (step + 1 < 2 * TM) {
          if ((step + 1) / TM != slab) read_b(slab + 1);
          read_a(step + 1);
        }
  #pragma unroll
        for (int t2 = 0; t2 < NT; t2++) {
  #pragma unroll
          for (int j = 0; j < TN; j++) mfma_terms<NPL, F16>(av[step & 1], bv[slab & 1][j], acc[i][j], t2);
  #pragma unroll
          for (int q = 0; q < PPG; q++) piece((step * NT + t2) * PPG + q);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      // N = 64 tile: 6 MFMAs per step and three waves per SIMD — occupancy hides the LDS latency, the second fragment set
      // would cost a wave
#pragma unroll
      for (int slab = 0; slab < 2; slab++) {
        s16x8 bv[TN][NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; pl++)
#pragma unroll
          for (int j = 0; j < TN; j++)
            bv[j][pl] = *reinterpret_cast<const s16x8*>(bh_rd + pl * B_PLANE + j * 32 * LDH + 8 * (gsw ^ (2 * slab)));
#pragma unroll
        for (int i = 0; i < TM; i++) {
          s16x8 av[NPL];
#pragma unroll
          for (int pl = 0; pl < NPL; pl++)
            av[pl] = *reinterpret_cast<const s16x8*>(Hh + pl * H_PLANE + a_rd[i] + tapoff + 16 * slab);
#pragma unroll
          for (int t2 = 0; t2 < NT; t2++) {
#pragma unroll
            for (int j = 0; j < TN; j++) mfma_terms<NPL, F16>(av, bv[j], acc[i][j], t2);
#pragma unroll
            for (int q = 0; q < PPG; q++) piece(((slab * TM + i) * NT + t2) * PPG + q);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    PHASE_STAMP(0);    // MFMA phase (fragment reads, 48 MFMAs, the next tile's loads issued)
    __syncthreads();   // every wave is done with this tile's weights (and, at a chunk end, with the halo)
    PHASE_STAMP(1);    // barrier 1
#ifdef UNFLOW_PHASE_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PHASE_STAMP(2);    // the next tile's loads landed
#endif
    store_b();
    if (new_chunk) store_h();
#ifdef UNFLOW_PHASE_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    PHASE_STAMP(3);    // LDS stores
    __syncthreads();
    PHASE_STAMP(4);    // barrier 2
    ty = ld_ty; tx = ld_tx;          // the tile just stored is the next one multiplied: its tap, its class's halo origin
    c_hy0 = p.dstep > 0 ? 0 : ltc.nty - 1;
    c_hx0 = p.dstep > 0 ? 0 : ltc.ntx - 1;
    ld_advance(kk + 1);
  }
  PHASE_FLUSH;
  pl_gather_epilogue<WM, WN>(p, acc, pix, smem16, wm, wn, wid, lane, n0, split);
}

}  // namespace
