// conv / conv_transpose stacks of FlowNetC/S (src/e2eflow/core/flownet.py:89-237) as implicit GEMMs whose operands are
// already 16-bit in HBM ("operand planes"):
//
//   n_planes == 3  bf16 hi / mid / lo with x = hi + mid + lo EXACTLY (3 x 8 significand bits): a*b is summed from the six
//                  terms hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid on v_mfma_f32_32x32x16_bf16 with fp32 accumulation —
//                  fp32-class accuracy (igemm_shared.h) at 6/16 of the matrix-core time of v_mfma_f32_32x32x2_f32;
//   n_planes == 1  fp16 activations and weights, fp32 accumulation on v_mfma_f32_32x32x16_f16 (BASELINE configs[4]).
//
// The planes are written ONCE per element by whoever produces the tensor (the epilogues of these kernels, of the
// flow-head / correlation / input kernels, and unflow_weight_planes for the parameters), so the K loops below contain no
// conversion work at all: a 16-byte load is 8 consecutive channels of one plane = one MFMA operand granule; it goes
// HBM/L2 -> register -> ds_write_b128 -> ds_read -> MFMA.  (conv_igemm.hip splits fp32 operands while staging them
// instead: ~240 VALU instructions per thread and K tile, as much issue time as the 48 MFMAs they feed.)
//
//   igemm_pl_gather_kernel  conv fwd, conv dgrad (stride 2: 4 output-parity classes), deconv fwd, deconv dgrad:
//        D[site, n] = sum_tap sum_c SRC[b, yg*sm + dy(tap), xg*sm + dx(tap), c] * W[tap][n][c]      (W: K-contiguous planes)
//        LDS image per plane [row][32 k] bf16, 64-byte rows, 16-byte granules XOR-swizzled by bits 2..3 of the row:
//        conflict-free ds_write_b128 / ds_read_b128 (the image of conv_igemm.hip's MATH == 1 path).
//   igemm_pl_wgrad_kernel   filter gradients dW[(tap,a), b] = sum_site SRC[gather(site,tap), a] * DST[site, b]: both
//        operands are site-major in HBM, so a tile is staged as [k = site][channel] rows (256 contiguous bytes per site)
//        and the MFMA fragments (8 consecutive k of one channel) come out of LDS through ds_read_b64_tr_b16, the
//        gfx950 transposing read (semantics probed in tools/microbench/probe_semantics.hip): no per-lane dword gathers.
#include "igemm_shared.h"
#include "planes_shared.h"
#include "halo_kernel.h"

namespace {
using namespace igemm;

// ------------------------------------------------------------------------------------------------ gather kernel
// 256 threads = 4 waves.  Loads: thread (kq = tid & 3, r = tid >> 2) fetches granule kq (8 consecutive k) of rows r, r + 64
// of each operand and plane: a wave instruction covers 16 rows x 64 contiguous bytes.  The loads of tile t+1 sit between
// the MFMA groups of tile t (sched_barrier fences); single LDS stage, 3 blocks per CU resident and out of phase.
// (Tried and dropped: the LDS-DMA form that pays for the filter gradients — K16 stages, double-buffered, one barrier per
// stage, 118 registers.  A K16 stage of a gathered operand is 32 bytes per row: conv2 / conv3 forward 261 -> 326 us,
// 251 -> 329 us, stride-2 data gradients 142 -> 173 us, the step 598 -> 570 image-pairs/s.  The 64-byte row pieces of a
// K32 tile are the smallest unit the L1 / TA path moves efficiently; two K32 stages are 96 KB of LDS, one block per CU.)
// (Round 3 measured and dropped: 768-thread workgroups whose three 4-wave K groups own one output tile and combine their
// accumulators through LDS — a third of the split-K partial traffic for conv4 .. conv6_1 and the deep deconvs: time-neutral
// where one round of such workgroups fills the chip, 20 % slower at 75 % fill; and skipping the MFMAs of 32-column
// sub-tiles beyond N (N = 388 / 772 / 1028): neutral.  profiles/r03_kgroups_ab.txt)
template <int BM, int BN, int WM, int WN, int NPL, bool F16>
__global__ __launch_bounds__(256, (NPL == 3 && BM == 128 && BN == 128) ? 3 : 1) void igemm_pl_gather_kernel(const PlGatherParams p) {
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves");
  static_assert(BM % 64 == 0 && BN % 64 == 0, "row passes of 64");
  constexpr int A_PLANE = BM * LDH, B_PLANE = BN * LDH;
  constexpr int AR = BM / 64, NB = BN / 64;
  constexpr int NT = NPL == 3 ? 6 : 1;           // product terms per K16 slab

  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  unsigned short* Ah = smem16;
  unsigned short* Bh = Ah + NPL * A_PLANE;
  int* pix = reinterpret_cast<int*>(reinterpret_cast<char*>(smem16) + pl_gather_main_bytes(BM, BN, WN, NPL));

  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  int mtile, ntile, cls_id, split;
  work_decode(xcd_remap(blockIdx.x, gridDim.x, p.xcd), p.mt, p.nt, p.ncls, p.nsplit, p.order, p.mgroup, mtile, ntile, cls_id, split);
  if (mtile < 0) return;
  const TapClass tc = p.cls[cls_id];
  const int M = p.B * p.Hg * p.Wg;
  const int m0 = mtile * BM, n0 = ntile * BN;
  const int ntaps = tc.nty * tc.ntx;
  const int Cg = p.Cs >> 3;                      // granules per tap
  const int Kg = ntaps * Cg;
  const int KT = (Kg + 3) >> 2;
  const int kt_per = (KT + p.nsplit - 1) / p.nsplit;
  const int kt0 = split * kt_per;
  const int kt1 = min(KT, kt0 + kt_per);

  __amdgpu_buffer_rsrc_t src_rs[NPL], w_rs[NPL];
#pragma unroll
  for (int pl = 0; pl < NPL; pl++) {
    src_rs[pl] = make_rsrc(p.src + pl * p.src_ps, (((size_t)p.B * p.Hs * p.Ws - 1) * (size_t)p.lds + (size_t)(p.gpx ? p.lds : p.Cs)) * 2);
    w_rs[pl] = make_rsrc(p.w + pl * p.w_ps, (size_t)p.wtaps * p.N * p.Cs * 2);
  }

  const int kq = tid & 3;
  const int lds2 = p.lds * 2;
  // row r of the tile -> site (b, yg, xg); valid: the site exists
  auto site_of = [&](int r, int& b, int& yg, int& xg) -> bool {
    if (p.tw_log) {
      // 2-D tiles: (BM / TW) rows x TW sites.  With a source stride of 2 a 4 x 32 tile of a 5x5 conv touches 11 x 67 source
      // pixels, a 1 x 128 tile 5 x 259: 43 % fewer bytes through L2 for the layers whose source does not stay in L2
      // between taps (conv2 forward moved 1.5 GB at 5.9 TB/s, the Infinity-Cache rate).
      int t = mtile;
      const int txi = t % p.tiles_x; t /= p.tiles_x;
      const int tyi = t % p.tiles_y;
      b = t / p.tiles_y;
      yg = tyi * (BM >> p.tw_log) + (r >> p.tw_log);
      xg = (txi << p.tw_log) + (r & ((1 << p.tw_log) - 1));
      return true;
    }
    const int m = m0 + r;
    xg = m % p.Wg;
    const int t = m / p.Wg;
    yg = t % p.Hg;
    b = t / p.Hg;
    return m < M;
  };
  int a_yx[AR], a_lin[AR];     // (y << 16 | x) of the site's source origin; byte offset of that pixel (or the OOB mark)
#pragma unroll
  for (int i = 0; i < AR; i++) {
    int b, yg, xg;
    if (site_of((tid >> 2) + 64 * i, b, yg, xg)) {
      const int y = yg * p.sm, x = xg * p.sm;
      a_yx[i] = (y << 16) | x;
      a_lin[i] = (b * p.Hs * p.Ws + y * p.Ws + x) * lds2;
    } else {
      a_yx[i] = 0;
      a_lin[i] = OOB_MARK;   // rows past M: every load out of range (zeros)
    }
  }
  if (tid < BM) {
    int b, yg, xg, v = -1;
    if (site_of(tid, b, yg, xg)) v = (b * p.Hd + yg * p.so + tc.py) * p.Wd + xg * p.so + tc.px;
    pix[tid] = v;
  }
  int b_row[NB];
#pragma unroll
  for (int i = 0; i < NB; i++) {
    const int n = n0 + (tid >> 2) + 64 * i;
    b_row[i] = n < p.N ? n * p.Cs * 2 : OOB_MARK;
  }

  // K walker in granules: (tap, granule-in-tap) of the 16-byte column this thread loads, advanced by 4 per tile
  const int adv_t = 4 / Cg, adv_c = 4 - adv_t * Cg;
  const unsigned ntx_magic = (65536u + (unsigned)tc.ntx - 1u) / (unsigned)tc.ntx;  // exact for tap < 2^10
  int q_tap, q_cg;
  {
    const unsigned q = (unsigned)kt0 * 4u + (unsigned)kq;
    q_tap = (int)(q / (unsigned)Cg);
    q_cg = (int)(q - (unsigned)q_tap * (unsigned)Cg);
  }
  bool live = kt0 < kt1;

  u32x4 ra[AR][NPL], rb[NB][NPL];
  int dy, dx, a_tile, w_tile;

  auto piece_begin = [&]() {
    const int ty = (int)(((unsigned)q_tap * ntx_magic) >> 16), tx = q_tap - ty * tc.ntx;
    const bool kvalid = live && q_tap < ntaps;
    dy = tc.dy0 + ty * p.dstep;
    // two-pixel granules (rgb4_form, row length 4): granule q_cg of a tap row starts gpx * q_cg pixels to the right, which
    // is also its address (gpx * lds2 = 16 bytes) — the bounds check of piece_a then covers each granule separately
    dx = tc.dx0 + tx * p.dstep + p.gpx * q_cg;
    const int cofs = q_cg * 16;
    a_tile = kvalid ? (dy * p.Ws + dx) * lds2 + (p.gpx ? 0 : cofs) : OOB_MARK;
    const int widx = (tc.ky0 + ty * p.kstep) * p.KW + tc.kx0 + tx * p.kstep;
    w_tile = kvalid ? widx * p.N * p.Cs * 2 + cofs : OOB_MARK;
  };
  auto piece_a = [&](int i) {
    const int y = (a_yx[i] >> 16) + dy, x = (a_yx[i] & 0xffff) + dx;
    const bool inb = (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;
    const int voff = inb ? a_lin[i] + a_tile : OOB_MARK;
#pragma unroll
    for (int pl = 0; pl < NPL; pl++) ra[i][pl] = buf_ld16(src_rs[pl], voff);
  };
  auto piece_b = [&](int i) {
    const int voff = b_row[i] + w_tile;
#pragma unroll
    for (int pl = 0; pl < NPL; pl++) rb[i][pl] = buf_ld16(w_rs[pl], voff);
  };
  auto piece_end = [&]() {
    q_cg += adv_c;
    q_tap += adv_t;
    const bool wrap = q_cg >= Cg;
    q_cg -= wrap ? Cg : 0;
    q_tap += wrap ? 1 : 0;
  };
  constexpr int NPIECE = AR + NB + 2;
  auto piece = [&](int step) {
    if (step == 0) piece_begin();
    if (step >= 1 && step <= AR) piece_a(step - 1);
    if (step > AR && step <= AR + NB) piece_b(step - AR - 1);
    if (step == AR + NB + 1) piece_end();
  };
  auto swz = [](int row, int g) { return row * LDH + 8 * (g ^ ((row >> 2) & 3)); };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < AR; i++)
#pragma unroll
      for (int pl = 0; pl < NPL; pl++)
        *reinterpret_cast<u32x4*>(Ah + pl * A_PLANE + swz((tid >> 2) + 64 * i, kq)) = ra[i][pl];
#pragma unroll
    for (int i = 0; i < NB; i++)
#pragma unroll
      for (int pl = 0; pl < NPL; pl++)
        *reinterpret_cast<u32x4*>(Bh + pl * B_PLANE + swz((tid >> 2) + 64 * i, kq)) = rb[i][pl];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int st = 0; st < NPIECE; st++) piece(st);
  store_tile();
  __syncthreads();
  const unsigned short* ah_rd = Ah + (wm * WM + l31) * LDH;
  const unsigned short* bh_rd = Bh + (wn * WN + l31) * LDH;
  const int gsw = lh ^ ((l31 >> 2) & 3);     // swizzled granule of K16 slab 0 (slab 1: ^ 2); tile bases are multiples of 32
  constexpr int NGROUP = 2 * TM * NT;        // MFMA groups per tile between which the load pieces are placed
  constexpr int PPG = (NPIECE + NGROUP - 1) / NGROUP;

  PHASE_DECL;
  for (int kt = kt0; kt < kt1; kt++) {
    PHASE_STAMP(6);
    live = kt + 1 < kt1;
#pragma unroll
    for (int slab = 0; slab < 2; slab++) {
      // B fragments of the slab stay live; A fragments are read per 32-row sub-tile (keeps the 128x128 kernel at 3 waves/SIMD)
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
          av[pl] = *reinterpret_cast<const s16x8*>(ah_rd + pl * A_PLANE + i * 32 * LDH + 8 * (gsw ^ (2 * slab)));
#pragma unroll
        for (int t = 0; t < NT; t++) {
#pragma unroll
          for (int j = 0; j < TN; j++) mfma_terms<NPL, F16>(av, bv[j], acc[i][j], t);
#pragma unroll
          for (int q = 0; q < PPG; q++) piece(((slab * TM + i) * NT + t) * PPG + q);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    PHASE_STAMP(0);   // MFMA phase (fragment reads, the MFMAs, the next tile's loads issued)
    __syncthreads();  // every wave is done reading this tile
    PHASE_STAMP(1);
#ifdef UNFLOW_PHASE_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PHASE_STAMP(2);   // the next tile's loads landed
#endif
    store_tile();     // (last iteration: zeros, never read)
    PHASE_STAMP(3);
    __syncthreads();
    PHASE_STAMP(4);
  }
  PHASE_FLUSH;

  pl_gather_epilogue<WM, WN>(p, acc, pix, smem16, wm, wn, wid, lane, n0, split);
}

// ------------------------------------------------------------------------------------------------ gather kernel, ping-pong form
// Two 4-wave INSTANCES of the 128 x 128 gather tile in one workgroup of 8 waves (neighbouring M tiles, the same N tile /
// class / K split, so both walk the same K tiles and share the weight tile), scheduled against each other like the
// filter-gradient kernel above: while instance 0 runs the 48 MFMAs of K tile t, instance 1 reads ITS fragments of tile t
// from LDS into registers, stores tile t + 1 (global loads issued one slot pair earlier) and issues the loads of tile
// t + 2; then they swap.  One workgroup barrier per slot.  A tiles are private and double-buffered, the weight tile is shared
// and double-buffered (each instance stores half of it; the other half arrives one slot later, a full slot before its first
// reader).  LDS rows are unpadded (64 B) with the XOR swizzle alone: 6 x 24.6 KB = 147 KB, one workgroup per CU.
template <int NPL, bool F16>
__global__ __launch_bounds__(512, 1) void igemm_pl_gather_pp_kernel(const PlGatherParams p) {
  constexpr int BM = 128, BN = 128, WM = 64, WN = 64;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LDP = 32;                        // row pitch (elements): no padding
  constexpr int A_PLANE = BM * LDP, B_PLANE = BN * LDP;
  constexpr int A_TILE = NPL * A_PLANE, B_TILE = NPL * B_PLANE;
  constexpr int AR = BM / 64;
  constexpr int NT = NPL == 3 ? 6 : 1;
  static_assert(NPL == 3 && !F16, "bf16 x 3 only");

  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  const int inst = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6;
  unsigned short* Abase = smem16 + inst * 2 * A_TILE;          // [buffer][plane][row][32]
  unsigned short* Bbase = smem16 + 4 * A_TILE;                 // [buffer][plane][row][32], shared
  int* pix = reinterpret_cast<int*>(smem16 + 4 * A_TILE + 2 * B_TILE) + inst * BM;

  const int wm = wid >> 1, wn = wid & 1;
  int mpair, ntile, cls_id, split;
  work_decode(xcd_remap(blockIdx.x, gridDim.x, p.xcd), p.mt, p.nt, p.ncls, p.nsplit, p.order, p.mgroup, mpair, ntile, cls_id, split);
  if (mpair < 0) return;                         // (padding workgroup of order 2: both instances leave)
  const int mtile = 2 * mpair + inst;
  const TapClass tc = p.cls[cls_id];
  const int M = p.B * p.Hg * p.Wg;
  const int mt_real = p.tw_log ? p.B * p.tiles_y * p.tiles_x : (M + BM - 1) / BM;
  const bool tile_ok = mtile < mt_real;          // an odd tile count leaves the last instance without a tile: it runs on zeros
  const int m0 = mtile * BM, n0 = ntile * BN;
  const int ntaps = tc.nty * tc.ntx;
  const int Cg = p.Cs >> 3;
  const int Kg = ntaps * Cg;
  const int KT = (Kg + 3) >> 2;
  const int kt_per = (KT + p.nsplit - 1) / p.nsplit;
  const int kt0 = split * kt_per;
  const int kt1 = min(KT, kt0 + kt_per);
  const int T = max(kt1 - kt0, 0);

  __amdgpu_buffer_rsrc_t src_rs[NPL], w_rs[NPL];
#pragma unroll
  for (int pl = 0; pl < NPL; pl++) {
    src_rs[pl] = make_rsrc(p.src + pl * p.src_ps, (((size_t)p.B * p.Hs * p.Ws - 1) * (size_t)p.lds + (size_t)(p.gpx ? p.lds : p.Cs)) * 2);
    w_rs[pl] = make_rsrc(p.w + pl * p.w_ps, (size_t)p.wtaps * p.N * p.Cs * 2);
  }
  const int kq = tid & 3;
  const int lds2 = p.lds * 2;
  auto site_of = [&](int r, int& b, int& yg, int& xg) -> bool {
    if (!tile_ok) return false;
    if (p.tw_log) {
      int t = mtile;
      const int txi = t % p.tiles_x; t /= p.tiles_x;
      const int tyi = t % p.tiles_y;
      b = t / p.tiles_y;
      yg = tyi * (BM >> p.tw_log) + (r >> p.tw_log);
      xg = (txi << p.tw_log) + (r & ((1 << p.tw_log) - 1));
      return true;
    }
    const int m = m0 + r;
    xg = m % p.Wg;
    const int t = m / p.Wg;
    yg = t % p.Hg;
    b = t / p.Hg;
    return m < M;
  };
  int a_yx[AR], a_lin[AR];
#pragma unroll
  for (int i = 0; i < AR; i++) {
    int b, yg, xg;
    if (site_of((tid >> 2) + 64 * i, b, yg, xg)) {
      const int y = yg * p.sm, x = xg * p.sm;
      a_yx[i] = (y << 16) | x;
      a_lin[i] = (b * p.Hs * p.Ws + y * p.Ws + x) * lds2;
    } else {
      a_yx[i] = 0;
      a_lin[i] = OOB_MARK;
    }
  }
  if (tid < BM) {
    int b, yg, xg, v = -1;
    if (site_of(tid, b, yg, xg)) v = (b * p.Hd + yg * p.so + tc.py) * p.Wd + xg * p.so + tc.px;
    pix[tid] = v;
  }
  // this instance's half of the weight tile: rows 64 inst + tid / 4
  const int b_r = 64 * inst + (tid >> 2);
  const int b_row = n0 + b_r < p.N ? (n0 + b_r) * p.Cs * 2 : OOB_MARK;

  const int adv_t = 4 / Cg, adv_c = 4 - adv_t * Cg;
  const unsigned ntx_magic = (65536u + (unsigned)tc.ntx - 1u) / (unsigned)tc.ntx;
  int q_tap, q_cg;
  {
    const unsigned q = (unsigned)kt0 * 4u + (unsigned)kq;
    q_tap = (int)(q / (unsigned)Cg);
    q_cg = (int)(q - (unsigned)q_tap * (unsigned)Cg);
  }
  u32x4 ra[AR][NPL], rb[NPL];
  // loads of the next not-yet-requested K tile (the walker advances by one tile per call); past the block's range: zeros
  int ld_left = T;
  auto load_tile = [&]() {
    const int ty = (int)(((unsigned)q_tap * ntx_magic) >> 16), tx = q_tap - ty * tc.ntx;
    const bool kvalid = ld_left > 0 && q_tap < ntaps;
    const int dy = tc.dy0 + ty * p.dstep;
    const int dx = tc.dx0 + tx * p.dstep + p.gpx * q_cg;
    const int cofs = q_cg * 16;
    const int a_tile = kvalid ? (dy * p.Ws + dx) * lds2 + (p.gpx ? 0 : cofs) : OOB_MARK;
    const int widx = (tc.ky0 + ty * p.kstep) * p.KW + tc.kx0 + tx * p.kstep;
    const int w_tile = kvalid ? widx * p.N * p.Cs * 2 + cofs : OOB_MARK;
#pragma unroll
    for (int i = 0; i < AR; i++) {
      const int y = (a_yx[i] >> 16) + dy, x = (a_yx[i] & 0xffff) + dx;
      const bool inb = (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;
      const int voff = inb ? a_lin[i] + a_tile : OOB_MARK;
#pragma unroll
      for (int pl = 0; pl < NPL; pl++) ra[i][pl] = buf_ld16(src_rs[pl], voff);
    }
    {
      const int voff = b_row + w_tile;
#pragma unroll
      for (int pl = 0; pl < NPL; pl++) rb[pl] = buf_ld16(w_rs[pl], voff);
    }
    q_cg += adv_c;
    q_tap += adv_t;
    const bool wrap = q_cg >= Cg;
    q_cg -= wrap ? Cg : 0;
    q_tap += wrap ? 1 : 0;
    ld_left--;
  };
  auto swz = [](int row, int g) { return row * LDP + 8 * (g ^ ((row >> 2) & 3)); };
  auto store_tile = [&](int buf) {
    unsigned short* Ah = Abase + buf * A_TILE;
    unsigned short* Bh = Bbase + buf * B_TILE;
#pragma unroll
    for (int i = 0; i < AR; i++)
#pragma unroll
      for (int pl = 0; pl < NPL; pl++)
        *reinterpret_cast<u32x4*>(Ah + pl * A_PLANE + swz((tid >> 2) + 64 * i, kq)) = ra[i][pl];
#pragma unroll
    for (int pl = 0; pl < NPL; pl++) *reinterpret_cast<u32x4*>(Bh + pl * B_PLANE + swz(b_r, kq)) = rb[pl];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  const int gsw = lh ^ ((l31 >> 2) & 3);
  const int a_rd = (wm * WM + l31) * LDP, b_rd = (wn * WN + l31) * LDP;
  s16x8 av[2][TM][NPL], bv[2][TN][NPL];
  auto read_frags = [&](int buf) {
    const unsigned short* Ah = Abase + buf * A_TILE + a_rd;
    const unsigned short* Bh = Bbase + buf * B_TILE + b_rd;
#pragma unroll
    for (int slab = 0; slab < 2; slab++)
#pragma unroll
      for (int pl = 0; pl < NPL; pl++) {
#pragma unroll
        for (int j = 0; j < TN; j++)
          bv[slab][j][pl] = *reinterpret_cast<const s16x8*>(Bh + pl * B_PLANE + j * 32 * LDP + 8 * (gsw ^ (2 * slab)));
#pragma unroll
        for (int i = 0; i < TM; i++)
          av[slab][i][pl] = *reinterpret_cast<const s16x8*>(Ah + pl * A_PLANE + i * 32 * LDP + 8 * (gsw ^ (2 * slab)));
      }
  };
  auto slot_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  if (T > 0) {
    load_tile();                                 // tile 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_tile(0);
    load_tile();                                 // tile 1 (in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    slot_end();                                  // both halves of weight tile 0 are stored
    if (inst == 1) slot_end();
#pragma unroll 1
    for (int t = 0; t < T; t++) {
      // read slot: fragments of tile t -> registers; tile t + 1 (loads requested one slot pair ago) -> LDS; request tile t + 2
      read_frags(t & 1);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      store_tile((t + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      load_tile();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // stores done (visible after the barrier), fragments in registers
      slot_end();
      // multiply slot
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int slab = 0; slab < 2; slab++)
#pragma unroll
        for (int tt = 0; tt < NT; tt++)
#pragma unroll
          for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) mfma_terms<NPL, F16>(av[slab][i], bv[slab][j], acc[i][j], tt);
      __builtin_amdgcn_s_setprio(0);
      slot_end();
    }
    if (inst == 0) slot_end();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the zero loads past the last tile
  }
  __syncthreads();                               // every wave is done with the tiles: the epilogue stages through them
  if (!tile_ok) return;
  pl_gather_epilogue<WM, WN>(p, acc, pix, smem16 + inst * 2 * A_TILE, wm, wn, wid, lane, n0, split);
}

// (the halo gather kernel: halo_kernel.h)

// (Round 3 measured and dropped, profiles/r03_halo_tall_ab.txt: a 256-site (8 x 32) form of this kernel with 8 waves per
// workgroup, the weight tile double-buffered (one barrier per tap) and half the weight bytes per MFMA — passes the same
// tests, slower on every layer it applied to (conv2 forward +39 us, conv3 data gradient +56 us; step 607 vs 620 pairs/s):
// one barrier domain of 8 waves stalls more than two independent 4-wave workgroups per CU.)
// Fixed-order sum of the split-K partials + the epilogue (+ the output planes).  One float4 per thread.
__global__ void pl_splitk_reduce_epilogue_kernel(const PlGatherParams p, int vec) {
  const size_t npix = (size_t)p.B * p.Hd * p.Wd;
  const size_t total = npix * p.N;
  if (vec) {
    const unsigned nq = (unsigned)(p.N >> 2);
    const size_t totq = total >> 2;
    for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < totq; q += (size_t)gridDim.x * blockDim.x) {
      const size_t px = q / nq;
      const int n = (int)(q - px * nq) * 4;
      const float4* src = reinterpret_cast<const float4*>(p.partial) + q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
      for (int s = 0; s < p.nsplit; s++) {
        const f32x4_nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(src) + (size_t)s * totq);   // read once
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      epi_store4(p, px, n, v);
    }
    return;
  }
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t px = e / p.N;
    const int n = (int)(e - px * p.N);
    float v = 0.f;
    for (int s = 0; s < p.nsplit; s++) v += p.partial[(size_t)s * total + e];
    v *= p.src_inv;
    if (p.bias) v += p.bias[n];
    if (p.leaky) v = leaky_relu(v);
    float* d = p.dst + px * p.ldd + n;
    if (p.accumulate) v += *d;
    if (n >= p.act_lo && n < p.act_hi) {
      if (p.act_src) v *= leaky_grad_from_out(p.act_src[px * p.ld_act + n]);
      else if (p.act_pl) v *= leaky_grad_from_bits(p.act_pl[px * p.ld_act + n]);
    }
    if (p.dst) *d = v;
    store_planes(p.pl, px, n, v);
  }
}


// ------------------------------------------------------------------------------------------------ filter gradient
struct PlWgradParams : WgradGeom {   // Ca: plane channels walked per tap (multiple of 8); Cb: columns of dW
  const unsigned short* src;   // gathered operand planes [B,Hs,Ws,lds]
  long src_ps;
  const unsigned short* dst;   // dense operand planes [B,Hg,Wg,ldd]
  long dst_ps;
  float* out;                  // dW [(tap, a < Ca_out)][b < Cb]
  float* partial;              // [nsplit][taps*Ca_out][Cb]
  int lds, ldd;
  int Ca_out;                  // rows per tap of dW (the weight tensor's own channel padding, <= Ca)
  float out_scale;             // 1 / (scale of the gathered planes x scale of the dense planes)
  int nsplit;
  int gpx;                     // conv1 form: granule ag of a tap row starts gpx * ag pixels to the right
  unsigned cag_magic;          // ceil(2^32 / (Ca/8))
  int mt, nt, xcd;             // 1-D grid of mt * nt * nsplit workgroups, XCD-contiguous in (split, N tile, M tile) order
};

// LDS image of one operand plane: [k = 32 sites][ROWS channels], rows of ROWS*2 bytes; 32-byte pairs of granules
// XOR-swizzled by the row so that the 8 (row, pair) segments one half-wave's ds_read_b64_tr_b16 touches are distinct
// 32-byte slots of a 256-byte bank row.
template <int ROWS>
__device__ __forceinline__ int tr_swz(int k, int granule) {
  const int pair = granule >> 1;
  const int sw = ROWS == 128 ? ((k & 3) << 1) : (((k >> 1) & 1) << 1);   // 256-byte rows: 4 rows differ; 128-byte: 2 per bank row
  return k * ROWS + (((pair ^ sw) << 1) | (granule & 1)) * 8;
}

template <int BM, int BN, int WM, int WN, int NPL, bool F16>
__global__ __launch_bounds__(256, (NPL == 3 && BM == 128 && BN == 128) ? 3 : 1) void igemm_pl_wgrad_kernel(const PlWgradParams p) {
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves");
  static_assert((BM == 128 || BM == 64) && (BN == 128 || BN == 64), "swizzle forms");
  constexpr int A_PLANE = BK * BM, B_PLANE = BK * BN;
  constexpr int AJ = BM / 8, BJ = BN / 8;        // granules per k row
  constexpr int AKS = 256 / AJ, BKS = 256 / BJ;  // k rows per pass
  constexpr int AR = BK / AKS, BR = BK / BKS;    // passes
  constexpr int NT = NPL == 3 ? 6 : 1;

  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  unsigned short* Ah = smem16;
  unsigned short* Bh = Ah + NPL * A_PLANE;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int taps = p.KH * p.KW;
  const int Cag = p.Ca >> 3;
  const int Mg = taps * Cag;                     // row granules of the (padded) problem
  // Work order: all (M tile, N tile) blocks of one K split are adjacent and an XCD runs a contiguous run of that order, so
  // the blocks that read the same range of sites (every tap / channel chunk of the gathered operand, every column tile
  // of the dense one) share one L2.  With the (x, y, z) grid dealt round-robin to the XCDs they did not: the filter
  // gradients of conv2 / conv3 / conv3_1 moved 1.6-1.9 GB each (6-7 TB/s from the Infinity Cache) for 0.25 GB of operands.
  int v = xcd_remap(blockIdx.x, gridDim.x, p.xcd);
  const int mtile = v % p.mt; v /= p.mt;
  const int ntile = v % p.nt;
  const int split = v / p.nt;
  const int m0 = mtile * BM, n0 = ntile * BN;
  const int S = p.B * p.Hg * p.Wg;               // reduction length (sites)
  const int KT = (S + BK - 1) / BK;
  const int kt_per = (KT + p.nsplit - 1) / p.nsplit;
  const int kt0 = split * kt_per, kt1 = min(KT, kt0 + kt_per);

  __amdgpu_buffer_rsrc_t src_rs[NPL], dst_rs[NPL];
#pragma unroll
  for (int pl = 0; pl < NPL; pl++) {
    src_rs[pl] = make_rsrc(p.src + pl * p.src_ps, (((size_t)p.B * p.Hs * p.Ws - 1) * (size_t)p.lds + (size_t)(p.gpx ? p.lds : p.Ca)) * 2);
    dst_rs[pl] = make_rsrc(p.dst + pl * p.dst_ps, (((size_t)S - 1) * (size_t)p.ldd + (size_t)((p.Cb + 7) & ~7)) * 2);
  }

  // gathered operand: this thread's row granule (tap, 8 channels) — fixed
  const int aj = tid % AJ, ak = tid / AJ;
  const int mg = (m0 >> 3) + aj;
  const bool m_ok = mg < Mg;
  const unsigned tap = p.cag_magic ? fast_div((unsigned)mg, p.cag_magic) : (unsigned)mg;   // magic 0: Ca == 8, one granule per tap
  const int ag = mg - (int)tap * Cag;
  const int ky = (int)tap / p.KW, kx = (int)tap - ky * p.KW;
  const int dy = p.dy0 + ky, dx = p.dx0 + kx + p.gpx * ag;
  const int lds2 = p.lds * 2;
  const int a_lane_off = (dy * p.Ws + (p.dx0 + kx)) * lds2 + ag * 16;   // may be negative; added to the site's base offset
  // dense operand: this thread's column granule
  const int bj = tid % BJ, bk = tid / BJ;
  const int nbq = (n0 >> 3) + bj;
  const bool b_ok = nbq * 8 < p.Cb;
  const int b_lane_off = nbq * 16;
  const int ldd2 = p.ldd * 2;
  const unsigned magW = (unsigned)((0x100000000ull + p.Wg - 1) / p.Wg), magH = (unsigned)((0x100000000ull + p.Hg - 1) / p.Hg);

  u32x4 ra[AR][NPL], rb[BR][NPL];
  auto piece_a = [&](int i, int kt) {
    const unsigned sidx = (unsigned)(kt * BK + ak + AKS * i);
    const unsigned q = fast_div(sidx, magW);
    const int xg = (int)(sidx - q * (unsigned)p.Wg);
    const unsigned bb = fast_div(q, magH);
    const int yg = (int)(q - bb * (unsigned)p.Hg);
    const int yb = yg * p.sm, xb = xg * p.sm;
    const bool ok = m_ok && (int)bb < p.B && (unsigned)(yb + dy) < (unsigned)p.Hs && (unsigned)(xb + dx) < (unsigned)p.Ws;
    const int voff = ok ? (((int)bb * p.Hs + yb) * p.Ws + xb) * lds2 + a_lane_off : OOB_MARK;
#pragma unroll
    for (int pl = 0; pl < NPL; pl++) ra[i][pl] = buf_ld16(src_rs[pl], voff);
  };
  auto piece_b = [&](int i, int kt) {
    const int sidx = kt * BK + bk + BKS * i;
    const int voff = (b_ok && sidx < S) ? sidx * ldd2 + b_lane_off : OOB_MARK;
#pragma unroll
    for (int pl = 0; pl < NPL; pl++) rb[i][pl] = buf_ld16(dst_rs[pl], voff);
  };
  constexpr int NPIECE = AR + BR;
  auto piece = [&](int step, int kt) {
    if (step < AR) piece_a(step, kt);
    else if (step < AR + BR) piece_b(step - AR, kt);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < AR; i++)
#pragma unroll
      for (int pl = 0; pl < NPL; pl++)
        *reinterpret_cast<u32x4*>(Ah + pl * A_PLANE + tr_swz<BM>(ak + AKS * i, aj)) = ra[i][pl];
#pragma unroll
    for (int i = 0; i < BR; i++)
#pragma unroll
      for (int pl = 0; pl < NPL; pl++)
        *reinterpret_cast<u32x4*>(Bh + pl * B_PLANE + tr_swz<BN>(bk + BKS * i, bj)) = rb[i][pl];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // transposing reads: within a 16-lane group, lane i16 supplies the address of 4 consecutive channels (i16 & 3) of k row
  // (i16 >> 2) and receives channel i16 of those 4 k rows.  Lane l: group = l >> 4 -> channels 16*(group & 1).., k half
  // group >> 1 (the MFMA's lane >> 5).
  const int i16 = lane & 15, grp = lane >> 4, lh = grp >> 1;
  const int krow = 8 * lh + (i16 >> 2);     // + 16*slab + 4*half at read time (multiples of 4: the swizzle term is unchanged)
  int a_rd[TM], b_rd[TN];
#pragma unroll
  for (int i = 0; i < TM; i++) {
    const int c = wm * WM + i * 32 + 16 * (grp & 1) + 4 * (i16 & 3);    // first of the 4 channels this lane addresses
    a_rd[i] = tr_swz<BM>(krow, c >> 3) + (c & 7);
  }
#pragma unroll
  for (int j = 0; j < TN; j++) {
    const int c = wn * WN + j * 32 + 16 * (grp & 1) + 4 * (i16 & 3);
    b_rd[j] = tr_swz<BN>(krow, c >> 3) + (c & 7);
  }
#pragma unroll
  for (int st = 0; st < NPIECE; st++) piece(st, kt0);
  store_tile();
  __syncthreads();
  constexpr int NGROUP = 2 * NT;
  constexpr int PPG = (NPIECE + NGROUP - 1) / NGROUP;
  const int l31 = lane & 31;
  for (int kt = kt0; kt < kt1; kt++) {
    const int ktn = kt + 1 < kt1 ? kt + 1 : KT + 1;   // past the last tile: every load is out of range (zeros)
    s16x8 av[TM][NPL], bv[TN][NPL];
    auto read_slab = [&](int slab) {
#pragma unroll
      for (int pl = 0; pl < NPL; pl++) {
#pragma unroll
        for (int i = 0; i < TM; i++) {
          const unsigned short* b0 = Ah + pl * A_PLANE + a_rd[i] + (16 * slab) * BM;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(b0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(b0 + 4 * BM));
          av[i][pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
          const unsigned short* b0 = Bh + pl * B_PLANE + b_rd[j] + (16 * slab) * BN;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(b0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(b0 + 4 * BN));
          bv[j][pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
    };
    // one fragment set at a time: three waves per SIMD hide the read latency (both slabs' sets in flight, two waves per SIMD,
    // measured slower)
#pragma unroll
    for (int slab = 0; slab < 2; slab++) {
      read_slab(slab);
#pragma unroll
      for (int t = 0; t < NT; t++) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) mfma_terms<NPL, F16>(av[i], bv[j], acc[i][j], t);
#pragma unroll
        for (int q2 = 0; q2 < PPG; q2++) piece((slab * NT + t) * PPG + q2, ktn);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    store_tile();
    __syncthreads();
  }

  float* o = p.nsplit > 1 ? p.partial + (size_t)split * taps * p.Ca_out * p.Cb : p.out;
  const int lh5 = lane >> 5;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh5;
      const int mgr = m >> 3;
      if (mgr >= Mg) continue;
      const unsigned tp = p.cag_magic ? fast_div((unsigned)mgr, p.cag_magic) : (unsigned)mgr;
      const int a = (mgr - (int)tp * Cag) * 8 + (m & 7);
      if (a >= p.Ca_out) continue;
      float* orow = o + ((size_t)tp * p.Ca_out + a) * p.Cb;
#pragma unroll
      for (int j = 0; j < TN; j++) {
        const int n = n0 + wn * WN + j * 32 + l31;
        if (n < p.Cb) orow[n] = acc[i][j][r] * p.out_scale;
      }
    }
}

// ------------------------------------------------------------------------------------------------ filter gradient, LDS-DMA form
// Same product, tiles, LDS image and MFMA loop as igemm_pl_wgrad_kernel (3 planes), different staging: the operand rows
// ([site][channels], 256 contiguous bytes per site and plane) go HBM/L2 -> LDS directly (buffer_load_dwordx4 ... lds, no
// staging registers, no ds_write pass), in stages of 16 sites, double-buffered: the six loads of stage s+1 are in flight
// while the 24 MFMAs of stage s run, and ONE barrier per stage publishes them (the register-staged kernel needs two per 32
// sites around its ds_write pass, and its loads could only start once the previous tile's registers were stored).  A wave
// instruction writes lane-linear: 64 lanes x 16 bytes = 4 site rows x 16 granule slots, so the XOR swizzle of the
// transposing-read image is applied on the SOURCE side — lane (row, slot) fetches the granule that belongs in that slot.
// Out-of-range rows (image border taps, the tail of the site range, channel padding) are buffer offsets >= num_records:
// the hardware writes zeros.  48 KB of LDS and <= 168 registers per K group.
//
// (Round 3 measured and dropped, profiles/r03_kgroups_ab.txt: 768-thread workgroups = three 4-wave K groups owning the same
// tile and combining through the stage buffers after the loop — a third of the 50 MB of partial sums per layer: neutral
// where one round of such workgroups fills the chip (conv1 / conv2 / conv3 / deconv2), 8 % slower on conv3_1 at 80 % fill;
// the shared per-stage barrier couples the 12 waves.  No MFMAs for sub-tiles beyond the last M / N tile: neutral.)
// F16 (round 6, BASELINE configs[4]): the operands are ONE fp16 plane; the three "planes" of a stage are then three CONSECUTIVE
// 16-site groups (a stage = 48 sites), each fetched with its own site offsets through the same descriptor, and a fragment set
// takes three products (group t of A with group t of B: mfma_terms<3, true>) instead of the six cross terms — the staging, the
// LDS image, the transposing reads and the loop are the bf16 kernel's.  (The register-staged fp16 kernel it replaces ran 8
// MFMAs between two barriers around a ds_write pass.)
template <int BN, int WN, bool F16 = false>
__global__ __launch_bounds__(256, 3) void igemm_pl_wgrad_dma_kernel(const PlWgradParams p) {
  constexpr int BM = 128, WM = 64, NPL = 3, KS = 16;
  constexpr int GPS = F16 ? 3 : 1;                               // 16-site groups per stage
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves");
  static_assert(BN == 128 || BN == 64, "swizzle forms");
  constexpr int A_PLANE = KS * BM, B_PLANE = KS * BN;            // elements
  constexpr int STAGE = NPL * (A_PLANE + B_PLANE);
  constexpr int B_RPI = 512 / BN;                                // site rows per wave instruction: 4 (256-byte rows) / 8
  constexpr int B_NI = KS / B_RPI;                               // wave instructions per plane and stage: 4 / 2
  constexpr int B_SLOTS = BN / 8;

  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int taps = p.KH * p.KW;
  const int Cag = p.Ca >> 3;
  const int Mg = taps * Cag;
  int v = xcd_remap(blockIdx.x, gridDim.x, p.xcd);               // work order: see igemm_pl_wgrad_kernel
  const int mtile = v % p.mt; v /= p.mt;
  const int ntile = v % p.nt;
  const int split = v / p.nt;
  const int m0 = mtile * BM, n0 = ntile * BN;
  const int S = p.B * p.Hg * p.Wg;
  const int KT = ((S + KS - 1) / KS + GPS - 1) / GPS;            // stages
  const int kt_per = F16 ? (KT + p.nsplit - 1) / p.nsplit
                         : (((KT + 1) / 2 + p.nsplit - 1) / p.nsplit) * 2;   // whole 32-site tiles per split, as the planner counts them
  const int kt0 = split * kt_per, kt1 = min(KT, kt0 + kt_per);

  // buffer descriptors as plain dwords (the LDS-DMA loads are inline asm: hipcc would otherwise wait vmcnt(0) in front of
  // every ds_read that follows a load into the same LDS array, i.e. drain the stage in flight)
  u32x4 src_rs[NPL], dst_rs[NPL];
#pragma unroll
  for (int pl = 0; pl < NPL; pl++) {
    src_rs[pl] = raw_rsrc(p.src + (F16 ? 0 : pl * p.src_ps), (((size_t)p.B * p.Hs * p.Ws - 1) * (size_t)p.lds + (size_t)(p.gpx ? p.lds : p.Ca)) * 2);
    dst_rs[pl] = raw_rsrc(p.dst + (F16 ? 0 : pl * p.dst_ps), (((size_t)S - 1) * (size_t)p.ldd + (size_t)((p.Cb + 7) & ~7)) * 2);
  }

  // gathered operand: lane -> (site row 4*wid + lane/16 of the stage, slot lane%16); the granule stored in that slot
  const int a_k = 4 * wid + (lane >> 4);
  const int a_slot = lane & 15;
  const int a_g = ((((a_slot >> 1) ^ ((a_k & 3) << 1)) << 1) | (a_slot & 1));
  const int mg = (m0 >> 3) + a_g;
  const bool m_ok = mg < Mg;
  const unsigned tap = p.cag_magic ? fast_div((unsigned)mg, p.cag_magic) : (unsigned)mg;
  const int ag = mg - (int)tap * Cag;
  const int ky = (int)tap / p.KW, kx = (int)tap - ky * p.KW;
  const int dy = p.dy0 + ky, dx = p.dx0 + kx + p.gpx * ag;
  const int lds2 = p.lds * 2;
  const int a_lane_off = (dy * p.Ws + (p.dx0 + kx)) * lds2 + ag * 16;
  // dense operand
  const int b_k = B_RPI * wid + lane / B_SLOTS;                  // (waves >= B_NI issue no B load)
  const int b_slot = lane % B_SLOTS;
  const int b_sw = BN == 128 ? ((b_k & 3) << 1) : (((b_k >> 1) & 1) << 1);
  const int b_g = ((((b_slot >> 1) ^ b_sw) << 1) | (b_slot & 1));
  const int nbq = (n0 >> 3) + b_g;
  const bool b_ok = nbq * 8 < p.Cb;
  const int b_lane_off = nbq * 16;
  const int ldd2 = p.ldd * 2;
  const unsigned magW = (unsigned)((0x100000000ull + p.Wg - 1) / p.Wg), magH = (unsigned)((0x100000000ull + p.Hg - 1) / p.Hg);

  const unsigned lds0 = lds_addr(smem16);
  auto a_voff = [&](int kt) {                                    // kt: index of the 16-site group
    const unsigned sidx = (unsigned)(kt * KS + a_k);
    const unsigned q = fast_div(sidx, magW);
    const int xg = (int)(sidx - q * (unsigned)p.Wg);
    const unsigned bb = fast_div(q, magH);
    const int yg = (int)(q - bb * (unsigned)p.Hg);
    const int yb = yg * p.sm, xb = xg * p.sm;
    const bool ok = m_ok && (int)bb < p.B && (unsigned)(yb + dy) < (unsigned)p.Hs && (unsigned)(xb + dx) < (unsigned)p.Ws;
    return ok ? (((int)bb * p.Hs + yb) * p.Ws + xb) * lds2 + a_lane_off : OOB_MARK;
  };
  auto b_voff = [&](int kt) {
    const int sidx = kt * KS + b_k;
    return (b_ok && sidx < S) ? sidx * ldd2 + b_lane_off : OOB_MARK;
  };
  auto issue = [&](int kt, int buf) {
    const unsigned st = lds0 + (unsigned)(buf * STAGE * 2);      // byte address of the stage in LDS
    const unsigned d = st + (unsigned)(4 * wid * BM * 2);
    if constexpr (F16) {
#pragma unroll
      for (int g = 0; g < 3; g++) dma1(a_voff(3 * kt + g), src_rs[0], d + g * A_PLANE * 2);
    } else {
      dma3(a_voff(kt), src_rs[0], src_rs[1], src_rs[2], d, d + A_PLANE * 2, d + 2 * A_PLANE * 2);
    }
    if (wid < B_NI) {
      const unsigned e = st + (unsigned)((NPL * A_PLANE + B_RPI * wid * BN) * 2);
      if constexpr (F16) {
#pragma unroll
        for (int g = 0; g < 3; g++) dma1(b_voff(3 * kt + g), dst_rs[0], e + g * B_PLANE * 2);
      } else {
        dma3(b_voff(kt), dst_rs[0], dst_rs[1], dst_rs[2], e, e + B_PLANE * 2, e + 2 * B_PLANE * 2);
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

  const int i16 = lane & 15, grp16 = lane >> 4, lh = grp16 >> 1;
  const int krow = 8 * lh + (i16 >> 2);
  int a_rd[TM], b_rd[TN];
#pragma unroll
  for (int i = 0; i < TM; i++) {
    const int c = wm * WM + i * 32 + 16 * (grp16 & 1) + 4 * (i16 & 3);
    a_rd[i] = tr_swz<BM>(krow, c >> 3) + (c & 7);
  }
#pragma unroll
  for (int j = 0; j < TN; j++) {
    const int c = wn * WN + j * 32 + 16 * (grp16 & 1) + 4 * (i16 & 3);
    b_rd[j] = NPL * A_PLANE + tr_swz<BN>(krow, c >> 3) + (c & 7);
  }
  if (kt0 < kt1) issue(kt0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  PHASE_DECL;
  for (int kt = kt0; kt < kt1; kt++) {
    PHASE_STAMP(6);
    const int cur = (kt - kt0) & 1;
    // this stage's fragment reads first, then stage kt+1 -> the other buffer (every wave passed the barrier after its reads
    // of that buffer): the address arithmetic while the reads return, the loads between the first MFMA groups — a wave's
    // issue port is idle while its MFMAs execute; past the last stage the loads are all out of range (zeros into a buffer
    // nobody reads)
    const unsigned short* st = smem16 + cur * STAGE;
    s16x8 av[TM][NPL], bv[TN][NPL];
    auto read_a = [&](int pl) {
#pragma unroll
      for (int i = 0; i < TM; i++) {
        const unsigned short* b0 = st + pl * A_PLANE + a_rd[i];
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(b0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(b0 + 4 * BM));
        av[i][pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    };
    auto read_b = [&](int pl) {
#pragma unroll
      for (int j = 0; j < TN; j++) {
        const unsigned short* b0 = st + pl * B_PLANE + b_rd[j];
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(b0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(b0 + 4 * BN));
        bv[j][pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    };
    // fragment reads in the order the six terms want them (mfma_terms: A2 B0, A0 B2, A1 B1, ...)
    read_a(2); read_b(0); read_a(0); read_b(2); read_a(1); read_b(1);
    __builtin_amdgcn_sched_barrier(0);
    PHASE_STAMP(4);    // fragment reads issued
    // stage kt+1: addresses while the reads return, then the six loads spread between the first MFMA groups
    const int ktn = kt + 1 < kt1 ? kt + 1 : KT + 1;
    const unsigned stn = lds0 + (unsigned)((cur ^ 1) * STAGE * 2);
    const unsigned da = stn + (unsigned)(4 * wid * BM * 2);
    const unsigned db = stn + (unsigned)((NPL * A_PLANE + B_RPI * wid * BN) * 2);
    auto group = [&](int t) {
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) mfma_terms<NPL, F16>(av[i], bv[j], acc[i][j], t);
    };
    if constexpr (F16) {
      // three 16-site groups per stage, each with its own site offsets; three products per fragment set
      const int voa0 = a_voff(3 * ktn), voa1 = a_voff(3 * ktn + 1), voa2 = a_voff(3 * ktn + 2);
      const int vob0 = b_voff(3 * ktn), vob1 = b_voff(3 * ktn + 1), vob2 = b_voff(3 * ktn + 2);
      dma1(voa0, src_rs[0], da);
      dma1(voa1, src_rs[0], da + A_PLANE * 2);
      __builtin_amdgcn_sched_barrier(0);
      PHASE_STAMP(0);
      group(0);
      __builtin_amdgcn_sched_barrier(0);
      dma1(voa2, src_rs[0], da + 2 * A_PLANE * 2);
      if (wid < B_NI) dma1(vob0, dst_rs[0], db);
      __builtin_amdgcn_sched_barrier(0);
      group(1);
      __builtin_amdgcn_sched_barrier(0);
      if (wid < B_NI) {
        dma1(vob1, dst_rs[0], db + B_PLANE * 2);
        dma1(vob2, dst_rs[0], db + 2 * B_PLANE * 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      group(2);
    } else {
      const int voa = a_voff(ktn);
      const int vob = b_voff(ktn);
      dma1(voa, src_rs[0], da);
      dma1(voa, src_rs[1], da + A_PLANE * 2);
      __builtin_amdgcn_sched_barrier(0);
      PHASE_STAMP(0);
      group(0);
      __builtin_amdgcn_sched_barrier(0);
      dma1(voa, src_rs[2], da + 2 * A_PLANE * 2);
      if (wid < B_NI) dma1(vob, dst_rs[0], db);
      __builtin_amdgcn_sched_barrier(0);
      group(1);
      __builtin_amdgcn_sched_barrier(0);
      if (wid < B_NI) {
        dma1(vob, dst_rs[1], db + B_PLANE * 2);
        dma1(vob, dst_rs[2], db + 2 * B_PLANE * 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      group(2);
      group(3);
      group(4);
      group(5);
    }
    // own loads landed + own LDS reads retired, then the barrier: stage kt+1 is complete and buffer `cur` is free (the
    // scheduling fences keep the MFMAs above and the next stage's loads below it)
    __builtin_amdgcn_sched_barrier(0);
    PHASE_STAMP(1);    // fragment reads + 24 MFMAs issued
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    PHASE_STAMP(2);    // own DMA landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PHASE_STAMP(3);    // barrier
  }
  PHASE_FLUSH;

  float* o = p.nsplit > 1 ? p.partial + (size_t)split * taps * p.Ca_out * p.Cb : p.out;
  const int lh5 = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh5;
      const int mgr = m >> 3;
      if (mgr >= Mg) continue;
      const unsigned tp = p.cag_magic ? fast_div((unsigned)mgr, p.cag_magic) : (unsigned)mgr;
      const int a = (mgr - (int)tp * Cag) * 8 + (m & 7);
      if (a >= p.Ca_out) continue;
      float* orow = o + ((size_t)tp * p.Ca_out + a) * p.Cb;
#pragma unroll
      for (int j = 0; j < TN; j++) {
        const int n = n0 + wn * WN + j * 32 + l31;
        if (n < p.Cb) orow[n] = acc[i][j][r] * p.out_scale;
      }
    }
}

// ------------------------------------------------------------------------------------------------ filter gradient, ping-pong form
// The two waves a SIMD holds in the kernels above belong to different workgroups and meet at the matrix pipe by chance: both
// in their MFMA phase (one waits) or both in their load phase (the pipe idles) — 60 % utilisation in the closed-queue model of
// DESIGN.md §4.1d.  Here they belong to ONE workgroup of 8 waves (256 x 128 tile, wave tile 64 x 64 as before) and are
// scheduled against each other: group 0 multiplies stage s while group 1 reads its fragments of stage s and issues the
// LDS-DMA of stage s + 2, then they swap; one workgroup barrier per slot keeps the two groups one slot apart (the 8-phase
// GEMM template of cdna_hip_programming.md §5, reduced to two phases).  Three stage buffers (3 x 36 KB): the DMA of stage
// s + 2 overwrites stage s - 1, whose last reads retired two slots earlier.  One workgroup per CU.
//   slot t (ends with a barrier):   group 0: A(t/2) on even t, B((t-1)/2) on odd t;   group 1: the same one slot later
//   A(x): fragment reads of stage x -> registers; DMA of stage x+2; wait: reads retired, DMA of stage x+1 landed (vmcnt = own
//         loads of stage x+2 still in flight)            B(x): 24 MFMAs at raised priority
// BN = 128: wave tile 64 x 64 (4 x 2 waves); BN = 256: wave tile 128 x 64 (2 x 4 waves: 48 MFMAs against 36 fragment reads and
// 6 loads per stage instead of 24 : 24 : 4.5)
// F16: as in igemm_pl_wgrad_dma_kernel — one fp16 plane, a stage = three consecutive 16-site groups, three products per fragment set.
template <int BN, bool F16 = false>
__global__ __launch_bounds__(512, 1) void igemm_pl_wgrad_pp_kernel(const PlWgradParams p, int grp_mode) {
  constexpr int BM = 256, WM = BN == 256 ? 128 : 64, WN = 64, NPL = 3, NT = F16 ? 3 : 6, KS = 16;
  constexpr int GPS = F16 ? 3 : 1;                                 // 16-site groups per stage
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_HALF = KS * 128;                                 // one 128-row half of an operand plane (tr_swz<128> image)
  constexpr int A_PLANE = 2 * A_HALF, B_PLANE = KS * BN;           // elements
  constexpr int STAGE = NPL * (A_PLANE + B_PLANE);
  constexpr int NSTAGE = 3;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // 0 .. 7
  // ping-pong group and the wave's index inside it; which waves share a SIMD is the hardware's choice: grp_mode picks the
  // pairing (0: waves w and w + 4, 1: waves 2k and 2k + 1)
  const int grp = grp_mode == 0 ? (wid >> 2) : (wid & 1);
  const int idx = grp_mode == 0 ? (wid & 3) : (wid >> 1);
  // wave tile: BN = 128: M quarter idx (64 rows), N half grp;  BN = 256: M half grp (128 rows), N quarter idx (64 columns)
  const int wmq = BN == 256 ? 2 * grp : idx, wn = BN == 256 ? idx : grp;
  const int taps = p.KH * p.KW;
  const int Cag = p.Ca >> 3;
  const int Mg = taps * Cag;
  int v = xcd_remap(blockIdx.x, gridDim.x, p.xcd);
  const int mtile = v % p.mt; v /= p.mt;
  const int ntile = v % p.nt;
  const int split = v / p.nt;
  const int m0 = mtile * BM, n0 = ntile * BN;
  const int S = p.B * p.Hg * p.Wg;
  const int KT = ((S + KS - 1) / KS + GPS - 1) / GPS;              // stages
  const int kt_per = F16 ? (KT + p.nsplit - 1) / p.nsplit : (((KT + 1) / 2 + p.nsplit - 1) / p.nsplit) * 2;
  const int kt0 = split * kt_per, kt1 = min(KT, kt0 + kt_per);
  const int n = max(kt1 - kt0, 0);

  u32x4 src_rs[NPL], dst_rs[NPL];
#pragma unroll
  for (int pl = 0; pl < NPL; pl++) {
    src_rs[pl] = raw_rsrc(p.src + (F16 ? 0 : pl * p.src_ps), (((size_t)p.B * p.Hs * p.Ws - 1) * (size_t)p.lds + (size_t)(p.gpx ? p.lds : p.Ca)) * 2);
    dst_rs[pl] = raw_rsrc(p.dst + (F16 ? 0 : pl * p.dst_ps), (((size_t)S - 1) * (size_t)p.ldd + (size_t)((p.Cb + 7) & ~7)) * 2);
  }
  // DMA roles (by wave id, independent of the groups): wave w loads site rows 4 (w & 3) .. + 3 of M half w >> 2 of the
  // gathered operand; waves 0 .. 3 also load the dense operand's site rows 4 w .. + 3
  const int dh = wid >> 2, dq = wid & 3;
  const int a_k = 4 * dq + (lane >> 4);
  const int a_slot = lane & 15;
  const int a_g = ((((a_slot >> 1) ^ ((a_k & 3) << 1)) << 1) | (a_slot & 1));
  const int mg = (m0 >> 3) + 16 * dh + a_g;
  const bool m_ok = mg < Mg;
  const unsigned tap = p.cag_magic ? fast_div((unsigned)mg, p.cag_magic) : (unsigned)mg;
  const int ag = mg - (int)tap * Cag;
  const int ky = (int)tap / p.KW, kx = (int)tap - ky * p.KW;
  const int dy = p.dy0 + ky, dx = p.dx0 + kx + p.gpx * ag;
  const int lds2 = p.lds * 2;
  const int a_lane_off = (dy * p.Ws + (p.dx0 + kx)) * lds2 + ag * 16;
  const int b_k = 4 * dq + (lane >> 4);
  const int b_slot = lane & 15;
  const int b_g = ((((b_slot >> 1) ^ ((b_k & 3) << 1)) << 1) | (b_slot & 1));
  const int nbq = (n0 >> 3) + (BN == 256 ? 16 * dh : 0) + b_g;
  const bool b_ok = nbq * 8 < p.Cb;
  const int b_lane_off = nbq * 16;
  const int ldd2 = p.ldd * 2;
  const unsigned magW = (unsigned)((0x100000000ull + p.Wg - 1) / p.Wg), magH = (unsigned)((0x100000000ull + p.Hg - 1) / p.Hg);
  const unsigned lds0 = lds_addr(smem16);
  auto a_voff = [&](int kt) {
    const unsigned sidx = (unsigned)(kt * KS + a_k);
    const unsigned q = fast_div(sidx, magW);
    const int xg = (int)(sidx - q * (unsigned)p.Wg);
    const unsigned bb = fast_div(q, magH);
    const int yg = (int)(q - bb * (unsigned)p.Hg);
    const int yb = yg * p.sm, xb = xg * p.sm;
    const bool ok = m_ok && (int)bb < p.B && (unsigned)(yb + dy) < (unsigned)p.Hs && (unsigned)(xb + dx) < (unsigned)p.Ws;
    return ok ? (((int)bb * p.Hs + yb) * p.Ws + xb) * lds2 + a_lane_off : OOB_MARK;
  };
  auto b_voff = [&](int kt) {
    const int sidx = kt * KS + b_k;
    return (b_ok && sidx < S) ? sidx * ldd2 + b_lane_off : OOB_MARK;
  };
  // the three loads of one operand of stage kt: three planes at the same site offsets, or (F16) three 16-site groups of the one plane
  auto load_a = [&](int kt, unsigned da) {
    if constexpr (F16) {
#pragma unroll
      for (int g = 0; g < 3; g++) dma1(a_voff(3 * kt + g), src_rs[0], da + g * A_PLANE * 2);
    } else {
      dma3(a_voff(kt), src_rs[0], src_rs[1], src_rs[2], da, da + A_PLANE * 2, da + 2 * A_PLANE * 2);
    }
  };
  auto load_b = [&](int kt, unsigned db) {
    if constexpr (F16) {
#pragma unroll
      for (int g = 0; g < 3; g++) dma1(b_voff(3 * kt + g), dst_rs[0], db + g * B_PLANE * 2);
    } else {
      dma3(b_voff(kt), dst_rs[0], dst_rs[1], dst_rs[2], db, db + B_PLANE * 2, db + 2 * B_PLANE * 2);
    }
  };
  auto issue = [&](int x) {                      // stage x of this block (kt0 + x) -> buffer x % 3; past the end: zeros
    const int kt = x < n ? kt0 + x : KT + 1;
    const unsigned st = lds0 + (unsigned)((x % NSTAGE) * STAGE * 2);
    const unsigned da = st + (unsigned)((dh * A_HALF + 4 * dq * 128) * 2);
    load_a(kt, da);
    if (BN == 256 || wid < 4) {                  // (BN = 256: the dense operand is two 128-column halves like the gathered one)
      const unsigned db = st + (unsigned)((NPL * A_PLANE + (BN == 256 ? dh * A_HALF : 0) + 4 * dq * 128) * 2);
      load_b(kt, db);
    }
  };
  auto wait_older = [&]() {                      // everything but this wave's most recent issue() has landed
    if (BN == 256 || wid < 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int i16 = lane & 15, grp16 = lane >> 4, lh = grp16 >> 1;
  const int krow = 8 * lh + (i16 >> 2);
  int a_rd[TM], b_rd[TN];
#pragma unroll
  for (int i = 0; i < TM; i++) {
    const int c = (BN == 256 ? 0 : (wmq & 1) * 64) + i * 32 + 16 * (grp16 & 1) + 4 * (i16 & 3);      // row inside the 128-row half
    a_rd[i] = (wmq >> 1) * A_HALF + tr_swz<128>(krow, c >> 3) + (c & 7);
  }
#pragma unroll
  for (int j = 0; j < TN; j++) {
    const int c = (BN == 256 ? (wn & 1) : wn) * WN + j * 32 + 16 * (grp16 & 1) + 4 * (i16 & 3);     // column inside the 128-column half
    b_rd[j] = NPL * A_PLANE + (BN == 256 ? (wn >> 1) * A_HALF : 0) + tr_swz<128>(krow, c >> 3) + (c & 7);
  }

  // fragments of ONE 64-row half of the wave tile at a time (BN = 256: the 128 x 64 wave tile is two such halves, which share
  // the dense operand's fragments): a stage is HALVES x (read slot, multiply slot)
  constexpr int HALVES = TM / 2;
  s16x8 av[2][NPL], bv[TN][NPL];
  auto read_frags = [&](int x, int half) {
    const unsigned short* st = smem16 + (x % NSTAGE) * STAGE;
    auto read_a = [&](int pl) {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const unsigned short* b0 = st + pl * A_PLANE + a_rd[2 * half + i];
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(b0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(b0 + 4 * 128));
        av[i][pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    };
    auto read_b = [&](int pl) {
#pragma unroll
      for (int j = 0; j < TN; j++) {
        const unsigned short* b0 = st + pl * B_PLANE + b_rd[j];
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(b0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(b0 + 4 * 128));
        bv[j][pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    };
    if (half == 0) { read_a(2); read_b(0); read_a(0); read_b(2); read_a(1); read_b(1); }
    else { read_a(2); read_a(0); read_a(1); }
    __builtin_amdgcn_sched_barrier(0);
  };
  // the loads of stage x + 2, spread over the read slots of stage x: the gathered operand in the first, the dense one in the
  // last (one slot: both)
  auto issue_part = [&](int x, int part) {
    const int kt = x < n ? kt0 + x : KT + 1;
    const unsigned st = lds0 + (unsigned)((x % NSTAGE) * STAGE * 2);
    if (part == 0 || HALVES == 1) {
      const unsigned da = st + (unsigned)((dh * A_HALF + 4 * dq * 128) * 2);
      load_a(kt, da);
    }
    if ((part == HALVES - 1) && (BN == 256 || wid < 4)) {
      const unsigned db = st + (unsigned)((NPL * A_PLANE + (BN == 256 ? dh * A_HALF : 0) + 4 * dq * 128) * 2);
      load_b(kt, db);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto multiply = [&](int half) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
          if (half == 0) mfma_terms<NPL, F16>(av[i], bv[j], acc[i][j], t);
          else mfma_terms<NPL, F16>(av[i], bv[j], acc[(TM > 2 ? 2 : 0) + i][j], t);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  if (n > 0) {
    issue(0);
    issue(1);
    wait_older();                                // stage 0 landed (stage 1 may be in flight)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // group 1 runs the same slot sequence one barrier later (and group 0 takes the last barrier alone)
    auto slot_end = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    if (grp == 1) slot_end();
    PHASE_DECL;
#pragma unroll 1
    for (int x = 0; x < n; x++) {
      PHASE_STAMP(6);
      // (the fragment reads of the FIRST read slot are not waited for before its barrier: they return while the other group
      // starts its read slot; the MFMAs wait for them by register dependence)
      read_frags(x, 0);
      issue_part(x + 2, 0);
      if (HALVES == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (last reads of the stage: see below)
        wait_older();
      }
      PHASE_STAMP(0);    // read slot 0: fragment reads + loads issued
      slot_end();
      PHASE_STAMP(1);    // its barrier
      multiply(0);
      PHASE_STAMP(2);    // 24 MFMAs issued
      slot_end();
      PHASE_STAMP(3);    // its barrier
      if (HALVES == 2) {
        read_frags(x, 1);
        issue_part(x + 2, 1);
        // the LAST reads of stage x retire before the barrier: the other group restages this buffer (stage x + 3) in its very
        // next slot, and an LDS-DMA write is ordered against a ds_read only through lgkmcnt + barrier (cdna_hip_programming.md §5)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_older();
        slot_end();
        PHASE_STAMP(4);  // read slot 1 + barrier
        multiply(1);
        slot_end();
      }
    }
    PHASE_FLUSH;
    if (grp == 0) slot_end();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero-fill loads past the last stage
  }

  float* o = p.nsplit > 1 ? p.partial + (size_t)split * taps * p.Ca_out * p.Cb : p.out;
  const int lh5 = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = m0 + wmq * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh5;
      const int mgr = m >> 3;
      if (mgr >= Mg) continue;
      const unsigned tp = p.cag_magic ? fast_div((unsigned)mgr, p.cag_magic) : (unsigned)mgr;
      const int a = (mgr - (int)tp * Cag) * 8 + (m & 7);
      if (a >= p.Ca_out) continue;
      float* orow = o + ((size_t)tp * p.Ca_out + a) * p.Cb;
#pragma unroll
      for (int j = 0; j < TN; j++) {
        const int nn = n0 + wn * WN + j * 32 + l31;
        if (nn < p.Cb) orow[nn] = acc[i][j][r] * p.out_scale;
      }
    }
}

// ------------------------------------------------------------------------------------------------ plane producers
// fp32 [npix][ldx] (C channels used) -> planes [npix][ldp] with channels C .. Cp-1 zero-filled (Cp a multiple of 4, >= C)
__global__ void planes_from_f32_kernel(const float* __restrict__ x, int ldx, long npix, int C, int Cp, PlaneOut o) {
  const int nq = Cp >> 2;
  const long total = npix * nq;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long px = e / nq;
    const int n = (int)(e - px * nq) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* s = x + px * ldx + n;
    if (n < C) v.x = s[0];
    if (n + 1 < C) v.y = s[1];
    if (n + 2 < C) v.z = s[2];
    if (n + 3 < C) v.w = s[3];
    store_planes4(o, (size_t)px, n, v);
  }
}

// Weight planes of one tensor W[taps][R][Cc] (fp32):
//   direct     D[p][tap][R][Cc8]   (K = Cc contiguous; Cc8 = round-up-8(Cc), zero padded)
//   transposed T[p][tap][Cc][R8]   (K = R contiguous)
// One launch covers a table of tensors (all layers of a network): block -> (tensor, tap, 64 x 64 tile).
struct WPlaneDesc {
  const float* w;
  unsigned short* direct;
  unsigned short* transposed;
  int taps, R, Cc;
  int tiles_r, tiles_c;
  int block0;
};
constexpr int MAX_WDESC = 64;
struct WPlaneBatch {
  WPlaneDesc d[MAX_WDESC];
  int n, n_planes;
};

// eight consecutive plane elements (16 bytes per plane) from eight fp32 values
__device__ __forceinline__ void put_planes8(unsigned short* base, long ps, int n_planes, size_t idx, const float (&v)[8]) {
  if (n_planes == 1) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = (unsigned)to_f16_bits(v[2 * i]) | ((unsigned)to_f16_bits(v[2 * i + 1]) << 16);
    __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(base + idx));
    return;
  }
  u32x4 h, m, l;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float a = v[2 * i], b = v[2 * i + 1];
    const unsigned hh = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(hh << 16), rb = b - __uint_as_float(hh & 0xffff0000u);
    const unsigned mm = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(mm << 16), sb = rb - __uint_as_float(mm & 0xffff0000u);
    h[i] = hh; m[i] = mm; l[i] = cvt_pk_bf16(sa, sb);
  }
  // (streamed once per step, read by the conv kernels much later: non-temporal)
  __builtin_nontemporal_store(h, reinterpret_cast<u32x4*>(base + idx));
  __builtin_nontemporal_store(m, reinterpret_cast<u32x4*>(base + idx + ps));
  __builtin_nontemporal_store(l, reinterpret_cast<u32x4*>(base + idx + 2 * ps));
}

__global__ __launch_bounds__(256) void weight_planes_kernel(const WPlaneBatch b) {
  __shared__ float tile[64][65];
  int di = 0;
  while (di + 1 < b.n && (int)blockIdx.x >= b.d[di + 1].block0) di++;
  const WPlaneDesc d = b.d[di];
  int lb = blockIdx.x - d.block0;
  const int tc = lb % d.tiles_c; lb /= d.tiles_c;
  const int tr = lb % d.tiles_r; lb /= d.tiles_r;
  const int tap = lb;
  const int r0 = tr * 64, c0 = tc * 64;
  const int Cc8 = (d.Cc + 7) & ~7, R8 = (d.R + 7) & ~7;
  const float* w = d.w + (size_t)tap * d.R * d.Cc;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    tile[r][c] = (r0 + r < d.R && c0 + c < d.Cc) ? w[(size_t)(r0 + r) * d.Cc + c0 + c] : 0.f;
  }
  __syncthreads();
  // one 16-byte store per plane and thread: 8 consecutive K elements (zero beyond the tensor: the tile is zero-filled)
  if (d.direct) {
    const long ps = (long)d.taps * d.R * Cc8;
    for (int e = threadIdx.x; e < 64 * 8; e += 256) {
      const int r = e >> 3, c8 = (e & 7) * 8;
      if (r0 + r >= d.R || c0 + c8 >= Cc8) continue;
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = tile[r][c8 + i];
      put_planes8(d.direct, ps, b.n_planes, ((size_t)tap * d.R + r0 + r) * Cc8 + c0 + c8, v);
    }
  }
  if (d.transposed) {
    const long ps = (long)d.taps * d.Cc * R8;
    for (int e = threadIdx.x; e < 64 * 8; e += 256) {
      const int r8 = (e & 7) * 8, c = e >> 3;      // 8 lanes write 128 contiguous bytes of one output row
      if (c0 + c >= d.Cc || r0 + r8 >= R8) continue;
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = tile[r8 + i][c];
      put_planes8(d.transposed, ps, b.n_planes, ((size_t)tap * d.Cc + c0 + c) * R8 + r0 + r8, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------ fused L2 + Adam + weight planes
// Round 6: the optimizer update and the re-split of the updated weights in ONE pass.  adam_kernel
// (train_misc.hip) streams P, G, M, V in and P, M, V out; weight_planes_kernel then reads the 157 MB of P it just wrote to
// write 471 MB of planes.  Here a block owns a 64 x 64 tile of one tap of one tensor (weight_planes_kernel's own mapping),
// updates it — the same expressions in the same order as adam_kernel: bit-identical parameters — keeps the new values in
// its LDS tile and writes both plane copies from there: the second read of P and a launch are gone.  The table also takes
// tensors WITHOUT planes (direct == transposed == NULL: the Cout = 2 layers) and the bias block (reg == 0: one pseudo-tensor of
// taps = R = 1), so one launch covers a whole flat range of the parameter vector.
struct AdamPlaneArgs {
  float* P; const float* G; float* M; float* V;      // the four flat buffers (identical layout: a tensor's offset is w - P)
  float gscale, l2, lr_t, b1, b2, eps;
  float* loss_acc;                                   // non-NULL: += l2 * 0.5 * sum(p^2) over the regularised tensors (pre-update)
};

__global__ __launch_bounds__(256) void adam_planes_kernel(const WPlaneBatch b, const AdamPlaneArgs a, const unsigned long long reg_mask) {
  __shared__ float tile[64][65];
  __shared__ float red[4];
  int di = 0;
  while (di + 1 < b.n && (int)blockIdx.x >= b.d[di + 1].block0) di++;
  const WPlaneDesc d = b.d[di];
  const bool reg = (reg_mask >> di) & 1ull;
  int lb = blockIdx.x - d.block0;
  const int tc = lb % d.tiles_c; lb /= d.tiles_c;
  const int tr = lb % d.tiles_r; lb /= d.tiles_r;
  const int tap = lb;
  const int r0 = tr * 64, c0 = tc * 64;
  const int Cc8 = (d.Cc + 7) & ~7, R8 = (d.R + 7) & ~7;
  const size_t base = (size_t)(d.w - a.P) + (size_t)tap * d.R * d.Cc;
  float sq = 0.f;
  const bool vec = (d.Cc & 3) == 0 && (base & 3) == 0;       // rows of whole, 16-byte aligned quads (every tensor of the engine)
  if (vec) {
    for (int e = threadIdx.x; e < 64 * 16; e += 256) {
      const int r = e >> 4, c = (e & 15) * 4;
      float pv[4] = {0.f, 0.f, 0.f, 0.f};
      if (r0 + r < d.R && c0 + c < d.Cc) {
        const size_t i = base + (size_t)(r0 + r) * d.Cc + c0 + c;
        const f32x4_nt pq = *reinterpret_cast<const f32x4_nt*>(a.P + i);
        const f32x4_nt gq = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(a.G + i));
        f32x4_nt mq = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(a.M + i));
        f32x4_nt vq = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(a.V + i));
        f32x4_nt po;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          float gr = gq[j] * a.gscale;
          if (reg) {
            gr += a.l2 * pq[j];
            sq += pq[j] * pq[j];
          }
          mq[j] = a.b1 * mq[j] + (1.f - a.b1) * gr;
          vq[j] = a.b2 * vq[j] + (1.f - a.b2) * (gr * gr);
          po[j] = pq[j] - a.lr_t * mq[j] / (sqrtf(vq[j]) + a.eps);
          pv[j] = po[j];
        }
        *reinterpret_cast<f32x4_nt*>(a.P + i) = po;
        __builtin_nontemporal_store(mq, reinterpret_cast<f32x4_nt*>(a.M + i));
        __builtin_nontemporal_store(vq, reinterpret_cast<f32x4_nt*>(a.V + i));
      }
#pragma unroll
      for (int j = 0; j < 4; j++) tile[r][c + j] = pv[j];
    }
  } else {
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
      const int r = e >> 6, c = e & 63;
      float pn = 0.f;
      if (r0 + r < d.R && c0 + c < d.Cc) {
        const size_t i = base + (size_t)(r0 + r) * d.Cc + c0 + c;
        const float pq = a.P[i];
        float gr = a.G[i] * a.gscale;
        if (reg) {
          gr += a.l2 * pq;
          sq += pq * pq;
        }
        const float mn = a.b1 * a.M[i] + (1.f - a.b1) * gr;
        const float vn = a.b2 * a.V[i] + (1.f - a.b2) * (gr * gr);
        a.M[i] = mn;
        a.V[i] = vn;
        pn = pq - a.lr_t * mn / (sqrtf(vn) + a.eps);
        a.P[i] = pn;
      }
      tile[r][c] = pn;
    }
  }
  __syncthreads();
  if (d.direct) {
    const long ps = (long)d.taps * d.R * Cc8;
    for (int e = threadIdx.x; e < 64 * 8; e += 256) {
      const int r = e >> 3, c8 = (e & 7) * 8;
      if (r0 + r >= d.R || c0 + c8 >= Cc8) continue;
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = tile[r][c8 + i];
      put_planes8(d.direct, ps, b.n_planes, ((size_t)tap * d.R + r0 + r) * Cc8 + c0 + c8, v);
    }
  }
  if (d.transposed) {
    const long ps = (long)d.taps * d.Cc * R8;
    for (int e = threadIdx.x; e < 64 * 8; e += 256) {
      const int r8 = (e & 7) * 8, c = e >> 3;
      if (c0 + c >= d.Cc || r0 + r8 >= R8) continue;
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = tile[r8 + i][c];
      put_planes8(d.transposed, ps, b.n_planes, ((size_t)tap * d.Cc + c0 + c) * R8 + r0 + r8, v);
    }
  }
  if (a.loss_acc && reg) {          // (block-uniform condition: block_sum's barriers are safe)
    const float t = block_sum(sq, red);
    if (threadIdx.x == 0) atomicAdd(a.loss_acc, t * 0.5f * a.l2);
  }
}

// ------------------------------------------------------------------------------------------------ host side
struct PlPlan {
  int cfg;  // 0: 128x128, 1: 128x64, 2: 64x64
  int nsplit;
  bool pp = false;   // ping-pong form (pairs of 128 x 128 tiles)
};

inline int pl_smem_gather(int bm, int bn, int npl) { return pl_gather_main_bytes(bm, bn, bm == bn ? bm / 2 : 32, npl) + bm * 4 + 16; }   // + the pixel table + the split-K flag

// blocks per CU by LDS (160 KB) and registers (<= 168: 3 waves per SIMD)
inline int pl_blocks_per_cu(int bm, int bn, int npl) {
  const int byl = (160 * 1024) / pl_smem_gather(bm, bn, npl);
  const int byr = bm == 128 && bn == 128 ? 3 : bm == 128 ? 4 : 6;
  return byl < byr ? byl : byr;
}

// ping-pong gather form (igemm_pl_gather_pp_kernel): bf16 x 3, 128 x 128 tiles in pairs
inline bool pl_gather_pp_ok(const GatherGeom& p, int npl, int cfg) {
  return unflow::options().gather_pp > 0 && npl == 3 && cfg == 0;
}

inline PlPlan plan_pl_gather(const GatherGeom& p, int npl) {
  PlPlan pl;
  const long M = (long)p.B * p.Hg * p.Wg;
  if (p.N <= 32) pl.cfg = 2;
  else if (p.N <= 64) pl.cfg = 1;
  else pl.cfg = 0;
  int maxtaps = 0;
  for (int c = 0; c < p.ncls; c++) maxtaps = max(maxtaps, p.cls[c].nty * p.cls[c].ntx);
  const int KT = (maxtaps * (p.Cs >> 3) + 3) >> 2;
  const int min_kt = max(1, unflow::options().gather_min_kt), max_split = max(1, unflow::options().gather_max_split);
  const int max_by_k = min(max_split, KT / min_kt > 0 ? KT / min_kt : 1);
  if (pl.cfg == 0) {
    const long b128 = ((M + 127) / 128) * ((p.N + 127) / 128) * p.ncls;
    if (b128 * max_by_k < 384) pl.cfg = 2;  // cannot fill half the chip with 128x128 tiles: smaller tiles
  }
  // a K loop of a few tiles (FlowNetC's first layer: 7) is all prologue and epilogue: more, smaller workgroups in flight
  // (conv1 forward 98 -> 91 us)
  if (pl.cfg == 1 && KT <= 8) pl.cfg = 2;
  const int bm = pl.cfg == 2 ? 64 : 128, bn = pl.cfg == 0 ? 128 : 64;
  const int slots = 256 * pl_blocks_per_cu(bm, bn, npl);
  const long blocks = ((M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * p.ncls;
  pl.nsplit = fill_one_round(blocks, slots, max_by_k);
  if (pl_gather_pp_ok(p, npl, pl.cfg)) {
    // one 8-wave workgroup per CU: 256 slots.  Taken where the launch fills them (measured per layer, profiles/
    // r03_gather_pingpong_per_layer.txt: the even-class deconv layers and the deep stride-1 layers gain 5-16 %; conv4's forward at
    // 75 % fill and the uneven parity classes of the stride-2 data gradients lose) — option gather_pp = 2 forces it (tests)
    const long pairs = (((M + 127) / 128 + 1) / 2) * ((p.N + 127) / 128) * p.ncls;
    const int ns = (int)max(1L, min((long)max_by_k, 256 / max(pairs, 1L)));      // ONE round of workgroups
    const long b = pairs * ns;
    const double fill = (double)b / (double)(((b + 255) / 256) * 256);
    bool even = true;
    for (int c = 1; c < p.ncls; c++) even = even && p.cls[c].nty * p.cls[c].ntx == p.cls[0].nty * p.cls[0].ntx;
    if (unflow::options().gather_pp >= 2 || (fill >= 0.01 * unflow::options().gather_pp_fill && even)) { pl.pp = true; pl.nsplit = ns; }
  }
  return pl;
}

inline size_t pl_gather_slab_bytes(const GatherGeom& p, int nsplit) {
  return (((size_t)nsplit * p.B * p.Hd * p.Wd * p.N * sizeof(float)) + 255) & ~(size_t)255;
}
inline size_t pl_gather_partial_bytes(const GatherGeom& p, int nsplit) {
  return nsplit > 1 ? pl_gather_slab_bytes(p, nsplit) : 0;
}

// (pl_grid, pl_grid_classes: planes_shared.h)

// 2-D site tiles of the plain gather kernels when they divide the site grid exactly (TW = min(32, Wg) sites wide)
template <int BM>
inline void pl_gather_tiles2d(PlGatherParams& q) {
  const int tiles2d = unflow::options().gather_tile2d;
  int twl = 0;
  while ((2 << twl) <= q.Wg && (2 << twl) <= 32) twl++;
  const int tw = 1 << twl, th = BM >> twl;
  q.tw_log = 0;
  if (tiles2d && twl >= 3 && th >= 2 && q.Wg % tw == 0 && q.Hg % th == 0 && th * tw == BM) {
    q.tw_log = twl;
    q.tiles_x = q.Wg / tw;
    q.tiles_y = q.Hg / th;
  }
}

template <int BM, int BN, int WM, int WN, int NPL, bool F16>
int launch_pl_gather(const PlGatherParams& p, hipStream_t st) {
  const int M = p.B * p.Hg * p.Wg;
  const int smem = pl_smem_gather(BM, BN, NPL);
  static DynLdsBook attr_book{};
  (void)ensure_dyn_lds(reinterpret_cast<const void*>(&igemm_pl_gather_kernel<BM, BN, WM, WN, NPL, F16>), smem, attr_book);
  PlGatherParams q = p;
  q.mt = cdiv(M, BM); q.nt = cdiv(p.N, BN);
  pl_gather_tiles2d<BM>(q);
  const int grid = pl_grid(q);
  igemm_pl_gather_kernel<BM, BN, WM, WN, NPL, F16><<<grid, 256, smem, st>>>(q);
  return launch_status();
}

// ping-pong form: pairs of 128-site tiles per workgroup
int launch_pl_gather_pp(const PlGatherParams& p, hipStream_t st) {
  const int M = p.B * p.Hg * p.Wg;
  constexpr int smem = 6 * 3 * 128 * 32 * 2 + 2 * 128 * 4;
  static DynLdsBook attr_book{};
  (void)ensure_dyn_lds(reinterpret_cast<const void*>(&igemm_pl_gather_pp_kernel<3, false>), smem, attr_book);
  PlGatherParams q = p;
  q.mt = cdiv(M, 128); q.nt = cdiv(p.N, 128);
  pl_gather_tiles2d<128>(q);
  if (q.tw_log) q.mt = q.B * q.tiles_y * q.tiles_x;
  q.mt = cdiv(q.mt, 2);                          // M tile PAIRS in the work order
  const int grid = pl_grid(q);
  igemm_pl_gather_pp_kernel<3, false><<<grid, 512, smem, st>>>(q);
  return launch_status();
}

template <int NPL, bool F16>
int run_pl_gather_mode(PlGatherParams& p, int cfg, hipStream_t st) {
  switch (cfg) {
    case 0: return launch_pl_gather<128, 128, 64, 64, NPL, F16>(p, st);
    case 1: return launch_pl_gather<128, 64, 64, 32, NPL, F16>(p, st);
    default: return launch_pl_gather<64, 64, 32, 32, NPL, F16>(p, st);
  }
}

// ---- halo kernel: eligibility, plan, launch
inline bool pl_halo_ok(const GatherGeom& p) {
  const bool off = !unflow::options().halo;
  if (off || p.sm != 1 || (p.dstep != 1 && p.dstep != -1) || p.Hg < 2 * TH || p.Wg < TW || p.N <= 32) return false;
  if (p.acc && (p.dstep != 1 || !unflow::options().halo_s2)) return false;
  for (int c = 0; c < p.ncls; c++)
    if (p.cls[c].nty < 1 || p.cls[c].ntx < 1) return false;
  return pl_halo_pixels(p) <= 256;
}
// The accumulating-class form of a source-stride-2 layer (build_conv_fwd_s2acc / build_deconv_dgrad_acc) against the plain
// gather kernel: the halo kernel runs two 128 x 128 blocks per CU (the gather kernel three) and splits K only over whole
// channel chunks, so it pays where the layer has tiles to spare — measured per layer on MI355X (profiles/r03_halo_s2_per_layer.txt):
// conv2 fwd 243 -> 226 us (768 tiles), conv3 fwd 254 -> 244 (384), deconv2 dgrad 138 -> 131 (768), but conv4 fwd 98 -> 108
// (192 tiles) and deconv3 dgrad 134 -> 141 (336).
inline bool pl_halo_acc_pays(const GatherGeom& a) {
  if (!a.acc || !pl_halo_ok(a)) return false;
  const int bn = a.N <= 64 ? 64 : 128;
  const long tiles = (long)a.B * cdiv(a.Hg, TH) * cdiv(a.Wg, TW) * cdiv(a.N, bn);
  return tiles >= 384 || unflow::options().halo_s2 >= 2;      // (halo_s2 = 2: wherever the kernel applies — tests)
}

// fp16 launches of the halo kernel may run with two chunk planes: K tiles of 64 channels (mfma_terms, planes_shared.h)
constexpr int F16_KCH = 2;
inline int f16_kch(const GatherGeom& p, int npl) {
  if (npl != 1) return 1;
  const int o = unflow::options().f16_k64;
  if (o >= 2) return F16_KCH;
  if (o <= 0) return 1;
  // Where it pays (per-layer A/B at B = 8, profiles/r06_f16_per_layer_ab.txt): SHORT items whose K loop is mostly loop overhead at 8
  // MFMAs per tile — the conv_transpose forwards (four classes of exactly 2 x 2 taps: -9 .. -19 us) and the 3 x 3 stride-2 forwards
  // (accumulating classes, 9 taps in all: conv4 -19 us).  Long items lose (the halo of a 64-channel chunk arrives in one burst:
  // conv3_1 +22 us), and so do the uneven parity classes of the stride-2 data gradients (+11 us).
  int sum = 0;
  bool four = true;
  for (int c = 0; c < p.ncls; c++) {
    sum += p.cls[c].nty * p.cls[c].ntx;
    four = four && p.cls[c].nty * p.cls[c].ntx == 4;
  }
  return (p.acc ? sum <= 9 : (p.ncls == 4 && four)) ? F16_KCH : 1;
}
// fp16 layers may run on the 256-site x 128 tile of conv_halo_tall.hip (option f16_tall)
inline bool f16_tall(const GatherGeom& p, int npl) {
  const int o = unflow::options().f16_tall;
  if (npl != 1 || o <= 0 || f16_kch(p, npl) != 1 || p.N <= 64 || p.Hg < TALL_TH || pl_halo_pixels(p, TALL_TH) > 384) return false;
  if (o >= 2) return true;
  const long tiles = (long)p.B * cdiv(p.Hg, TALL_TH) * cdiv(p.Wg, TW) * cdiv(p.N, 128) * pl_grid_classes(p);
  return tiles >= 512;
}
inline int plan_pl_halo(const GatherGeom& p, int npl, int* bn_out, bool* tall_out = nullptr) {
  const int bn = p.N <= 64 ? 64 : 128;
  const bool tall = f16_tall(p, npl);
  if (tall_out) *tall_out = tall;
  const int th = tall ? TALL_TH : TH;
  *bn_out = bn;
  const int kch = f16_kch(p, npl);
  const int nbuf = (npl == 1 && kch == 1 && unflow::options().f16_db > 0) ? 2 : 1;
  const int smem = pl_halo_main_bytes(bn, bn >= 128 ? 64 : 32, npl == 1 ? kch : npl, pl_halo_pixels(p, th), nbuf) + th * 32 * 4 + 16;
  const int per_cu = min((160 * 1024) / smem, bn >= 128 ? 2 : 3);
  const long blocks = (long)p.B * cdiv(p.Hg, th) * cdiv(p.Wg, TW) * cdiv(p.N, bn) * pl_grid_classes(p);
  const int nchunk = ((((p.Cs >> 3) + 3) >> 2) + kch - 1) / kch;
  int maxtaps = 0, sumtaps = 0;
  for (int c = 0; c < p.ncls; c++) {
    maxtaps = max(maxtaps, p.cls[c].nty * p.cls[c].ntx);
    sumtaps += p.cls[c].nty * p.cls[c].ntx;
  }
  if (p.acc) maxtaps = sumtaps;                                       // a block walks every class
  const int max_by_k = max(1, min(min(16, max(1, unflow::options().halo_max_split)), nchunk * maxtaps / 16));      // >= 16 K tiles per split, whole chunks
  return min(nchunk, fill_one_round(blocks, 256 * per_cu, max_by_k));
}

template <int BN, int WN, int NPL, bool F16, int WM = 64, bool DB = false>
int launch_pl_halo(const PlGatherParams& p, hipStream_t st) {
  const int hp = pl_halo_pixels(p);
  const int smem = pl_halo_main_bytes(BN, WN, NPL, hp, DB ? 2 : 1) + 128 * 4 + 16;
  static DynLdsBook book{};     // grow-only per device (the halo size depends on the layer)
  (void)ensure_dyn_lds(reinterpret_cast<const void*>(&igemm_pl_halo_kernel<BN, WM, WN, NPL, F16, DB>), smem, book);
  PlGatherParams q = p;
  if (F16 && NPL > 1) q.src_ps = q.w_ps = 32;          // chunk planes: plane pl = channels 32 pl .. of the fp16 plane
  q.mt = p.B * p.tiles_y * p.tiles_x; q.nt = cdiv(p.N, BN);
  const int grid = pl_grid(q);
  igemm_pl_halo_kernel<BN, WM, WN, NPL, F16, DB><<<grid, 256, smem, st>>>(q, hp);
  return launch_status();
}

int run_pl_gather(PlGatherParams& p, int npl, void* ws, size_t ws_bytes, hipStream_t st) {
  const unflow::Options& opt = unflow::options();
  if (p.src_inv == 0.f) p.src_inv = 1.f;
  p.xcd = opt.xcd_swizzle;
  // order 2 (M groups per XCD, M tile fastest inside) measured best or tied on every layer of FlowNetC 384x512 B=4 against
  // 0 and 1 (profiles/r02_xcd_order_per_layer.txt)
  p.order = 2;
  const bool halo = pl_halo_ok(p);
  int halo_bn = 128;
  bool tall = false;
  PlPlan pl = plan_pl_gather(p, npl);
  if (halo) {
    pl.nsplit = plan_pl_halo(p, npl, &halo_bn, &tall);
    p.tiles_y = cdiv(p.Hg, TH);
    p.tiles_x = cdiv(p.Wg, TW);
  }
  p.nsplit = pl.nsplit;
  p.partial = nullptr;
  if (p.nsplit > 1) {
    if (!ws || ws_bytes < pl_gather_partial_bytes(p, p.nsplit)) p.nsplit = 1;  // no scratch: un-split (same result up to fp32 order)
    else p.partial = reinterpret_cast<float*>(ws);
  }
  {
    const uintptr_t al = reinterpret_cast<uintptr_t>(p.dst) | reinterpret_cast<uintptr_t>(p.partial) |
                         reinterpret_cast<uintptr_t>(p.act_src) | (p.pl.n_planes ? reinterpret_cast<uintptr_t>(p.pl.base) * 2 : 0) |
                         reinterpret_cast<uintptr_t>(p.act_pl) * 2;
    p.vec_epi = p.N % 4 == 0 && (!p.dst || p.ldd % 4 == 0) && ((!p.act_src && !p.act_pl) || p.ld_act % 4 == 0) && (al & 15) == 0 &&
                (!p.pl.n_planes || p.pl.ld % 4 == 0);
    if (!p.dst && (!p.vec_epi || p.accumulate || !p.pl.n_planes)) return UNFLOW_ERR_UNSUPPORTED;   // planes-only output
  }
  if (halo && p.vec_epi && pl_halo_sk_ok(p, npl, halo_bn) && ws && ws_bytes >= pl_halo_sk_ws_bytes())
    return launch_pl_halo_sk(p, ws, ws_bytes, st);      // persistent stream-K form (conv_streamk.hip): no split, no reduce pass
  int code;
  if (halo) {
    if (npl == 3) code = halo_bn == 128 ? launch_pl_halo<128, 64, 3, false>(p, st) : launch_pl_halo<64, 32, 3, false>(p, st);
    else if (tall) code = launch_pl_halo_f16_tall(p, st);
    else if (f16_kch(p, npl) == 1 && opt.f16_db > 0)
      code = halo_bn == 128 ? launch_pl_halo<128, 64, 1, true, 64, true>(p, st) : launch_pl_halo<64, 32, 1, true, 64, true>(p, st);
    else if (f16_kch(p, npl) == 1) code = halo_bn == 128 ? launch_pl_halo<128, 64, 1, true>(p, st) : launch_pl_halo<64, 32, 1, true>(p, st);
    else code = halo_bn == 128 ? launch_pl_halo<128, 64, F16_KCH, true>(p, st) : launch_pl_halo<64, 32, F16_KCH, true>(p, st);
  } else {
    code = pl.pp ? launch_pl_gather_pp(p, st) : npl == 3 ? run_pl_gather_mode<3, false>(p, pl.cfg, st) : run_pl_gather_mode<1, true>(p, pl.cfg, st);
  }
  if (code != UNFLOW_OK) return code;
  if (p.nsplit > 1) {
    const size_t total = (size_t)p.B * p.Hd * p.Wd * p.N;
    pl_splitk_reduce_epilogue_kernel<<<stream_grid((long)(p.vec_epi ? total / 4 : total)), 256, 0, st>>>(p, p.vec_epi);
    return launch_status();
  }
  return UNFLOW_OK;
}

inline int pl_wgrad_cfg(const WgradGeom& p) { return p.Cb <= 64 ? 1 : 0; }   // 0: 128x128, 1: 128x64

// ping-pong form (igemm_pl_wgrad_pp_kernel): bf16 x 3 planes, N > 64, at least one full 256-row tile
// Measured per layer (FlowNetC 384 x 512, B = 4, profiles/r03_wgrad_pingpong_per_layer.txt): wins where a workgroup runs many
// stages and N > 128 (conv3_1 260 -> 238 us, the 1028 -> 256 deconv 109 -> 94); loses on the 6 x 8 / 12 x 16 layers (a
// split leaves 1-3 stages per workgroup: nothing to pipeline) and is neutral-to-worse with the 64 x 64 wave tile (N <= 128).
// Option wgrad_pp: 0 off, 1 on by this rule, 3 everywhere it can run (tests), 2 / 4 the same with waves 2k / 2k+1 paired.
inline bool pl_wgrad_pp_ok(const WgradGeom& p, int npl) {
  const int o = unflow::options().wgrad_pp;
  if (o <= 0 || !(npl == 3 || (npl == 1 && unflow::options().f16_wgrad_dma >= 2)) || p.Cb <= 64 || p.KH * p.KW * p.Ca < 256) return false;
  if (o >= 3) return true;
  // (round 6, per-layer A/B of wgrad_pp = 3 against this rule, profiles/r06_planner_per_layer_ab.txt: the 1028 -> 256
  // conv_transpose at 1536 sites gains 17 us of 106, the layers at 384 sites lose 14-18 us, everything between is neutral)
  return p.Cb > 128 && (long)p.B * p.Hg * p.Wg >= 1536;
}
inline int pl_wgrad_pp_bn(const WgradGeom& p) { return p.Cb > 128 ? 256 : 128; }

inline int plan_pl_wgrad(const WgradGeom& p, int npl) {
  const int Mp = p.KH * p.KW * p.Ca;
  if (pl_wgrad_pp_ok(p, npl)) {
    const long blocks = (long)cdiv(Mp, 256) * cdiv(p.Cb, pl_wgrad_pp_bn(p));
    const long S = (long)p.B * p.Hg * p.Wg;
    const int KT = (int)((S + BK - 1) / BK);
    const int min_kt = max(1, unflow::options().wgrad_min_kt);
    return fill_one_round(blocks, 256, min(256, KT / min_kt > 0 ? KT / min_kt : 1));      // one workgroup per CU
  }
  const int cfg = pl_wgrad_cfg(p);
  const int bn = cfg == 1 ? 64 : 128;
  const long blocks = (long)cdiv(Mp, 128) * cdiv(p.Cb, bn);
  const long S = (long)p.B * p.Hg * p.Wg;
  const int KT = (int)((S + BK - 1) / BK);
  // fp16 (one product per fragment pair: a stage is a third of the bf16 kernel's matrix-core time): four times the sites per split
  // (per-layer A/B at B = 8, profiles/r06_f16_per_layer_ab.txt: the two deep conv_transpose filter gradients -30 us each)
  const int min_kt = max(1, unflow::options().wgrad_min_kt) * (npl == 1 ? 4 : 1);
  const int max_by_k = min(256, KT / min_kt > 0 ? KT / min_kt : 1);
  const int per_cu = min((160 * 1024) / (npl * (128 + bn) * BK * 2), cfg == 1 ? 4 : 3);
  // (Fewer blocks to shrink the partial sums — blocks x 64 KB whatever the layer, 50 MB at 768 blocks — lose more than the
  // traffic costs: 587 / 594 / 607 image-pairs/s at 256 / 384 / 512 blocks against ~620 at 768, profiles/r02_knob_sweep.txt.)
  return fill_one_round(blocks, 256 * per_cu, max_by_k);
}

inline size_t pl_wgrad_partial_bytes(const WgradGeom& p, int ca_out, int nsplit) {
  const size_t n = (size_t)p.KH * p.KW * ca_out * p.Cb;
  return nsplit > 1 ? (size_t)nsplit * n * sizeof(float) + reduce_scratch_bytes(n, nsplit) : 0;
}
inline size_t pl_wgrad_plan_bytes(const WgradGeom& p, int ca_out, int npl) {
  return pl_wgrad_partial_bytes(p, ca_out, plan_pl_wgrad(p, npl));
}

template <int BM, int BN, int WM, int WN, int NPL, bool F16>
int launch_pl_wgrad(const PlWgradParams& p, hipStream_t st) {
  const int Mp = p.KH * p.KW * p.Ca;
  const int smem = NPL * (BM + BN) * BK * 2;
  static DynLdsBook attr_book{};
  (void)ensure_dyn_lds(reinterpret_cast<const void*>(&igemm_pl_wgrad_kernel<BM, BN, WM, WN, NPL, F16>), smem, attr_book);
  PlWgradParams q = p;
  q.mt = cdiv(Mp, BM); q.nt = cdiv(p.Cb, BN);
  igemm_pl_wgrad_kernel<BM, BN, WM, WN, NPL, F16><<<q.mt * q.nt * p.nsplit, 256, smem, st>>>(q);
  return launch_status();
}

template <int BN, int WN, bool F16 = false>
int launch_pl_wgrad_dma(const PlWgradParams& p, hipStream_t st) {
  const int Mp = p.KH * p.KW * p.Ca;
  const int smem = 2 * 3 * (128 + BN) * 16 * 2;
  static DynLdsBook attr_book{};
  (void)ensure_dyn_lds(reinterpret_cast<const void*>(&igemm_pl_wgrad_dma_kernel<BN, WN, F16>), smem, attr_book);
  PlWgradParams q = p;
  q.mt = cdiv(Mp, 128); q.nt = cdiv(p.Cb, BN);
  igemm_pl_wgrad_dma_kernel<BN, WN, F16><<<q.mt * q.nt * p.nsplit, 256, smem, st>>>(q);
  return launch_status();
}

template <int BN, bool F16 = false>
int launch_pl_wgrad_pp(const PlWgradParams& p, hipStream_t st) {
  const int Mp = p.KH * p.KW * p.Ca;
  const int smem = 3 * 3 * (256 + BN) * 16 * 2;
  static DynLdsBook attr_book{};
  (void)ensure_dyn_lds(reinterpret_cast<const void*>(&igemm_pl_wgrad_pp_kernel<BN, F16>), smem, attr_book);
  PlWgradParams q = p;
  q.mt = cdiv(Mp, 256); q.nt = cdiv(p.Cb, BN);
  igemm_pl_wgrad_pp_kernel<BN, F16><<<q.mt * q.nt * p.nsplit, 512, smem, st>>>(q, (unflow::options().wgrad_pp & 1) ? 0 : 1);
  return launch_status();
}

int run_pl_wgrad(PlWgradParams& p, int npl, void* ws, size_t ws_bytes, size_t* used, hipStream_t st) {
  const unflow::Options& opt = unflow::options();
  if (p.out_scale == 0.f) p.out_scale = 1.f;
  p.xcd = opt.xcd_swizzle;
  p.cag_magic = p.Ca == 8 ? 0u : magic_u32((unsigned)(p.Ca >> 3));   // 2^32 / 1 does not fit: 0 marks 'no division'
  const size_t wsize = (size_t)p.KH * p.KW * p.Ca_out * p.Cb;
  int ns = plan_pl_wgrad(p, npl);
  if (ns > 1 && (!ws || ws_bytes < pl_wgrad_partial_bytes(p, p.Ca_out, ns))) {
    ns = ws ? (int)min((size_t)REDUCE_FAN, ws_bytes / (wsize * sizeof(float))) : 1;
    if (ns < 1) ns = 1;
  }
  p.nsplit = ns;
  p.partial = ns > 1 ? reinterpret_cast<float*>(ws) : nullptr;
  *used = pl_wgrad_partial_bytes(p, p.Ca_out, ns);
  const int cfg = pl_wgrad_cfg(p);
  int code;
  if (pl_wgrad_pp_ok(p, npl) && npl == 1) code = pl_wgrad_pp_bn(p) == 256 ? launch_pl_wgrad_pp<256, true>(p, st) : launch_pl_wgrad_pp<128, true>(p, st);
  else if (pl_wgrad_pp_ok(p, npl)) code = pl_wgrad_pp_bn(p) == 256 ? launch_pl_wgrad_pp<256>(p, st) : launch_pl_wgrad_pp<128>(p, st);
  else if (npl == 3 && opt.wgrad_dma) code = cfg == 1 ? launch_pl_wgrad_dma<64, 32>(p, st) : launch_pl_wgrad_dma<128, 64>(p, st);
  else if (npl == 3) code = cfg == 1 ? launch_pl_wgrad<128, 64, 64, 32, 3, false>(p, st) : launch_pl_wgrad<128, 128, 64, 64, 3, false>(p, st);
  else if (opt.wgrad_dma && opt.f16_wgrad_dma) code = cfg == 1 ? launch_pl_wgrad_dma<64, 32, true>(p, st) : launch_pl_wgrad_dma<128, 64, true>(p, st);
  else code = cfg == 1 ? launch_pl_wgrad<128, 64, 64, 32, 1, true>(p, st) : launch_pl_wgrad<128, 128, 64, 64, 1, true>(p, st);
  if (code != UNFLOW_OK) return code;
  if (ns > 1) return reduce_partials(p.partial, p.partial + (size_t)ns * wsize, p.out, wsize, ns, st);
  return UNFLOW_OK;
}

inline bool planes_ok(const unflow_planes* t, int channels) {
  return t && t->base && (t->n_planes == 1 || t->n_planes == 3) && t->ld % 4 == 0 && t->ld >= ((channels + 7) & ~7) &&
         (reinterpret_cast<uintptr_t>(t->base) & 7) == 0;
}

inline PlaneOut plane_out(const unflow_planes* t, int lo, int hi) {
  PlaneOut o{};
  if (t && t->base && t->n_planes) {
    o.base = reinterpret_cast<unsigned short*>(t->base);
    o.plane_stride = t->plane_stride;
    o.ld = t->ld;
    o.lo = lo;
    o.hi = hi;
    o.n_planes = t->n_planes;
    o.scale = t->n_planes == 1 ? t->scale : 0.f;
  }
  return o;
}
inline float plane_scale(const unflow_planes* t) { return (t && t->n_planes == 1 && t->scale != 0.f) ? t->scale : 1.f; }

// source of the leaky-ReLU derivative of a data gradient: the fp32 activation if the caller has one, else the first operand
// plane of the activation (same sign, a third of the bytes)
inline void set_act(PlGatherParams& p, const float* act_src, int ld_act, const unflow_planes* act_pl) {
  p.act_src = act_src; p.ld_act = ld_act; p.act_pl = nullptr;
  if (!act_src && act_pl && act_pl->base) {
    p.act_pl = reinterpret_cast<const unsigned short*>(act_pl->base);
    p.ld_act = act_pl->ld;
  }
}

}  // namespace

// The fp32-operand kernels of conv_igemm.hip, used by the *_pl entry points for the shapes the plane kernels do not take
// (Cout <= 4 flow heads, 2 -> 2 deconvs, 1x1 x 32 data gradient): same arithmetic as the plain entry points, plus the
// optional output planes.
int unflow_conv2d_fwd_po(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int B, int H, int W,
                         int Cin, int Cout, int k, int stride, int leaky, const igemm::PlaneOut& po, void* workspace,
                         size_t workspace_bytes, unflow_stream_t stream);
int unflow_conv2d_bwd_data_po(const float* dz, int lddz, const float* w, float* dx, int lddx, int B, int H, int W, int Cin,
                              int Cout, int k, int stride, int accumulate, const float* act_src, int ld_act, int act_lo,
                              int act_hi, const igemm::PlaneOut& po, void* workspace, size_t workspace_bytes,
                              unflow_stream_t stream);
int unflow_conv2d_transpose_fwd_po(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int B, int H,
                                   int W, int Cin, int Cout, int leaky, const igemm::PlaneOut& po, void* workspace,
                                   size_t workspace_bytes, unflow_stream_t stream);
int unflow_conv2d_transpose_bwd_data_po(const float* dz, int lddz, const float* w, float* dx, int lddx, int B, int H, int W,
                                        int Cin, int Cout, int accumulate, const float* act_src, int ld_act, int act_lo,
                                        int act_hi, const igemm::PlaneOut& po, void* workspace, size_t workspace_bytes,
                                        unflow_stream_t stream);

// ===================================================================== C ABI
// Host-side replay of the work order of the gather / halo kernels (xcd_remap + work_decode, the same functions the kernels
// run): out[4 * b .. 4 * b + 3] = (M tile, N tile, class, split) of workgroup b, M tile = -1 for the padding blocks of order 2.
// Returns the grid size; out may be NULL to query it.  For tests: every (M tile, N tile, class, split) exactly once.
UNFLOW_API int unflow_debug_work_order(int mt, int nt, int ncls, int nsplit, int order, int xcd, int* out, int out_blocks) {
  if (mt <= 0 || nt <= 0 || ncls <= 0 || nsplit <= 0 || order < 0 || order > 2) return UNFLOW_ERR_SHAPE;
  PlGatherParams q{};
  q.mt = mt; q.nt = nt; q.ncls = ncls; q.nsplit = nsplit; q.order = order;
  const int grid = pl_grid(q);
  if (!out) return grid;
  if (out_blocks < grid) return UNFLOW_ERR_WORKSPACE;
  for (int b = 0; b < grid; b++) {
    int m, n, c, sp;
    work_decode(xcd_remap(b, grid, xcd), mt, nt, ncls, nsplit, order, q.mgroup, m, n, c, sp);
    out[4 * b] = m; out[4 * b + 1] = n; out[4 * b + 2] = c; out[4 * b + 3] = sp;
  }
  return grid;
}

#ifdef UNFLOW_PHASE_TRACE
UNFLOW_API int unflow_debug_phase_trace(unsigned long long* host_out, int clear) {      // [8 waves][8]: 6 phase sums, tiles
  if (clear) {
    static unsigned long long zero[8 * 8];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase_trace), zero, sizeof(zero)) == hipSuccess ? UNFLOW_OK : UNFLOW_ERR_LAUNCH;
  }
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_trace), sizeof(unsigned long long) * 8 * 8) == hipSuccess ? UNFLOW_OK
                                                                                                                       : UNFLOW_ERR_LAUNCH;
}
#endif

UNFLOW_API int unflow_planes_from_f32(const float* x, int ldx, long npix, int C, int C_fill, const unflow_planes* out,
                                      unflow_stream_t stream) {
  if (!x || !out || !out->base) return UNFLOW_ERR_NULL;
  if (npix <= 0 || C <= 0) return UNFLOW_OK;
  const int Cp = (C_fill + 3) & ~3;
  if ((out->n_planes != 1 && out->n_planes != 3) || out->ld % 4 != 0 || (reinterpret_cast<uintptr_t>(out->base) & 7) != 0 ||
      ldx < C || Cp < C || Cp > ((C + 7) & ~7) || out->ld < Cp)
    return UNFLOW_ERR_UNSUPPORTED;
  planes_from_f32_kernel<<<stream_grid(npix * (Cp / 4)), 256, 0, as_stream(stream)>>>(x, ldx, npix, C, Cp, plane_out(out, 0, Cp));
  return launch_status();
}

UNFLOW_API size_t unflow_weight_planes_elems(int taps, int R, int Cc, int transposed) {
  return transposed ? (size_t)taps * Cc * ((R + 7) & ~7) : (size_t)taps * R * ((Cc + 7) & ~7);
}

UNFLOW_API int unflow_weight_planes_batched(int n, const float* const* w, const int* taps, const int* R, const int* Cc,
                                            void* const* direct, void* const* transposed, int n_planes,
                                            unflow_stream_t stream) {
  if (!w || !taps || !R || !Cc || !direct || !transposed) return UNFLOW_ERR_NULL;
  if (n_planes != 1 && n_planes != 3) return UNFLOW_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  for (int i0 = 0; i0 < n; i0 += MAX_WDESC) {
    WPlaneBatch b{};
    b.n = min(n - i0, MAX_WDESC);
    b.n_planes = n_planes;
    int blocks = 0;
    for (int i = 0; i < b.n; i++) {
      WPlaneDesc& d = b.d[i];
      if (!w[i0 + i] || taps[i0 + i] <= 0 || R[i0 + i] <= 0 || Cc[i0 + i] <= 0) return UNFLOW_ERR_SHAPE;
      d.w = w[i0 + i];
      d.direct = reinterpret_cast<unsigned short*>(direct[i0 + i]);
      d.transposed = reinterpret_cast<unsigned short*>(transposed[i0 + i]);
      d.taps = taps[i0 + i]; d.R = R[i0 + i]; d.Cc = Cc[i0 + i];
      d.tiles_r = cdiv((d.R + 7) & ~7, 64);
      d.tiles_c = cdiv((d.Cc + 7) & ~7, 64);
      d.block0 = blocks;
      blocks += d.taps * d.tiles_r * d.tiles_c;
    }
    if (blocks > 0) weight_planes_kernel<<<blocks, 256, 0, st>>>(b);
  }
  return launch_status();
}

UNFLOW_API int unflow_adam_planes_batched(int n, float* const* w, const int* taps, const int* R, const int* Cc, void* const* direct,
                                          void* const* transposed, const int* regularized, int n_planes, float* P, const float* G,
                                          float* M, float* V, float grad_scale, float l2_scale, float lr_t, float beta1, float beta2,
                                          float eps, float* loss_acc, unflow_stream_t stream) {
  if (!w || !taps || !R || !Cc || !direct || !transposed || !regularized || !P || !G || !M || !V) return UNFLOW_ERR_NULL;
  if (n_planes != 0 && n_planes != 1 && n_planes != 3) return UNFLOW_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  AdamPlaneArgs a{P, G, M, V, grad_scale, l2_scale, lr_t, beta1, beta2, eps, loss_acc};
  for (int i0 = 0; i0 < n; i0 += MAX_WDESC) {
    WPlaneBatch b{};
    b.n = min(n - i0, MAX_WDESC);
    b.n_planes = n_planes ? n_planes : 3;
    unsigned long long reg_mask = 0;
    int blocks = 0;
    for (int i = 0; i < b.n; i++) {
      WPlaneDesc& d = b.d[i];
      if (!w[i0 + i] || taps[i0 + i] <= 0 || R[i0 + i] <= 0 || Cc[i0 + i] <= 0 || w[i0 + i] < P) return UNFLOW_ERR_SHAPE;
      if (!n_planes && (direct[i0 + i] || transposed[i0 + i])) return UNFLOW_ERR_UNSUPPORTED;
      d.w = w[i0 + i];
      d.direct = reinterpret_cast<unsigned short*>(direct[i0 + i]);
      d.transposed = reinterpret_cast<unsigned short*>(transposed[i0 + i]);
      d.taps = taps[i0 + i]; d.R = R[i0 + i]; d.Cc = Cc[i0 + i];
      d.tiles_r = cdiv(d.R, 64);
      d.tiles_c = cdiv(d.Cc, 64);
      d.block0 = blocks;
      blocks += d.taps * d.tiles_r * d.tiles_c;
      if (regularized[i0 + i]) reg_mask |= 1ull << i;
    }
    if (blocks > 0) adam_planes_kernel<<<blocks, 256, 0, st>>>(b, a, reg_mask);
  }
  return launch_status();
}

static int pl_gather_nsplit(const GatherGeom& g, int npl) {      // (workspace sizing: the larger of the forms that may run)
  int bn;
  if (!pl_halo_ok(g)) return plan_pl_gather(g, npl).nsplit;
  return plan_pl_halo(g, npl, &bn);
}

UNFLOW_API size_t unflow_conv_pl_workspace_bytes(int B, int H, int W, int Cin, int Cout, int k, int stride, int n_planes) {
  // Exact requirement of the *_pl entry points of a conv2d with these dims (and, for k == 4 && stride == 2, of the
  // conv2d_transpose whose OUTPUT is [B,H,W,Cout]); Cin / Cout as passed to those entry points.
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || stride <= 0) return 0;
  size_t need = unflow_conv_workspace_bytes(B, H, W, (Cin + 3) & ~3, Cout, k, stride);     // fp32-operand fallbacks
  if (Cout <= 4 || (n_planes != 1 && n_planes != 3)) return need;
  const int Ci8 = (Cin + 7) & ~7, Co8 = (Cout + 7) & ~7;
  GatherGeom g{};
  build_conv_fwd(g, B, H, W, Ci8, Cout, k, stride);
  need = max(need, pl_gather_partial_bytes(g, pl_gather_nsplit(g, n_planes)));
  if (unflow::options().streamk > 0 && n_planes == 3) need = max(need, pl_halo_sk_ws_bytes());      // slabs + flags of the stream-K kernels
  if (stride == 2 && build_conv_fwd_s2acc(g, B, H, W, Ci8, Cout, k) && pl_halo_acc_pays(g))
    need = max(need, pl_gather_partial_bytes(g, pl_gather_nsplit(g, n_planes)));
  GatherGeom d{};
  if ((stride == 1 || stride == 2) && build_conv_dgrad(d, B, H, W, Cin, Co8, k, stride) == UNFLOW_OK)
    need = max(need, pl_gather_partial_bytes(d, pl_gather_nsplit(d, n_planes)));
  WgradGeom wg{};
  build_conv_wgrad(wg, B, H, W, Ci8, Cout, k, stride);
  need = max(need, pl_wgrad_plan_bytes(wg, Cin, n_planes));
  if (Cin == 4 && k == 7 && stride == 2 && W % 2 == 0) {   // the two-pixel-granule form of FlowNetC's first layer (rgb4_form)
    g.Cs = 32; g.KW = 1; g.wtaps = 7; g.cls[0].ntx = 1;
    need = max(need, pl_gather_partial_bytes(g, pl_gather_nsplit(g, n_planes)));
    wg.Ca = 32; wg.KW = 1;
    need = max(need, pl_wgrad_plan_bytes(wg, 28, n_planes));
  }
  if (k == 4 && stride == 2 && H % 2 == 0 && W % 2 == 0) {
    GatherGeom tf{};
    build_deconv_fwd(tf, B, H / 2, W / 2, Ci8, Cout);
    need = max(need, pl_gather_partial_bytes(tf, pl_gather_nsplit(tf, n_planes)));
    GatherGeom td{};
    build_deconv_dgrad(td, B, H / 2, W / 2, Cin, Co8);
    need = max(need, pl_gather_partial_bytes(td, pl_gather_nsplit(td, n_planes)));
    build_deconv_dgrad_acc(td, B, H / 2, W / 2, Cin, Co8);
    if (pl_halo_acc_pays(td)) need = max(need, pl_gather_partial_bytes(td, pl_gather_nsplit(td, n_planes)));
    WgradGeom tw{};
    build_deconv_wgrad(tw, B, H / 2, W / 2, Cin, Co8);
    need = max(need, pl_wgrad_plan_bytes(tw, Cout, n_planes));
  }
  return need + 1024;
}

int conv_first7_fwd(const unflow_planes* x_pl, const unflow_planes* w_pl, const float* bias, const unflow_planes* y_pl, int B, int H,
                    int W, int Cout, int leaky, hipStream_t st);      // conv_first.hip

// The first layer of a FlowNetC (7x7 stride 2 over RGB0, flownet.py:204): with 8-channel K granules half of every granule
// is padding (49 taps x 8 = 392 K slots for 147 weights).  Stored with row length 4, a 16-byte granule of the input planes is
// TWO neighbouring pixels, and because the SAME padding on the left is even (2) and W is even, the pairs a tap row needs —
// pixels (2xg-2, 2xg-1) .. (2xg+4, 2xg+5) — never straddle the image border: the layer becomes a 7x1 convolution over
// "pixels" of 32 channels (4 granules, the last pixel of the last one meets a zero weight row): K = 7 x 32 = 224 slots.
// The weight tensor [7][7][4][Cout] read as [7 taps][28 rows][Cout] gives the matching planes (rows 28..31 zero).
static bool rgb4_form(const unflow_planes* x_pl, int W, int Cin, int k, int stride) {
  return x_pl && x_pl->base && Cin == 4 && k == 7 && stride == 2 && x_pl->ld == 4 && W % 2 == 0 &&
         (x_pl->n_planes == 1 || x_pl->n_planes == 3) && (reinterpret_cast<uintptr_t>(x_pl->base) & 7) == 0;
}

static bool use_planes(const unflow_planes* a, int ca, const unflow_planes* b, int cb, int Cout_or_n) {
  return Cout_or_n > 4 && planes_ok(a, ca) && planes_ok(b, cb) && a->n_planes == b->n_planes;
}

UNFLOW_API int unflow_conv2d_fwd_pl(const float* x, int ldx, const unflow_planes* x_pl, const float* w,
                                    const unflow_planes* w_pl, const float* bias, float* y, int ldy,
                                    const unflow_planes* y_pl, int B, int H, int W, int Cin, int Cout, int k, int stride,
                                    int leaky, void* workspace, size_t workspace_bytes, unflow_stream_t stream) {
  if ((!y && !(y_pl && y_pl->base)) || (!x && !x_pl)) return UNFLOW_ERR_NULL;      // y == NULL: planes-only output
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || stride <= 0) return UNFLOW_ERR_SHAPE;
  const int Ci8 = (Cin + 7) & ~7;
  const bool rgb4 = rgb4_form(x_pl, W, Cin, k, stride) && Cout > 4 && planes_ok(w_pl, 28) && w_pl->ld == 32 &&
                    w_pl->n_planes == x_pl->n_planes;
  if (!rgb4 && (!use_planes(x_pl, Cin, w_pl, Cin, Cout) || w_pl->ld != Ci8))
    return !y ? UNFLOW_ERR_UNSUPPORTED : unflow_conv2d_fwd_po(x, ldx, w, bias, y, ldy, B, H, W, Cin, Cout, k, stride, leaky, plane_out(y_pl, 0, Cout), workspace,
                                workspace_bytes, stream);
  if (y && ldy < Cout) return UNFLOW_ERR_UNSUPPORTED;
  if (rgb4 && !y) {                 // planes-only first layer: its own kernel where it applies (conv_first.hip)
    const int code = conv_first7_fwd(x_pl, w_pl, bias, y_pl, B, H, W, Cout, leaky, as_stream(stream));
    if (code != UNFLOW_ERR_UNSUPPORTED) return code;
  }
  PlGatherParams p{};
  build_conv_fwd(p, B, H, W, Ci8, Cout, k, stride);
  if (stride == 2 && !rgb4) {       // source stride 2: the halo kernel over four accumulating parity classes, where it applies
    GatherGeom a{};
    if (build_conv_fwd_s2acc(a, B, H, W, Ci8, Cout, k) && pl_halo_acc_pays(a)) static_cast<GatherGeom&>(p) = a;
  }
  if (rgb4) {
    p.Cs = 32; p.KW = 1; p.wtaps = 7; p.gpx = 2;
    p.cls[0].ntx = 1;
  }
  p.src = reinterpret_cast<const unsigned short*>(x_pl->base); p.src_ps = x_pl->plane_stride; p.lds = x_pl->ld;
  p.w = reinterpret_cast<const unsigned short*>(w_pl->base); p.w_ps = w_pl->plane_stride;
  p.src_inv = 1.f / (plane_scale(x_pl) * plane_scale(w_pl));
  p.bias = bias; p.dst = y; p.ldd = ldy; p.act_src = nullptr; p.leaky = leaky; p.accumulate = 0;
  p.pl = plane_out(y_pl, 0, Cout);
  return run_pl_gather(p, x_pl->n_planes, workspace, workspace_bytes, as_stream(stream));
}

UNFLOW_API int unflow_conv2d_bwd_data_pl(const float* dz, int lddz, const unflow_planes* dz_pl, const float* w,
                                         const unflow_planes* w_pl, float* dx, int lddx, const unflow_planes* dx_pl,
                                         int pl_lo, int pl_hi, int B, int H, int W, int Cin, int Cout, int k, int stride,
                                         int accumulate, const float* act_src, int ld_act, const unflow_planes* act_pl,
                                         int act_lo, int act_hi, void* workspace, size_t workspace_bytes,
                                         unflow_stream_t stream) {
  if ((!dx && !(dx_pl && dx_pl->base)) || (!dz && !dz_pl)) return UNFLOW_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || (stride != 1 && stride != 2)) return UNFLOW_ERR_SHAPE;
  const int Co8 = (Cout + 7) & ~7;
  const bool pointwise32 = k == 1 && stride == 1 && Cout == 32;      // conv_redir: the streaming kernel of conv_igemm.hip
  if (pointwise32 || !use_planes(dz_pl, Cout, w_pl, Cout, Cout) || w_pl->ld != Co8)
    return (!dx || (!act_src && act_pl && act_hi > act_lo)) ? UNFLOW_ERR_UNSUPPORTED : unflow_conv2d_bwd_data_po(dz, lddz, w, dx, lddx, B, H, W, Cin, Cout, k, stride, accumulate, act_src, ld_act, act_lo,
                                     act_hi, plane_out(dx_pl, pl_lo, pl_hi), workspace, workspace_bytes, stream);
  if (dx && lddx < Cin) return UNFLOW_ERR_UNSUPPORTED;
  PlGatherParams p{};
  const int bc = build_conv_dgrad(p, B, H, W, Cin, Co8, k, stride);
  if (bc != UNFLOW_OK) return bc;
  p.src = reinterpret_cast<const unsigned short*>(dz_pl->base); p.src_ps = dz_pl->plane_stride; p.lds = dz_pl->ld;
  p.w = reinterpret_cast<const unsigned short*>(w_pl->base); p.w_ps = w_pl->plane_stride;
  p.src_inv = 1.f / (plane_scale(dz_pl) * plane_scale(w_pl));
  p.bias = nullptr; p.dst = dx; p.ldd = lddx; p.act_lo = act_lo; p.act_hi = act_hi;
  set_act(p, act_src, ld_act, act_pl);
  p.leaky = 0; p.accumulate = accumulate;
  p.pl = plane_out(dx_pl, pl_lo, pl_hi);
  return run_pl_gather(p, dz_pl->n_planes, workspace, workspace_bytes, as_stream(stream));
}

UNFLOW_API int unflow_conv2d_bwd_filter_pl(const float* x, int ldx, const unflow_planes* x_pl, const float* dz, int lddz,
                                           const unflow_planes* dz_pl, float* dw, int B, int H, int W, int Cin, int Cout,
                                           int k, int stride, void* workspace, size_t workspace_bytes,
                                           unflow_stream_t stream) {
  if (!dw || (!x && !x_pl) || (!dz && !dz_pl)) return UNFLOW_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || stride <= 0) return UNFLOW_ERR_SHAPE;
  const bool rgb4 = rgb4_form(x_pl, W, Cin, k, stride) && planes_ok(dz_pl, Cout) && dz_pl->n_planes == x_pl->n_planes &&
                    Cout > 4 && Cout % 4 == 0;
  if (!rgb4 && (!use_planes(x_pl, Cin, dz_pl, Cout, Cout) || Cout % 4 != 0))
    return (!x || !dz) ? UNFLOW_ERR_UNSUPPORTED :      // planes-only operand: no fp32 copy to fall back on
           unflow_conv2d_bwd_filter(x, ldx, dz, lddz, dw, nullptr, B, H, W, Cin, Cout, k, stride, workspace, workspace_bytes, stream);
  PlWgradParams p{};
  build_conv_wgrad(p, B, H, W, (Cin + 7) & ~7, Cout, k, stride);
  if (rgb4) { p.Ca = 32; p.KW = 1; p.gpx = 2; }
  p.src = reinterpret_cast<const unsigned short*>(x_pl->base); p.src_ps = x_pl->plane_stride; p.lds = x_pl->ld;
  p.dst = reinterpret_cast<const unsigned short*>(dz_pl->base); p.dst_ps = dz_pl->plane_stride; p.ldd = dz_pl->ld;
  p.out_scale = 1.f / (plane_scale(x_pl) * plane_scale(dz_pl));
  p.out = dw; p.Ca_out = rgb4 ? 28 : Cin;
  size_t used = 0;
  return run_pl_wgrad(p, x_pl->n_planes, workspace, workspace_bytes, &used, as_stream(stream));
}

UNFLOW_API int unflow_conv2d_transpose_fwd_pl(const float* x, int ldx, const unflow_planes* x_pl, const float* w,
                                              const unflow_planes* w_pl, const float* bias, float* y, int ldy,
                                              const unflow_planes* y_pl, int B, int H, int W, int Cin, int Cout, int leaky,
                                              void* workspace, size_t workspace_bytes, unflow_stream_t stream) {
  if ((!y && !(y_pl && y_pl->base)) || (!x && !x_pl)) return UNFLOW_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UNFLOW_ERR_SHAPE;
  const int Ci8 = (Cin + 7) & ~7;
  if (!use_planes(x_pl, Cin, w_pl, Cin, Cout) || w_pl->ld != Ci8)
    return !y ? UNFLOW_ERR_UNSUPPORTED : unflow_conv2d_transpose_fwd_po(x, ldx, w, bias, y, ldy, B, H, W, Cin, Cout, leaky, plane_out(y_pl, 0, Cout), workspace,
                                          workspace_bytes, stream);
  if (y && ldy < Cout) return UNFLOW_ERR_UNSUPPORTED;
  PlGatherParams p{};
  build_deconv_fwd(p, B, H, W, Ci8, Cout);
  p.src = reinterpret_cast<const unsigned short*>(x_pl->base); p.src_ps = x_pl->plane_stride; p.lds = x_pl->ld;
  p.w = reinterpret_cast<const unsigned short*>(w_pl->base); p.w_ps = w_pl->plane_stride;
  p.src_inv = 1.f / (plane_scale(x_pl) * plane_scale(w_pl));
  p.bias = bias; p.dst = y; p.ldd = ldy; p.act_src = nullptr; p.leaky = leaky; p.accumulate = 0;
  p.pl = plane_out(y_pl, 0, Cout);
  return run_pl_gather(p, x_pl->n_planes, workspace, workspace_bytes, as_stream(stream));
}

UNFLOW_API int unflow_conv2d_transpose_bwd_data_pl(const float* dz, int lddz, const unflow_planes* dz_pl, const float* w,
                                                   const unflow_planes* w_pl, float* dx, int lddx,
                                                   const unflow_planes* dx_pl, int pl_lo, int pl_hi, int B, int H, int W,
                                                   int Cin, int Cout, int accumulate, const float* act_src, int ld_act,
                                                   const unflow_planes* act_pl, int act_lo, int act_hi, void* workspace,
                                                   size_t workspace_bytes, unflow_stream_t stream) {
  if ((!dx && !(dx_pl && dx_pl->base)) || (!dz && !dz_pl)) return UNFLOW_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UNFLOW_ERR_SHAPE;
  const int Co8 = (Cout + 7) & ~7;
  if (!use_planes(dz_pl, Cout, w_pl, Cout, Cout) || w_pl->ld != Co8 || Cin <= 4)
    return (!dx || (!act_src && act_pl && act_hi > act_lo)) ? UNFLOW_ERR_UNSUPPORTED : unflow_conv2d_transpose_bwd_data_po(dz, lddz, w, dx, lddx, B, H, W, Cin, Cout, accumulate, act_src, ld_act, act_lo,
                                               act_hi, plane_out(dx_pl, pl_lo, pl_hi), workspace, workspace_bytes, stream);
  if (dx && lddx < Cin) return UNFLOW_ERR_UNSUPPORTED;
  PlGatherParams p{};
  build_deconv_dgrad(p, B, H, W, Cin, Co8);
  {
    GatherGeom a{};
    build_deconv_dgrad_acc(a, B, H, W, Cin, Co8);
    if (pl_halo_acc_pays(a)) static_cast<GatherGeom&>(p) = a;
  }
  p.src = reinterpret_cast<const unsigned short*>(dz_pl->base); p.src_ps = dz_pl->plane_stride; p.lds = dz_pl->ld;
  p.w = reinterpret_cast<const unsigned short*>(w_pl->base); p.w_ps = w_pl->plane_stride;
  p.src_inv = 1.f / (plane_scale(dz_pl) * plane_scale(w_pl));
  p.bias = nullptr; p.dst = dx; p.ldd = lddx; p.act_lo = act_lo; p.act_hi = act_hi;
  set_act(p, act_src, ld_act, act_pl);
  p.leaky = 0; p.accumulate = accumulate;
  p.pl = plane_out(dx_pl, pl_lo, pl_hi);
  return run_pl_gather(p, dz_pl->n_planes, workspace, workspace_bytes, as_stream(stream));
}

UNFLOW_API int unflow_conv2d_transpose_bwd_filter_pl(const float* x, int ldx, const unflow_planes* x_pl, const float* dz,
                                                     int lddz, const unflow_planes* dz_pl, float* dw, int B, int H, int W,
                                                     int Cin, int Cout, void* workspace, size_t workspace_bytes,
                                                     unflow_stream_t stream) {
  if (!dw || (!x && !x_pl) || (!dz && !dz_pl)) return UNFLOW_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UNFLOW_ERR_SHAPE;
  if (!use_planes(x_pl, Cin, dz_pl, Cout, Cout) || Cin % 4 != 0 || Cin <= 4)
    return (!x || !dz) ? UNFLOW_ERR_UNSUPPORTED :
           unflow_conv2d_transpose_bwd_filter(x, ldx, dz, lddz, dw, nullptr, B, H, W, Cin, Cout, workspace, workspace_bytes, stream);
  // dW[ky,kx,co,ci]: gathered operand = dz (rows (tap, co)), dense operand = x (columns ci)
  PlWgradParams p{};
  build_deconv_wgrad(p, B, H, W, Cin, (Cout + 7) & ~7);
  p.src = reinterpret_cast<const unsigned short*>(dz_pl->base); p.src_ps = dz_pl->plane_stride; p.lds = dz_pl->ld;
  p.dst = reinterpret_cast<const unsigned short*>(x_pl->base); p.dst_ps = x_pl->plane_stride; p.ldd = x_pl->ld;
  p.out_scale = 1.f / (plane_scale(x_pl) * plane_scale(dz_pl));
  p.out = dw; p.Ca_out = Cout;
  size_t used = 0;
  return run_pl_wgrad(p, x_pl->n_planes, workspace, workspace_bytes, &used, as_stream(stream));
}
