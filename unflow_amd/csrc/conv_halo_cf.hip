// Class-fused halo kernel: the four OUTPUT-PARITY classes of a stride-2 conv's data gradient / a conv_transpose's forward pass
// (flownet.py:195-206 conv2..conv6 backward, :89-131 deconv5..deconv2) in ONE workgroup.
//
// igemm_pl_halo_kernel gives every parity class its own workgroups: dx[2y + py, 2x + px] of a 5 x 5 stride-2 conv sums 3 x 3,
// 3 x 2, 2 x 3 or 2 x 2 taps of dz around (y, x), a conv_transpose output pixel 2 x 2 taps.  All four classes read the SAME
// neighbourhood of the source — the union of their tap windows, 3 x 3 (5 x 5 conv, conv_transpose) or 2 x 2 (3 x 3 conv)
// source offsets ("virtual taps") — so per-class workgroups load that halo four times, and a layer with N = 64 output channels
// is left with 128 x 64 tiles whose fragment reads (0.75 ds_read_b128 per MFMA) saturate the LDS pipe long before the matrix
// pipe (conv2's data gradient: 145 TFLOP/s, the longest launch of the step).
//
// Here the layer is what it is algebraically: a stride-1 conv over the virtual taps from Cs channels to 4 x N columns (class
// major), followed by a depth-to-space scatter — with a block-sparse filter: column block `class` has a tap at virtual tap v
// only if the class's window covers it (25 of 36 blocks for 5 x 5, 16 of 36 for conv_transpose, 9 of 16 for 3 x 3).  A
// workgroup owns 4 x 32 sites x (4 classes x 32 channels): the halo is staged once per 32-channel chunk for all classes, an A
// fragment (32 sites x K16 x 3 planes) read from LDS feeds up to two classes' products (0.5 reads per MFMA, the 128 x 128
// ratio, whatever N is), the empty (virtual tap, class) blocks are skipped by wave-uniform branches around the products (their
// weight loads carry an out-of-range offset: zeros, no traffic), and work items are four times larger.  Waves: 2 (sites) x 2 (class pairs {0, 3}, {1, 2}:
// the diagonal pairing balances the tap counts, 13 : 12 for 5 x 5).
//
// Everything else is the one-shot halo kernel's: [pixel][32 ch] halo image at the 80-byte pitch, XOR-swizzled 64-byte weight
// rows, two barriers per K32 tile with the next tile's loads issued between the MFMA groups, split-K over whole chunks with
// the chip-wide reduce + epilogue pass, vector epilogue through wave-private staging.  bf16 x 3 planes only.
#include <type_traits>
#include "igemm_shared.h"
#include "planes_shared.h"

namespace {
using namespace igemm;

constexpr int CF_CN = 32;          // channels per class of a workgroup's tile
constexpr int CF_BN = 4 * CF_CN;   // columns: class-major

struct CfGeom {
  int vty, vtx;          // virtual taps: the union of the classes' tap windows
  int dmy, dmx;          // source offset of virtual tap (0, 0)
  int oy[4], ox[4];      // class c's tap (ty, tx) sits at virtual tap (oy[c] - ty, ox[c] - tx)      (dstep = -1)
  unsigned act[4];       // bit vy * vtx + vx: class c has a tap at that virtual tap
};

inline bool cf_geom(const GatherGeom& p, CfGeom& g) {
  if (p.ncls != 4 || p.acc || p.dstep != -1 || p.sm != 1 || p.sp != 1 || p.so != 2) return false;
  int lo_y = 1 << 30, lo_x = 1 << 30, hi_y = -(1 << 30), hi_x = -(1 << 30);
  for (int c = 0; c < 4; c++) {
    const TapClass& t = p.cls[c];
    if (t.nty < 1 || t.ntx < 1 || t.py != (c >> 1) || t.px != (c & 1)) return false;
    lo_y = min(lo_y, t.dy0 - (t.nty - 1)); hi_y = max(hi_y, t.dy0);
    lo_x = min(lo_x, t.dx0 - (t.ntx - 1)); hi_x = max(hi_x, t.dx0);
  }
  g.vty = hi_y - lo_y + 1; g.vtx = hi_x - lo_x + 1; g.dmy = lo_y; g.dmx = lo_x;
  if (g.vty * g.vtx > 16) return false;
  for (int c = 0; c < 4; c++) {
    const TapClass& t = p.cls[c];
    g.oy[c] = t.dy0 - lo_y; g.ox[c] = t.dx0 - lo_x;
    unsigned m = 0;
    for (int vy = 0; vy < g.vty; vy++)
      for (int vx = 0; vx < g.vtx; vx++) {
        const int ty = g.oy[c] - vy, tx = g.ox[c] - vx;
        if (ty >= 0 && ty < t.nty && tx >= 0 && tx < t.ntx) m |= 1u << (vy * g.vtx + vx);
      }
    g.act[c] = m;
  }
  return true;
}
inline int cf_halo_pixels(const CfGeom& g) { return (TH + g.vty - 1) * (TW + g.vtx - 1); }
inline int cf_smem(const CfGeom& g) { return pl_halo_main_bytes(CF_BN, 64, 3, cf_halo_pixels(g)) + 128 * 4 + 16 + 64 * 4; }      // tiles | pixel table | weight-offset table

__global__ __launch_bounds__(256, 2) void igemm_pl_halo_cf_kernel(const PlGatherParams p, const CfGeom g, int HPmax) {
  constexpr int NPL = 3, NT = 6, BM = 128, BN = CF_BN, TM = 2, TN = 2, WM = 64;
  constexpr int B_PLANE = BN * LDH;
  constexpr int NB = BN / 64;                    // weight granules per thread and plane
  constexpr int NH = 4;                          // halo granules per thread and plane (covers 256 pixels)

  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  const int H_PLANE = HPmax * HPITCH;
  unsigned short* Hh = smem16;
  unsigned short* Bh = Hh + NPL * H_PLANE;
  int* pix = reinterpret_cast<int*>(reinterpret_cast<char*>(smem16) + pl_halo_main_bytes(BN, 64, NPL, HPmax));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  int t, ntile, cls_unused, split;
  work_decode(xcd_remap(blockIdx.x, gridDim.x, p.xcd), p.mt, p.nt, 1, p.nsplit, p.order, p.mgroup, t, ntile, cls_unused, split);
  if (t < 0) return;
  const int n0 = ntile * CF_CN;
  const int Cg = p.Cs >> 3;
  const int nchunk = (Cg + 3) >> 2;
  const int ch_per = (nchunk + p.nsplit - 1) / p.nsplit;
  const int ch0 = split * ch_per, ch1 = min(nchunk, (split + 1) * ch_per);
  const int VT = g.vty * g.vtx;
  const int T = max(ch1 - ch0, 0) * VT;          // K tiles (chunk x virtual tap) of this block

  const int txi = t % p.tiles_x; t /= p.tiles_x;
  const int tyi = t % p.tiles_y;
  const int b = t / p.tiles_y;
  const int y0 = tyi * TH, x0 = txi * TW;
  const int HC = TW + g.vtx - 1, HR = TH + g.vty - 1;

  __amdgpu_buffer_rsrc_t src_rs[NPL], w_rs[NPL];
#pragma unroll
  for (int pl = 0; pl < NPL; pl++) {
    src_rs[pl] = make_rsrc(p.src + pl * p.src_ps, (((size_t)p.B * p.Hs * p.Ws - 1) * (size_t)p.lds + (size_t)p.Cs) * 2);
    w_rs[pl] = make_rsrc(p.w + pl * p.w_ps, (size_t)p.wtaps * p.N * p.Cs * 2);
  }
  const int kq = tid & 3;
  const int lds2 = p.lds * 2;
  // this thread's halo granules: pixel hp = (tid >> 2) + 64 j of the union halo, granule kq of the chunk
  int h_off[NH];
#pragma unroll
  for (int j = 0; j < NH; j++) {
    const int hp = (tid >> 2) + 64 * j;
    const int hy = hp / HC, hx = hp - hy * HC;
    const int y = y0 + hy + g.dmy, x = x0 + hx + g.dmx;
    const bool ok = hy < HR && (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;
    h_off[j] = ok ? ((b * p.Hs + y) * p.Ws + x) * lds2 + kq * 16 : OOB_MARK;
  }
  if (tid < BM) {      // destination pixel of class (0, 0); class (py, px) lies py * Wd + px further
    const int yg = y0 + (tid >> TWL), xg = x0 + (tid & (TW - 1));
    pix[tid] = (yg < p.Hg && xg < p.Wg) ? (b * p.Hd + yg * p.so) * p.Wd + xg * p.so : -1;
  }
  // weight-offset table [class][virtual tap]: byte offset of the class's tap there within a weight plane, or -1 (no tap)
  int* wtab = pix + BM + 4;
  if (tid < 64) {
    const int c = tid >> 4, v = tid & 15;
    const int vy = v / g.vtx, vx = v - vy * g.vtx;
    const int ty = g.oy[c] - vy, tx = g.ox[c] - vx;
    const bool on = v < g.vty * g.vtx && ty >= 0 && ty < p.cls[c].nty && tx >= 0 && tx < p.cls[c].ntx;
    wtab[tid] = on ? ((p.cls[c].ky0 + ty * p.kstep) * p.KW + p.cls[c].kx0 + tx * p.kstep) * p.N * p.Cs * 2 : -1;
  }
  // this thread's weight rows: row (tid >> 2) + 64 i of the [class][32 channels] tile (class (tid >> 7) + 2 i: wave-uniform)
  int b_row[NB];
#pragma unroll
  for (int i = 0; i < NB; i++) {
    const int n = n0 + (((tid >> 2) + 64 * i) & 31);
    b_row[i] = n < p.N ? n * p.Cs * 2 + kq * 16 : OOB_MARK;
  }
  const int* wrow = wtab + (tid >> 7) * 16;      // + 32 i: the row of this thread's class
  __syncthreads();                               // the table is read by the first loads

  u32x4 rh[NH][NPL], rb[NB][NPL];
  // loads of the NEXT K tile (chunk ld_chunk, virtual tap (ld_vy, ld_vx)): weights always, the halo when the tile opens a chunk
  int ld_chunk = ch0, ld_v = 0;
  bool ld_live = T > 0;
  auto load_b = [&](int i) {      // branch-free: a (virtual tap, class) block that does not exist loads zeros (out-of-range offset: no traffic)
    const int wo = wrow[32 * i + ld_v];
    const bool on = ld_live && wo >= 0 && ld_chunk * 4 + kq < Cg;
    const int voff = on ? b_row[i] + wo + ld_chunk * 64 : OOB_MARK;
#pragma unroll
    for (int pl = 0; pl < NPL; pl++) rb[i][pl] = buf_ld16(w_rs[pl], voff);
  };
  auto load_h = [&](int j) {
    const bool ok = ld_live && ld_chunk * 4 + kq < Cg;
    const int voff = ok ? h_off[j] + ld_chunk * 64 : OOB_MARK;
#pragma unroll
    for (int pl = 0; pl < NPL; pl++) rh[j][pl] = buf_ld16(src_rs[pl], voff);
  };
  auto ld_advance = [&](int kk_next) {
    ld_v++;
    if (ld_v == VT) { ld_v = 0; ld_chunk++; }
    ld_live = kk_next + 1 < T;
  };
  auto swz = [](int row, int gq) { return row * LDH + 8 * (gq ^ ((row >> 2) & 3)); };
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
  // classes of this wave: j = 0 -> wn (0 or 1), j = 1 -> 3 - wn (3 or 2)
  const int cls_j[TN] = {wn, 3 - wn};
  const unsigned act_j[TN] = {g.act[wn], g.act[3 - wn]};
  int a_rd[TM];
#pragma unroll
  for (int i = 0; i < TM; i++) {
    const int sidx = wm * WM + i * 32 + l31;
    a_rd[i] = ((sidx >> TWL) * HC + (sidx & (TW - 1))) * HPITCH + lh * 8;
  }
  const unsigned short* bh_rd[TN] = {Bh + (cls_j[0] * CF_CN + l31) * LDH, Bh + (cls_j[1] * CF_CN + l31) * LDH};
  const int gsw = lh ^ ((l31 >> 2) & 3);

#pragma unroll
  for (int j = 0; j < NH; j++) load_h(j);
#pragma unroll
  for (int i = 0; i < NB; i++) load_b(i);
  ld_advance(0);
  store_h();
  store_b();
  __syncthreads();
  constexpr int NPIECE = NB + NH;
  int v = 0, vy = 0, vx = 0;                     // virtual tap of the tile being multiplied
  for (int kk = 0; kk < T; kk++) {
    const int tapoff = (vy * HC + vx) * HPITCH;
    const unsigned vbit = 1u << v;
    const bool new_chunk = ld_v == 0;            // the next tile opens a chunk: its halo is loaded during this tile
    auto piece = [&](int step) {
      if (step < NB) load_b(step);
      else if (step < NPIECE && new_chunk) load_h(step - NB);
    };
    // J0 / J1: which of the wave's two classes have a tap here (wave-uniform); the (slab, sub-tile) steps are software-
    // pipelined — the fragments of step s + 1 are read from LDS while the MFMAs of step s run
    auto body = [&](auto j0_tag, auto j1_tag) __attribute__((always_inline)) {
      constexpr bool J0 = decltype(j0_tag)::value, J1 = decltype(j1_tag)::value;
      constexpr int NJ = (J0 ? 1 : 0) + (J1 ? 1 : 0);
      // A fragments one step ahead (two register sets); the B fragments of a slab in ONE set, re-read at the slab boundary — the
      // registers of a second set are what this kernel does not have (256 per wave at two workgroups per CU), and the read's
      // latency falls into the partner wave's products (two waves per SIMD share the matrix pipe)
      s16x8 bv[TN][NPL], av[2][NPL];
      auto read_b = [&](int slab) {
#pragma unroll
        for (int pl = 0; pl < NPL; pl++) {
          if (J0) bv[0][pl] = *reinterpret_cast<const s16x8*>(bh_rd[0] + pl * B_PLANE + 8 * (gsw ^ (2 * slab)));
          if (J1) bv[1][pl] = *reinterpret_cast<const s16x8*>(bh_rd[1] + pl * B_PLANE + 8 * (gsw ^ (2 * slab)));
        }
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
        if (step + 1 < 2 * TM) read_a(step + 1);
#pragma unroll
        for (int t2 = 0; t2 < NT; t2++) {
          if (J0) mfma_terms<NPL, false>(av[step & 1], bv[0], acc[i][0], t2);
          if (J1) mfma_terms<NPL, false>(av[step & 1], bv[1], acc[i][1], t2);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (step + 1 < 2 * TM && (step + 1) / TM != slab) read_b(slab + 1);
      }
      static_assert(NJ >= 1, "a class to multiply");
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    // the next tile's loads first, in the code all three product variants share (issued inside the variants their destination
    // registers differ per variant and every K tile ends in ~100 register copies); two waves per SIMD: the partner's products cover
    // the issue time
#pragma unroll
    for (int s = 0; s < NPIECE; s++) piece(s);
    __builtin_amdgcn_sched_barrier(0);
    const bool j0 = (act_j[0] & vbit) != 0, j1 = (act_j[1] & vbit) != 0;
    if (j0 && j1) body(T_{}, T_{});
    else if (j0) body(T_{}, F_{});
    else if (j1) body(F_{}, T_{});
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();   // every wave is done with this tile's weights (and, at a chunk end, with the halo)
    store_b();
    if (new_chunk) store_h();
    __syncthreads();
    v = ld_v;                        // the tile just stored is the next one multiplied
    vx++;
    if (vx == g.vtx) { vx = 0; vy++; }
    if (v == 0) { vx = 0; vy = 0; }
    ld_advance(kk + 1);
  }

  // ---- epilogue: bias / leaky-ReLU / leaky derivative, fp32 result + output planes, or the split-K partial; class j of the
  // wave goes to destination pixel pix + py * Wd + px.  Through wave-private staging (pl_gather_epilogue's vector form): the
  // accumulator layout (lane = column, 16 scattered rows) becomes lane = (row, 8 consecutive channels).
  {
    constexpr int EP = CF_CN + 4;
    float* stg = reinterpret_cast<float*>(smem16) + wid * (32 * EP);
    const bool to_partial = p.nsplit > 1;
    const size_t npix_d = (size_t)p.B * p.Hd * p.Wd;
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int coff = (cls_j[j] >> 1) * p.Wd + (cls_j[j] & 1);
#pragma unroll
      for (int i = 0; i < TM; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * EP + l31] = acc[i][j][r];
        if (!to_partial) {
          constexpr int OPR = CF_CN / 8, RPI8 = 64 / OPR;      // 4 lanes per row, 16 rows per pass
#pragma unroll
          for (int it = 0; it < 32 / RPI8; it++) {
            const int rr = it * RPI8 + lane / OPR, q = lane % OPR;
            const float4 v0 = *reinterpret_cast<const float4*>(stg + rr * EP + 8 * q);
            const float4 v1 = *reinterpret_cast<const float4*>(stg + rr * EP + 8 * q + 4);
            const int px = pix[wm * WM + i * 32 + rr];
            const int n = n0 + 8 * q;
            if (px < 0 || n >= p.N) continue;
            if (n + 8 <= p.N) epi_store8(p, (size_t)(px + coff), n, v0, v1);
            else epi_store4(p, (size_t)(px + coff), n, v0);
          }
        } else {
          constexpr int QPR = CF_CN / 4, RPI = 64 / QPR;       // 8 lanes per row, 8 rows per pass
#pragma unroll
          for (int it = 0; it < 32 / RPI; it++) {
            const int rr = it * RPI + lane / QPR, q = lane % QPR;
            const float4 v = *reinterpret_cast<const float4*>(stg + rr * EP + 4 * q);
            const int px = pix[wm * WM + i * 32 + rr];
            const int n = n0 + 4 * q;
            if (px < 0 || n >= p.N) continue;
            *reinterpret_cast<float4*>(p.partial + ((size_t)split * npix_d + (size_t)(px + coff)) * p.N + n) = v;
          }
        }
      }
    }
  }
}

}  // namespace

namespace igemm {

// eligible: four output-parity classes (stride-2 data gradient, conv_transpose forward), bf16 x 3 planes, tiles that fit
bool pl_halo_cf_ok(const GatherGeom& p, int npl) {
  const int o = unflow::options().halo_cf;
  if (o <= 0 || npl != 3 || p.N % CF_CN != 0 || p.Hg < 2 * TH || p.Wg < TW) return false;
  CfGeom g{};
  if (!cf_geom(p, g) || cf_halo_pixels(g) > 256 || 2 * cf_smem(g) > 160 * 1024) return false;
  return true;
}

// K split (whole chunks) so that the launch is about one round of 2 workgroups per CU
int plan_pl_halo_cf(const GatherGeom& p) {
  CfGeom g{};
  cf_geom(p, g);
  const long blocks = (long)p.B * cdiv(p.Hg, TH) * cdiv(p.Wg, TW) * (p.N / CF_CN);
  const int nchunk = ((p.Cs >> 3) + 3) >> 2;
  const int max_by_k = max(1, min(16, nchunk * g.vty * g.vtx / 16));      // >= 16 K tiles per split
  return min(nchunk, fill_one_round(blocks, 512, max_by_k));
}

int launch_pl_halo_cf(const PlGatherParams& p, hipStream_t st) {
  CfGeom g{};
  if (!cf_geom(p, g)) return UNFLOW_ERR_UNSUPPORTED;
  const int hp = cf_halo_pixels(g);
  const int smem = cf_smem(g);
  static int smem_set = 0;      // grow-only: benign race, idempotent
  if (smem > smem_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_pl_halo_cf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    smem_set = smem;
  }
  PlGatherParams q = p;
  q.mt = p.B * p.tiles_y * p.tiles_x; q.nt = p.N / CF_CN;
  q.fused_splitk = 0; q.counters = nullptr;
  int mt = q.mt;
  if (q.order == 2) {
    q.mgroup = q.mt >= 16 ? cdiv(q.mt, 8) : q.mt;
    mt = cdiv(q.mt, q.mgroup) * q.mgroup;
  }
  const int grid = mt * q.nt * q.nsplit;
  igemm_pl_halo_cf_kernel<<<grid, 256, smem, st>>>(q, g, hp);
  return launch_status();
}

}  // namespace igemm
