// Persistent stream-K form of the halo implicit-GEMM kernel (conv_planes.hip: igemm_pl_halo_pp_kernel) for the source-stride-1
// layers of FlowNetC/S (src/e2eflow/core/flownet.py:89-237: 3x3 stride-1 convs, every conv data gradient incl. the four
// output-parity classes of stride 2, conv_transpose forward).
//
// One 512-thread workgroup per CU for the whole launch.  The launch's work is the list of ITEMS (tap class, N tile, pair of
// 4 x 32-site M tiles), each a K loop of nchunk 32-channel chunks x ntaps(class) K32 tiles.  The list, laid end to end in K
// tiles, is cut into G equal contiguous ranges (+- one K tile: sk_pos) — a workgroup walks its range segment by segment:
//     a segment that covers a whole item           -> the ordinary epilogue;
//     a segment that does NOT start the item       -> the accumulators go to this workgroup's slab (write-through stores),
//                                                     then its arrival flag is raised; nothing is waited for;
//     a segment that starts an item but not ends it -> (always the LAST segment of a workgroup) the accumulators stay in
//                                                     registers; the slabs of the workgroups that hold the rest of the item
//                                                     (w + 1, w + 2, ...: each computed that segment FIRST) are added in
//                                                     workgroup order, then the ordinary epilogue runs.
// So: no split-K reduce launch, no partial sums for the part of an item its finishing workgroup computed itself, every
// workgroup runs the same number of K tiles (+- one) whatever the tile count or the tap counts of the parity classes
// (4 / 2 / 2 / 1 taps for a stride-2 3 x 3 data gradient), and the result is a fixed-order sum: bit-identical run to run.
// Forward progress: only item-starting segments wait, and only for workgroups with a HIGHER logical id, whose awaited
// segment is their first and waits for nothing; the grid is <= the CU count (one workgroup per CU by LDS), so every
// workgroup becomes resident without any other one having to finish.  Every spin is bounded (g_sk_timeouts).
// Hand-off (cdna_hip_programming.md §5 / Guideline 16, write-through form): 16-byte sc1 stores -> s_waitcnt vmcnt(0) in
// every wave -> workgroup barrier -> relaxed agent-scope flag store; the reader polls relaxed, passes a barrier, and reads
// the slab with sc1 loads (served by L2 / fabric, never a stale L1 line).  The flags clean themselves (g_sk_flags below).
#include <mutex>

#include "planes_shared.h"

namespace {
using namespace igemm;

__device__ int g_sk_timeouts;      // spins that gave up (a result is then wrong; the tests assert 0)

// Arrival flags.  Round 6: they live in the library's own device memory instead of the caller's workspace and CLEAN THEMSELVES —
// every raised flag has exactly one waiter (the workgroup that started the item), which resets it once it has passed — so a
// launch finds them zero (module load zero-initialises them) and the memset node in front of every stream-K launch (5 per
// training step, 5-6 us each on the critical path) is gone.  One row of flags per WORKSPACE: launches that share a workspace
// share its slabs and are serialised by their caller anyway, so keying the flags by the workspace pointer adds no constraint
// (host table sk_flag_slot; more than SK_FLAG_SLOTS distinct workspaces or G > SK_MAX_G: the old way, flags behind the slabs
// + memset).  A spin that timed out leaves its flag raised: unflow_debug_streamk_timeouts() — which every caller of this path
// polls, and whose non-zero result is fatal — clears the flags along with the counter.
constexpr int SK_FLAG_SLOTS = 64, SK_MAX_G = 512;
__device__ int g_sk_flags[SK_FLAG_SLOTS][SK_MAX_G];

struct SkPlan {
  int G;                   // persistent workgroups
  int ncls;                // tap classes
  int nt;                  // N tiles
  int mtp;                 // M tile pairs
  int ng, mg, mlast;       // the M tile pairs are cut into ng groups of mg (the last one: mlast) — one group ~ one XCD's share
  int nchunk;              // 32-channel chunks per item
  int ntaps[4];            // K tiles per chunk, by class
  unsigned wfull, wall;    // K tiles of a full group / of the launch
  int* flags;              // [G] arrival flags of the slabs (workspace form), or nullptr: row `slot` of g_sk_flags
  int slot;
  float* slabs;            // [G][2 instances][4 waves][16 float4 rows][64 lanes] float4
};

constexpr int SK_SLAB_BYTES = 2 * 4 * 16 * 64 * 16;       // 128 KB per workgroup
constexpr int SK_SPIN_LIMIT = 1 << 22;

// The item list, in the order the workgroups walk it: M group (slowest) > tap class > N tile > M tile pair of the group.
// The workgroups of an XCD hold a contiguous eighth of the list = (about) one M group with ALL its classes and N tiles: the
// parity classes of a tile read the same source pixels and write interleaved pixels of the same destination rows, so they
// belong on one L2 at one time (class-major over the whole launch cost the stride-2 data gradients 10 %).
// First K-tile position of workgroup w's range (w = G: the end of the list).
__host__ __device__ __forceinline__ unsigned sk_pos(const SkPlan& s, int w) {
  return w >= s.G ? s.wall : (unsigned)(((unsigned long long)s.wall * (unsigned)w) / (unsigned)s.G);
}

// K-tile position -> the item it lies in (class, N tile, M tile pair) and its K tile inside that item
__host__ __device__ __forceinline__ void sk_decode(const SkPlan& s, unsigned pos, int& cls, int& ntile, int& mp, int& k) {
  unsigned g = pos / s.wfull;
  if (g > (unsigned)(s.ng - 1)) g = (unsigned)(s.ng - 1);
  unsigned rem = pos - g * s.wfull;
  const unsigned sz = (unsigned)(g == (unsigned)(s.ng - 1) ? s.mlast : s.mg);
  const unsigned per_cls = sz * (unsigned)s.nt;                                   // items of one class in this group
  int c = 0;
  while (c + 1 < s.ncls && rem >= per_cls * (unsigned)(s.nchunk * s.ntaps[c])) {
    rem -= per_cls * (unsigned)(s.nchunk * s.ntaps[c]);
    c++;
  }
  const unsigned per_item = (unsigned)(s.nchunk * s.ntaps[c]);
  const unsigned it = rem / per_item;
  const unsigned nti = it / sz;
  cls = c;
  ntile = (int)nti;
  mp = (int)(g * (unsigned)s.mg + (it - nti * sz));
  k = (int)(rem - it * per_item);
}

// host: fill the partition fields (G, ncls, nt, mtp, nchunk, ntaps given); false: outside the 31-bit arithmetic of sk_pos
static bool sk_fill(SkPlan& s, int ngroups) {
  if (ngroups < 1) ngroups = 1;
  s.mg = (s.mtp + ngroups - 1) / ngroups;
  s.ng = (s.mtp + s.mg - 1) / s.mg;
  s.mlast = s.mtp - (s.ng - 1) * s.mg;
  unsigned long long tapsum = 0;
  for (int c = 0; c < s.ncls; c++) tapsum += (unsigned long long)s.ntaps[c];
  const unsigned long long per_pair = (unsigned long long)s.nt * s.nchunk * tapsum;
  const unsigned long long all = per_pair * (unsigned long long)s.mtp;
  if (all == 0 || all * (unsigned long long)s.G >= (1ull << 31)) return false;
  s.wfull = (unsigned)(per_pair * (unsigned long long)s.mg);
  s.wall = (unsigned)all;
  return true;
}

// ACC: the tap classes of the launch ACCUMULATE into one output tile (GatherGeom.acc: forward of a stride-2 conv, data
// gradient of a conv_transpose — four stride-1 problems on the parity sub-lattices of the source, pixel pitch p.sp = 2): an
// item is then (N tile, M tile pair) and its K loop walks class after class (class-major, then chunk, then tap), each class
// with its own halo geometry and tap grid; the plan sees one class of sum-of-taps K tiles per chunk.
template <int NPL, bool F16, bool ACC>
__global__ __launch_bounds__(512, 1) void igemm_pl_halo_sk_kernel(const PlGatherParams p, const SkPlan sk, int HPmax) {
  constexpr int BM = 128, BN = 128, WM = 64, WN = 64;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LDP = 32;
  constexpr int B_PLANE = BN * LDP, B_TILE = NPL * B_PLANE;
  constexpr int NH = 4;
  constexpr int NT = NPL == 3 ? 6 : 1;
  static_assert(NPL == 3 && !F16, "bf16 x 3 only");

  extern __shared__ __attribute__((aligned(16))) unsigned short smem_all[];
  const int inst = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6;
  const int H_PLANE = HPmax * HPITCH;
  unsigned short* Hh = smem_all + inst * NPL * H_PLANE;
  unsigned short* Bbase = smem_all + 2 * NPL * H_PLANE;
  int* pix = reinterpret_cast<int*>(smem_all + 2 * NPL * H_PLANE + 2 * B_TILE) + inst * BM;

  const int wm = wid >> 1, wn = wid & 1;
  const int wg = xcd_remap(blockIdx.x, gridDim.x, p.xcd);          // logical id: an XCD runs a contiguous run of the item list
  const unsigned pos_begin = sk_pos(sk, wg), pos_end = sk_pos(sk, wg + 1);
  if (pos_begin >= pos_end) return;
  int* const flags = sk.flags ? sk.flags : g_sk_flags[sk.slot];

  __amdgpu_buffer_rsrc_t src_rs[NPL], w_rs[NPL];
#pragma unroll
  for (int pl = 0; pl < NPL; pl++) {
    src_rs[pl] = make_rsrc(p.src + pl * p.src_ps, (((size_t)p.B * p.Hs * p.Ws - 1) * (size_t)p.lds + (size_t)p.Cs) * 2);
    w_rs[pl] = make_rsrc(p.w + pl * p.w_ps, (size_t)p.wtaps * p.N * p.Cs * 2);
  }
  const __amdgpu_buffer_rsrc_t slab_rs = make_rsrc(sk.slabs, (size_t)sk.G * SK_SLAB_BYTES);
  const int kq = tid & 3;
  const int lds2 = p.lds * 2;
  const int Cg = p.Cs >> 3;
  const int l31 = lane & 31, lh = lane >> 5;
  const int b_r = 64 * inst + (tid >> 2);       // this instance's half of the weight tile
  const int b_rd = (wn * WN + l31) * LDP;
  const int gsw = lh ^ ((l31 >> 2) & 3);
  const int slab_lane = ((inst * 4 + wid) * 16 * 64 + lane) * 16;       // byte offset of this lane's first float4 in a slab
  auto swz = [](int row, int g) { return row * LDP + 8 * (g ^ ((row >> 2) & 3)); };
  auto slot_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

#pragma unroll 1
  for (unsigned pos = pos_begin; pos < pos_end;) {
    // ---- the segment: K tiles [k0, k1) of item (class, N tile, M tile pair)
    int cls_id, ntile, mpair, k0;
    sk_decode(sk, pos, cls_id, ntile, mpair, k0);
    const int nk = sk.nchunk * sk.ntaps[ACC ? 0 : cls_id];
    const int k1 = min(nk, k0 + (int)(pos_end - pos));
    const unsigned item_end = pos - (unsigned)k0 + (unsigned)nk;      // K-tile position one past this item
    pos += (unsigned)(k1 - k0);
    // the class K tile k0 lies in (ACC: classes follow each other inside the item) and its K tile inside that class
    int cls0 = cls_id, kc0 = k0;
    if (ACC) {
      cls0 = 0;
      for (;;) {
        const int nkc = sk.nchunk * p.cls[cls0].nty * p.cls[cls0].ntx;
        if (kc0 < nkc || cls0 + 1 >= p.ncls) break;
        kc0 -= nkc;
        cls0++;
      }
    }
    int t = 2 * mpair + inst;
    const bool tile_ok = t < p.B * p.tiles_y * p.tiles_x;
    const int n0 = ntile * BN;
    const int T = k1 - k0;
    const int txi = t % p.tiles_x; t /= p.tiles_x;
    const int tyi = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int y0 = tyi * TH, x0 = txi * TW;
    int HC = TW + p.cls[cls0].ntx - 1;           // halo row pitch (pixels); ACC: the widest class's, for every class
    if (ACC)
      for (int c = 0; c < p.ncls; c++) HC = max(HC, TW + p.cls[c].ntx - 1);

    __syncthreads();       // the previous segment's epilogue is done with the LDS staging areas and the pixel table
    TapClass ltc = p.cls[cls0], mtc = ltc;       // the class of the tile being LOADED / MULTIPLIED
    int h_off[NH];
    auto set_load_class = [&]() {
      const int dmy = p.dstep > 0 ? ltc.dy0 : ltc.dy0 - (ltc.nty - 1);
      const int dmx = p.dstep > 0 ? ltc.dx0 : ltc.dx0 - (ltc.ntx - 1);
      const int HRc = TH + ltc.nty - 1, HCc = TW + ltc.ntx - 1;
#pragma unroll
      for (int j = 0; j < NH; j++) {
        const int hp = (tid >> 2) + 64 * j;
        const int hy = hp / HC, hx = hp - hy * HC;
        const int y = (y0 + hy) * p.sp + dmy, x = (x0 + hx) * p.sp + dmx;
        const bool ok = tile_ok && hy < HRc && hx < HCc && (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;
        h_off[j] = ok ? ((b * p.Hs + y) * p.Ws + x) * lds2 + kq * 16 : OOB_MARK;
      }
    };
    set_load_class();
    if (tid < BM) {
      const int yg = y0 + (tid >> TWL), xg = x0 + (tid & (TW - 1));
      pix[tid] = (tile_ok && yg < p.Hg && xg < p.Wg) ? (b * p.Hd + yg * p.so + ltc.py) * p.Wd + xg * p.so + ltc.px : -1;
    }
    const int b_row = n0 + b_r < p.N ? (n0 + b_r) * p.Cs * 2 + kq * 16 : OOB_MARK;

    // walkers over the segment's K tiles (chunk-major, then tap): L = the tile the next load_b() requests, M = the tile multiplied
    const int ntaps0 = ltc.nty * ltc.ntx;
    const int tap0 = kc0 % ntaps0;               // a range may start in the middle of a chunk: its halo is loaded all the same
    int l_cls = cls0, l_chunk = kc0 / ntaps0, l_ty = tap0 / ltc.ntx, l_tx = tap0 % ltc.ntx, l_left = T;
    int m_cls = cls0, m_chunk = l_chunk, m_ty = l_ty, m_tx = l_tx;
    u32x4 rh[NH][NPL], rb[NPL];
    auto load_b = [&]() {
      const int widx = (ltc.ky0 + l_ty * p.kstep) * p.KW + ltc.kx0 + l_tx * p.kstep;
      const bool ok = l_left > 0 && l_chunk * 4 + kq < Cg;
      const int voff = ok ? b_row + widx * p.N * p.Cs * 2 + l_chunk * 64 : OOB_MARK;
#pragma unroll
      for (int pl = 0; pl < NPL; pl++) rb[pl] = buf_ld16(w_rs[pl], voff);
    };
    auto load_h = [&]() {
      const bool okc = l_left > 0 && l_chunk * 4 + kq < Cg;
#pragma unroll
      for (int j = 0; j < NH; j++) {
        const int voff = okc ? h_off[j] + l_chunk * 64 : OOB_MARK;
#pragma unroll
        for (int pl = 0; pl < NPL; pl++) rh[j][pl] = buf_ld16(src_rs[pl], voff);
      }
    };
    auto l_advance = [&]() {
      l_tx++;
      if (l_tx == ltc.ntx) { l_tx = 0; l_ty++; }
      if (l_ty == ltc.nty) { l_ty = 0; l_chunk++; }
      if (ACC && l_chunk == sk.nchunk && l_cls + 1 < p.ncls) {      // next class: its own halo geometry and tap grid
        l_chunk = 0;
        l_cls++;
        ltc = p.cls[l_cls];
        set_load_class();
      }
      l_left--;
    };
    auto store_b = [&](int buf) {
#pragma unroll
      for (int pl = 0; pl < NPL; pl++) *reinterpret_cast<u32x4*>(Bbase + buf * B_TILE + pl * B_PLANE + swz(b_r, kq)) = rb[pl];
    };
    auto store_h = [&]() {
#pragma unroll
      for (int j = 0; j < NH; j++) {
        const int hp = (tid >> 2) + 64 * j;
        if (hp < HPmax) {
#pragma unroll
          for (int pl = 0; pl < NPL; pl++) *reinterpret_cast<u32x4*>(Hh + pl * H_PLANE + hp * HPITCH + kq * 8) = rh[j][pl];
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

    int a_rd[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) {
      const int sidx = wm * WM + i * 32 + l31;
      a_rd[i] = ((sidx >> TWL) * HC + (sidx & (TW - 1))) * HPITCH + lh * 8;
    }

    s16x8 av[TM][NPL], bv[TN][NPL];
    auto read_frags = [&](int buf, int slab) {
      const int c_hy0 = p.dstep > 0 ? 0 : mtc.nty - 1, c_hx0 = p.dstep > 0 ? 0 : mtc.ntx - 1;
      const int tapoff = ((c_hy0 + m_ty * p.dstep) * HC + c_hx0 + m_tx * p.dstep) * HPITCH;
#pragma unroll
      for (int pl = 0; pl < NPL; pl++) {
#pragma unroll
        for (int j = 0; j < TN; j++)
          bv[j][pl] = *reinterpret_cast<const s16x8*>(Bbase + buf * B_TILE + pl * B_PLANE + b_rd + j * 32 * LDP + 8 * (gsw ^ (2 * slab)));
#pragma unroll
        for (int i = 0; i < TM; i++)
          av[i][pl] = *reinterpret_cast<const s16x8*>(Hh + pl * H_PLANE + a_rd[i] + tapoff + 16 * slab);
      }
    };
    auto multiply = [&]() {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int tt = 0; tt < NT; tt++)
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) mfma_terms<NPL, F16>(av[i], bv[j], acc[i][j], tt);
      __builtin_amdgcn_s_setprio(0);
    };

    // ---- K loop of the segment (the slot schedule of igemm_pl_halo_pp_kernel; T > 0 always)
    load_h();
    load_b();
    l_advance();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_h();
    store_b(0);
    bool h_pending = l_left > 0 && l_ty == 0 && l_tx == 0;
    if (h_pending) load_h();
    load_b();
    l_advance();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    slot_end();
    if (inst == 1) slot_end();
#pragma unroll 1
    for (int kk = 0; kk < T; kk++) {
      // read slot 0: slab 0 of tile kk; tile kk + 1's weights -> the other buffer; request tile kk + 2
      read_frags(kk & 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      store_b((kk + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      const bool store_halo_now = h_pending;
      const bool next_opens = l_left > 0 && l_ty == 0 && l_tx == 0;
      load_b();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      slot_end();
      multiply();                                               // slab 0
      slot_end();
      // read slot 1: slab 1 of tile kk
      read_frags(kk & 1, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // every halo read of this tile has returned
      slot_end();
      if (store_halo_now) {                                     // the old chunk is done: its successor's halo -> LDS
        store_h();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      h_pending = false;
      if (next_opens) {
        load_h();
        h_pending = true;
      }
      l_advance();
      __builtin_amdgcn_sched_barrier(0);
      multiply();                                               // slab 1
      slot_end();
      m_tx++;
      if (m_tx == mtc.ntx) { m_tx = 0; m_ty++; }
      if (m_ty == mtc.nty) { m_ty = 0; m_chunk++; }
      if (ACC && m_chunk == sk.nchunk && m_cls + 1 < p.ncls) {
        m_chunk = 0;
        m_cls++;
        mtc = p.cls[m_cls];
      }
    }
    if (inst == 0) slot_end();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the zero loads past the last tile
    __syncthreads();                                            // both instances are done with the LDS tiles

    if (k0 > 0) {
      // ---- not the start of its item: accumulators -> this workgroup's slab, raise the flag, go on
      const int soff = wg * SK_SLAB_BYTES;
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
          for (int r4 = 0; r4 < 4; r4++) {
            u32x4 v;
            v.x = __float_as_uint(acc[i][j][4 * r4]); v.y = __float_as_uint(acc[i][j][4 * r4 + 1]);
            v.z = __float_as_uint(acc[i][j][4 * r4 + 2]); v.w = __float_as_uint(acc[i][j][4 * r4 + 3]);
            buf_st16_held<AUX_SC1>(v, slab_rs, slab_lane + (((i * TN + j) * 4 + r4) * 64) * 16, soff);
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(flags + wg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    if (k1 < nk) {
      // ---- starts its item but does not end it: add the slabs of the workgroups that hold the rest, in workgroup order
#pragma unroll 1
      for (int w2 = wg + 1; w2 < sk.G; w2++) {
        const unsigned pa = sk_pos(sk, w2);
        if (pa >= item_end) break;
        if (sk_pos(sk, w2 + 1) <= pa) continue;                  // an empty range: holds nothing
        if (threadIdx.x == 0) {
          int spins = 0;
          while (__hip_atomic_load(flags + w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > SK_SPIN_LIMIT) { atomicAdd(&g_sk_timeouts, 1); break; }
          }
          // the one waiter of this flag has passed: lower it for the next launch on this workspace (nobody raises it again in
          // this one: a workgroup writes at most one slab per launch)
          if (spins <= SK_SPIN_LIMIT) __hip_atomic_store(flags + w2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const int soff = w2 * SK_SLAB_BYTES;
        u32x4 v[TM * TN * 4];
#pragma unroll
        for (int q = 0; q < TM * TN * 4; q++) v[q] = __builtin_amdgcn_raw_buffer_load_b128(slab_rs, slab_lane + q * 64 * 16, soff, AUX_SC1);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
              const u32x4 t4 = v[(i * TN + j) * 4 + r4];
              acc[i][j][4 * r4] += __uint_as_float(t4.x); acc[i][j][4 * r4 + 1] += __uint_as_float(t4.y);
              acc[i][j][4 * r4 + 2] += __uint_as_float(t4.z); acc[i][j][4 * r4 + 3] += __uint_as_float(t4.w);
            }
      }
    }
    (void)m_chunk;
    if (tile_ok) pl_gather_epilogue<WM, WN>(p, acc, pix, smem_all + inst * (4 * 32 * (WN + 4) * 2), wm, wn, wid, lane, n0, 0);
  }
}


}  // namespace

namespace igemm {

// LDS of the stream-K halo kernel = the ping-pong halo kernel's: two private halos, the shared double-buffered weight tile, two pixel tables
int pl_halo_sk_smem(const GatherGeom& p) { return 2 * 3 * pl_halo_pixels(p) * HPITCH * 2 + 2 * 3 * 128 * 32 * 2 + 2 * 128 * 4; }

static int sk_workgroups() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

size_t pl_halo_sk_ws_bytes() { return (size_t)sk_workgroups() * SK_SLAB_BYTES + (size_t)sk_workgroups() * sizeof(int) + 256; }

static void sk_plan_of(const GatherGeom& p, SkPlan& sk) {
  sk.G = sk_workgroups();
  sk.nt = cdiv(p.N, 128);
  sk.mtp = cdiv((long)p.B * cdiv(p.Hg, TH) * cdiv(p.Wg, TW), 2);
  sk.nchunk = ((p.Cs >> 3) + 3) >> 2;
  for (int c = 0; c < 4; c++) sk.ntaps[c] = c < p.ncls ? p.cls[c].nty * p.cls[c].ntx : 1;
  sk.ncls = p.ncls;
  if (p.acc) {                                 // accumulating classes: one item walks them all
    int sum = 0;
    for (int c = 0; c < p.ncls; c++) sum += sk.ntaps[c];
    sk.ncls = 1;
    sk.ntaps[0] = sum;
  }
}

bool pl_halo_sk_ok(const GatherGeom& p, int npl, int bn) {
  const int o = unflow::options().streamk;
  if (o <= 0 || npl != 3 || bn != 128 || (p.acc && p.dstep != 1) || pl_halo_sk_smem(p) > 160 * 1024) return false;
  SkPlan sk{};
  sk_plan_of(p, sk);
  if (!sk_fill(sk, unflow::options().streamk_groups)) return false;
  // Where it pays (MI355X, FlowNetC 384x512 B=4, profiles/r04_streamk_per_layer.txt): items with LONG K loops, walked by one
  // output tile — conv3_1 forward 288 -> 252 us, conv4_1 forward / data gradient 173 -> 153 / 173 -> 158, conv3_1 data
  // gradient 281 -> 272, conv3 forward (four accumulating classes, 100 K tiles) 237 -> 208.  It loses where an item is short:
  // conv2 forward (50 K tiles) 223 -> 236, deconv2's data gradient (32) 131 -> 143, and on the output-parity classes of the
  // stride-2 data gradients / conv_transpose forward (16 .. 72 K tiles per item; conv4's data gradient 129 -> 150 us, conv3's
  // 245 -> 277).  There a one-shot launch keeps two 4-wave workgroups per CU, so one's prologue / epilogue hides under the
  // other's K loop, and its uneven blocks balance themselves over 512 slots without any partial sums; one 8-wave workgroup per
  // CU exposes every segment boundary (2.5 - 3 per workgroup there).  Option streamk = 2 forces it (tests).
  int min_nk = 1 << 30;
  for (int c = 0; c < sk.ncls; c++) min_nk = min(min_nk, sk.nchunk * sk.ntaps[c]);
  return o >= 2 || (sk.ncls == 1 && min_nk >= 64 && sk.wall >= 48u * (unsigned)sk.G);
}

// row of g_sk_flags that belongs to workspace `ws` on the current device (first come, first served; -1: table full)
static int sk_flag_slot(const void* ws) {
  static std::mutex mu;
  static const void* owner[UNFLOW_MAX_DEVICES][SK_FLAG_SLOTS];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= UNFLOW_MAX_DEVICES) return -1;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < SK_FLAG_SLOTS; i++) {
    if (owner[dev][i] == ws) return i;
    if (!owner[dev][i]) { owner[dev][i] = ws; return i; }
  }
  return -1;
}

int launch_pl_halo_sk(const PlGatherParams& p, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!ws || ws_bytes < pl_halo_sk_ws_bytes()) return UNFLOW_ERR_WORKSPACE;
  const int hp = pl_halo_pixels(p);
  const int smem = pl_halo_sk_smem(p);
  static DynLdsBook book_a{}, book_b{};
  (void)ensure_dyn_lds(reinterpret_cast<const void*>(&igemm_pl_halo_sk_kernel<3, false, false>), smem, book_a);
  (void)ensure_dyn_lds(reinterpret_cast<const void*>(&igemm_pl_halo_sk_kernel<3, false, true>), smem, book_b);
  PlGatherParams q = p;
  q.nsplit = 1; q.partial = nullptr;
  SkPlan sk{};
  sk_plan_of(p, sk);
  if (!sk_fill(sk, unflow::options().streamk_groups)) return UNFLOW_ERR_UNSUPPORTED;
  sk.slabs = reinterpret_cast<float*>(ws);
  sk.slot = sk.G <= SK_MAX_G ? sk_flag_slot(ws) : -1;
  sk.flags = nullptr;
  if (sk.slot < 0) {                             // no row left for this workspace: flags behind the slabs, zeroed per launch
    sk.slot = 0;
    sk.flags = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + (size_t)sk.G * SK_SLAB_BYTES);
    if (hipMemsetAsync(sk.flags, 0, (size_t)sk.G * sizeof(int), st) != hipSuccess) return UNFLOW_ERR_LAUNCH;
  }
  if (p.acc) igemm_pl_halo_sk_kernel<3, false, true><<<sk.G, 512, smem, st>>>(q, sk, hp);
  else igemm_pl_halo_sk_kernel<3, false, false><<<sk.G, 512, smem, st>>>(q, sk, hp);
  return launch_status();
}

}  // namespace igemm

UNFLOW_API int unflow_debug_streamk_plan(int G, int mtp, int nt, int nchunk, int ncls, const int* ntaps, int ngroups, int* range_pos,
                                         int npos, const int* pos, int* decoded) {
  if (G <= 0 || mtp <= 0 || nt <= 0 || nchunk <= 0 || ncls < 1 || ncls > 4 || !ntaps || !range_pos) return UNFLOW_ERR_SHAPE;
  SkPlan sk{};
  sk.G = G; sk.ncls = ncls; sk.nt = nt; sk.mtp = mtp; sk.nchunk = nchunk;
  for (int c = 0; c < 4; c++) {
    sk.ntaps[c] = c < ncls ? ntaps[c] : 1;
    if (sk.ntaps[c] <= 0) return UNFLOW_ERR_SHAPE;
  }
  if (!sk_fill(sk, ngroups)) return UNFLOW_ERR_SHAPE;
  for (int w = 0; w <= G; w++) range_pos[w] = (int)sk_pos(sk, w);
  for (int i = 0; i < npos && pos && decoded; i++) {
    if (pos[i] < 0 || (unsigned)pos[i] >= sk.wall) return UNFLOW_ERR_SHAPE;
    sk_decode(sk, (unsigned)pos[i], decoded[4 * i], decoded[4 * i + 1], decoded[4 * i + 2], decoded[4 * i + 3]);
  }
  return UNFLOW_OK;
}

// spins of the stream-K kernels that hit their bound since the last call (tests: must be 0)
UNFLOW_API int unflow_debug_streamk_timeouts(void) {
  int v = 0, zero = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_sk_timeouts), sizeof(int)) != hipSuccess) return -1;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sk_timeouts), &zero, sizeof(int));
  if (v != 0) {                                  // a waiter gave up: its flag may still be raised (or be raised later)
    static const int zeros[SK_FLAG_SLOTS * SK_MAX_G] = {0};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sk_flags), zeros, sizeof(zeros));
  }
  return v;
}
