// conv / conv_transpose stacks of FlowNetC/S (src/e2eflow/core/flownet.py:89-237) as implicit GEMMs for gfx950,
// channels-last — the kernels that take fp32 tensors as operands.  The default path of the training step is
// csrc/conv_planes.hip (pre-split 16-bit operand planes); this file is what runs with UNFLOW_CONV_MATH=fp32 and for
// operands without planes (the plain fp32-tensor entry points of the C ABI — the seam a TF op shell binds, INTEGRATION.md
// 2b — and unaligned channel counts), and it holds the
// kernels of the layers that are no GEMM at all: the Cout = 2 flow heads (head3_*), the 2 -> 2 flow upsamplers
// (tiny_deconv_*), conv_redir's data gradient (pointwise32) and the batched bias-gradient column sums.
//
// One "gather GEMM" kernel covers conv fwd, conv dgrad, deconv fwd and deconv dgrad (geometry: igemm_shared.h):
//   D[site, n] = sum_{tap} sum_{c} SRC[b, yg*sm + dy(tap), xg*sm + dx(tap), c] * W(tap, c, n)
// with the rows (sites) a regular grid, out-of-image taps contributing zero (TF 'SAME'), and the stride-2 transposed
// cases split into the 4 output-parity classes (each a dense stride-1 gather with its own tap subset) — no
// zero-insertion, no im2col buffer, no col2im.  One "wgrad" kernel covers conv and deconv filter gradients:
//   dW[(tap,a), b] = sum_{site} SRC[gather(site, tap), a] * DST[site, b]
// K is the flattened (tap, channel) axis walked in 16-byte quads.
//
// Math (template MATH): 0 = v_mfma_f32_32x32x2_f32 (exact fp32 products, 157 TFLOP/s peak), operands in LDS as fp32
// [row][K+4]; 1 = the 3-way bf16 split done WHILE staging (split_store: fp32 -> hi/mid/lo planes in LDS, six
// v_mfma_f32_32x32x16_bf16 terms) — the round-1 default, superseded by the pre-split planes of conv_planes.hip, which take
// the split and two thirds of the LDS store traffic out of the K loop.
// Tiling: 256 threads = 4 waves, K-tiles of 32, register-staged double buffering, one barrier per K-tile; split-K for
// the deep layers with partials in a caller workspace and a fixed-order reduce that applies the epilogue (deterministic,
// no float atomics).
#include "igemm_shared.h"
#include "options.h"

namespace {
using namespace igemm;

constexpr int LDK = BK + 4;

struct GatherParams : GatherGeom {
  const float* src;
  const float* w;
  const float* bias;
  float* dst;
  float* partial;  // split-K partials [nsplit][dst pixels][N] (nsplit > 1)
  const float* act_src;
  int lds, ldd, ld_act, act_lo, act_hi;
  int nsplit;
  int leaky, accumulate;
  unsigned cs_magic;  // ceil(2^32 / Cs)
  PlaneOut pl;        // optional 16-bit operand planes of the output (conv_planes.hip consumers)
};

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---- fp32-equivalent products on the bf16 matrix cores (MATH == 1) -------------------------------------------------------
// x = hi + mid + lo with three bf16 values (3 x 8 significand bits = the 24 of fp32, same exponent range); a*b is summed
// from the six terms hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid on v_mfma_f32_32x32x16_bf16 (fp32 accumulation); the
// dropped terms are <= 2^-23 |a||b|.  Measured (tools/microbench/bf16x3_accuracy.hip, K = 4608): rms error 2.5e-8 of
// sum|a||b| against 2.8e-8 for v_mfma_f32_32x32x2_f32 — the same accuracy class, at 1/16 of the matrix-core time per
// product term.  The split happens once per element while the tile is staged into LDS (v_cvt_pk_bf16_f32, gfx950).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr bool kEarlyLoads = true;   // load pieces go out after the FIRST term groups of a tile (measured better than evenly spread)
constexpr int LDH = BK;       // bf16 row pitch of one plane: 64 bytes, no padding (49 KB per 128x128 block -> 3 blocks per CU)
// ... made conflict-free by an XOR swizzle of the 16-byte granule index with bits 2..3 of the row: the 16 lanes a
// ds_read_b128 serves per cycle (rows r .. r+15, same logical granule) then touch 16 different granule slots of 256 bytes.
__device__ __forceinline__ int swz_off(int row, int kquad) {   // offset (in bf16 elements) of k = 4*kquad of `row`
  return row * LDH + 8 * ((kquad >> 1) ^ ((row >> 2) & 3)) + 4 * (kquad & 1);
}
// four consecutive k of one row -> three planes, 8 bytes each
__device__ __forceinline__ void split_store(unsigned short* __restrict__ dst, int plane_stride, const float4 v) {
  const unsigned h0 = cvt_pk_bf16(v.x, v.y), h1 = cvt_pk_bf16(v.z, v.w);
  const float r0 = v.x - __uint_as_float(h0 << 16), r1 = v.y - __uint_as_float(h0 & 0xffff0000u);
  const float r2 = v.z - __uint_as_float(h1 << 16), r3 = v.w - __uint_as_float(h1 & 0xffff0000u);
  const unsigned m0 = cvt_pk_bf16(r0, r1), m1 = cvt_pk_bf16(r2, r3);
  const float s0 = r0 - __uint_as_float(m0 << 16), s1 = r1 - __uint_as_float(m0 & 0xffff0000u);
  const float s2 = r2 - __uint_as_float(m1 << 16), s3 = r3 - __uint_as_float(m1 & 0xffff0000u);
  *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(dst + plane_stride) = make_uint2(m0, m1);
  *reinterpret_cast<uint2*>(dst + 2 * plane_stride) = make_uint2(cvt_pk_bf16(s0, s1), cvt_pk_bf16(s2, s3));
}

// LDS operand layout: [row][LDK] with LDK = 36 floats (16-byte aligned rows, conflict-free for the
// 128-bit reads and writes used below).  K order inside a tile is permuted: wave half lh (lanes 32*lh..)
// supplies k = 16*lh + s at MFMA step s, so one ds_read_b128 feeds 4 consecutive steps.  Both operands
// use the same permutation, so the sum over k is unchanged (only its fp32 order).
//
// Schedule: the global loads of tile t+1 (address arithmetic, bounds predicates, loads into registers) are
// cut into pieces that sit BETWEEN the MFMA groups of tile t, fenced with sched_barrier so they stay
// there: an in-order wave then always has an MFMA within a few instructions, instead of a ~300
// instruction load prologue during which its SIMD's matrix pipe idles (and co-resident waves lock-step).
// All predication is by address select + value select — no divergent branches in the loop.
template <int BM, int BN, int WM, int WN, bool B_NK, int MATH = 0>
__global__ __launch_bounds__(256) void igemm_gather_kernel(const GatherParams p) {
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves");
  constexpr int A_ELEMS = BM * LDK;
  constexpr int A_PLANE = BM * LDH, B_PLANE = BN * LDH;   // MATH == 1: bf16 planes [3][rows][LDH]
  constexpr int AR = BM / 32;                 // A rows per thread (quad column kq fixed)
  constexpr int BR_NK = BN / 32;              // B rows per thread, K-contiguous weights
  constexpr int KG = 256 / BN;                // KN weights: thread = (n, k-group); k-quads kq = kg + KG*i
  constexpr int BQ_KN = 8 / KG;               // quads per thread (KN)
  constexpr int NB = B_NK ? BR_NK : BQ_KN;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + A_ELEMS;
  unsigned short* Ah = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Bh = Ah + 3 * A_PLANE;
  int* pix = MATH ? reinterpret_cast<int*>(Bh + 3 * B_PLANE) : reinterpret_cast<int*>(smem + A_ELEMS + BN * LDK);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int cls_id = blockIdx.z % p.ncls, split = blockIdx.z / p.ncls;
  const TapClass tc = p.cls[cls_id];
  const int M = p.B * p.Hg * p.Wg;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int ntaps = tc.nty * tc.ntx;
  const int Cq = p.Cs >> 2;
  const int Ktot = ntaps * p.Cs;
  const int KT = (Ktot + BK - 1) / BK;
  const int kt_per = (KT + p.nsplit - 1) / p.nsplit;
  const int kt0 = split * kt_per;
  const int kt1 = min(KT, kt0 + kt_per);

  const __amdgpu_buffer_rsrc_t src_rs =
      make_rsrc(p.src, ((size_t)p.B * p.Hs * p.Ws - 1) * (size_t)p.lds * 4 + (size_t)p.Cs * 4);
  const __amdgpu_buffer_rsrc_t w_rs = make_rsrc(p.w, (size_t)p.wtaps * p.Cs * p.N * 4);

  // ---- per-thread A row decode (fixed for the whole K loop)
  const int kq = tid & 7;
  int a_y[AR], a_x[AR], a_lin[AR];
#pragma unroll
  for (int i = 0; i < AR; i++) {
    const int m = m0 + (tid >> 3) + 32 * i;
    if (m < M) {
      const int xg = m % p.Wg, t = m / p.Wg;
      const int yg = t % p.Hg, b = t / p.Hg;
      a_y[i] = yg * p.sm;
      a_x[i] = xg * p.sm;
      a_lin[i] = b * p.Hs * p.Ws + a_y[i] * p.Ws + a_x[i];
    } else {
      a_y[i] = -(1 << 28);  // forces out-of-bounds
      a_x[i] = 0;
      a_lin[i] = 0;
    }
  }
  if (tid < BM) {
    const int m = m0 + tid;
    int v = -1;
    if (m < M) {
      const int xg = m % p.Wg, t = m / p.Wg;
      const int yg = t % p.Hg, b = t / p.Hg;
      v = (b * p.Hd + yg * p.so + tc.py) * p.Wd + xg * p.so + tc.px;
    }
    pix[tid] = v;
  }
  // weight-side per-thread constants
  int b_row[NB];   // NK: byte offset of row n (or mark); KN: unused
  int kn_voff = 0; // KN: byte offset of column n (or mark)
  int kn_kg = 0;   // KN: k-group of this wave (wave-uniform)
  if constexpr (B_NK) {
#pragma unroll
    for (int i = 0; i < NB; i++) {
      const int n = n0 + (tid >> 3) + 32 * i;
      b_row[i] = n < p.N ? n * p.Cs * 4 : OOB_MARK;
    }
  } else {
    const int n = n0 + (tid % BN);
    kn_voff = n < p.N ? n * 4 : OOB_MARK;
    kn_kg = __builtin_amdgcn_readfirstlane(tid / BN);
  }
  const int lds4 = p.lds * 4;

  // ---- K walker: (tap, quad-in-tap) of the 16-byte column this thread loads; advanced by 8 quads per tile
  // with selects only.  8 = adv_t * Cq + adv_c.
  const int adv_t = 8 / Cq, adv_c = 8 - adv_t * Cq;
  const unsigned ntx_magic = (65536u + (unsigned)tc.ntx - 1u) / (unsigned)tc.ntx;  // exact for tap < 2^10
  int q_tap, q_c4;
  {
    const unsigned q = (unsigned)kt0 * 8u + (unsigned)kq;
    q_tap = (int)(q / (unsigned)Cq);
    q_c4 = (int)(q - (unsigned)q_tap * (unsigned)Cq);
  }
  bool live = kt0 < kt1;  // false while "loading" past the last tile: every load is forced out of range

  float4 ra[AR];
  float4 rb[NB];
  // state of the tile being loaded (set by piece 0)
  int dy, dx, a_tile, w_tile;

  auto piece_begin = [&]() {
    const int ty = (int)(((unsigned)q_tap * ntx_magic) >> 16), tx = q_tap - ty * tc.ntx;
    const bool kvalid = live && q_tap < ntaps;
    dy = tc.dy0 + ty * p.dstep;
    dx = tc.dx0 + tx * p.dstep;
    const int cofs = q_c4 * 16;
    a_tile = kvalid ? (dy * p.Ws + dx) * lds4 + cofs : OOB_MARK;
    if constexpr (B_NK) {
      const int widx = (tc.ky0 + ty * p.kstep) * p.KW + tc.kx0 + tx * p.kstep;
      w_tile = kvalid ? widx * p.N * p.Cs * 4 + cofs : OOB_MARK;
    }
  };
  auto piece_a = [&](int i, float4* RA) {
    const int y = a_y[i] + dy, x = a_x[i] + dx;
    const bool inb = (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;
    const int voff = inb ? a_lin[i] * lds4 + a_tile : OOB_MARK;
    RA[i] = buf_ld4(src_rs, voff, 0);
  };
  auto piece_b = [&](int i, int kt_next, float4* RB) {
    if constexpr (B_NK) {
      RB[i] = buf_ld4(w_rs, b_row[i] + w_tile, 0);
    } else {
      // W[kk][n] (conv fwd, deconv dgrad: the class walks all taps in weight order, so the flattened K
      // index is the weight row).  Lane = n (coalesced 256 B per wave and k); the row offset is a scalar
      // (rows past Ktot are clamped to the last row: the A operand is zero there).
      const int kk = kt_next * BK + 4 * (kn_kg + KG * i);
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = buf_ld1(w_rs, kn_voff, min(kk + j, Ktot - 1) * p.N * 4);
      RB[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  auto piece_end = [&]() {
    q_c4 += adv_c;
    q_tap += adv_t;
    const bool wrap = q_c4 >= Cq;
    q_c4 -= wrap ? Cq : 0;
    q_tap += wrap ? 1 : 0;
  };
  // piece schedule over the 16 MFMA steps of a tile: 0: begin, 1..AR: A rows, then B, then end
  auto piece_set = [&](int step, int kt_next, float4* RA, float4* RB) {
    if (step == 0) piece_begin();
    if (step >= 1 && step <= AR) piece_a(step - 1, RA);
    if (step > AR && step <= AR + NB) piece_b(step - AR - 1, kt_next, RB);
    if (step == AR + NB + 1) piece_end();
  };
  auto piece = [&](int step, int kt_next) { piece_set(step, kt_next, ra, rb); };
  static_assert(AR + NB + 2 <= 16, "pieces must fit the 16 steps");

  auto store_set = [&](const float4* RA, const float4* RB) {     // MATH == 1
#pragma unroll
    for (int i = 0; i < AR; i++) split_store(Ah + swz_off((tid >> 3) + 32 * i, kq), A_PLANE, RA[i]);
    if constexpr (B_NK) {
#pragma unroll
      for (int i = 0; i < BR_NK; i++) split_store(Bh + swz_off((tid >> 3) + 32 * i, kq), B_PLANE, RB[i]);
    } else {
#pragma unroll
      for (int i = 0; i < BQ_KN; i++) split_store(Bh + swz_off(tid % BN, (tid / BN) + KG * i), B_PLANE, RB[i]);
    }
  };
  auto store_tile = [&]() {
    if constexpr (MATH == 1) {
      store_set(ra, rb);
      return;
    }
#pragma unroll
    for (int i = 0; i < AR; i++)
      *reinterpret_cast<float4*>(As + ((tid >> 3) + 32 * i) * LDK + kq * 4) = ra[i];
    if constexpr (B_NK) {
#pragma unroll
      for (int i = 0; i < BR_NK; i++)
        *reinterpret_cast<float4*>(Bs + ((tid >> 3) + 32 * i) * LDK + kq * 4) = rb[i];
    } else {
#pragma unroll
      for (int i = 0; i < BQ_KN; i++)
        *reinterpret_cast<float4*>(Bs + (tid % BN) * LDK + 4 * ((tid / BN) + KG * i)) = rb[i];
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
  // prologue: tile kt0
#pragma unroll
  for (int st = 0; st < 16; st++) piece(st, kt0);
  store_tile();
  __syncthreads();
  const float* a_rd = As + (wm * WM + l31) * LDK + 16 * lh;
  const float* b_rd = Bs + (wn * WN + l31) * LDK + 16 * lh;
  const unsigned short* ah_rd = Ah + (wm * WM + l31) * LDH;
  const unsigned short* bh_rd = Bh + (wn * WN + l31) * LDH;
  const int gsw = lh ^ ((l31 >> 2) & 3);     // swizzled granule of K16 slab 0 (slab 1: ^ 2); tile bases are multiples of 32
  // MATH == 1: the 12 term groups of the tile in LDS, with the load pieces of tile `kload` (into RA/RB) between them.
  // Two K16 slabs per tile; lane (row l31, half lh) holds k = 16*slab + 8*lh .. +7 of its row for both operands.
  constexpr bool EARLY_LOADS = kEarlyLoads;
  auto mfma_phase = [&](int kload, float4* RA, float4* RB) {
#pragma unroll
    for (int slab = 0; slab < 2; slab++) {
      bf16x8 av[3][TM], bv[3][TN];
#pragma unroll
      for (int pl = 0; pl < 3; pl++) {
#pragma unroll
        for (int i = 0; i < TM; i++)
          av[pl][i] = *reinterpret_cast<const bf16x8*>(ah_rd + pl * A_PLANE + i * 32 * LDH + 8 * (gsw ^ (2 * slab)));
#pragma unroll
        for (int j = 0; j < TN; j++)
          bv[pl][j] = *reinterpret_cast<const bf16x8*>(bh_rd + pl * B_PLANE + j * 32 * LDH + 8 * (gsw ^ (2 * slab)));
      }
      constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
#pragma unroll
      for (int t = 0; t < 6; t++) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ta[t]][i], bv[tb[t]][j], acc[i][j], 0, 0, 0);
        // two load pieces after each of the first eight term groups: a bf16x3 tile is ~0.6 us of matrix-core time, so the
        // requests go out early in the phase and have the rest of it (plus the other resident waves) to come back
        if (EARLY_LOADS) {
          piece_set(2 * (slab * 6 + t), kload, RA, RB);
          piece_set(2 * (slab * 6 + t) + 1, kload, RA, RB);
        } else {
          piece_set(slab * 6 + t, kload, RA, RB);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!EARLY_LOADS) {
#pragma unroll
      for (int st = 12; st < 16; st++) piece_set(st, kload, RA, RB);
    }
  };
  for (int kt = kt0; kt < kt1; kt++) {
    live = kt + 1 < kt1;
    if constexpr (MATH == 1) {
      mfma_phase(kt + 1, ra, rb);
    } else {
#pragma unroll
    for (int j4 = 0; j4 < 4; j4++) {
      float4 av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; i++) av[i] = *reinterpret_cast<const float4*>(a_rd + i * 32 * LDK + 4 * j4);
#pragma unroll
      for (int j = 0; j < TN; j++) bv[j] = *reinterpret_cast<const float4*>(b_rd + j * 32 * LDK + 4 * j4);
#pragma unroll
      for (int e = 0; e < 4; e++) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32((&av[i].x)[e], (&bv[j].x)[e], acc[i][j], 0, 0, 0);
        piece(j4 * 4 + e, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    }
    __syncthreads();  // every wave is done reading this tile
    store_tile();     // (last iteration: zeros, never read)
    __syncthreads();
  }

  // ---- epilogue. C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool to_partial = p.nsplit > 1;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int px = pix[row];
      if (px < 0) continue;
#pragma unroll
      for (int j = 0; j < TN; j++) {
        const int n = n0 + wn * WN + j * 32 + l31;
        if (n >= p.N) continue;
        float v = acc[i][j][r];
        if (to_partial) {
          p.partial[((size_t)split * ((size_t)p.B * p.Hd * p.Wd) + px) * p.N + n] = v;
        } else {
          if (p.bias) v += p.bias[n];
          if (p.leaky) v = leaky_relu(v);
          float* d = p.dst + (size_t)px * p.ldd + n;
          if (p.accumulate) v += *d;
          if (p.act_src && n >= p.act_lo && n < p.act_hi) v *= leaky_grad_from_out(p.act_src[(size_t)px * p.ld_act + n]);
          *d = v;
          store_planes(p.pl, (size_t)px, n, v);
        }
      }
    }
}

// Scalar fallback of the kernel below for destinations whose rows are not 16-byte aligned.
__global__ void splitk_reduce_epilogue_scalar_kernel(const GatherParams p) {
  const size_t npix = (size_t)p.B * p.Hd * p.Wd;
  const size_t total = npix * p.N;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t px = e / p.N;
    const int n = (int)(e - px * p.N);
    float v = 0.f;
    for (int s = 0; s < p.nsplit; s++) v += p.partial[(size_t)s * total + e];
    if (p.bias) v += p.bias[n];
    if (p.leaky) v = leaky_relu(v);
    float* d = p.dst + px * p.ldd + n;
    if (p.accumulate) v += *d;
    if (p.act_src && n >= p.act_lo && n < p.act_hi) v *= leaky_grad_from_out(p.act_src[px * p.ld_act + n]);
    *d = v;
    store_planes(p.pl, px, n, v);
  }
}

// Fixed-order sum of the split-K partials + the epilogue.  One float4 (4 consecutive output channels) per thread; the
// split loop is unrolled so its loads are in flight together (the kernel is a pure HBM/L2 stream).
__global__ void splitk_reduce_epilogue_kernel(const GatherParams p) {
  const size_t npix = (size_t)p.B * p.Hd * p.Wd;
  const size_t total = npix * p.N;
  const unsigned nq = (unsigned)(p.N >> 2);
  const size_t totq = total >> 2;          // N % 4 == 0 (checked by the launchers)
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < totq; q += (size_t)gridDim.x * blockDim.x) {
    const size_t px = q / nq;
    const int n = (int)(q - px * nq) * 4;
    const float4* src = reinterpret_cast<const float4*>(p.partial) + q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int s = 0; s < p.nsplit; s++) {
      const float4 t = src[(size_t)s * totq];
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (p.bias) {   // bias vectors are not necessarily 16-byte aligned: scalar loads (L1-resident)
      v.x += p.bias[n]; v.y += p.bias[n + 1]; v.z += p.bias[n + 2]; v.w += p.bias[n + 3];
    }
    if (p.leaky) { v.x = leaky_relu(v.x); v.y = leaky_relu(v.y); v.z = leaky_relu(v.z); v.w = leaky_relu(v.w); }
    float4* d = reinterpret_cast<float4*>(p.dst + px * p.ldd + n);
    if (p.accumulate) {
      const float4 e = *d;
      v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
    }
    if (p.act_src && n + 3 >= p.act_lo && n < p.act_hi) {
      const float4 a = *reinterpret_cast<const float4*>(p.act_src + px * p.ld_act + n);
      if (n >= p.act_lo && n < p.act_hi) v.x *= leaky_grad_from_out(a.x);
      if (n + 1 >= p.act_lo && n + 1 < p.act_hi) v.y *= leaky_grad_from_out(a.y);
      if (n + 2 >= p.act_lo && n + 2 < p.act_hi) v.z *= leaky_grad_from_out(a.z);
      if (n + 3 >= p.act_lo && n + 3 < p.act_hi) v.w *= leaky_grad_from_out(a.w);
    }
    *d = v;
    store_planes4(p.pl, px, n, v);
  }
}

// ------------------------------------------------------------------ wgrad
struct WgradParams : WgradGeom {
  const float* src;  // gathered operand [B,Hs,Ws,lds], channels a
  const float* dst;  // dense operand   [B,Hg,Wg,ldd], channels b
  float* out;        // dW [(tap,a)][b] (ld = Cb) when nsplit == 1
  float* partial;    // [nsplit][Mp][Cb] otherwise
  int lds, ldd;
  int nsplit;
  unsigned ca_magic;
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_wgrad_kernel(const WgradParams p) {
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves");
  constexpr int A_ELEMS = BK * BM;
  constexpr int AQ = BM / 4, BQ = BN / 4;
  constexpr int AR = (BK * AQ) / 256, BR = (BK * BQ) / 256;
  constexpr int ASTEP = 256 / AQ, BSTEP = 256 / BQ;

  // Both operands are row(site)-major in HBM, so the LDS image is [k = site][rows] (16-byte stores, and
  // the MFMA operand read of lane i is the i-th consecutive dword: conflict-free ds_read_b32).
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + A_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int Mp = p.KH * p.KW * p.Ca;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int S = p.B * p.Hg * p.Wg;  // reduction length (sites)
  const int KT = (S + BK - 1) / BK;
  const int kt_per = (KT + p.nsplit - 1) / p.nsplit;
  const int kt0 = blockIdx.z * kt_per, kt1 = min(KT, kt0 + kt_per);

  const __amdgpu_buffer_rsrc_t src_rs =
      make_rsrc(p.src, ((size_t)p.B * p.Hs * p.Ws - 1) * (size_t)p.lds * 4 + (size_t)p.Ca * 4);
  const __amdgpu_buffer_rsrc_t dst_rs = make_rsrc(p.dst, ((size_t)S - 1) * (size_t)p.ldd * 4 + (size_t)p.Cb * 4);

  // this thread's (tap, a) quad — fixed
  const int mm = m0 + (tid % AQ) * 4;
  const bool m_ok = mm < Mp;
  const unsigned tap = fast_div((unsigned)mm, p.ca_magic);
  const int a_ch = mm - (int)tap * p.Ca;
  const int ky = (int)tap / p.KW, kx = (int)tap - ky * p.KW;
  const int dy = p.dy0 + ky, dx = p.dx0 + kx;
  const int a_base = m_ok ? a_ch * 4 : OOB_MARK;
  const int nb = n0 + (tid % BQ) * 4;
  const int lds4 = p.lds * 4;
  const unsigned magW = (unsigned)((0x100000000ull + p.Wg - 1) / p.Wg), magH = (unsigned)((0x100000000ull + p.Hg - 1) / p.Hg);

  int bvoff[BR];  // dense operand: byte offset of (site row, 4 columns); rows past S fall out of the buffer
#pragma unroll
  for (int i = 0; i < BR; i++)
    bvoff[i] = nb < p.Cb ? ((kt0 * BK + tid / BQ + BSTEP * i) * p.ldd + nb) * 4 : OOB_MARK;
  const int bstep = BK * p.ldd * 4;

  float4 ra[AR], rb[BR];
  // gathered operand row i of tile kt: site -> (b, yg, xg) by magic division (exact: site * Wg < 2^32)
  auto piece_a = [&](int i, int kt) {
    const unsigned sidx = (unsigned)(kt * BK + tid / AQ + ASTEP * i);
    const unsigned q = fast_div(sidx, magW);
    const int xg = (int)(sidx - q * (unsigned)p.Wg);
    const unsigned bb = fast_div(q, magH);
    const int yg = (int)(q - bb * (unsigned)p.Hg);
    const int y = yg * p.sm + dy, x = xg * p.sm + dx;
    const bool ok = (int)bb < p.B && (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;
    const int voff = ok ? (((int)bb * p.Hs + y) * p.Ws + x) * lds4 + a_base : OOB_MARK;
    ra[i] = buf_ld4(src_rs, voff, 0);
  };
  auto piece_b = [&](int i) {
    rb[i] = buf_ld4(dst_rs, bvoff[i], 0);
    bvoff[i] += bstep;   // may wrap past 2^31 only for tensors > 1 GB (then marked out of range anyway)
  };
  auto piece = [&](int step, int kt) {
    if (step < AR) piece_a(step, kt);
    else if (step < AR + BR) piece_b(step - AR);
  };
  static_assert(AR + BR <= 16, "pieces must fit the 16 steps");
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < AR; i++)
      *reinterpret_cast<float4*>(As + (tid / AQ + ASTEP * i) * BM + (tid % AQ) * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < BR; i++)
      *reinterpret_cast<float4*>(Bs + (tid / BQ + BSTEP * i) * BN + (tid % BQ) * 4) = rb[i];
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
  for (int st = 0; st < 16; st++) piece(st, kt0);
  store_tile();
  __syncthreads();
  const float* a = As + lh * BM + wm * WM + l31;
  const float* b = Bs + lh * BN + wn * WN + l31;
  for (int kt = kt0; kt < kt1; kt++) {
    // tile kt+1 is loaded while tile kt is multiplied (past the last tile the loads read zeros / are harmless)
    const int ktn = kt + 1 < kt1 ? kt + 1 : KT + 1;
#pragma unroll
    for (int s = 0; s < BK / 2; s++) {
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; i++) av[i] = a[2 * s * BM + i * 32];
#pragma unroll
      for (int j = 0; j < TN; j++) bv[j] = b[2 * s * BN + j * 32];
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      piece(s, ktn);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    store_tile();
    __syncthreads();
  }

  float* o = p.nsplit > 1 ? p.partial + (size_t)blockIdx.z * Mp * p.Cb : p.out;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= Mp) continue;
#pragma unroll
      for (int j = 0; j < TN; j++) {
        const int n = n0 + wn * WN + j * 32 + l31;
        if (n < p.Cb) o[(size_t)m * p.Cb + n] = acc[i][j][r];
      }
    }
}

// Filter gradient with the fp32-equivalent 3-way bf16 split (see split_store).  The MFMA wants 8 consecutive k (= sites)
// of one row per lane, but both operands are site-major in HBM, so here a LANE IS A ROW: lane m loads its channel of 4
// consecutive sites as four dword loads — each instruction still covers 64 consecutive channels of one site, 256
// contiguous bytes — splits the float4 of 4 consecutive k and stores it like the gather kernel does.  The site
// decode is wave-uniform (scalar); the per-lane part of the gathered address is a constant of the thread.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_wgrad_b3_kernel(const WgradParams p) {
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves");
  constexpr int A_PLANE = BM * LDH, B_PLANE = BN * LDH;
  constexpr int ASETS = 256 / BM, BSETS = 256 / BN;   // thread sets per operand; a wave lies inside one set (BM, BN >= 64)
  constexpr int AG = 8 / ASETS, BG = 8 / BSETS;       // groups of 4 consecutive sites per thread (BK / 4 = 8 groups)
  constexpr int NLA = 4 * AG, NL = 4 * AG + 4 * BG;   // dword loads per thread and K tile
  static_assert(BM >= 64 && BN >= 64 && NL <= 36, "load schedule");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* Ah = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Bh = Ah + 3 * A_PLANE;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int Mp = p.KH * p.KW * p.Ca;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int S = p.B * p.Hg * p.Wg;
  const int KT = (S + BK - 1) / BK;
  const int kt_per = (KT + p.nsplit - 1) / p.nsplit;
  const int kt0 = blockIdx.z * kt_per, kt1 = min(KT, kt0 + kt_per);

  const __amdgpu_buffer_rsrc_t src_rs =
      make_rsrc(p.src, ((size_t)p.B * p.Hs * p.Ws - 1) * (size_t)p.lds * 4 + (size_t)p.Ca * 4);
  const __amdgpu_buffer_rsrc_t dst_rs = make_rsrc(p.dst, ((size_t)S - 1) * (size_t)p.ldd * 4 + (size_t)p.Cb * 4);

  // gathered operand: this lane's row m = (tap, a)
  const int arow = tid % BM;
  const int aset = __builtin_amdgcn_readfirstlane(tid / BM);
  const int mm = m0 + arow;
  const bool m_ok = mm < Mp;
  const unsigned tap = fast_div((unsigned)mm, p.ca_magic);
  const int a_ch = mm - (int)tap * p.Ca;
  const int ky = (int)tap / p.KW, kx = (int)tap - ky * p.KW;
  const int dy = p.dy0 + ky, dx = p.dx0 + kx;
  const int lds4 = p.lds * 4;
  const int a_lane_off = (dy * p.Ws + dx) * lds4 + a_ch * 4;      // may be negative; added to the site's base offset
  // dense operand: this lane's column n
  const int brow = tid % BN;
  const int bset = __builtin_amdgcn_readfirstlane(tid / BN);
  const int nb = n0 + brow;
  const bool b_ok = nb < p.Cb;
  const unsigned magW = (unsigned)((0x100000000ull + p.Wg - 1) / p.Wg), magH = (unsigned)((0x100000000ull + p.Hg - 1) / p.Hg);

  float ra[AG][4], rb[BG][4];
  auto load_one = [&](int l, int kt) {
    if (l < NLA) {
      const int gi = l >> 2, j = l & 3;
      const unsigned sidx = (unsigned)(kt * BK + 4 * (aset + ASETS * gi) + j);      // wave-uniform
      const unsigned q = fast_div(sidx, magW);
      const int xg = (int)(sidx - q * (unsigned)p.Wg);
      const unsigned bb = fast_div(q, magH);
      const int yg = (int)(q - bb * (unsigned)p.Hg);
      const int yb = yg * p.sm, xb = xg * p.sm;
      const int base = (((int)bb * p.Hs + yb) * p.Ws + xb) * lds4;                  // uniform
      const bool ok = m_ok && (int)bb < p.B && (unsigned)(yb + dy) < (unsigned)p.Hs && (unsigned)(xb + dx) < (unsigned)p.Ws;
      ra[gi][j] = buf_ld1(src_rs, ok ? base + a_lane_off : OOB_MARK, 0);
    } else {
      const int lb = l - NLA;
      const int gi = lb >> 2, j = lb & 3;
      const int sidx = kt * BK + 4 * (bset + BSETS * gi) + j;                       // wave-uniform
      const bool ok = b_ok && sidx < S;
      rb[gi][j] = buf_ld1(dst_rs, ok ? (sidx * p.ldd + nb) * 4 : OOB_MARK, 0);
    }
  };
  // 12 MFMA term groups per K tile: the loads go out after the first ones (six per group with kEarlyLoads)
  constexpr int LPG = kEarlyLoads ? 6 : 3;
  auto piece = [&](int step, int kt) {
#pragma unroll
    for (int l = LPG * step; l < LPG * step + LPG; l++)
      if (l < NL) load_one(l, kt);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int gi = 0; gi < AG; gi++)
      split_store(Ah + swz_off(arow, aset + ASETS * gi), A_PLANE, make_float4(ra[gi][0], ra[gi][1], ra[gi][2], ra[gi][3]));
#pragma unroll
    for (int gi = 0; gi < BG; gi++)
      split_store(Bh + swz_off(brow, bset + BSETS * gi), B_PLANE, make_float4(rb[gi][0], rb[gi][1], rb[gi][2], rb[gi][3]));
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
  for (int st = 0; st < 12; st++) piece(st, kt0);
  store_tile();
  __syncthreads();
  const unsigned short* ah_rd = Ah + (wm * WM + l31) * LDH;
  const unsigned short* bh_rd = Bh + (wn * WN + l31) * LDH;
  const int gsw = lh ^ ((l31 >> 2) & 3);
  for (int kt = kt0; kt < kt1; kt++) {
    const int ktn = kt + 1 < kt1 ? kt + 1 : KT + 1;   // past the last tile: every load is out of range (zeros)
#pragma unroll
    for (int slab = 0; slab < 2; slab++) {
      bf16x8 av[3][TM], bv[3][TN];
#pragma unroll
      for (int pl = 0; pl < 3; pl++) {
#pragma unroll
        for (int i = 0; i < TM; i++)
          av[pl][i] = *reinterpret_cast<const bf16x8*>(ah_rd + pl * A_PLANE + i * 32 * LDH + 8 * (gsw ^ (2 * slab)));
#pragma unroll
        for (int j = 0; j < TN; j++)
          bv[pl][j] = *reinterpret_cast<const bf16x8*>(bh_rd + pl * B_PLANE + j * 32 * LDH + 8 * (gsw ^ (2 * slab)));
      }
      constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
#pragma unroll
      for (int t = 0; t < 6; t++) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ta[t]][i], bv[tb[t]][j], acc[i][j], 0, 0, 0);
        piece(slab * 6 + t, ktn);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    store_tile();
    __syncthreads();
  }

  float* o = p.nsplit > 1 ? p.partial + (size_t)blockIdx.z * Mp * p.Cb : p.out;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= Mp) continue;
#pragma unroll
      for (int j = 0; j < TN; j++) {
        const int n = n0 + wn * WN + j * 32 + l31;
        if (n < p.Cb) o[(size_t)m * p.Cb + n] = acc[i][j][r];
      }
    }
}

// Column sums of a [npix, C] slice (bias gradients): block = 64 columns x 4 row-lanes.
__global__ void colsum_partial_kernel(const float* __restrict__ x, int ld, long npix, int C,
                                      float* __restrict__ partial) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const long rows_per = (npix + gridDim.y - 1) / gridDim.y;
  const long r0 = blockIdx.y * rows_per, r1 = min(npix, r0 + rows_per);
  float s = 0.f;
  if (col < C)
    for (long r = r0 + rl; r < r1; r += 4) s += x[r * ld + col];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && col < C)
    partial[(size_t)blockIdx.y * C + col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------ skinny kernels (Cout <= 4: flow heads)
struct SkinnyParams {
  const float* x; int ldx;
  const float* w;       // [k,k,Cin,CO]
  const float* bias;
  float* y; int ldy;
  int B, H, W, Cin, k, pt, pl;  // stride 1 'SAME'
};

template <int CO>
__global__ __launch_bounds__(256) void skinny_conv_fwd_kernel(const SkinnyParams p) {
  const int lane = threadIdx.x & 63;
  const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const long npix = (long)p.B * p.H * p.W;
  const int Cq = p.Cin >> 2, nq = p.k * p.k * Cq;
  for (long pxl = wave; pxl < npix; pxl += nwaves) {
    const int ox = (int)(pxl % p.W), oy = (int)((pxl / p.W) % p.H);
    const long b = pxl / ((long)p.W * p.H);
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; c++) acc[c] = 0.f;
    for (int q = lane; q < nq; q += 64) {
      const int tap = q / Cq, c4 = q - tap * Cq;
      const int ky = tap / p.k, kx = tap - ky * p.k;
      const int iy = oy - p.pt + ky, ix = ox - p.pl + kx;
      if ((unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.W) continue;
      const float4 xv = ldg4(p.x + ((b * p.H + iy) * p.W + ix) * p.ldx + c4 * 4);
      const float* wp = p.w + ((size_t)tap * p.Cin + c4 * 4) * CO;
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int c = 0; c < CO; c++) acc[c] += xs[j] * wp[j * CO + c];
    }
#pragma unroll
    for (int c = 0; c < CO; c++) acc[c] = wave_sum(acc[c]);
    if (lane == 0)
#pragma unroll
      for (int c = 0; c < CO; c++) p.y[pxl * p.ldy + c] = acc[c] + (p.bias ? p.bias[c] : 0.f);
  }
}

struct SkinnyBwdParams {
  const float* dz; int lddz;   // [B,H,W,lddz], CO channels
  const float* w;              // [k,k,Cin,CO]
  float* dx; int lddx;
  const float* act_src; int ld_act, act_lo, act_hi;
  int accumulate;
  int B, H, W, Cin, k, pt, pl;
  PlaneOut po;                 // optional 16-bit operand planes of dx (final pre-activation gradients)
};

// dx[pix, ci] (+)= sum_tap sum_co dz[pix + pad - tap, co] * w[tap, ci, co]; thread = (pixel, ci-quad).
template <int CO>
__global__ __launch_bounds__(256) void skinny_conv_dgrad_kernel(const SkinnyBwdParams p) {
  const int Cq = p.Cin >> 2;
  const long total = (long)p.B * p.H * p.W * Cq;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long pxl = e / Cq;
    const int c4 = (int)(e - pxl * Cq);
    const int ix = (int)(pxl % p.W), iy = (int)((pxl / p.W) % p.H);
    const long b = pxl / ((long)p.W * p.H);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < p.k; ky++) {
      const int oy = iy + p.pt - ky;
      if ((unsigned)oy >= (unsigned)p.H) continue;
      for (int kx = 0; kx < p.k; kx++) {
        const int ox = ix + p.pl - kx;
        if ((unsigned)ox >= (unsigned)p.W) continue;
        const float* dzp = p.dz + ((b * p.H + oy) * p.W + ox) * p.lddz;
        const float* wp = p.w + ((size_t)(ky * p.k + kx) * p.Cin + c4 * 4) * CO;
#pragma unroll
        for (int c = 0; c < CO; c++) {
          const float g = dzp[c];
#pragma unroll
          for (int j = 0; j < 4; j++) a[j] += g * wp[j * CO + c];
        }
      }
    }
    float* d = p.dx + pxl * p.lddx + c4 * 4;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float v = a[j];
      const int n = c4 * 4 + j;
      if (p.accumulate) v += d[j];
      if (p.act_src && n >= p.act_lo && n < p.act_hi) v *= leaky_grad_from_out(p.act_src[pxl * p.ld_act + n]);
      d[j] = v;
      store_planes(p.po, (size_t)pxl, n, v);
    }
  }
}

// dw[tap, ci, co] partials: wave = 64 ci-quads, loops over a chunk of INPUT pixels, each pixel's
// x quad is read once and scattered into all k*k taps (k == 3 only).
struct SkinnyWgradParams {
  const float* x; int ldx;
  const float* dz; int lddz;
  float* partial;  // [nchunk][9][Cin][CO]
  int B, H, W, Cin, pt, pl;
};

template <int CO>
__global__ __launch_bounds__(64) void skinny_conv_wgrad3_kernel(const SkinnyWgradParams p) {
  const int Cq = p.Cin >> 2;
  const int c4 = blockIdx.x * 64 + threadIdx.x;
  const long npix = (long)p.B * p.H * p.W;
  const long per = (npix + gridDim.y - 1) / gridDim.y;
  const long p0 = blockIdx.y * per, p1 = min(npix, p0 + per);
  float acc[9][4][CO];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int c = 0; c < CO; c++) acc[t][j][c] = 0.f;
  const bool active = c4 < Cq;
  for (long pxl = p0; pxl < p1; pxl++) {
    const int ix = (int)(pxl % p.W), iy = (int)((pxl / p.W) % p.H);
    const long b = pxl / ((long)p.W * p.H);
    float4 xv = make_float4(0, 0, 0, 0);
    if (active) xv = ldg4(p.x + pxl * p.ldx + c4 * 4);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int ky = 0; ky < 3; ky++) {
      const int oy = iy + p.pt - ky;
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const int ox = ix + p.pl - kx;
        if ((unsigned)oy < (unsigned)p.H && (unsigned)ox < (unsigned)p.W) {  // wave-uniform
          const float* dzp = p.dz + ((b * p.H + oy) * p.W + ox) * p.lddz;
#pragma unroll
          for (int c = 0; c < CO; c++) {
            const float g = dzp[c];
#pragma unroll
            for (int j = 0; j < 4; j++) acc[ky * 3 + kx][j][c] += xs[j] * g;
          }
        }
      }
    }
  }
  if (active) {
    float* o = p.partial + (size_t)blockIdx.y * 9 * p.Cin * CO;
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int c = 0; c < CO; c++) o[((size_t)t * p.Cin + c4 * 4 + j) * CO + c] = acc[t][j][c];
  }
}

// ------------------------------------------------------------------ strip kernels for the large flow heads
// k = 3, stride 1, Cout = 2 (flow2 / flow3 at 384x512: >= 24k pixels).  The per-pixel kernels above re-read the 14-28 KB
// filter from L1 for EVERY pixel (L1-bandwidth bound: 21 KB per output pixel through a 64 B/clk port).  Here a lane owns
// one channel quad and keeps its 9 x 4 x 2 filter values in registers while the wave walks a strip of S pixels of one
// image row, so the filter is read once per strip and every input quad once per (row, strip).
// Explicit fmaf here (the file is built with -ffp-contract=off): these kernels are VALU-bound, the fused form halves
// the instruction count and is the more accurate of the two roundings.
// S = pixels per strip: 8 on the large levels, 4 on the small ones (W % S == 0 is a dispatch condition)

__device__ __forceinline__ void load_head_weights(const float* __restrict__ w, int Cin, int c4, bool active,
                                                  float (&wr)[9][8]) {
#pragma unroll
  for (int t = 0; t < 9; t++) {
    float4 a = make_float4(0, 0, 0, 0), b = a;
    if (active) {
      const float* wp = w + ((size_t)t * Cin + c4 * 4) * 2;   // [tap][ci][co], 8 contiguous floats for the quad
      a = ldg4(wp);
      b = ldg4(wp + 4);
    }
    wr[t][0] = a.x; wr[t][1] = a.y; wr[t][2] = a.z; wr[t][3] = a.w;
    wr[t][4] = b.x; wr[t][5] = b.y; wr[t][6] = b.z; wr[t][7] = b.w;
  }
}

// y[b, yy, x0+o, co] = bias + sum_{ky,kx,c} x[b, yy-1+ky, x0+o-1+kx, c] * w[ky,kx,c,co]
// WSPLIT (small levels): one strip per BLOCK, the four waves split the channel-quad groups and their sums meet in LDS.
template <int S, bool WSPLIT>
__global__ __launch_bounds__(256) void head3_fwd_strip_kernel(const SkinnyParams p) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int strips = p.W / S;
  const long njobs = (long)p.B * p.H * strips;
  const long job = WSPLIT ? (long)blockIdx.x
                          : (long)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6));
  if (job >= njobs) return;
  const int xs = (int)(job % strips);
  const long row = job / strips;
  const int yy = (int)(row % p.H);
  const long b = row / p.H;
  const int x0 = xs * S;
  const int Cq = p.Cin >> 2;
  float acc[2 * S];   // [o][co]
#pragma unroll
  for (int o = 0; o < 2 * S; o++) acc[o] = 0.f;
  for (int g0 = WSPLIT ? 64 * wv : 0; g0 < Cq; g0 += WSPLIT ? 256 : 64) {
    const bool active = g0 + lane < Cq;
    const int c4 = min(g0 + lane, Cq - 1);          // clamped: every load below is unconditional and in range
#pragma unroll
    for (int ky = 0; ky < 3; ky++) {
      const int iy = yy - 1 + ky;
      if ((unsigned)iy >= (unsigned)p.H) continue;   // wave-uniform
      float wr[3][8];                                // this filter row only: 24 registers instead of 72
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const float* wp = p.w + ((size_t)(ky * 3 + kx) * p.Cin + c4 * 4) * 2;
        const float4 wa = ldg4(wp), wb = ldg4(wp + 4);
        wr[kx][0] = wa.x; wr[kx][1] = wa.y; wr[kx][2] = wa.z; wr[kx][3] = wa.w;
        wr[kx][4] = wb.x; wr[kx][5] = wb.y; wr[kx][6] = wb.z; wr[kx][7] = wb.w;
      }
      const float* xrow = p.x + ((b * p.H + iy) * p.W) * p.ldx + c4 * 4;
      float4 xv[S + 2];
#pragma unroll
      for (int i = 0; i < S + 2; i++) {
        const int ix = x0 - 1 + i;
        const float4 t = ldg4(xrow + (long)min(max(ix, 0), p.W - 1) * p.ldx);
        const bool ok = active && (unsigned)ix < (unsigned)p.W;
        xv[i] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
      }
#pragma unroll
      for (int o = 0; o < S; o++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
          const float4 v = xv[o + kx];
          const float* q = wr[kx];
          acc[2 * o] = fmaf(v.x, q[0], acc[2 * o]); acc[2 * o + 1] = fmaf(v.x, q[1], acc[2 * o + 1]);
          acc[2 * o] = fmaf(v.y, q[2], acc[2 * o]); acc[2 * o + 1] = fmaf(v.y, q[3], acc[2 * o + 1]);
          acc[2 * o] = fmaf(v.z, q[4], acc[2 * o]); acc[2 * o + 1] = fmaf(v.z, q[5], acc[2 * o + 1]);
          acc[2 * o] = fmaf(v.w, q[6], acc[2 * o]); acc[2 * o + 1] = fmaf(v.w, q[7], acc[2 * o + 1]);
        }
    }
  }
  // 2S wave sums as one halving butterfly (S=8: 8+4+2+1 exchanges + 2 plain steps = 17 shuffles instead of 96): after
  // the xor-32 step a lane keeps half of the values, ... until exactly one is left: value id = the top bits of the lane.
  constexpr int NV = 2 * S;
  constexpr int LG = NV == 16 ? 4 : NV == 8 ? 3 : 2;
  static_assert(NV == 16 || NV == 8 || NV == 4, "S in {8, 4, 2}");
  float v1;
  if constexpr (NV == 16) {
    float v8[8], v4[4], v2[2];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const bool hi = lane & 32;
      v8[i] = (hi ? acc[8 + i] : acc[i]) + __shfl_xor(hi ? acc[i] : acc[8 + i], 32);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bool hi = lane & 16;
      v4[i] = (hi ? v8[4 + i] : v8[i]) + __shfl_xor(hi ? v8[i] : v8[4 + i], 16);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const bool hi = lane & 8;
      v2[i] = (hi ? v4[2 + i] : v4[i]) + __shfl_xor(hi ? v4[i] : v4[2 + i], 8);
    }
    const bool hi = lane & 4;
    v1 = (hi ? v2[1] : v2[0]) + __shfl_xor(hi ? v2[0] : v2[1], 4);
  } else if constexpr (NV == 8) {
    float v4[4], v2[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bool hi = lane & 32;
      v4[i] = (hi ? acc[4 + i] : acc[i]) + __shfl_xor(hi ? acc[i] : acc[4 + i], 32);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const bool hi = lane & 16;
      v2[i] = (hi ? v4[2 + i] : v4[i]) + __shfl_xor(hi ? v4[i] : v4[2 + i], 16);
    }
    const bool hi = lane & 8;
    v1 = (hi ? v2[1] : v2[0]) + __shfl_xor(hi ? v2[0] : v2[1], 8);
  } else {
    float v2[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const bool hi = lane & 32;
      v2[i] = (hi ? acc[2 + i] : acc[i]) + __shfl_xor(hi ? acc[i] : acc[2 + i], 32);
    }
    const bool hi = lane & 16;
    v1 = (hi ? v2[1] : v2[0]) + __shfl_xor(hi ? v2[0] : v2[1], 16);
  }
#pragma unroll
  for (int m = 32 >> LG; m >= 1; m >>= 1) v1 += __shfl_xor(v1, m);
  const int id = lane >> (6 - LG);      // value index = [o][co]
  const bool holder = (lane & ((64 >> LG) - 1)) == 0;
  if (WSPLIT) {
    __shared__ float red[4][NV];
    if (holder) red[wv][id] = v1;
    __syncthreads();
    if (threadIdx.x < NV) {
      const int t = threadIdx.x;
      const float sum = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
      p.y[((b * p.H + yy) * p.W + x0 + (t >> 1)) * p.ldy + (t & 1)] = sum + (p.bias ? p.bias[t & 1] : 0.f);
    }
  } else if (holder) {
    p.y[((b * p.H + yy) * p.W + x0 + (id >> 1)) * p.ldy + (id & 1)] = v1 + (p.bias ? p.bias[id & 1] : 0.f);
  }
}

// 2-D form of the strip kernel for the large levels (flow2, flow3: the activation is 38-77 MB and the op is its HBM time): a wave
// owns a tile of HT_R rows x 8 pixels.  The 72 filter values of a lane's channel quad stay in registers for the whole tile (the
// strip kernel reloads a filter row per strip: 96 of 256 bytes per lane and input row), and an input row is loaded ONCE and
// feeds the up to three output rows it belongs to — (HT_R + 2) / HT_R = 1.5 reads per activation byte instead of 3.
// acc[r][o][co]: 64 wave sums, reduced by one halving butterfly after which lane l holds value l = (r, o, co).
constexpr int HT_R = 4;
__global__ __launch_bounds__(256) void head3_fwd_tile_kernel(const SkinnyParams p) {
  constexpr int S = 8;
  const int lane = threadIdx.x & 63;
  const int strips = p.W / S, rtiles = (p.H + HT_R - 1) / HT_R;
  const long njobs = (long)p.B * rtiles * strips;
  const long job = (long)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6));
  if (job >= njobs) return;
  const int xs = (int)(job % strips);
  const long rowt = job / strips;
  const int y0 = (int)(rowt % rtiles) * HT_R;
  const long b = rowt / rtiles;
  const int x0 = xs * S;
  const int Cq = p.Cin >> 2;
  float acc[HT_R][2 * S];
#pragma unroll
  for (int r = 0; r < HT_R; r++)
#pragma unroll
    for (int o = 0; o < 2 * S; o++) acc[r][o] = 0.f;
  for (int g0 = 0; g0 < Cq; g0 += 64) {
    const bool active = g0 + lane < Cq;
    const int c4 = min(g0 + lane, Cq - 1);          // clamped: every load below is unconditional and in range
    float wr[9][8];
#pragma unroll
    for (int t = 0; t < 9; t++) {
      const float* wp = p.w + ((size_t)t * p.Cin + c4 * 4) * 2;
      const float4 wa = ldg4(wp), wb = ldg4(wp + 4);
      wr[t][0] = wa.x; wr[t][1] = wa.y; wr[t][2] = wa.z; wr[t][3] = wa.w;
      wr[t][4] = wb.x; wr[t][5] = wb.y; wr[t][6] = wb.z; wr[t][7] = wb.w;
    }
#pragma unroll
    for (int ri = 0; ri < HT_R + 2; ri++) {          // input row y0 - 1 + ri: output row r = ri - ky takes it through filter row ky
      const int iy = y0 - 1 + ri;
      if ((unsigned)iy >= (unsigned)p.H) continue;   // wave-uniform (zero padding: nothing to add)
      const float* xrow = p.x + ((b * p.H + iy) * p.W) * p.ldx + c4 * 4;
      float4 xv[S + 2];
#pragma unroll
      for (int i = 0; i < S + 2; i++) {
        const int ix = x0 - 1 + i;
        const float4 t = ldg4(xrow + (long)min(max(ix, 0), p.W - 1) * p.ldx);
        const bool ok = active && (unsigned)ix < (unsigned)p.W;
        xv[i] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
      }
#pragma unroll
      for (int ky = 0; ky < 3; ky++) {
        const int r = ri - ky;
        if (r < 0 || r >= HT_R) continue;            // compile-time after unrolling
#pragma unroll
        for (int o = 0; o < S; o++)
#pragma unroll
          for (int kx = 0; kx < 3; kx++) {
            const float4 v = xv[o + kx];
            const float* q = wr[ky * 3 + kx];
            float* a = acc[r];
            a[2 * o] = fmaf(v.x, q[0], a[2 * o]); a[2 * o + 1] = fmaf(v.x, q[1], a[2 * o + 1]);
            a[2 * o] = fmaf(v.y, q[2], a[2 * o]); a[2 * o + 1] = fmaf(v.y, q[3], a[2 * o + 1]);
            a[2 * o] = fmaf(v.z, q[4], a[2 * o]); a[2 * o + 1] = fmaf(v.z, q[5], a[2 * o + 1]);
            a[2 * o] = fmaf(v.w, q[6], a[2 * o]); a[2 * o + 1] = fmaf(v.w, q[7], a[2 * o + 1]);
          }
      }
    }
  }
  // 64 wave sums as one halving butterfly: afterwards lane l holds the wave total of value l = r * 16 + o * 2 + co
  float v32[32], v16[16], v8[8], v4[4], v2[2];
  {
    const float* f = &acc[0][0];
    const bool hi = lane & 32;
#pragma unroll
    for (int i = 0; i < 32; i++) v32[i] = (hi ? f[32 + i] : f[i]) + __shfl_xor(hi ? f[i] : f[32 + i], 32);
  }
  {
    const bool hi = lane & 16;
#pragma unroll
    for (int i = 0; i < 16; i++) v16[i] = (hi ? v32[16 + i] : v32[i]) + __shfl_xor(hi ? v32[i] : v32[16 + i], 16);
  }
  {
    const bool hi = lane & 8;
#pragma unroll
    for (int i = 0; i < 8; i++) v8[i] = (hi ? v16[8 + i] : v16[i]) + __shfl_xor(hi ? v16[i] : v16[8 + i], 8);
  }
  {
    const bool hi = lane & 4;
#pragma unroll
    for (int i = 0; i < 4; i++) v4[i] = (hi ? v8[4 + i] : v8[i]) + __shfl_xor(hi ? v8[i] : v8[4 + i], 4);
  }
  {
    const bool hi = lane & 2;
#pragma unroll
    for (int i = 0; i < 2; i++) v2[i] = (hi ? v4[2 + i] : v4[i]) + __shfl_xor(hi ? v4[i] : v4[2 + i], 2);
  }
  const bool hi1 = lane & 1;
  const float tot = (hi1 ? v2[1] : v2[0]) + __shfl_xor(hi1 ? v2[0] : v2[1], 1);
  const int r = lane >> 4, o = (lane >> 1) & 7, co = lane & 1;
  if (y0 + r < p.H) p.y[((b * p.H + y0 + r) * p.W + x0 + o) * p.ldy + co] = tot + (p.bias ? p.bias[co] : 0.f);
}

// One row of the wave-uniform dz window (columns x0-1 .. x0+S, 2 channels = 2*(S+2) floats, zero outside the image):
// lane l < 2*(S+2) loads element l in ONE coalesced instruction; consumers broadcast with v_readlane (an SGPR operand).
template <int S>
__device__ __forceinline__ float load_dz_window_row(const float* __restrict__ dz, int lddz, long b, int oy, int x0, int H,
                                                    int W, int lane) {
  const int col = lane >> 1, co = lane & 1;
  const int ox = x0 - 1 + col;
  const bool ok = lane < 2 * (S + 2) && (unsigned)oy < (unsigned)H && (unsigned)ox < (unsigned)W;
  const float v = dz[((b * H + (ok ? oy : 0)) * W + (ok ? ox : 0)) * lddz + co];
  return ok ? v : 0.f;
}
__device__ __forceinline__ float lane_bcast(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

// dx[b, yy, x0+o, ci] (+)= sum_{ky,kx,co} dz[b, yy+1-ky, x0+o+1-kx, co] * w[ky,kx,ci,co]; wave = (row, strip, quad group);
// the dz window of the strip is wave-uniform (scalar loads), lanes differ only in the channel quad.
// A wave walks `ns` consecutive strips of its row with the quad's 72 filter values resident (round 6: one strip per wave
// re-loaded them — 18 16-byte loads per lane — for every 8 pixels = 8 16-byte stores).
template <int S>
__global__ __launch_bounds__(256) void head3_dgrad_strip_kernel(const SkinnyBwdParams p, int ns) {
  const int lane = threadIdx.x & 63;
  const int strips = p.W / S, sblocks = strips / ns;
  const int Cq = p.Cin >> 2, groups = (Cq + 63) / 64;
  const long njobs = (long)p.B * p.H * sblocks * groups;
  const long job = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6));
  if (job >= njobs) return;
  const int g = (int)(job % groups);
  const long j2 = job / groups;
  const int xsb = (int)(j2 % sblocks);
  const long row = j2 / sblocks;
  const int yy = (int)(row % p.H);
  const long b = row / p.H;
  const int c4 = g * 64 + lane;
  const bool active = c4 < Cq;
  float wr[9][8];
  load_head_weights(p.w, p.Cin, c4, active, wr);
#pragma unroll 1
  for (int si = 0; si < ns; si++) {
  const int x0 = (xsb * ns + si) * S;
  // dz window: rows yy-1..yy+1, columns x0-1..x0+S, 2 channels; zero outside the image
  float dzv[3];
#pragma unroll
  for (int r = 0; r < 3; r++) dzv[r] = load_dz_window_row<S>(p.dz, p.lddz, b, yy - 1 + r, x0, p.H, p.W, lane);
  float dzw[3][S + 2][2];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int i = 0; i < S + 2; i++) {
      dzw[r][i][0] = lane_bcast(dzv[r], 2 * i);
      dzw[r][i][1] = lane_bcast(dzv[r], 2 * i + 1);
    }
  if (!active) continue;
#pragma unroll
  for (int o = 0; o < S; o++) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        // oy = yy + 1 - ky -> window row 2 - ky;  ox = x0 + o + 1 - kx -> window column o + 2 - kx
        const float g0 = dzw[2 - ky][o + 2 - kx][0], g1 = dzw[2 - ky][o + 2 - kx][1];
        const float* q = wr[ky * 3 + kx];
#pragma unroll
        for (int j = 0; j < 4; j++) { a[j] = fmaf(g0, q[2 * j], a[j]); a[j] = fmaf(g1, q[2 * j + 1], a[j]); }
      }
    const long pxl = (b * p.H + yy) * p.W + x0 + o;
    float* d = p.dx + pxl * p.lddx + c4 * 4;
    float4 v = make_float4(a[0], a[1], a[2], a[3]);
    if (p.accumulate) {
      const float4 e = ldg4(d);
      v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
    }
    if (p.act_src) {
      const int n = c4 * 4;
      if (n + 3 >= p.act_lo && n < p.act_hi) {
        const float4 s = ldg4(p.act_src + pxl * p.ld_act + n);
        if (n >= p.act_lo && n < p.act_hi) v.x *= leaky_grad_from_out(s.x);
        if (n + 1 >= p.act_lo && n + 1 < p.act_hi) v.y *= leaky_grad_from_out(s.y);
        if (n + 2 >= p.act_lo && n + 2 < p.act_hi) v.z *= leaky_grad_from_out(s.z);
        if (n + 3 >= p.act_lo && n + 3 < p.act_hi) v.w *= leaky_grad_from_out(s.w);
      }
    }
    *reinterpret_cast<float4*>(d) = v;
    store_planes4(p.po, (size_t)pxl, c4 * 4, v);
  }
  }
}

// dw[ky,kx,ci,co] partials = sum over a chunk of strips of x[b,iy,ix,ci] * dz[b, iy+1-ky, ix+1-kx, co].
// block = 4 waves x the same 64 channel quads; each wave walks its own strips, then the four register tiles are
// summed through LDS in a fixed order (deterministic) and the block writes ONE partial.
template <int S>
__device__ __forceinline__ void head3_wgrad_strip_body(const SkinnyWgradParams& p, int strips_per_wave, int bx, int by,
                                                       float* __restrict__ red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int Cq = p.Cin >> 2;
  const int c4 = bx * 64 + lane;
  const bool active = c4 < Cq;
  const int cc = min(c4, Cq - 1);   // clamped: loads are unconditional, inactive lanes contribute zeros
  const int strips = p.W / S;
  const long nstrips = (long)p.B * p.H * strips;
  const long s0 = __builtin_amdgcn_readfirstlane((int)(((long)by * 4 + wv) * strips_per_wave));
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int j = 0; j < 8; j++) acc[t][j] = 0.f;
  for (long sidx = s0; sidx < min(nstrips, s0 + strips_per_wave); sidx++) {
    const int xs = (int)(sidx % strips);
    const long row = sidx / strips;
    const int iy = (int)(row % p.H);
    const long b = row / p.H;
    const int x0 = xs * S;
    float4 xv[S];
    const float* xrow = p.x + ((b * p.H + iy) * p.W + x0) * p.ldx + cc * 4;
#pragma unroll
    for (int i = 0; i < S; i++) {
      const float4 t = ldg4(xrow + (long)i * p.ldx);
      xv[i] = make_float4(active ? t.x : 0.f, active ? t.y : 0.f, active ? t.z : 0.f, active ? t.w : 0.f);
    }
    float dzv[3];   // window rows iy+1, iy, iy-1 for ky = 0, 1, 2
#pragma unroll
    for (int ky = 0; ky < 3; ky++) dzv[ky] = load_dz_window_row<S>(p.dz, p.lddz, b, iy + 1 - ky, x0, p.H, p.W, lane);
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++)
#pragma unroll
        for (int i = 0; i < S; i++) {
          // ox = x0 + i + 1 - kx -> window column i + 2 - kx
          const float g0 = lane_bcast(dzv[ky], 2 * (i + 2 - kx)), g1 = lane_bcast(dzv[ky], 2 * (i + 2 - kx) + 1);
          float* a = acc[ky * 3 + kx];
          a[0] = fmaf(xv[i].x, g0, a[0]); a[1] = fmaf(xv[i].x, g1, a[1]);
          a[2] = fmaf(xv[i].y, g0, a[2]); a[3] = fmaf(xv[i].y, g1, a[3]);
          a[4] = fmaf(xv[i].z, g0, a[4]); a[5] = fmaf(xv[i].z, g1, a[5]);
          a[6] = fmaf(xv[i].w, g0, a[6]); a[7] = fmaf(xv[i].w, g1, a[7]);
        }
  }
  // fixed-order cross-wave sum: wave 0 stores, waves 1..3 add in turn
  for (int turn = 0; turn < 4; turn++) {
    if (wv == turn) {
#pragma unroll
      for (int t = 0; t < 9; t++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
          float* r = red + (t * 8 + j) * 64 + lane;
          *r = (turn == 0 ? 0.f : *r) + acc[t][j];
        }
    }
    __syncthreads();
  }
  // partial layout [chunk][tap][Cin][2]: the quad's 8 values per tap are contiguous
  float* o = p.partial + (size_t)by * 9 * p.Cin * 2;
  for (int e = threadIdx.x; e < 72 * 64; e += 256) {
    const int ln = e & 63, tj = e >> 6, t = tj >> 3, j = tj & 7;
    const int cq = bx * 64 + ln;
    if (cq < Cq) o[((size_t)t * p.Cin + cq * 4) * 2 + j] = red[e];
  }
}
template <int S>
__global__ __launch_bounds__(256) void head3_wgrad_strip_kernel(const SkinnyWgradParams p, int strips_per_wave) {
  __shared__ float red[72 * 64];
  head3_wgrad_strip_body<S>(p, strips_per_wave, blockIdx.x, blockIdx.y, red);
}

// ------------------------------------------------------------------ 1x1 conv, 32 output channels: data gradient
// conv_redir (flownet.py:224, 256 -> 32, k1): dx[px, ci] (+)= sum_co dz[px, co] * w[ci, co].  K = 32 is ONE K-tile of the
// implicit-GEMM kernel, which then is all prologue + epilogue (55 us, 7 TFLOP/s); this op is a stream over dx.  A lane owns
// a channel quad with its 4 x 32 filter values in registers; the wave walks PW_S pixels whose dz rows are wave-uniform
// (one coalesced 128-byte load + v_readlane broadcasts).
constexpr int PW_S = 8, PW_CO = 32;
__global__ __launch_bounds__(256) void pointwise32_dgrad_kernel(const SkinnyBwdParams p) {
  const int lane = threadIdx.x & 63;
  const int Cq = p.Cin >> 2, groups = (Cq + 63) / 64;
  const long npix = (long)p.B * p.H * p.W;
  const long nstrips = (npix + PW_S - 1) / PW_S;
  const long job = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6));
  if (job >= nstrips * groups) return;
  const int g = (int)(job % groups);
  const long p0 = (job / groups) * PW_S;
  const bool active = g * 64 + lane < Cq;
  const int c4 = min(g * 64 + lane, Cq - 1);
  float wr[4][PW_CO];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int c = 0; c < PW_CO; c += 4) {
      const float4 t = ldg4(p.w + (size_t)(c4 * 4 + j) * PW_CO + c);
      wr[j][c] = t.x; wr[j][c + 1] = t.y; wr[j][c + 2] = t.z; wr[j][c + 3] = t.w;
    }
  float dzv[PW_S];
#pragma unroll
  for (int i = 0; i < PW_S; i++) {
    const long px = min(p0 + i, npix - 1);
    dzv[i] = p.dz[px * p.lddz + (lane & (PW_CO - 1))];
  }
#pragma unroll
  for (int i = 0; i < PW_S; i++) {
    if (p0 + i >= npix) break;     // wave-uniform
    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < PW_CO; c++) {
      const float gz = lane_bcast(dzv[i], c);
#pragma unroll
      for (int j = 0; j < 4; j++) a[j] = fmaf(gz, wr[j][c], a[j]);
    }
    if (!active) continue;
    const long pxl = p0 + i;
    float* d = p.dx + pxl * p.lddx + c4 * 4;
    float4 v = make_float4(a[0], a[1], a[2], a[3]);
    if (p.accumulate) {
      const float4 e = ldg4(d);
      v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
    }
    if (p.act_src) {
      const int n = c4 * 4;
      if (n + 3 >= p.act_lo && n < p.act_hi) {
        const float4 sa = ldg4(p.act_src + pxl * p.ld_act + n);
        if (n >= p.act_lo && n < p.act_hi) v.x *= leaky_grad_from_out(sa.x);
        if (n + 1 >= p.act_lo && n + 1 < p.act_hi) v.y *= leaky_grad_from_out(sa.y);
        if (n + 2 >= p.act_lo && n + 2 < p.act_hi) v.z *= leaky_grad_from_out(sa.z);
        if (n + 3 >= p.act_lo && n + 3 < p.act_hi) v.w *= leaky_grad_from_out(sa.w);
      }
    }
    *reinterpret_cast<float4*>(d) = v;
    store_planes4(p.po, (size_t)pxl, c4 * 4, v);
  }
}

// ------------------------------------------------------------------ tiny deconv (flowN_upM: 2 -> 2 channels, k4 s2)
// y[b,oy,ox,co] = bias + sum over the <=4 valid taps: oy = 2*iy + ky - 1.  These kernels sit in the serial chain of the decoder
// (four forward, four backward launches per step) and are pure latency: every tap is loaded unconditionally from a clamped
// address and masked afterwards (a `continue` per tap made the loads a chain of dependent L2 round trips), the pixel decode is
// 32-bit (a 64-bit div / mod pair is ~200 instructions), and invalid taps add +0 in the same order as before.
template <int CI, int CO>
__global__ void tiny_deconv_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                       const float* __restrict__ bias, float* __restrict__ y, int ldy, int B, int H,
                                       int W, const PlaneOut po) {
  const unsigned OH = 2u * (unsigned)H, OW = 2u * (unsigned)W;
  const unsigned n = (unsigned)B * OH * OW;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const Pix pp = decode_pix(e, OW, OH);
    const int ox = pp.x, oy = pp.y, b = pp.n;
    const int ky0 = (oy + 1) & 1, kx0 = (ox + 1) & 1;
    float xv[2][2][CI];
    bool ok[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const int iy = ((oy + 1 - ky0) >> 1) - a, ix = ((ox + 1 - kx0) >> 1) - c;       // taps (ky0 + 2a, kx0 + 2c)
        ok[a][c] = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const float* xp = x + ((size_t)(b * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)) * ldx;
#pragma unroll
        for (int ci = 0; ci < CI; ci++) xv[a][c][ci] = xp[ci];
      }
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; c++) acc[c] = bias ? bias[c] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const float* wp = w + ((ky0 + 2 * a) * 4 + kx0 + 2 * c) * CO * CI;
#pragma unroll
        for (int co = 0; co < CO; co++)
#pragma unroll
          for (int ci = 0; ci < CI; ci++) acc[co] += (ok[a][c] ? xv[a][c][ci] : 0.f) * wp[co * CI + ci];
      }
#pragma unroll
    for (int c = 0; c < CO; c++) {
      y[(size_t)e * ldy + c] = acc[c];
      store_planes(po, (size_t)e, c, acc[c]);
    }
  }
}

template <int CI, int CO>
__global__ void tiny_deconv_dgrad_kernel(const float* __restrict__ dz, int lddz, const float* __restrict__ w,
                                         float* __restrict__ dx, int lddx, int accumulate, int B, int H, int W) {
  const int OH = 2 * H, OW = 2 * W;
  const unsigned n = (unsigned)B * (unsigned)H * (unsigned)W;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const Pix pp = decode_pix(e, (unsigned)W, (unsigned)H);
    const int ix = pp.x, iy = pp.y, b = pp.n;
    float g[16][CO];
    bool ok[16];
#pragma unroll
    for (int ky = 0; ky < 4; ky++)
#pragma unroll
      for (int kx = 0; kx < 4; kx++) {
        const int oy = 2 * iy + ky - 1, ox = 2 * ix + kx - 1;
        ok[ky * 4 + kx] = (unsigned)oy < (unsigned)OH && (unsigned)ox < (unsigned)OW;
        const float* gp = dz + ((size_t)(b * OH + min(max(oy, 0), OH - 1)) * OW + min(max(ox, 0), OW - 1)) * lddz;
#pragma unroll
        for (int co = 0; co < CO; co++) g[ky * 4 + kx][co] = gp[co];
      }
    float acc[CI];
#pragma unroll
    for (int c = 0; c < CI; c++) acc[c] = 0.f;
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const float* wp = w + t * CO * CI;
#pragma unroll
      for (int co = 0; co < CO; co++)
#pragma unroll
        for (int ci = 0; ci < CI; ci++) acc[ci] += (ok[t] ? g[t][co] : 0.f) * wp[co * CI + ci];
    }
#pragma unroll
    for (int c = 0; c < CI; c++) {
      float* d = dx + (size_t)e * lddx + c;
      *d = accumulate ? *d + acc[c] : acc[c];
    }
  }
}

// dw[ky,kx,co,ci] partial sums per block -> partial[block][16*CO*CI]
template <int CI, int CO>
__device__ __forceinline__ void tiny_deconv_wgrad_body(const float* __restrict__ x, int ldx, const float* __restrict__ dz, int lddz,
                                                       float* __restrict__ partial, int B, int H, int W, int bx, int nblocks,
                                                       float (*red)[16 * CO * CI]) {
  constexpr int NV = 16 * CO * CI;
  const int OH = 2 * H, OW = 2 * W;
  const long n = (long)B * H * W;
  float acc[16][CO][CI];
#pragma unroll
  for (int t = 0; t < 16; t++)
#pragma unroll
    for (int a = 0; a < CO; a++)
#pragma unroll
      for (int c = 0; c < CI; c++) acc[t][a][c] = 0.f;
  for (long e = bx * (long)blockDim.x + threadIdx.x; e < n; e += (long)nblocks * blockDim.x) {
    const int ix = (int)(e % W), iy = (int)((e / W) % H);
    const long b = e / ((long)W * H);
    float xv[CI];
#pragma unroll
    for (int c = 0; c < CI; c++) xv[c] = x[e * ldx + c];
#pragma unroll
    for (int ky = 0; ky < 4; ky++) {
      const int oy = 2 * iy + ky - 1;
#pragma unroll
      for (int kx = 0; kx < 4; kx++) {
        const int ox = 2 * ix + kx - 1;
        if ((unsigned)oy < (unsigned)OH && (unsigned)ox < (unsigned)OW) {
          const float* gp = dz + ((b * OH + oy) * OW + ox) * lddz;
#pragma unroll
          for (int co = 0; co < CO; co++)
#pragma unroll
            for (int ci = 0; ci < CI; ci++) acc[ky * 4 + kx][co][ci] += gp[co] * xv[ci];
        }
      }
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  static_assert(NV == 64, "the transpose-reduce below maps value index == lane");
  // 64 wave sums as ONE halving butterfly (63 independent-per-level exchanges instead of 64 x 6 dependent ones, which
  // made this kernel ~20 us whatever the image size): afterwards lane l holds the wave total of value l.
  float v32[32], v16[16], v8[8], v4[4], v2[2];
  {
    const float* f = &acc[0][0][0];
    const bool hi = lane & 32;
#pragma unroll
    for (int i = 0; i < 32; i++) v32[i] = (hi ? f[32 + i] : f[i]) + __shfl_xor(hi ? f[i] : f[32 + i], 32);
  }
  {
    const bool hi = lane & 16;
#pragma unroll
    for (int i = 0; i < 16; i++) v16[i] = (hi ? v32[16 + i] : v32[i]) + __shfl_xor(hi ? v32[i] : v32[16 + i], 16);
  }
  {
    const bool hi = lane & 8;
#pragma unroll
    for (int i = 0; i < 8; i++) v8[i] = (hi ? v16[8 + i] : v16[i]) + __shfl_xor(hi ? v16[i] : v16[8 + i], 8);
  }
  {
    const bool hi = lane & 4;
#pragma unroll
    for (int i = 0; i < 4; i++) v4[i] = (hi ? v8[4 + i] : v8[i]) + __shfl_xor(hi ? v8[i] : v8[4 + i], 4);
  }
  {
    const bool hi = lane & 2;
#pragma unroll
    for (int i = 0; i < 2; i++) v2[i] = (hi ? v4[2 + i] : v4[i]) + __shfl_xor(hi ? v4[i] : v4[2 + i], 2);
  }
  {
    const bool hi = lane & 1;
    red[wid][lane] = (hi ? v2[1] : v2[0]) + __shfl_xor(hi ? v2[0] : v2[1], 1);
  }
  __syncthreads();
  if (threadIdx.x < NV)
    partial[(size_t)bx * NV + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
template <int CI, int CO>
__global__ __launch_bounds__(256) void tiny_deconv_wgrad_kernel(const float* __restrict__ x, int ldx,
                                                                const float* __restrict__ dz, int lddz,
                                                                float* __restrict__ partial, int B, int H, int W) {
  __shared__ float red[4][16 * CO * CI];
  tiny_deconv_wgrad_body<CI, CO>(x, ldx, dz, lddz, partial, B, H, W, blockIdx.x, gridDim.x, red);
}

// ---- every Cout = 2 filter gradient of a decoder in ONE launch (+ one launch for their partial sums) ----------------------
// flowN (3x3, C -> 2) and flowN_upM (conv_transpose 2 -> 2): ~0.7 GFLOP in total, but nine layers x (kernel + one or two
// partial-sum launches), each too small to fill the chip and 8-40 us long whatever its size.  All nine read data that is
// final once the coarsest head's gradient is (the loss pyramid wrote d flowN, the 2 -> 2 data gradients added theirs), so
// they run as one batch: a block finds its layer in a prefix table and runs that layer's body; <= 64 partials per layer,
// summed in a fixed order by the second launch.
constexpr int MAX_FLOW_WGRAD = 16;
struct FlowWgradDesc {
  SkinnyWgradParams p;      // head: x, dz, partial, dims; 2 -> 2: x = layer input (flow), dz = gradient of the 2H x 2W output
  float* out;               // dw
  int kind;                 // 0: head, strips of 8; 1: head, strips of 4; 2: conv_transpose 2 -> 2
  int spw, colblocks, chunks, block0, wsz, rblock0;
};
struct FlowWgradBatch {
  FlowWgradDesc d[MAX_FLOW_WGRAD];
  int n;
};
__global__ __launch_bounds__(256) void flow_wgrad_batched_kernel(const FlowWgradBatch b) {
  __shared__ float red[72 * 64];
  int di = 0;
  while (di + 1 < b.n && (int)blockIdx.x >= b.d[di + 1].block0) di++;
  const FlowWgradDesc& d = b.d[di];
  const int lb = blockIdx.x - d.block0;
  if (d.kind == 2) {
    tiny_deconv_wgrad_body<2, 2>(d.p.x, d.p.ldx, d.p.dz, d.p.lddz, d.p.partial, d.p.B, d.p.H, d.p.W, lb, d.chunks,
                                 reinterpret_cast<float(*)[64]>(red));
    return;
  }
  const int bx = lb % d.colblocks, by = lb / d.colblocks;
  if (d.kind == 0) head3_wgrad_strip_body<8>(d.p, d.spw, bx, by, red);
  else head3_wgrad_strip_body<4>(d.p, d.spw, bx, by, red);
}
// Sixteen lanes per element: lane q sums the partials q, q + 16, q + 32, ... (<= 16 loads, all in flight at once — under the
// backward pass's memory traffic a serial chain of 64 loads per thread made this kernel 80-100 us on the filter-gradient
// stream, 7 us alone), then the sixteen lane sums are combined by a fixed butterfly ((q0 + q1) + (q2 + q3)) + ... — the same
// value in every run.
__global__ __launch_bounds__(256) void flow_wgrad_sum_kernel(const FlowWgradBatch b) {
  int di = 0;
  while (di + 1 < b.n && (int)blockIdx.x >= b.d[di + 1].rblock0) di++;
  const FlowWgradDesc& d = b.d[di];
  const int e = (blockIdx.x - d.rblock0) * 16 + (threadIdx.x >> 4), q = threadIdx.x & 15;
  const bool ok = e < d.wsz;
  float v = 0.f;
  if (ok) {
#pragma unroll 16
    for (int k = q; k < d.chunks; k += 16) v += d.p.partial[(size_t)k * d.wsz + e];
  }
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) v += __shfl_down(v, off, 64);      // lane q = 0 of each 16-lane group ends with the tree sum
  if (ok && q == 0) d.out[e] = v;
}

// Batched column sums (all bias gradients of a step in one launch): block -> (descriptor, 64-col tile, row chunk).
constexpr int MAX_COLSUM = 32;
constexpr int COLSUM_MAX_CHUNKS = 256;
struct ColsumBatch {
  const float* x[MAX_COLSUM];
  float* out[MAX_COLSUM];
  long npix[MAX_COLSUM];
  int ld[MAX_COLSUM], C[MAX_COLSUM], chunks[MAX_COLSUM];
  int block0[MAX_COLSUM + 1];   // first block of each descriptor
  long part0[MAX_COLSUM];       // offset (floats) of each descriptor's partials [chunks][C]
  int n;
};

__global__ __launch_bounds__(256) void colsum_batched_kernel(const ColsumBatch d, float* __restrict__ partial) {
  __shared__ float red[16][64];
  int di = 0;
  while (di + 1 < d.n && (int)blockIdx.x >= d.block0[di + 1]) di++;
  const int lb = blockIdx.x - d.block0[di];
  const int C = d.C[di], ld = d.ld[di];
  const int ctiles = (C + 63) / 64;
  const int ct = lb % ctiles, chunk = lb / ctiles;
  const long per = (d.npix[di] + d.chunks[di] - 1) / d.chunks[di];
  const long r0 = chunk * per, r1 = min(d.npix[di], r0 + per);
  const float* x = d.x[di];
  const bool vec = (C % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<size_t>(x) & 15) == 0);
  if (vec) {
    // thread = (4 columns, one of 16 row lanes): 16-byte loads, 2 rows in flight per thread
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = ct * 64 + cq * 4;
    float4 s0 = make_float4(0, 0, 0, 0), s1 = s0;
    if (col < C) {
      long r = r0 + rl;
      for (; r + 16 < r1; r += 32) {
        const float4 a = *reinterpret_cast<const float4*>(x + r * ld + col);
        const float4 b = *reinterpret_cast<const float4*>(x + (r + 16) * ld + col);
        s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
      }
      if (r < r1) {
        const float4 a = *reinterpret_cast<const float4*>(x + r * ld + col);
        s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
      }
    }
    red[rl][cq * 4] = s0.x + s1.x; red[rl][cq * 4 + 1] = s0.y + s1.y;
    red[rl][cq * 4 + 2] = s0.z + s1.z; red[rl][cq * 4 + 3] = s0.w + s1.w;
  } else {
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = ct * 64 + cl;
    float s0 = 0.f;
    if (col < C)
      for (long r = r0 + rl; r < r1; r += 4) s0 += x[r * ld + col];
    red[rl][cl] = s0;
    for (int k = 4 + rl; k < 16; k += 4) red[k][cl] = 0.f;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int col = ct * 64 + threadIdx.x;
    if (col < C) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < 16; k++) v += red[k][threadIdx.x];
      partial[d.part0[di] + (size_t)chunk * C + col] = v;
    }
  }
}

// one block per (descriptor, 64 columns): 4 row lanes stride over the chunks, then a fixed-order combine
__global__ __launch_bounds__(256) void colsum_batched_final_kernel(const ColsumBatch d, const float* __restrict__ partial) {
  __shared__ float red[4][64];
  const int di = blockIdx.y;
  const int C = d.C[di];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  if (blockIdx.x * 64 >= C) return;
  float v = 0.f;
  if (col < C)
    for (int k = rl; k < d.chunks[di]; k += 4) v += partial[d.part0[di] + (size_t)k * C + col];
  red[rl][threadIdx.x & 63] = v;
  __syncthreads();
  if (rl == 0 && col < C)
    d.out[di][col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void leaky_bwd_inplace_kernel(float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy,
                                         long npix, int C) {
  const long n = npix * C;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const long px = e / C;
    const int c = (int)(e - px * C);
    dy[px * lddy + c] *= leaky_grad_from_out(y[px * ldy + c]);
  }
}

// ------------------------------------------------------------------ host side
// ---- planners

struct GatherPlan {
  int cfg;  // 0: 128x128, 1: 128x64, 2: 64x64
  int nsplit;
};

// The 128-row gather tiles (conv fwd / dgrad, deconv fwd / dgrad) compute on the bf16 matrix cores by default (3-way
// split, six terms: fp32-equivalent, see split_store); UNFLOW_CONV_MATH=fp32 selects v_mfma_f32_32x32x2_f32 for them too.
inline bool conv_math_bf16x3() { return !unflow::options().conv_math_fp32; }

// option wgrad_math_fp32 keeps the filter gradients on v_mfma_f32_32x32x2_f32
inline bool wgrad_math_bf16x3() { return conv_math_bf16x3() && !unflow::options().wgrad_math_fp32; }

inline GatherPlan plan_gather(const GatherParams& p) {
  GatherPlan pl;
  const long M = (long)p.B * p.Hg * p.Wg;
  if (p.N <= 32) pl.cfg = 2;
  else if (p.N <= 64) pl.cfg = 1;
  else pl.cfg = 0;
  int maxtaps = 0;
  for (int c = 0; c < p.ncls; c++) maxtaps = max(maxtaps, p.cls[c].nty * p.cls[c].ntx);
  const int KT = (maxtaps * p.Cs + BK - 1) / BK;
  const int min_kt = max(1, unflow::options().gather_min_kt);
  const int max_by_k = min(16, KT / min_kt > 0 ? KT / min_kt : 1);  // keep >= 8 K-tiles per split
  if (pl.cfg == 0) {
    const long b128 = ((M + 127) / 128) * ((p.N + 127) / 128) * p.ncls;
    if (b128 * max_by_k < 384) pl.cfg = 2;  // cannot fill half the chip with 128x128 tiles: smaller tiles
  }
  // (A 256x64 tile for N <= 64 with a long M — 4 accumulators per wave, the MFMA : LDS-read ratio of the 128x128 tile —
  // measured no gain on conv1 fwd / conv2 dgrad, 438 vs 441 pairs/s: dropped.)
  const int bm = pl.cfg == 2 ? 64 : 128, bn = pl.cfg == 0 ? 128 : 64;
  // bf16x3 tiles: 49 KB / 37 KB of LDS per block (swizzled, unpadded planes): the same residency as the fp32 tiles
  const int slots = 256 * (pl.cfg == 0 ? 3 : pl.cfg == 1 ? 4 : 6);
  const long blocks = ((M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * p.ncls;
  pl.nsplit = fill_one_round(blocks, slots, max_by_k);
  return pl;
}

inline size_t gather_partial_bytes(const GatherParams& p, int nsplit) {
  return nsplit > 1 ? (size_t)nsplit * p.B * p.Hd * p.Wd * p.N * sizeof(float) : 0;
}

// tile of the filter-gradient kernel: 0 = 128x128, 1 = 128x64 (a 256x128 tile, 8 accumulators per wave, measured slower)
inline int wgrad_cfg(const WgradParams& p) { return p.Cb <= 64 ? 1 : 0; }

inline int plan_wgrad(const WgradParams& p) {
  const int Mp = p.KH * p.KW * p.Ca;
  const int cfg = wgrad_cfg(p);
  const int bn = cfg == 1 ? 64 : 128;
  const long blocks = (long)cdiv(Mp, 128) * cdiv(p.Cb, bn);
  const long S = (long)p.B * p.Hg * p.Wg;
  const int KT = (int)((S + BK - 1) / BK);
  const int min_kt = max(1, unflow::options().wgrad_min_kt);    // 8 vs 4: +0.8 %
  const int max_by_k = min(256, KT / min_kt > 0 ? KT / min_kt : 1);
  const bool b3 = wgrad_math_bf16x3();                            // bf16x3: 49 / 37 KB of LDS per block
  const int slots = 256 * (b3 ? (cfg == 1 ? 4 : 3) : cfg == 1 ? 5 : 4);   // fp32: single-stage LDS 32 KB / 122 regs: 4 per CU
  return fill_one_round(blocks, slots, max_by_k);
}

inline size_t wgrad_partial_bytes(const WgradParams& p, int nsplit) {
  const size_t n = (size_t)p.KH * p.KW * p.Ca * p.Cb;
  return nsplit > 1 ? (size_t)nsplit * n * sizeof(float) + reduce_scratch_bytes(n, nsplit) : 0;
}

// pixel chunks of the skinny (Cout == 2) filter-gradient kernel: enough single-wave blocks to fill the chip
inline int skinny_wgrad_chunks(long npix, int Cin) {
  const long colblocks = (Cin / 4 + 63) / 64;
  long c = (4096 + colblocks - 1) / colblocks;
  c = min(c, max((long)1, npix / 8));
  return (int)min(c, (long)1024);
}

// strip size of the flow-head kernels: 8 when the level fills the chip with (row, strip) waves, else 4 (then the forward
// kernel also splits the channel groups over the 4 waves of a block); 0 = use the per-pixel kernels
inline int head_strip(int B, int H, int W, int k, int Cout) {
  if (k != 3 || Cout != 2) return 0;
  if (W % 8 == 0 && (long)B * H * W >= 16384) return 8;
  return W % 4 == 0 ? 4 : 0;
}
// blocks (= partials) of head3_wgrad_strip_kernel per channel-quad group: <= 1024 blocks, >= 4 strips per wave when
// there are enough strips
inline int head_wgrad_blocks(int B, int H, int W, int Cin, int S, int* strips_per_wave) {
  const long nstrips = (long)B * H * (W / S);
  const long colblocks = (Cin / 4 + 63) / 64;
  long blocks = min((long)1024, max((long)1, 1024 / colblocks));
  long spw = max((long)(nstrips >= 4096 ? 4 : 1), (nstrips + blocks * 4 - 1) / (blocks * 4));
  blocks = (nstrips + spw * 4 - 1) / (spw * 4);
  *strips_per_wave = (int)spw;
  return (int)blocks;
}

constexpr size_t COLSUM_SCRATCH_BYTES(int C) { return (size_t)REDUCE_FAN * C * sizeof(float) + 512; }

// ---- launchers
template <int BM, int BN, int WM, int WN, bool B_NK, int MATH = 0>
int launch_gather_cfg(const GatherParams& p, hipStream_t st) {
  const int M = p.B * p.Hg * p.Wg;
  const size_t smem = (MATH ? (size_t)3 * (BM + BN) * LDH * sizeof(unsigned short) : (size_t)(BM + BN) * LDK * sizeof(float)) +
                      BM * sizeof(int);
  static DynLdsBook attr_book{};   // once per instantiation
  (void)ensure_dyn_lds(reinterpret_cast<const void*>(&igemm_gather_kernel<BM, BN, WM, WN, B_NK, MATH>), (int)smem, attr_book);
  dim3 grid(cdiv(M, BM), cdiv(p.N, BN), p.ncls * p.nsplit);
  igemm_gather_kernel<BM, BN, WM, WN, B_NK, MATH><<<grid, 256, smem, st>>>(p);
  return launch_status();
}

template <bool B_NK>
int run_gather(GatherParams& p, void* ws, size_t ws_bytes, hipStream_t st) {
  p.cs_magic = magic_u32((unsigned)p.Cs);
  const GatherPlan pl = plan_gather(p);
  p.nsplit = pl.nsplit;
  p.partial = nullptr;
  if (p.nsplit > 1) {
    if (!ws || ws_bytes < gather_partial_bytes(p, p.nsplit)) p.nsplit = 1;  // no scratch: un-split (slower, same result up to fp32 order)
    else p.partial = reinterpret_cast<float*>(ws);
  }
  int code;
  switch (pl.cfg) {
    case 0: code = conv_math_bf16x3() ? launch_gather_cfg<128, 128, 64, 64, B_NK, 1>(p, st)
                                      : launch_gather_cfg<128, 128, 64, 64, B_NK>(p, st); break;
    case 1: code = conv_math_bf16x3() ? launch_gather_cfg<128, 64, 64, 32, B_NK, 1>(p, st)
                                      : launch_gather_cfg<128, 64, 64, 32, B_NK>(p, st); break;
    default: code = launch_gather_cfg<64, 64, 32, 32, B_NK>(p, st); break;
  }
  if (code != UNFLOW_OK) return code;
  if (p.nsplit > 1) {
    const size_t total = (size_t)p.B * p.Hd * p.Wd * p.N;
    const uintptr_t al = reinterpret_cast<uintptr_t>(p.dst) | reinterpret_cast<uintptr_t>(p.partial) |
                         reinterpret_cast<uintptr_t>(p.act_src);
    const bool vec = p.N % 4 == 0 && p.ldd % 4 == 0 && (!p.act_src || p.ld_act % 4 == 0) && (al & 15) == 0;
    if (vec) splitk_reduce_epilogue_kernel<<<stream_grid((long)(total / 4)), 256, 0, st>>>(p);
    else splitk_reduce_epilogue_scalar_kernel<<<stream_grid((long)total), 256, 0, st>>>(p);
    return launch_status();
  }
  return UNFLOW_OK;
}

template <int BM, int BN, int WM, int WN>
int launch_wgrad_b3_cfg(const WgradParams& p, hipStream_t st) {
  const int Mp = p.KH * p.KW * p.Ca;
  const size_t smem = (size_t)3 * (BM + BN) * LDH * sizeof(unsigned short);
  static DynLdsBook attr_book{};   // once per instantiation
  (void)ensure_dyn_lds(reinterpret_cast<const void*>(&igemm_wgrad_b3_kernel<BM, BN, WM, WN>), (int)smem, attr_book);
  dim3 grid(cdiv(Mp, BM), cdiv(p.Cb, BN), p.nsplit);
  igemm_wgrad_b3_kernel<BM, BN, WM, WN><<<grid, 256, smem, st>>>(p);
  return launch_status();
}

template <int BM, int BN, int WM, int WN>
int launch_wgrad_cfg(const WgradParams& p, hipStream_t st) {
  const int Mp = p.KH * p.KW * p.Ca;
  const size_t smem = (size_t)(BK * BM + BK * BN) * sizeof(float);
  static DynLdsBook attr_book{};   // once per instantiation
  (void)ensure_dyn_lds(reinterpret_cast<const void*>(&igemm_wgrad_kernel<BM, BN, WM, WN>), (int)smem, attr_book);
  dim3 grid(cdiv(Mp, BM), cdiv(p.Cb, BN), p.nsplit);
  igemm_wgrad_kernel<BM, BN, WM, WN><<<grid, 256, smem, st>>>(p);
  return launch_status();
}

// returns bytes of workspace consumed through *used
int run_wgrad(WgradParams& p, void* ws, size_t ws_bytes, size_t* used, hipStream_t st) {
  p.ca_magic = magic_u32((unsigned)p.Ca);
  const size_t wsize = (size_t)p.KH * p.KW * p.Ca * p.Cb;
  int ns = plan_wgrad(p);
  if (ns > 1 && (!ws || ws_bytes < wgrad_partial_bytes(p, ns))) {
    ns = ws ? (int)min((size_t)REDUCE_FAN, ws_bytes / (wsize * sizeof(float))) : 1;
    if (ns < 1) ns = 1;
  }
  p.nsplit = ns;
  p.partial = ns > 1 ? reinterpret_cast<float*>(ws) : nullptr;
  *used = wgrad_partial_bytes(p, ns);
  const int cfg = wgrad_cfg(p);
  const bool b3 = wgrad_math_bf16x3();
  const int code = b3 ? (cfg == 1 ? launch_wgrad_b3_cfg<128, 64, 64, 32>(p, st) : launch_wgrad_b3_cfg<128, 128, 64, 64>(p, st))
                 : cfg == 1 ? launch_wgrad_cfg<128, 64, 64, 32>(p, st)
                            : launch_wgrad_cfg<128, 128, 64, 64>(p, st);
  if (code != UNFLOW_OK) return code;
  if (ns > 1) return reduce_partials(p.partial, p.partial + (size_t)ns * wsize, p.out, wsize, ns, st);
  return UNFLOW_OK;
}

// bias gradient: column sums of dz [npix, C]; scratch placed after `ws_used` bytes of the workspace.
int run_colsum(const float* dz, int ld, long npix, int C, float* out, void* ws, size_t ws_bytes, size_t ws_used,
               hipStream_t st) {
  int chunks = (int)min((long)REDUCE_FAN, max((long)1, npix / 256));
  float* part = nullptr;
  const size_t need = (size_t)chunks * C * sizeof(float);
  const size_t off = (ws_used + 255) & ~(size_t)255;
  if (ws && ws_bytes >= off + need) part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + off);
  else chunks = 1;
  dim3 grid(cdiv(C, 64), chunks);
  colsum_partial_kernel<<<grid, 256, 0, st>>>(dz, ld, npix, C, chunks == 1 ? out : part);
  if (chunks > 1) sum_partials_kernel<<<stream_grid(C), 256, 0, st>>>(part, out, (size_t)C, chunks, REDUCE_FAN);
  return launch_status();
}

}  // namespace

// ===================================================================== C ABI
// *_po: the plain entry points plus optional output planes (called by the *_pl entry points of conv_planes.hip)
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

UNFLOW_API size_t unflow_conv_workspace_bytes(int B, int H, int W, int Cin, int Cout, int k, int stride) {
  // Exact requirement of the fwd / bwd_data / bwd_filter entry points of a conv2d with these dims
  // (and, for k == 4 && stride == 2, of the conv2d_transpose whose OUTPUT is [B,H,W,Cout]).
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || stride <= 0) return 0;
  size_t need = 4096;
  const int cmax = max(Cin, Cout);
  if (Cout <= 4 || Cin < 4) {
    // skinny kernels: wgrad partials [chunks][k*k*Cin*Cout] + tree-reduce scratch
    const size_t wsz = (size_t)k * k * max(Cin, 4) * max(Cout, 2);
    const int ch = skinny_wgrad_chunks((long)B * H * W, max(Cin, 4));
    need = max(need, (size_t)ch * wsz * sizeof(float) + reduce_scratch_bytes(wsz, ch));
  } else {
    GatherParams g{};
    build_conv_fwd(g, B, H, W, Cin, Cout, k, stride);
    need = max(need, gather_partial_bytes(g, plan_gather(g).nsplit));
    GatherParams d{};
    if ((stride == 1 || stride == 2) && build_conv_dgrad(d, B, H, W, Cin, Cout, k, stride) == UNFLOW_OK)
      need = max(need, gather_partial_bytes(d, plan_gather(d).nsplit));
    WgradParams w{};
    build_conv_wgrad(w, B, H, W, Cin, Cout, k, stride);
    need = max(need, wgrad_partial_bytes(w, plan_wgrad(w)));
    if (k == 4 && stride == 2 && H % 2 == 0 && W % 2 == 0) {
      GatherParams tf{};
      build_deconv_fwd(tf, B, H / 2, W / 2, Cin, Cout);
      need = max(need, gather_partial_bytes(tf, plan_gather(tf).nsplit));
      GatherParams td{};
      build_deconv_dgrad(td, B, H / 2, W / 2, Cin, Cout);
      need = max(need, gather_partial_bytes(td, plan_gather(td).nsplit));
      WgradParams tw{};
      build_deconv_wgrad(tw, B, H / 2, W / 2, Cin, Cout);
      need = max(need, wgrad_partial_bytes(tw, plan_wgrad(tw)));
    }
  }
  return need + COLSUM_SCRATCH_BYTES(cmax) + 1024;
}

UNFLOW_API int unflow_conv2d_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int B,
                                 int H, int W, int Cin, int Cout, int k, int stride, int leaky, void* workspace,
                                 size_t workspace_bytes, unflow_stream_t stream) {
  return unflow_conv2d_fwd_po(x, ldx, w, bias, y, ldy, B, H, W, Cin, Cout, k, stride, leaky, igemm::PlaneOut{}, workspace,
                              workspace_bytes, stream);
}

// The same op with optional 16-bit operand planes of the output (conv_planes.hip consumers); Cout <= 4 heads write none.
int unflow_conv2d_fwd_po(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int B, int H, int W,
                         int Cin, int Cout, int k, int stride, int leaky, const igemm::PlaneOut& po, void* workspace,
                         size_t workspace_bytes, unflow_stream_t stream) {
  if (!x || !w || !y) return UNFLOW_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || stride <= 0) return UNFLOW_ERR_SHAPE;
  if (Cin % 4 != 0 || ldx % 4 != 0 || ldx < Cin || ldy < Cout) return UNFLOW_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  if (Cout <= 4) {
    if (stride != 1 || leaky) return UNFLOW_ERR_UNSUPPORTED;
    int pt, pl, Ho, Wo;
    same_pads(H, k, 1, &pt, &Ho);
    same_pads(W, k, 1, &pl, &Wo);
    SkinnyParams p{x, ldx, w, bias, y, ldy, B, H, W, Cin, k, pt, pl};
    const int S = head_strip(B, H, W, k, Cout);
    if (S == 8) {
      const long jobs = (long)B * cdiv(H, HT_R) * (W / 8);      // 4 x 8-pixel tiles: filter resident, 1.5 reads per input row
      head3_fwd_tile_kernel<<<(int)((jobs + 3) / 4), 256, 0, st>>>(p);
      return launch_status();
    }
    if (S == 4) {
      head3_fwd_strip_kernel<4, true><<<B * H * (W / 4), 256, 0, st>>>(p);
      return launch_status();
    }
    const long waves = (long)B * H * W;
    const int grid = (int)min((long)4096, (waves + 3) / 4);
    if (Cout == 2) skinny_conv_fwd_kernel<2><<<grid, 256, 0, st>>>(p);
    else if (Cout == 1) skinny_conv_fwd_kernel<1><<<grid, 256, 0, st>>>(p);
    else if (Cout == 4) skinny_conv_fwd_kernel<4><<<grid, 256, 0, st>>>(p);
    else return UNFLOW_ERR_UNSUPPORTED;
    return launch_status();
  }
  if (Cout % 4 != 0) return UNFLOW_ERR_UNSUPPORTED;
  GatherParams p{};
  build_conv_fwd(p, B, H, W, Cin, Cout, k, stride);
  p.src = x; p.w = w; p.bias = bias; p.dst = y; p.act_src = nullptr;
  p.lds = ldx; p.ldd = ldy; p.leaky = leaky; p.accumulate = 0;
  p.pl = po;
  return run_gather<false>(p, workspace, workspace_bytes, st);
}

UNFLOW_API int unflow_conv2d_bwd_data(const float* dz, int lddz, const float* w, float* dx, int lddx, int B, int H,
                                      int W, int Cin, int Cout, int k, int stride, int accumulate,
                                      const float* act_src, int ld_act, int act_lo, int act_hi, void* workspace,
                                      size_t workspace_bytes, unflow_stream_t stream) {
  return unflow_conv2d_bwd_data_po(dz, lddz, w, dx, lddx, B, H, W, Cin, Cout, k, stride, accumulate, act_src, ld_act, act_lo,
                                   act_hi, igemm::PlaneOut{}, workspace, workspace_bytes, stream);
}

int unflow_conv2d_bwd_data_po(const float* dz, int lddz, const float* w, float* dx, int lddx, int B, int H, int W, int Cin,
                              int Cout, int k, int stride, int accumulate, const float* act_src, int ld_act, int act_lo,
                              int act_hi, const igemm::PlaneOut& po, void* workspace, size_t workspace_bytes,
                              unflow_stream_t stream) {
  if (!dz || !w || !dx) return UNFLOW_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || (stride != 1 && stride != 2)) return UNFLOW_ERR_SHAPE;
  if (Cin % 4 != 0 || lddx < Cin || lddz < Cout) return UNFLOW_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  if (Cout <= 4) {
    if (stride != 1) return UNFLOW_ERR_UNSUPPORTED;
    int pt, pl, Ho, Wo;
    same_pads(H, k, 1, &pt, &Ho);
    same_pads(W, k, 1, &pl, &Wo);
    SkinnyBwdParams p{dz, lddz, w, dx, lddx, act_src, ld_act, act_lo, act_hi, accumulate, B, H, W, Cin, k, pt, pl, po};
    const int S = (lddx % 4 == 0 && (!act_src || ld_act % 4 == 0)) ? head_strip(B, H, W, k, Cout) : 0;
    if (S) {
      // (round 6 measured and dropped: a wave walking 2 or 4 strips with its filter values resident — flow2's data gradient 37.8 ->
      // 44.8 us, flow3's 24.6 -> 26.7: fewer, longer waves hide less of the store latency than the filter re-loads cost)
      const int strips = W / S, ns = 1;
      const long jobs = (long)B * H * (strips / ns) * cdiv(Cin / 4, 64);
      if (S == 8) head3_dgrad_strip_kernel<8><<<(int)((jobs + 3) / 4), 256, 0, st>>>(p, ns);
      else head3_dgrad_strip_kernel<4><<<(int)((jobs + 3) / 4), 256, 0, st>>>(p, ns);
      return launch_status();
    }
    const long total = (long)B * H * W * (Cin / 4);
    if (Cout == 2) skinny_conv_dgrad_kernel<2><<<stream_grid(total), 256, 0, st>>>(p);
    else if (Cout == 1) skinny_conv_dgrad_kernel<1><<<stream_grid(total), 256, 0, st>>>(p);
    else if (Cout == 4) skinny_conv_dgrad_kernel<4><<<stream_grid(total), 256, 0, st>>>(p);
    else return UNFLOW_ERR_UNSUPPORTED;
    return launch_status();
  }
  if (Cout % 4 != 0 || lddz % 4 != 0) return UNFLOW_ERR_UNSUPPORTED;
  if (k == 1 && stride == 1 && Cout == PW_CO && lddx % 4 == 0 && (!act_src || ld_act % 4 == 0) &&
      (reinterpret_cast<uintptr_t>(dx) & 15) == 0 && (reinterpret_cast<uintptr_t>(act_src) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
    SkinnyBwdParams sp{dz, lddz, w, dx, lddx, act_src, ld_act, act_lo, act_hi, accumulate, B, H, W, Cin, 1, 0, 0, po};
    const long jobs = (((long)B * H * W + PW_S - 1) / PW_S) * cdiv(Cin / 4, 64);
    pointwise32_dgrad_kernel<<<(int)((jobs + 3) / 4), 256, 0, st>>>(sp);
    return launch_status();
  }
  GatherParams p{};
  const int bc = build_conv_dgrad(p, B, H, W, Cin, Cout, k, stride);
  if (bc != UNFLOW_OK) return bc;
  p.src = dz; p.w = w; p.bias = nullptr; p.dst = dx; p.act_src = act_src;
  p.lds = lddz; p.ldd = lddx; p.ld_act = ld_act; p.act_lo = act_lo; p.act_hi = act_hi;
  p.leaky = 0; p.accumulate = accumulate;
  p.pl = po;
  return run_gather<true>(p, workspace, workspace_bytes, st);
}

UNFLOW_API int unflow_conv2d_bwd_filter(const float* x, int ldx, const float* dz, int lddz, float* dw, float* dbias,
                                        int B, int H, int W, int Cin, int Cout, int k, int stride, void* workspace,
                                        size_t workspace_bytes, unflow_stream_t stream) {
  if (!x || !dz || !dw) return UNFLOW_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || stride <= 0) return UNFLOW_ERR_SHAPE;
  if (Cin % 4 != 0 || ldx % 4 != 0 || ldx < Cin || lddz < Cout) return UNFLOW_ERR_UNSUPPORTED;
  int pt, pl, Ho, Wo;
  same_pads(H, k, stride, &pt, &Ho);
  same_pads(W, k, stride, &pl, &Wo);
  hipStream_t st = as_stream(stream);
  size_t used = 0;
  if (Cout <= 4) {
    if (stride != 1 || k != 3 || Cout != 2) return UNFLOW_ERR_UNSUPPORTED;
    const long npix = (long)B * H * W;
    const size_t wsz = (size_t)9 * Cin * Cout;
    const int S = head_strip(B, H, W, k, Cout);
    const bool strip = S != 0;
    int spw = 1;
    const int chunks = strip ? head_wgrad_blocks(B, H, W, Cin, S, &spw) : skinny_wgrad_chunks(npix, Cin);
    used = (size_t)chunks * wsz * sizeof(float) + reduce_scratch_bytes(wsz, chunks);
    if (!workspace || workspace_bytes < used) return UNFLOW_ERR_WORKSPACE;
    SkinnyWgradParams p{x, ldx, dz, lddz, reinterpret_cast<float*>(workspace), B, H, W, Cin, pt, pl};
    dim3 grid(cdiv(Cin / 4, 64), chunks);
    if (S == 8) head3_wgrad_strip_kernel<8><<<grid, 256, 0, st>>>(p, spw);
    else if (S == 4) head3_wgrad_strip_kernel<4><<<grid, 256, 0, st>>>(p, spw);
    else skinny_conv_wgrad3_kernel<2><<<grid, 64, 0, st>>>(p);
    const int rc = reduce_partials(p.partial, p.partial + (size_t)chunks * wsz, dw, wsz, chunks, st);
    if (rc != UNFLOW_OK) return rc;
  } else {
    if (Cout % 4 != 0 || lddz % 4 != 0) return UNFLOW_ERR_UNSUPPORTED;
    WgradParams p{};
    build_conv_wgrad(p, B, H, W, Cin, Cout, k, stride);
    p.src = x; p.dst = dz; p.out = dw; p.lds = ldx; p.ldd = lddz;
    const int code = run_wgrad(p, workspace, workspace_bytes, &used, st);
    if (code != UNFLOW_OK) return code;
  }
  if (dbias) return run_colsum(dz, lddz, (long)B * Ho * Wo, Cout, dbias, workspace, workspace_bytes, used, st);
  return launch_status();
}

UNFLOW_API int unflow_conv2d_transpose_fwd(const float* x, int ldx, const float* w, const float* bias, float* y,
                                           int ldy, int B, int H, int W, int Cin, int Cout, int leaky, void* workspace,
                                           size_t workspace_bytes, unflow_stream_t stream) {
  return unflow_conv2d_transpose_fwd_po(x, ldx, w, bias, y, ldy, B, H, W, Cin, Cout, leaky, igemm::PlaneOut{}, workspace,
                                        workspace_bytes, stream);
}

int unflow_conv2d_transpose_fwd_po(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int B, int H,
                                   int W, int Cin, int Cout, int leaky, const igemm::PlaneOut& po, void* workspace,
                                   size_t workspace_bytes, unflow_stream_t stream) {
  if (!x || !w || !y) return UNFLOW_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UNFLOW_ERR_SHAPE;
  hipStream_t st = as_stream(stream);
  if (Cin == 2 && Cout == 2) {
    if (leaky) return UNFLOW_ERR_UNSUPPORTED;
    tiny_deconv_fwd_kernel<2, 2><<<stream_grid((long)B * 4 * H * W), 256, 0, st>>>(x, ldx, w, bias, y, ldy, B, H, W, po);
    return launch_status();
  }
  if (Cin % 4 != 0 || ldx % 4 != 0 || ldx < Cin || ldy < Cout) return UNFLOW_ERR_UNSUPPORTED;
  GatherParams p{};
  build_deconv_fwd(p, B, H, W, Cin, Cout);
  p.src = x; p.w = w; p.bias = bias; p.dst = y; p.act_src = nullptr;
  p.lds = ldx; p.ldd = ldy; p.leaky = leaky; p.accumulate = 0;
  p.pl = po;
  return run_gather<true>(p, workspace, workspace_bytes, st);
}

UNFLOW_API int unflow_conv2d_transpose_bwd_data(const float* dz, int lddz, const float* w, float* dx, int lddx, int B,
                                                int H, int W, int Cin, int Cout, int accumulate, const float* act_src,
                                                int ld_act, int act_lo, int act_hi, void* workspace,
                                                size_t workspace_bytes, unflow_stream_t stream) {
  return unflow_conv2d_transpose_bwd_data_po(dz, lddz, w, dx, lddx, B, H, W, Cin, Cout, accumulate, act_src, ld_act, act_lo,
                                             act_hi, igemm::PlaneOut{}, workspace, workspace_bytes, stream);
}

int unflow_conv2d_transpose_bwd_data_po(const float* dz, int lddz, const float* w, float* dx, int lddx, int B, int H, int W,
                                        int Cin, int Cout, int accumulate, const float* act_src, int ld_act, int act_lo,
                                        int act_hi, const igemm::PlaneOut& po, void* workspace, size_t workspace_bytes,
                                        unflow_stream_t stream) {
  if (!dz || !w || !dx) return UNFLOW_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UNFLOW_ERR_SHAPE;
  hipStream_t st = as_stream(stream);
  if (Cin == 2 && Cout == 2) {
    if (act_src) return UNFLOW_ERR_UNSUPPORTED;
    tiny_deconv_dgrad_kernel<2, 2><<<stream_grid((long)B * H * W), 256, 0, st>>>(dz, lddz, w, dx, lddx, accumulate, B, H, W);
    return launch_status();
  }
  if (Cout % 4 != 0 || Cin % 4 != 0 || lddz % 4 != 0 || lddz < Cout || lddx < Cin) return UNFLOW_ERR_UNSUPPORTED;
  GatherParams p{};
  build_deconv_dgrad(p, B, H, W, Cin, Cout);
  p.src = dz; p.w = w; p.bias = nullptr; p.dst = dx; p.act_src = act_src;
  p.lds = lddz; p.ldd = lddx; p.ld_act = ld_act; p.act_lo = act_lo; p.act_hi = act_hi;
  p.leaky = 0; p.accumulate = accumulate;
  p.pl = po;
  return run_gather<false>(p, workspace, workspace_bytes, st);
}

UNFLOW_API int unflow_conv2d_transpose_bwd_filter(const float* x, int ldx, const float* dz, int lddz, float* dw,
                                                  float* dbias, int B, int H, int W, int Cin, int Cout, void* workspace,
                                                  size_t workspace_bytes, unflow_stream_t stream) {
  if (!x || !dz || !dw) return UNFLOW_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UNFLOW_ERR_SHAPE;
  hipStream_t st = as_stream(stream);
  size_t used = 0;
  if (Cin == 2 && Cout == 2) {
    const int blocks = (int)min((long)32, ((long)B * H * W + 255) / 256);   // <= 32 partials: the 1-block final sum stays short
    if (!workspace || workspace_bytes < (size_t)blocks * 64 * sizeof(float)) return UNFLOW_ERR_WORKSPACE;
    float* part = reinterpret_cast<float*>(workspace);
    tiny_deconv_wgrad_kernel<2, 2><<<blocks, 256, 0, st>>>(x, ldx, dz, lddz, part, B, H, W);
    sum_partials_kernel<<<1, 64, 0, st>>>(part, dw, 64, blocks, 128);
    used = (size_t)blocks * 64 * sizeof(float);
  } else {
    if (Cout % 4 != 0 || Cin % 4 != 0 || lddz % 4 != 0 || ldx % 4 != 0) return UNFLOW_ERR_UNSUPPORTED;
    WgradParams p{};
    build_deconv_wgrad(p, B, H, W, Cin, Cout);
    p.src = dz; p.dst = x; p.out = dw; p.lds = lddz; p.ldd = ldx;
    const int code = run_wgrad(p, workspace, workspace_bytes, &used, st);
    if (code != UNFLOW_OK) return code;
  }
  if (dbias) return run_colsum(dz, lddz, (long)B * 4 * H * W, Cout, dbias, workspace, workspace_bytes, used, st);
  return launch_status();
}

// Plan of one layer of the batch.  Blocks (= partials) per channel-quad group in proportion to the layer's strips, >= 4 strips
// per wave, <= 256: the kernel lasts as long as its longest block, and with the round-3 cap of 64 that was flow2's (12288 strips
// of 8 pixels x 196 channels = half the batch's bytes on 64 blocks = a quarter of the CUs: 97 us for 142 MB at B = 4, 210 us at
// B = 8).  Returns false when the layer needs the per-layer path (no strip form).
static bool flow_wgrad_plan(FlowWgradDesc& d, int kind_in, int B, int H, int W, int Cin) {
  if (kind_in == 1) {                      // conv_transpose 2 -> 2 over an H x W input
    d.kind = 2; d.colblocks = 1; d.spw = 0; d.wsz = 64;
    d.chunks = (int)min((long)32, ((long)B * H * W + 255) / 256);
    return true;
  }
  const int S = head_strip(B, H, W, 3, 2);
  if (S != 8 && S != 4) return false;
  d.kind = S == 8 ? 0 : 1;
  d.colblocks = cdiv(Cin / 4, 64);
  d.wsz = 9 * Cin * 2;
  const long nstrips = (long)B * H * (W / S);
  const long blocks = min((long)256, max((long)1, nstrips / 16));      // 4 waves per block, >= 4 strips per wave (a block ends in a cross-wave sum + an 18 KB partial)
  const long spw = (nstrips + blocks * 4 - 1) / (blocks * 4);
  d.spw = (int)spw;
  d.chunks = (int)((nstrips + spw * 4 - 1) / (spw * 4));
  return true;
}

UNFLOW_API size_t unflow_flow_wgrad_batched_workspace_bytes(int n, const int* kind, const int* B, const int* H, const int* W,
                                                            const int* Cin) {
  size_t need = 0;
  for (int i = 0; i < n; i++) {
    FlowWgradDesc d{};
    if (!flow_wgrad_plan(d, kind[i], B[i], H[i], W[i], Cin[i])) return 0;
    need += ((size_t)d.chunks * d.wsz * sizeof(float) + 255) & ~(size_t)255;
  }
  return need;
}

// kind[i] 0: flowN head (3x3, Cin -> 2, stride 1: x [B,H,W,Cin], dz [B,H,W,2], dw [3,3,Cin,2]);
//         1: flowN_upM (conv_transpose 2 -> 2: x [B,H,W,2], dz [B,2H,2W,2], dw [4,4,2,2]).
UNFLOW_API int unflow_flow_wgrad_batched(int n, const int* kind, const float* const* x, const int* ldx, const float* const* dz,
                                         const int* lddz, float* const* dw, const int* B, const int* H, const int* W, const int* Cin,
                                         void* workspace, size_t workspace_bytes, unflow_stream_t stream) {
  if (!kind || !x || !ldx || !dz || !lddz || !dw || !B || !H || !W || !Cin) return UNFLOW_ERR_NULL;
  if (n <= 0) return UNFLOW_OK;
  if (n > MAX_FLOW_WGRAD) return UNFLOW_ERR_UNSUPPORTED;
  FlowWgradBatch b{};
  b.n = n;
  size_t off = 0;
  int blocks = 0, rblocks = 0;
  for (int i = 0; i < n; i++) {
    FlowWgradDesc& d = b.d[i];
    if (!x[i] || !dz[i] || !dw[i]) return UNFLOW_ERR_NULL;
    if (B[i] <= 0 || H[i] <= 0 || W[i] <= 0 || Cin[i] <= 0) return UNFLOW_ERR_SHAPE;
    if (kind[i] == 0 && (Cin[i] % 4 != 0 || ldx[i] % 4 != 0 || ldx[i] < Cin[i] || lddz[i] < 2)) return UNFLOW_ERR_UNSUPPORTED;
    if (kind[i] == 1 && Cin[i] != 2) return UNFLOW_ERR_UNSUPPORTED;
    if (!flow_wgrad_plan(d, kind[i], B[i], H[i], W[i], Cin[i])) return UNFLOW_ERR_UNSUPPORTED;
    int pt, pl, ho, wo;
    same_pads(H[i], 3, 1, &pt, &ho);
    same_pads(W[i], 3, 1, &pl, &wo);
    const size_t bytes = ((size_t)d.chunks * d.wsz * sizeof(float) + 255) & ~(size_t)255;
    if (!workspace || off + bytes > workspace_bytes) return UNFLOW_ERR_WORKSPACE;
    d.p = SkinnyWgradParams{x[i], ldx[i], dz[i], lddz[i], reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + off),
                            B[i], H[i], W[i], Cin[i], pt, pl};
    off += bytes;
    d.out = dw[i];
    d.block0 = blocks;
    blocks += d.chunks * d.colblocks;
    d.rblock0 = rblocks;
    rblocks += cdiv(d.wsz, 16);
  }
  hipStream_t st = as_stream(stream);
  flow_wgrad_batched_kernel<<<blocks, 256, 0, st>>>(b);
  flow_wgrad_sum_kernel<<<rblocks, 256, 0, st>>>(b);
  return launch_status();
}

UNFLOW_API int unflow_leaky_bwd_inplace(float* dy, int lddy, const float* y, int ldy, long npix, int C,
                                        unflow_stream_t stream) {
  if (!dy || !y) return UNFLOW_ERR_NULL;
  if (npix <= 0 || C <= 0) return UNFLOW_OK;
  leaky_bwd_inplace_kernel<<<stream_grid(npix * C), 256, 0, as_stream(stream)>>>(dy, lddy, y, ldy, npix, C);
  return launch_status();
}

UNFLOW_API size_t unflow_colsum_batched_workspace_bytes(int n, const int* C) {
  size_t t = 0;
  for (int i = 0; i < n; i++) t += (size_t)COLSUM_MAX_CHUNKS * C[i];
  return t * sizeof(float) + 256;
}

UNFLOW_API int unflow_colsum_batched(int n, const float* const* x, const int* ld, const long* npix, const int* C,
                                     float* const* out, void* workspace, size_t workspace_bytes,
                                     unflow_stream_t stream) {
  if (!x || !ld || !npix || !C || !out || !workspace) return UNFLOW_ERR_NULL;
  if (n <= 0) return UNFLOW_OK;
  if (n > MAX_COLSUM) return UNFLOW_ERR_UNSUPPORTED;
  if (workspace_bytes < unflow_colsum_batched_workspace_bytes(n, C)) return UNFLOW_ERR_WORKSPACE;
  ColsumBatch d{};
  d.n = n;
  int blocks = 0;
  long poff = 0;
  for (int i = 0; i < n; i++) {
    if (!x[i] || !out[i] || C[i] <= 0 || npix[i] <= 0 || ld[i] < C[i]) return UNFLOW_ERR_SHAPE;
    d.x[i] = x[i]; d.out[i] = out[i]; d.npix[i] = npix[i]; d.ld[i] = ld[i]; d.C[i] = C[i];
    d.chunks[i] = (int)min((long)COLSUM_MAX_CHUNKS, max((long)1, npix[i] / 512));
    d.block0[i] = blocks;
    blocks += ((C[i] + 63) / 64) * d.chunks[i];
    d.part0[i] = poff;
    poff += (long)d.chunks[i] * C[i];
  }
  d.block0[n] = blocks;
  hipStream_t st = as_stream(stream);
  float* part = reinterpret_cast<float*>(workspace);
  colsum_batched_kernel<<<blocks, 256, 0, st>>>(d, part);
  int maxc = 0;
  for (int i = 0; i < n; i++) maxc = max(maxc, C[i]);
  colsum_batched_final_kernel<<<dim3((maxc + 63) / 64, n), 256, 0, st>>>(d, part);
  return launch_status();
}
