// Pieces shared by the operand-plane implicit-GEMM kernels (conv_planes.hip: one-shot workgroups; conv_streamk.hip: the
// persistent stream-K form): launch parameters, the work-order decode, the MFMA term schedule and the epilogues.
#pragma once
#include "igemm_shared.h"
#include "options.h"

namespace igemm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int LDH = BK;   // 16-bit elements per LDS row of the gather image (64 bytes)

struct PlGatherParams : GatherGeom {
  const unsigned short* src;   // source planes, channel 0 of the consumed slice; [pixel][lds] per plane
  long src_ps;                 // plane stride (elements)
  const unsigned short* w;     // weight planes [tap][N][Cs] (K-contiguous)
  long w_ps;
  const float* bias;
  float* dst;
  float* partial;
  const float* act_src;        // leaky-ReLU derivative taken from this fp32 activation (its sign), or
  const unsigned short* act_pl;   // ... from the activation's FIRST operand plane (bf16 hi / fp16: same sign, 2 bytes per element)
  int lds, ldd, ld_act, act_lo, act_hi;
  int nsplit;
  int leaky, accumulate;
  float src_inv;               // 1 / (scale of the source planes x scale of the weight planes): applied to the finished sums
  int vec_epi;                 // every row of dst / partial / act_src / output planes is 16-byte (planes: 8-byte) aligned, N % 4 == 0
  int tiles_y, tiles_x;        // halo kernel: 4 x 32-site tiles per image
  int xcd;                     // XCD-contiguous work order (xcd_remap + work_decode)
  int mt, nt;                  // M tiles, N tiles of the launch (the grid is 1-D: mt * nt * ncls * nsplit workgroups)
  int order;                   // work order (work_decode): 0 N tile fastest .. M tile slowest; 1 M tile fastest; 2 M groups
  int mgroup;                  // order 2: M tiles per group (one group per XCD)
  int tw_log;                  // gather kernel: 0 = an M tile is BM consecutive sites of the linear (b, y, x) order; else the
                               // tile is (BM >> tw_log) rows x (1 << tw_log) sites of one image (tiles_x, tiles_y per image)
  int gpx;                     // pixels per K granule along x (0: a granule is 8 channels of ONE pixel; 2: conv1 form, below)
  PlaneOut pl;
};

// LDS bytes of one gather block: the operand tiles, or (larger for n_planes == 1) the four wave-private staging areas of
// the epilogue (32 rows x (WN + 4) floats each); the destination-pixel table follows.
constexpr int pl_gather_main_bytes(int bm, int bn, int wn, int npl) {
  const int tiles = npl * (bm + bn) * LDH * 2, stage = 4 * 32 * (wn + 4) * 4;
  return tiles > stage ? tiles : stage;
}

// (xcd_remap: igemm_shared.h)
// The launch is a 1-D grid; the linear workgroup id is first made XCD-contiguous (xcd_remap: the workgroups one XCD runs
// are a contiguous run of the work order) and then decoded so that the workgroups resident together on an XCD share
// operands in its L2.  Orders (run_pl_gather picks one per layer):
//   0  N tile fastest, then parity class, K split, M tile: the consumers of one M tile's source pixels are adjacent;
//   1  M tile fastest, then N tile, class, split: the M tiles of one weight slice are adjacent and an XCD touches 1/8 of the
//      weights (deep layers: 28-56 MB of weight planes against 5 MB of activations; with the (x, y, z) grid dealt
//      round-robin the forward / data-gradient kernels of conv5..conv6_1 moved 210-345 MB each);
//   2  the M tiles are cut into 8 groups (one per XCD: neighbouring tiles share the vertical taps' rows); inside a group the
//      M tile runs fastest, then N tile, class, split: every weight slice is streamed once per XCD by all its M tiles in
//      step.  The grid is padded to whole groups; surplus workgroups return at once (m = -1).
__host__ __device__ __forceinline__ void work_decode(int v, int mt, int nt, int ncls, int nsplit, int order, int mgroup, int& m, int& n,
                                            int& c, int& s) {
  if (order == 0) {
    n = v % nt; v /= nt;
    c = v % ncls; v /= ncls;
    s = v % nsplit;
    m = v / nsplit;
    return;
  }
  if (order == 2) {
    const int per = mgroup * nt * ncls * nsplit;
    const int g = v / per;
    v -= g * per;
    const int mi = v % mgroup;
    v /= mgroup;
    m = g * mgroup + mi;
    if (m >= mt) m = -1;
  } else {
    m = v % mt; v /= mt;
  }
  n = v % nt; v /= nt;
  c = v % ncls;
  s = v / ncls;
}

// F16 with NPL > 1 (round 6, BASELINE configs[4]): the "planes" of an fp16 launch are NPL CONSECUTIVE 32-channel chunks of the
// one fp16 plane (plane stride = 32 elements): a K tile is then 32 NPL channels deep — NPL products (chunk t of A with chunk t
// of B) per fragment pair instead of the six cross terms of the bf16 split — with the kernel's staging, LDS image and loop
// unchanged: NPL x the MFMAs between two barriers (mfma_nt below is the term count of a (NPL, F16) pair).
constexpr int mfma_nt(int npl, bool f16) { return f16 ? npl : (npl == 3 ? 6 : 1); }
template <int NPL, bool F16>
__device__ __forceinline__ void mfma_terms(const s16x8 (&av)[NPL], const s16x8 (&bv)[NPL], f32x16& acc, int t) {
  if constexpr (F16) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[NPL > 1 ? t : 0]), __builtin_bit_cast(f16x8, bv[NPL > 1 ? t : 0]),
                                                 acc, 0, 0, 0);
  } else if constexpr (NPL == 1) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[0]), __builtin_bit_cast(bf16x8, bv[0]), acc, 0, 0, 0);
  } else {
    constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[ta[t]]), __builtin_bit_cast(bf16x8, bv[tb[t]]), acc,
                                                  0, 0, 0);
  }
}

// sign test on a 16-bit plane element (bf16 hi plane or fp16): the value is > 0 (tf.maximum(0.1 x, x) took the x branch)
__device__ __forceinline__ float leaky_grad_from_bits(unsigned h) { return ((h & 0x8000u) == 0u && (h & 0x7fffu) != 0u) ? 1.f : 0.1f; }

// bias / leaky-ReLU / accumulate / leaky derivative of four consecutive output channels; stores the fp32 result when the
// layer keeps one (dst may be NULL: tensors that only convolutions read live as operand planes alone) and returns it
__device__ __forceinline__ float4 epi_value4(const PlGatherParams& p, size_t px, int n, float4 v) {
  if (p.src_inv != 1.f) { v.x *= p.src_inv; v.y *= p.src_inv; v.z *= p.src_inv; v.w *= p.src_inv; }
  if (p.bias) { v.x += p.bias[n]; v.y += p.bias[n + 1]; v.z += p.bias[n + 2]; v.w += p.bias[n + 3]; }
  if (p.leaky) { v.x = leaky_relu(v.x); v.y = leaky_relu(v.y); v.z = leaky_relu(v.z); v.w = leaky_relu(v.w); }
  float4* d = reinterpret_cast<float4*>(p.dst + px * p.ldd + n);
  if (p.accumulate) {
    const float4 e = *d;
    v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
  }
  if (n + 3 >= p.act_lo && n < p.act_hi) {
    float gx = 1.f, gy = 1.f, gz = 1.f, gw = 1.f;
    bool have = false;
    if (p.act_src) {
      const float4 a = *reinterpret_cast<const float4*>(p.act_src + px * p.ld_act + n);
      gx = leaky_grad_from_out(a.x); gy = leaky_grad_from_out(a.y); gz = leaky_grad_from_out(a.z); gw = leaky_grad_from_out(a.w);
      have = true;
    } else if (p.act_pl) {
      const uint2 a = *reinterpret_cast<const uint2*>(p.act_pl + px * p.ld_act + n);
      gx = leaky_grad_from_bits(a.x & 0xffffu); gy = leaky_grad_from_bits(a.x >> 16);
      gz = leaky_grad_from_bits(a.y & 0xffffu); gw = leaky_grad_from_bits(a.y >> 16);
      have = true;
    }
    if (have) {
      if (n >= p.act_lo && n < p.act_hi) v.x *= gx;
      if (n + 1 >= p.act_lo && n + 1 < p.act_hi) v.y *= gy;
      if (n + 2 >= p.act_lo && n + 2 < p.act_hi) v.z *= gz;
      if (n + 3 >= p.act_lo && n + 3 < p.act_hi) v.w *= gw;
    }
  }
  if (p.dst) *d = v;
  return v;
}
__device__ __forceinline__ void epi_store4(const PlGatherParams& p, size_t px, int n, float4 v) {
  store_planes4(p.pl, px, n, epi_value4(p, px, n, v));
}
// eight consecutive channels n .. n+7 (n % 8 == 0 in the destination row): the fp32 halves as above, the planes as ONE
// 16-byte store per plane (the epilogues are store-issue bound on layers with few K tiles per output: conv1, the 64-channel
// decoder levels; a lane that owns 8 channels issues 5 stores where two 4-channel lanes issued 8)
__device__ __forceinline__ void epi_store8(const PlGatherParams& p, size_t px, int n, float4 v0, float4 v1) {
  v0 = epi_value4(p, px, n, v0);
  v1 = epi_value4(p, px, n + 4, v1);
  store_planes8(p.pl, px, n, v0, v1);
}

constexpr int AUX_SC1 = 16;    // cache-policy bits of the raw buffer intrinsics: sc1 = write-through store / L1-bypassing load

// Epilogue shared by the gather kernels: bias / leaky-ReLU / accumulate / leaky derivative, fp32 result + output planes, or
// the split-K partial.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <int WM, int WN>
__device__ __forceinline__ void pl_gather_epilogue(const PlGatherParams& p, f32x16 (&acc)[WM / 32][WN / 32], const int* pix,
                                                   unsigned short* smem16, int wm, int wn, int wid, int lane, int n0,
                                                   int split) {
  constexpr int TM = WM / 32, TN = WN / 32;
  const int l31 = lane & 31, lh = lane >> 5;
  const bool to_partial = p.nsplit > 1;
  if (p.vec_epi) {
    // Through LDS (free after the K loop; wave-private areas, no barrier): the accumulator layout — lane = column, 16
    // scattered rows — becomes lane = (row, 4 consecutive columns), so that every global access of the epilogue is a
    // 16-byte one (fp32) or an 8-byte one (each output plane) covering 256 / 128 contiguous bytes per row.  With one
    // dword + three 2-byte stores per ELEMENT the epilogue was store-issue bound: conv2's data gradient (25 M outputs)
    // took 614 us instead of 290.
    constexpr int EP = WN + 4;                 // staging row pitch (floats)
    constexpr int QPR = WN / 4, RPI = 64 / QPR;
    float* stg = reinterpret_cast<float*>(smem16) + wid * (32 * EP);
    const size_t npix_d = (size_t)p.B * p.Hd * p.Wd;
#pragma unroll
    for (int i = 0; i < TM; i++) {
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * EP + j * 32 + l31] = acc[i][j][r];
      if (!to_partial) {
        // final values: a lane owns 8 consecutive channels of a row (two 16-byte fp32 stores, one 16-byte store per plane)
        constexpr int OPR = WN / 8, RPI8 = 64 / OPR;
#pragma unroll
        for (int it = 0; it < 32 / RPI8; it++) {
          const int rr = it * RPI8 + lane / OPR, q = lane % OPR;
          const float4 v0 = *reinterpret_cast<const float4*>(stg + rr * EP + 8 * q);
          const float4 v1 = *reinterpret_cast<const float4*>(stg + rr * EP + 8 * q + 4);
          const int px = pix[wm * WM + i * 32 + rr];
          const int n = n0 + wn * WN + 8 * q;
          if (px < 0 || n >= p.N) continue;
          if (n + 8 <= p.N) epi_store8(p, (size_t)px, n, v0, v1);
          else epi_store4(p, (size_t)px, n, v0);          // (N % 4 == 0: the first half is whole)
        }
        continue;
      }
#pragma unroll
      for (int it = 0; it < 32 / RPI; it++) {
        const int rr = it * RPI + lane / QPR, q = lane % QPR;
        float4 v = *reinterpret_cast<const float4*>(stg + rr * EP + 4 * q);
        const int px = pix[wm * WM + i * 32 + rr];
        const int n = n0 + wn * WN + 4 * q;
        if (px < 0 || n >= p.N) continue;
        *reinterpret_cast<float4*>(p.partial + ((size_t)split * npix_d + px) * p.N + n) = v;
      }
    }
    return;
  }
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
          v *= p.src_inv;
          if (p.bias) v += p.bias[n];
          if (p.leaky) v = leaky_relu(v);
          float* d = p.dst + (size_t)px * p.ldd + n;
          if (p.accumulate) v += *d;
          if (n >= p.act_lo && n < p.act_hi) {
            if (p.act_src) v *= leaky_grad_from_out(p.act_src[(size_t)px * p.ld_act + n]);
            else if (p.act_pl) v *= leaky_grad_from_bits(p.act_pl[(size_t)px * p.ld_act + n]);
          }
          if (p.dst) *d = v;
          store_planes(p.pl, (size_t)px, n, v);
        }
      }
    }
}

// halo kernels: a workgroup instance owns a TH x TW-site tile; source pixels are staged as [pixel][32 channels + pad]
constexpr int TH = 4, TW = 32, TWL = 5;     // tile = TH x TW = 128 sites
constexpr int HPITCH = 40;                  // 16-bit elements per halo pixel row: 32 channels + 8 pad (80 bytes)

// LDS bytes of a 4-wave halo block before its pixel table: the operand tiles, or the epilogue's four wave-private staging areas
// (nbuf = 2: the fp16 double-buffered weight tile, halo_kernel.h)
constexpr int pl_halo_main_bytes(int bn, int wn, int npl, int hp, int nbuf = 1) {
  const int tiles = npl * (hp * HPITCH + nbuf * bn * LDH) * 2, stage = 4 * 32 * (wn + 4) * 4;
  return tiles > stage ? tiles : stage;
}

// source pixels of a block's halo image (largest class; accumulating classes share one image at the widest row pitch)
inline int pl_halo_pixels(const GatherGeom& p, int th = TH) {      // th: tile rows (8: the fp16 tall tile)
  int hp = 0, mty = 0, mtx = 0;
  for (int c = 0; c < p.ncls; c++) {
    hp = max(hp, (th + p.cls[c].nty - 1) * (TW + p.cls[c].ntx - 1));
    mty = max(mty, p.cls[c].nty); mtx = max(mtx, p.cls[c].ntx);
  }
  return p.acc ? (th + mty - 1) * (TW + mtx - 1) : hp;      // accumulating classes share one halo image (widest row pitch)
}

// workgroups of a launch (q.mt, q.nt set): order 2 pads the M tiles to whole groups
inline int pl_grid_classes(const GatherGeom& q) { return q.acc ? 1 : q.ncls; }   // accumulating classes share a block

inline int pl_grid(PlGatherParams& q) {
  int mt = q.mt;
  if (q.order == 2) {
    q.mgroup = q.mt >= 16 ? cdiv(q.mt, 8) : q.mt;
    mt = cdiv(q.mt, q.mgroup) * q.mgroup;
  }
  return mt * q.nt * pl_grid_classes(q) * q.nsplit;
}

// ---- persistent stream-K halo kernel (conv_streamk.hip)
bool pl_halo_sk_ok(const GatherGeom& p, int npl, int bn);      // eligible and expected to pay (option streamk)
size_t pl_halo_sk_ws_bytes();                                   // slabs + arrival flags
int launch_pl_halo_sk(const PlGatherParams& p, void* ws, size_t ws_bytes, hipStream_t st);      // p.tiles_y / tiles_x set
// ---- fp16 halo kernel with the 256-site x 128 tile (conv_halo_tall.hip): p.tiles_x / nsplit / partial set (tiles_y is its own)
constexpr int TALL_TH = 8;
int launch_pl_halo_f16_tall(const PlGatherParams& p, hipStream_t st);

}  // namespace igemm
