// Geometry, planning and reduction helpers shared by the implicit-GEMM conv kernels: conv_igemm.hip (fp32 operands from
// HBM: fp32 MFMA, or the 3-way bf16 split done while staging) and conv_planes.hip (operands pre-split into 16-bit planes).
// Layer geometry follows slim.conv2d / conv2d_transpose with TF 'SAME' padding (src/e2eflow/core/flownet.py:89-237).
#pragma once
#include <stdlib.h>
#include <cstdint>
#include <cstring>
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace igemm {

constexpr int BK = 32;

struct TapClass {
  int nty, ntx;  // taps of this class
  int dy0, dx0;  // source offset of tap (0,0); tap (ty,tx) -> dy0 + ty*dstep
  int ky0, kx0;  // weight index of tap (0,0);  -> ky0 + ty*kstep
  int py, px;    // destination parity offset
};

// D[site, n] = sum_tap sum_c SRC[b, yg*sm + dy(tap), xg*sm + dx(tap), c] * W(tap, c, n): sites = a regular grid
// [B,Hg,Wg] written to destination pixel (yg*so + py, xg*so + px) of [B,Hd,Wd]; up to 4 output-parity classes.
struct GatherGeom {
  int B, Hg, Wg, Hs, Ws, sm;
  int dstep, kstep, KW;
  int Cs, N;
  int Hd, Wd, so;
  int ncls;
  int wtaps;  // taps of the whole weight tensor (KH*KW)
  int sp;     // halo kernel: pitch of the class's source pixels (1; 2 for the parity sub-lattices of a source-stride-2 layer)
  int acc;    // halo kernel: 1 = the classes accumulate into ONE output tile (one block walks them all), 0 = one class per block
  TapClass cls[4];
};

// dW[(tap,a), b] = sum_site SRC[gather(site, tap), a] * DST[site, b]
struct WgradGeom {
  int B, Hg, Wg, Hs, Ws, sm;
  int KH, KW, dy0, dx0;  // tap (ky,kx) -> offset dy0 + ky
  int Ca, Cb;
};

// Optional second output of an epilogue: the value's 16-bit operand planes for channels [lo, hi) of the destination
// (n_planes 3: bf16 hi/mid/lo with x = hi + mid + lo exactly; 1: fp16), plane p at base + p*plane_stride (elements).
struct PlaneOut {
  unsigned short* base;
  long plane_stride;
  int ld, lo, hi, n_planes;   // n_planes == 0: no plane output
  float scale;                // fp16 planes hold scale * value (0 or 1: unscaled; gradient tensors: a power of two)
};
__device__ __forceinline__ float plane_scaled(const PlaneOut& o, float v) { return o.scale != 0.f ? v * o.scale : v; }

__device__ __forceinline__ unsigned fast_div(unsigned a, unsigned magic) { return __umulhi(a, magic); }

// Raw buffer loads: 32-bit byte offsets against a wave-uniform descriptor; an offset >= num_records returns 0
// (hardware bounds check), which replaces every predicate/select of the zero-padding logic.  OOB_MARK is added
// to (or used as) an offset to force that; two marks sum to 2^31, still out of range.
constexpr int OOB_MARK = 0x40000000;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x3fffffffu ? 0x3fffffffu : bytes), 0x00020000);
}
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ u32x4 buf_ld16(__amdgpu_buffer_rsrc_t r, int voff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
}
// 16-byte buffer store whose data registers stay allocated for two more issue cycles.
// WHY (root cause of conv_first.hip's round-4 "race"): a VMEM store of more than 64 data bits reads its data
// VGPRs over several cycles after issue (four lanes of each 16-lane row per cycle), so a VALU write of one of them in the
// next cycle is stored instead of the data in the rows' last lanes.  The ISA documents the hazard (1 wait state) and LLVM
// pads for it — EXCEPT when the store carries an SGPR soffset (GCNHazardRecognizer::createsVALUHazard: "the hardware takes an
// extra cycle"), and on gfx950 that exemption is wrong: `buffer_store_dwordx4 v[96:99], v118, s[28:31], s43 offen` followed at
// once by `v_mov_b32 v96, 0x40000000` stored 0x40000000 as dword 0 of lanes 12-15 / 28-31 / 44-47 / 60-63, a few hundred to a
// few thousand times per launch, timing-dependent.  The s_nop takes the data as an operand: the registers cannot be reused
// before it, whatever the scheduler or a future compiler does around the store.  tools/isa_store_hazard.py scans the
// library's assembly for the pattern (tests/test_isa_hazard_cpu.py).
template <int AUX>
__device__ __forceinline__ void buf_st16_held(u32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, AUX);
  asm volatile("s_nop 1" ::"v"(v));
}
__device__ __forceinline__ float buf_ld1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// ---- LDS-DMA (buffer_load ... lds) as inline asm --------------------------------------------------------------------------
// hipcc counts a __builtin_amdgcn_raw_ptr_buffer_load_lds as a pending LDS write and puts s_waitcnt vmcnt(0) in front of the
// next ds_read of the same array — which drains the stage a double-buffered loop wants in flight.  As inline asm the loads
// are invisible to that bookkeeping; the kernel waits for them itself (s_waitcnt vmcnt(0) + s_barrier before the stage is
// read).  M0 = LDS byte address of the wave's 1 KB destination (lane i lands at M0 + 16 i); M0 is compiler-reserved, so it
// is saved and restored inside the statement.  Out-of-range offsets (>= num_records) write zeros.
__device__ __forceinline__ u32x4 raw_rsrc(const void* p, size_t bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  u32x4 r;
  r.x = (unsigned)a;
  r.y = (unsigned)(a >> 32) & 0xffffu;
  r.z = (unsigned)(bytes > 0x3fffffffu ? 0x3fffffffu : bytes);
  r.w = 0x00020000u;
  return r;
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}
// three 16-byte-per-lane loads (one per plane) with the same per-lane offset into three descriptors / LDS destinations
__device__ __forceinline__ void dma3(int voff, u32x4 r0, u32x4 r1, u32x4 r2, unsigned d0, unsigned d1, unsigned d2) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %[keep], m0\n\t"
      "s_mov_b32 m0, %[d0]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[r0], 0 offen lds\n\t"
      "s_mov_b32 m0, %[d1]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[r1], 0 offen lds\n\t"
      "s_mov_b32 m0, %[d2]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[r2], 0 offen lds\n\t"
      "s_mov_b32 m0, %[keep]"
      : [keep] "=&s"(keep)
      : [v] "v"(voff), [r0] "s"(r0), [r1] "s"(r1), [r2] "s"(r2), [d0] "s"(d0), [d1] "s"(d1), [d2] "s"(d2)
      : "memory");
}

// one such load
__device__ __forceinline__ void dma1(int voff, u32x4 r, unsigned d) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %[keep], m0\n\t"
      "s_mov_b32 m0, %[d]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[r], 0 offen lds\n\t"
      "s_mov_b32 m0, %[keep]"
      : [keep] "=&s"(keep)
      : [v] "v"(voff), [r] "s"(r), [d] "s"(d)
      : "memory");
}

// Workgroups are handed to the 8 XCDs round-robin by linear id (MI355X_MICROARCH.md), so neighbouring tiles — which share
// source rows (the vertical taps) — sit on different L2s.  This bijective remap of blockIdx.x gives every XCD a contiguous
// run of tiles instead: its L2 then fetches each source row once, not once per XCD that holds one of the row's consumers.
__host__ __device__ __forceinline__ int xcd_remap(int b, int n, int on) {
  if (!on || n < 16) return b;
  const int xcd = b & 7, idx = b >> 3;
  const int q = n >> 3, r = n & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- fp32 -> 16-bit operand planes -------------------------------------------------------------------------------------
// x = hi + mid + lo with three bf16 values (3 x 8 significand bits = the 24 of fp32, same exponent range); a*b is summed
// from the six terms hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid on v_mfma_f32_32x32x16_bf16 (fp32 accumulation); the
// dropped terms are <= 2^-23 |a||b|.  Measured (tools/microbench/bf16x3_accuracy.hip, K = 4608): rms error 2.5e-8 of
// sum|a||b| against 2.8e-8 for v_mfma_f32_32x32x2_f32 — the same accuracy class.
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// one value -> its three bf16 planes (round-to-nearest-even at every level; the two subtractions are exact)
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  const unsigned hh = cvt_pk_bf16(x, 0.f) & 0xffffu;
  const float r = x - __uint_as_float(hh << 16);
  const unsigned mm = cvt_pk_bf16(r, 0.f) & 0xffffu;
  const float s = r - __uint_as_float(mm << 16);
  h = (unsigned short)hh;
  m = (unsigned short)mm;
  l = (unsigned short)(cvt_pk_bf16(s, 0.f) & 0xffffu);
}
__device__ __forceinline__ unsigned short to_f16_bits(float x) {
  const _Float16 h = (_Float16)x;   // v_cvt_f16_f32, round to nearest even
  return __builtin_bit_cast(unsigned short, h);
}
// store the planes of one output element (channel n of destination pixel px); no-op outside [lo, hi) / without planes
__device__ __forceinline__ void store_planes(const PlaneOut& o, size_t px, int n, float v) {
  if (o.n_planes == 0 || n < o.lo || n >= o.hi) return;
  unsigned short* d = o.base + px * (size_t)o.ld + n;
  if (o.n_planes == 1) {
    *d = to_f16_bits(plane_scaled(o, v));
  } else {
    unsigned short h, m, l;
    split3(v, h, m, l);
    d[0] = h;
    d[o.plane_stride] = m;
    d[2 * o.plane_stride] = l;
  }
}
// four consecutive channels n .. n+3 (n % 4 == 0, 8-byte aligned destination): one 8-byte store per plane
__device__ __forceinline__ void store_planes4(const PlaneOut& o, size_t px, int n, const float4 v) {
  if (o.n_planes == 0 || n + 3 < o.lo || n >= o.hi) return;
  if (n < o.lo || n + 4 > o.hi) {   // straddles the range: element-wise
    store_planes(o, px, n, v.x); store_planes(o, px, n + 1, v.y); store_planes(o, px, n + 2, v.z); store_planes(o, px, n + 3, v.w);
    return;
  }
  unsigned short* d = o.base + px * (size_t)o.ld + n;
  if (o.n_planes == 1) {
    const unsigned a = (unsigned)to_f16_bits(plane_scaled(o, v.x)) | ((unsigned)to_f16_bits(plane_scaled(o, v.y)) << 16);
    const unsigned b = (unsigned)to_f16_bits(plane_scaled(o, v.z)) | ((unsigned)to_f16_bits(plane_scaled(o, v.w)) << 16);
    *reinterpret_cast<uint2*>(d) = make_uint2(a, b);
    return;
  }
  const unsigned h0 = cvt_pk_bf16(v.x, v.y), h1 = cvt_pk_bf16(v.z, v.w);
  const float r0 = v.x - __uint_as_float(h0 << 16), r1 = v.y - __uint_as_float(h0 & 0xffff0000u);
  const float r2 = v.z - __uint_as_float(h1 << 16), r3 = v.w - __uint_as_float(h1 & 0xffff0000u);
  const unsigned m0 = cvt_pk_bf16(r0, r1), m1 = cvt_pk_bf16(r2, r3);
  const float s0 = r0 - __uint_as_float(m0 << 16), s1 = r1 - __uint_as_float(m0 & 0xffff0000u);
  const float s2 = r2 - __uint_as_float(m1 << 16), s3 = r3 - __uint_as_float(m1 & 0xffff0000u);
  *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(d + o.plane_stride) = make_uint2(m0, m1);
  *reinterpret_cast<uint2*>(d + 2 * o.plane_stride) = make_uint2(cvt_pk_bf16(s0, s1), cvt_pk_bf16(s2, s3));
}

// eight consecutive channels n .. n+7 (16-byte aligned destination): one 16-byte store per plane
__device__ __forceinline__ void store_planes8(const PlaneOut& o, size_t px, int n, const float4 a, const float4 b) {
  if (o.n_planes == 0 || n + 7 < o.lo || n >= o.hi) return;
  const size_t e = px * (size_t)o.ld + n;
  if (n < o.lo || n + 8 > o.hi || (e & 7) != 0) {   // straddles the range / not 16-byte aligned: two 8-byte halves
    store_planes4(o, px, n, a);
    store_planes4(o, px, n + 4, b);
    return;
  }
  unsigned short* d = o.base + e;
  if (o.n_planes == 1) {
    u32x4 h;
    h.x = (unsigned)to_f16_bits(plane_scaled(o, a.x)) | ((unsigned)to_f16_bits(plane_scaled(o, a.y)) << 16);
    h.y = (unsigned)to_f16_bits(plane_scaled(o, a.z)) | ((unsigned)to_f16_bits(plane_scaled(o, a.w)) << 16);
    h.z = (unsigned)to_f16_bits(plane_scaled(o, b.x)) | ((unsigned)to_f16_bits(plane_scaled(o, b.y)) << 16);
    h.w = (unsigned)to_f16_bits(plane_scaled(o, b.z)) | ((unsigned)to_f16_bits(plane_scaled(o, b.w)) << 16);
    *reinterpret_cast<u32x4*>(d) = h;
    return;
  }
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  u32x4 h, m, l;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float x = v[2 * i], y = v[2 * i + 1];
    const unsigned hh = cvt_pk_bf16(x, y);
    const float rx = x - __uint_as_float(hh << 16), ry = y - __uint_as_float(hh & 0xffff0000u);
    const unsigned mm = cvt_pk_bf16(rx, ry);
    const float sx = rx - __uint_as_float(mm << 16), sy = ry - __uint_as_float(mm & 0xffff0000u);
    h[i] = hh; m[i] = mm; l[i] = cvt_pk_bf16(sx, sy);
  }
  *reinterpret_cast<u32x4*>(d) = h;
  *reinterpret_cast<u32x4*>(d + o.plane_stride) = m;
  *reinterpret_cast<u32x4*>(d + 2 * o.plane_stride) = l;
}

// ------------------------------------------------------------------ host side: geometry
inline void same_pads(int in, int k, int s, int* before, int* out) {
  const int o = (in + s - 1) / s;
  int total = (o - 1) * s + k - in;
  if (total < 0) total = 0;
  *before = total / 2;
  *out = o;
}

inline unsigned magic_u32(unsigned d) { return (unsigned)((0x100000000ull + d - 1) / d); }

inline void build_conv_fwd(GatherGeom& p, int B, int H, int W, int Cin, int Cout, int k, int stride) {
  int pt, pl, Ho, Wo;
  same_pads(H, k, stride, &pt, &Ho);
  same_pads(W, k, stride, &pl, &Wo);
  p.B = B; p.Hg = Ho; p.Wg = Wo; p.Hs = H; p.Ws = W; p.sm = stride;
  p.dstep = 1; p.kstep = 1; p.KW = k; p.Cs = Cin; p.N = Cout; p.wtaps = k * k;
  p.Hd = Ho; p.Wd = Wo; p.so = 1; p.ncls = 1; p.sp = 1; p.acc = 0;
  p.cls[0] = TapClass{k, k, -pt, -pl, 0, 0, 0, 0};
}

// The same layer for the halo kernel when stride == 2: source pixel 2*yg - pt + ky with ky = py + 2a is pixel (yg + a) of
// the parity sub-lattice {2*y' + py - pt}, so the conv is the sum over the four (py, px) of a stride-1 conv with
// ceil((k - py) / 2) x ceil((k - px) / 2) taps on its sub-lattice (pixel pitch 2): four ACCUMULATING tap classes.
inline bool build_conv_fwd_s2acc(GatherGeom& p, int B, int H, int W, int Cin, int Cout, int k) {
  if (k < 2) return false;
  build_conv_fwd(p, B, H, W, Cin, Cout, k, 2);
  const int pt = -p.cls[0].dy0, pl = -p.cls[0].dx0;
  p.sm = 1; p.sp = 2; p.acc = 1; p.ncls = 4; p.kstep = 2; p.dstep = 1;
  for (int c = 0; c < 4; c++) {
    const int py = c >> 1, px = c & 1;
    const int nty = (k - py + 1) / 2, ntx = (k - px + 1) / 2;
    if (nty < 1 || ntx < 1) return false;
    p.cls[c] = TapClass{nty, ntx, py - pt, px - pl, py, px, 0, 0};
  }
  return true;
}

inline int build_conv_dgrad(GatherGeom& p, int B, int H, int W, int Cin, int Cout, int k, int stride) {
  int pt, pl, Ho, Wo;
  same_pads(H, k, stride, &pt, &Ho);
  same_pads(W, k, stride, &pl, &Wo);
  p.B = B; p.Hs = Ho; p.Ws = Wo; p.sm = 1;
  p.dstep = -1; p.KW = k; p.Cs = Cout; p.N = Cin; p.wtaps = k * k;
  p.Hd = H; p.Wd = W; p.sp = 1; p.acc = 0;
  if (stride == 1) {
    p.Hg = H; p.Wg = W; p.so = 1; p.ncls = 1; p.kstep = 1;
    p.cls[0] = TapClass{k, k, pt, pl, 0, 0, 0, 0};
    return UNFLOW_OK;
  }
  // input pixel (2*yg+py): contributing taps ky == (py+pt) mod 2, source row yg + (py+pt-ky)/2
  if (H % 2 != 0 || W % 2 != 0) return UNFLOW_ERR_UNSUPPORTED;
  p.Hg = H / 2; p.Wg = W / 2; p.so = 2; p.ncls = 4; p.kstep = 2;
  for (int c = 0; c < 4; c++) {
    const int py = c >> 1, px = c & 1;
    const int ky0 = (py + pt) & 1, kx0 = (px + pl) & 1;
    const int nty = ky0 < k ? (k - ky0 + 1) / 2 : 0, ntx = kx0 < k ? (k - kx0 + 1) / 2 : 0;
    if (nty == 0 || ntx == 0) return UNFLOW_ERR_UNSUPPORTED;
    p.cls[c] = TapClass{nty, ntx, (py + pt - ky0) / 2, (px + pl - kx0) / 2, ky0, kx0, py, px};
  }
  return UNFLOW_OK;
}

// conv_transpose k4 s2 'SAME': oy = 2*iy + ky - 1
inline void build_deconv_fwd(GatherGeom& p, int B, int H, int W, int Cin, int Cout) {
  p.B = B; p.Hg = H; p.Wg = W; p.Hs = H; p.Ws = W; p.sm = 1;
  p.dstep = -1; p.kstep = 2; p.KW = 4; p.Cs = Cin; p.N = Cout; p.wtaps = 16;
  p.Hd = 2 * H; p.Wd = 2 * W; p.so = 2; p.ncls = 4; p.sp = 1; p.acc = 0;
  for (int c = 0; c < 4; c++) {
    const int py = c >> 1, px = c & 1;
    const int ky0 = (py + 1) & 1, kx0 = (px + 1) & 1;
    p.cls[c] = TapClass{2, 2, (py + 1 - ky0) / 2, (px + 1 - kx0) / 2, ky0, kx0, py, px};
  }
}

inline void build_deconv_dgrad(GatherGeom& p, int B, int H, int W, int Cin, int Cout) {
  p.B = B; p.Hg = H; p.Wg = W; p.Hs = 2 * H; p.Ws = 2 * W; p.sm = 2;
  p.dstep = 1; p.kstep = 1; p.KW = 4; p.Cs = Cout; p.N = Cin; p.wtaps = 16;
  p.Hd = H; p.Wd = W; p.so = 1; p.ncls = 1; p.sp = 1; p.acc = 0;
  p.cls[0] = TapClass{4, 4, -1, -1, 0, 0, 0, 0};
}

// ... for the halo kernel: dz pixel 2*iy - 1 + ky, ky = py + 2a: four accumulating 2 x 2-tap classes on the parity
// sub-lattices of dz (see build_conv_fwd_s2acc)
inline void build_deconv_dgrad_acc(GatherGeom& p, int B, int H, int W, int Cin, int Cout) {
  build_deconv_dgrad(p, B, H, W, Cin, Cout);
  p.sm = 1; p.sp = 2; p.acc = 1; p.ncls = 4; p.kstep = 2; p.dstep = 1;
  for (int c = 0; c < 4; c++) {
    const int py = c >> 1, px = c & 1;
    p.cls[c] = TapClass{2, 2, py - 1, px - 1, py, px, 0, 0};
  }
}

inline void build_conv_wgrad(WgradGeom& p, int B, int H, int W, int Cin, int Cout, int k, int stride) {
  int pt, pl, Ho, Wo;
  same_pads(H, k, stride, &pt, &Ho);
  same_pads(W, k, stride, &pl, &Wo);
  p.B = B; p.Hg = Ho; p.Wg = Wo; p.Hs = H; p.Ws = W; p.sm = stride;
  p.KH = k; p.KW = k; p.dy0 = -pt; p.dx0 = -pl; p.Ca = Cin; p.Cb = Cout;
}

// dW[ky,kx,co,ci] = sum_{input sites} dz[2iy+ky-1, 2ix+kx-1, co] * x[iy,ix,ci]
inline void build_deconv_wgrad(WgradGeom& p, int B, int H, int W, int Cin, int Cout) {
  p.B = B; p.Hg = H; p.Wg = W; p.Hs = 2 * H; p.Ws = 2 * W; p.sm = 2;
  p.KH = 4; p.KW = 4; p.dy0 = -1; p.dx0 = -1; p.Ca = Cout; p.Cb = Cin;
}

// ------------------------------------------------------------------ host side: split planning
// Resident workgroups per launch ("slots"): 256 CUs x blocks per CU of a tile config.  A split count is chosen so the
// launch is ONE full round of resident blocks (blocks * nsplit <= slots, as close as possible): 1088 blocks on 768 slots
// run as 1.4 rounds.
inline int fill_one_round(long blocks, int slots, int max_split) {
  if (blocks >= slots) return 1;
  int ns = (int)(slots / blocks);
  if (ns > max_split) ns = max_split;
  if (ns < 1) ns = 1;
  // A badly filled single round (288 blocks on 512 slots): a split that runs 2-3 well-filled rounds wins.  Take it only
  // for a clear gain, and the fewest splits that get it.
  auto eff = [&](int n) { const long b = blocks * n; return (double)b / (double)(((b + slots - 1) / slots) * slots); };
  const double e1 = eff(ns);
  if (e1 < 0.8) {
    int best = ns;
    double be = e1;
    for (int n = ns + 1; n <= max_split && blocks * n <= 3L * slots; n++)
      if (eff(n) > be + 0.02) { best = n; be = eff(n); }
    if (be >= e1 + 0.15) ns = best;
  }
  return ns;
}

// ------------------------------------------------------------------ fixed-order partial sums (deterministic split-K)
constexpr int REDUCE_FAN = 32;

// out[g][e] = sum_{k < fan} partial[g*fan + k][e]  (fixed order; g < ceil(S/fan)).  With S <= fan this is the
// final sum.  Applied repeatedly it is a deterministic tree reduction whose serial depth is <= fan.
static __global__ void sum_partials_kernel(const float* __restrict__ partial, float* __restrict__ out, size_t n, int S,
                                           int fan) {
  const int G = (S + fan - 1) / fan;
  const size_t total = n * (size_t)G;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const size_t g = t / n, e = t - g * n;
    const int k1 = min(S, (int)(g + 1) * fan);
    float v = 0.f;
#pragma unroll 8
    for (int k = (int)g * fan; k < k1; k++) v += partial[(size_t)k * n + e];
    out[g * n + e] = v;
  }
}

typedef float f32x4_nt __attribute__((ext_vector_type(4)));
// float4 variant (n % 4 == 0): same fixed order per element, a quarter of the threads, 8 loads in flight.
static __global__ void sum_partials4_kernel(const float4* __restrict__ partial, float4* __restrict__ out, size_t nq, int S,
                                            int fan) {
  const int G = (S + fan - 1) / fan;
  const size_t total = nq * (size_t)G;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const size_t g = t / nq, e = t - g * nq;
    const int k1 = min(S, (int)(g + 1) * fan);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int k = (int)g * fan; k < k1; k++) {
      // partials are read once, the sums go to Adam a whole backward pass later: non-temporal both ways
      const f32x4_nt a = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(partial) + (size_t)k * nq + e);
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    f32x4_nt o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
    __builtin_nontemporal_store(o, reinterpret_cast<f32x4_nt*>(out) + g * nq + e);
  }
}

// Bytes of scratch reduce_partials needs after the S*n partials themselves.
inline size_t reduce_scratch_bytes(size_t n, int S) {
  return S > REDUCE_FAN ? 2 * (size_t)((S + REDUCE_FAN - 1) / REDUCE_FAN) * n * sizeof(float) : 0;
}

inline void launch_sum_partials(const float* partial, float* out, size_t n, int S, int G, hipStream_t st) {
  const bool vec = n % 4 == 0 && ((reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (vec)
    sum_partials4_kernel<<<stream_grid((long)(n / 4 * G)), 256, 0, st>>>(reinterpret_cast<const float4*>(partial),
                                                                        reinterpret_cast<float4*>(out), n / 4, S, REDUCE_FAN);
  else
    sum_partials_kernel<<<stream_grid((long)(n * G)), 256, 0, st>>>(partial, out, n, S, REDUCE_FAN);
}

// out[e] = sum_s partial[s][e]; `scratch` (reduce_scratch_bytes) is used when S > REDUCE_FAN.
inline int reduce_partials(const float* partial, float* scratch, float* out, size_t n, int S, hipStream_t st) {
  while (S > REDUCE_FAN) {
    const int G = (S + REDUCE_FAN - 1) / REDUCE_FAN;
    launch_sum_partials(partial, scratch, n, S, G, st);
    // next level reads `scratch`; its output must not alias: levels alternate between scratch halves
    partial = scratch;
    scratch = scratch + (size_t)G * n;
    S = G;
  }
  launch_sum_partials(partial, out, n, S, 1, st);
  return launch_status();
}

}  // namespace igemm
