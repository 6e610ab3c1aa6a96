// FlowNetC's first layer (7 x 7 stride 2 over RGB0 -> 64 channels, flownet.py:204) as its own kernel.
//
// Through the gather kernel (conv_planes.hip, rgb4_form: two-pixel K granules, a tap row = 8 pixels x 4 channels = one K32 tile)
// a 128 x 64 output tile moves 7 tap rows x 128 sites x 64 B x 3 planes = 172 KB of gathered rows plus the 86 KB filter through
// L2 -> LDS: 3072 tiles x 258 KB = 0.79 GB in 93 us = 8.5 TB/s — the operand-stream bound of the deep layers, with
// the matrix cores at 29 %.  But the seven tap rows of an output row read overlapping input rows, and the filter is the same for
// every tile.  Here a workgroup (8 waves) owns an 8-row x 32-column output tile:
//   * its 21 input rows x 70 pixels x 4 channels x 3 planes (36 KB) are staged ONCE per tile — registers -> LDS, the next tile's
//     loads in flight under this tile's products — and serve all seven tap rows: a K16 step of a tap row is four pixels, a lane's
//     half two pixels = one 16-byte granule, and the 32 sites of an output row read 32 CONSECUTIVE granules (conflict-free);
//   * the whole filter (7 tap rows x 64 channels x 32 K x 3 planes = 86 KB, XOR-swizzled 64-byte rows) is loaded once per
//     workgroup, and workgroups are persistent (one per CU, tiles strided by the grid): 0.08 GB of operands instead of 0.79;
//   * no barrier inside a tile's 14 K16 steps (168 MFMAs per wave); the MFMA takes the FILTER as its row operand, so a lane ends
//     up with four consecutive output channels of ONE site per accumulator group: bias, leaky ReLU and the 3-way bf16 split
//     happen in registers, the planes pass through a wave-private staging slab (8-byte stores, 16-byte reads) and leave as
//     whole 128-byte lines (one pixel's 64 channels of one plane).
// Output: the bf16 x 3 operand planes only (conv1's activation has no fp32 copy in the step: engine.py planes_only); any other
// request takes the gather kernel.
#include <type_traits>
#include "igemm_shared.h"
#include "options.h"

namespace {
using namespace igemm;

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
// the staging slab is written as 8-byte and read as 16-byte vectors: both through may_alias types — type-based alias analysis
// (also the machine scheduler's) would otherwise be free to move the read-back above the stores it reads
typedef unsigned u32x2_ma __attribute__((ext_vector_type(2), may_alias));
typedef unsigned u32x4_ma __attribute__((ext_vector_type(4), may_alias));

struct First7Params {
  const unsigned short* x;     // input planes [B,H,W,4]
  long x_ps;
  const unsigned short* w;     // filter planes [7 tap rows][64][32 K] (K = 4 kx + channel; kx = 7 and rows 28..31 zero)
  long w_ps;
  const float* bias;           // [64] or NULL
  unsigned short* y;           // output planes [B,Ho,Wo,ldy], channels 0..63
  long y_ps;
  int ldy;
  int B, H, W, Ho, Wo, pad;    // pad: rows / pixels of zero padding above / left (2 for even sizes)
  int tiles_y, tiles_x, ntiles;
  int leaky;
};

constexpr int F_TH = 8, F_TW = 32;               // output tile: one row per wave
constexpr int F_ROWS = 2 * F_TH + 5;             // input rows of a tile
constexpr int F_GR = F_TW + 3;                   // 16-byte granules (two pixels) per input row: 70 pixels
constexpr int F_PITCH = 576;                     // bytes per staged input row (36 granules)
constexpr int F_HPLANE = F_ROWS * F_PITCH;       // 12096
constexpr int F_WPLANE = 7 * 64 * 64;            // 28672 bytes: [tap row][channel][32 K]
constexpr int F_SPITCH = 144;                    // staging: bytes per site (64 channels + 16 pad)
constexpr int F_STAGE = F_TW * F_SPITCH;         // per wave
constexpr int F_SMEM = 3 * F_WPLANE + 3 * F_HPLANE + 8 * F_STAGE;      // 86016 + 36288 + 36864 = 159168
constexpr int F_NLOAD = (3 * F_ROWS * F_GR + 511) / 512;               // halo granules per thread

__global__ __launch_bounds__(512) void conv_first7_kernel(const First7Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  unsigned char* Wl = fsm;
  unsigned char* Hl = fsm + 3 * F_WPLANE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  unsigned char* Sl = fsm + 3 * F_WPLANE + 3 * F_HPLANE + wid * F_STAGE;

  // one descriptor per tensor: the three planes lie plane_stride apart (the host checks that everything stays below 1 GB, the
  // out-of-range mark)
  const int x_pb = (int)(p.x_ps * 2), w_pb = (int)(p.w_ps * 2), y_pb = (int)(p.y_ps * 2);      // plane pitch in bytes
  const __amdgpu_buffer_rsrc_t x_rs = make_rsrc(p.x, (size_t)2 * x_pb + (size_t)p.B * p.H * p.W * 8);
  const __amdgpu_buffer_rsrc_t w_rs = make_rsrc(p.w, (size_t)2 * w_pb + (size_t)7 * 64 * 64);
  const __amdgpu_buffer_rsrc_t y_rs = make_rsrc(p.y, (size_t)2 * y_pb + (((size_t)p.B * p.Ho * p.Wo - 1) * (size_t)p.ldy + 64) * 2);
  // ---- the filter, once: granule g of row (tap row, channel n) sits at slot g ^ ((n >> 2) & 3) of its 64-byte row
  for (int idx = tid; idx < 3 * 7 * 64 * 4; idx += 512) {
    const int pl = idx / (7 * 64 * 4), rem = idx - pl * (7 * 64 * 4);
    const int row = rem >> 2, g = rem & 3, n = row & 63;
    const u32x4 v = buf_ld16(w_rs, pl * w_pb + rem * 16);
    *reinterpret_cast<u32x4*>(Wl + pl * F_WPLANE + row * 64 + ((g ^ ((n >> 2) & 3)) << 4)) = v;
  }

  // ---- a thread's share of a tile's input rows: granule idx = (plane, row, granule)
  int h_lds[F_NLOAD], h_row[F_NLOAD], h_px[F_NLOAD], h_src[F_NLOAD];      // h_src: plane offset in the source, or the mark (no granule)
#pragma unroll
  for (int i = 0; i < F_NLOAD; i++) {
    const int idx = tid + i * 512;
    const bool any = idx < 3 * F_ROWS * F_GR;
    const int pl = idx / (F_ROWS * F_GR), rem = idx - pl * (F_ROWS * F_GR);
    const int r = rem / F_GR, g = rem - r * F_GR;
    h_src[i] = any ? pl * x_pb : OOB_MARK;
    h_row[i] = r;
    h_px[i] = 2 * g;
    h_lds[i] = any ? pl * F_HPLANE + r * F_PITCH + g * 16 : -1;
  }
  u32x4 hl[F_NLOAD];
  auto tile_of = [&](int t, int& b, int& oy0, int& ox0) {
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    b = t / p.tiles_y;
    oy0 = ty * F_TH;
    ox0 = tx * F_TW;
  };
  auto load_rows = [&](int t) {                 // -> registers (zeros outside the image and past the last tile)
    int b, oy0, ox0;
    tile_of(t, b, oy0, ox0);
    const bool live = t < p.ntiles;
#pragma unroll
    for (int i = 0; i < F_NLOAD; i++) {
      const int yi = 2 * oy0 - p.pad + h_row[i], px = 2 * ox0 - p.pad + h_px[i];
      const bool ok = live && (unsigned)yi < (unsigned)p.H && (unsigned)px < (unsigned)p.W;
      hl[i] = buf_ld16(x_rs, ok ? h_src[i] + ((b * p.H + yi) * p.W + px) * 8 : OOB_MARK);
    }
  };

  float bias4[2][4][4];                         // bias of the lane's channels: [subtile][group][4]
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++)
#pragma unroll
      for (int e = 0; e < 4; e++) bias4[c][g4][e] = p.bias ? p.bias[32 * c + 8 * g4 + 4 * h + e] : 0.f;

  constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
  const int st_site = lane >> 3, st_gr = lane & 7;         // read-back of the staging slab: lane -> (site 8 j + lane / 8, granule lane % 8)
  const int sw = (l31 >> 2) & 3;                 // (channel 32 c + l31: the same swizzle for both subtiles)
  f32x16 acc[2], accp[2];
  int pb = 0, poy = 0, pox0 = 0;                 // the tile whose accumulators wait in accp
  // One tile's straight-line block.  MF: its 14 K16 steps (wave = output row wid; operand A = filter, 32 channels per subtile,
  // operand B = the row's 32 sites).  EPI: the PREVIOUS tile's epilogue from accp — bias, leaky ReLU, 3-way split, staging slab,
  // line stores — in the same block, so that its ~300 vector instructions and its LDS / store traffic issue between this tile's
  // MFMAs instead of after them with the matrix pipes idle (all eight waves run their phases in step: there is nobody else to
  // fill the pipe).  accp[c][e] = channel 32 c + (e & 3) + 8 (e >> 2) + 4 h of site l31 in output row poy.
  auto block = [&](auto mf_tag, auto epi_tag) __attribute__((always_inline)) {
    constexpr bool MF = decltype(mf_tag)::value, EPI = decltype(epi_tag)::value;
    unsigned pk[3][2][4][2];                     // [plane][subtile][group][pair]: two packed bf16 each
    if (EPI) {
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++)
#pragma unroll
          for (int pr = 0; pr < 2; pr++) {
            float a = accp[c][4 * g4 + 2 * pr] + bias4[c][g4][2 * pr], bb = accp[c][4 * g4 + 2 * pr + 1] + bias4[c][g4][2 * pr + 1];
            if (p.leaky) { a = leaky_relu(a); bb = leaky_relu(bb); }
            const unsigned hh = cvt_pk_bf16(a, bb);
            const float ra = a - __uint_as_float(hh << 16), rb = bb - __uint_as_float(hh & 0xffff0000u);
            const unsigned mm = cvt_pk_bf16(ra, rb);
            const float sa = ra - __uint_as_float(mm << 16), sb = rb - __uint_as_float(mm & 0xffff0000u);
            pk[0][c][g4][pr] = hh; pk[1][c][g4][pr] = mm; pk[2][c][g4][pr] = cvt_pk_bf16(sa, sb);
          }
    }
    if (MF) {
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[c][e] = 0.f;
    }
    const unsigned char* xrow = Hl + (2 * wid) * F_PITCH + (l31 + h) * 16;
    const unsigned char* wrow = Wl + l31 * 64;
    // fragments one K16 step ahead of the MFMAs (two register sets): the LDS reads of step k + 1 run under the products of step k
    s16x8 xf[2][3], wf[2][2][3];
    auto rd = [&](int k, s16x8 (&x)[3], s16x8 (&w)[2][3]) __attribute__((always_inline)) {
      const int ky = k >> 1, sk = k & 1;
#pragma unroll
      for (int pl = 0; pl < 3; pl++) {
        x[pl] = *reinterpret_cast<const s16x8*>(xrow + pl * F_HPLANE + ky * F_PITCH + sk * 32);
#pragma unroll
        for (int c = 0; c < 2; c++)
          w[c][pl] = *reinterpret_cast<const s16x8*>(wrow + pl * F_WPLANE + (ky * 64 + 32 * c) * 64 + (((2 * sk + h) ^ sw) << 4));
      }
    };
    if (MF) rd(0, xf[0], wf[0]);
#pragma unroll
    for (int ky = 0; ky < 7; ky++) {
      if (MF) {
#pragma unroll
        for (int sk = 0; sk < 2; sk++) {
          const int k = 2 * ky + sk;
          if (k + 1 < 14) rd(k + 1, xf[(k + 1) & 1], wf[(k + 1) & 1]);
#pragma unroll
          for (int tt = 0; tt < 6; tt++)
#pragma unroll
            for (int c = 0; c < 2; c++)
              acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[k & 1][c][ta[tt]]),
                                                               __builtin_bit_cast(bf16x8_t, xf[k & 1][tb[tt]]), acc[c], 0, 0, 0);
        }
      }
      if (EPI && ky >= 1 && ky <= 6) {           // plane (ky - 1) / 2: its slab stores with the odd tap row, read-back + line stores with the even one
        const int pl = (ky - 1) >> 1;
        if ((ky & 1) == 1) {
#pragma unroll
          for (int c = 0; c < 2; c++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++)
              *reinterpret_cast<u32x2_ma*>(Sl + l31 * F_SPITCH + (32 * c + 8 * g4 + 4 * h) * 2) = u32x2_ma{pk[pl][c][g4][0], pk[pl][c][g4][1]};
          asm volatile("" ::: "memory");           // (the read-back below uses another access type: no reordering across this line)
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int site = 8 * j + st_site, ox = pox0 + site;
            const u32x4 v = *reinterpret_cast<const u32x4_ma*>(Sl + site * F_SPITCH + st_gr * 16);
            const bool ok = poy < p.Ho && ox < p.Wo;
            const int voff = ok ? (((pb * p.Ho + poy) * p.Wo + ox) * p.ldy) * 2 + st_gr * 16 : OOB_MARK;
            buf_st16_held<0>(v, y_rs, voff, pl * y_pb);
          }
          asm volatile("" ::: "memory");
        }
      }
      // (Round 4 needed one scheduling region per tap row here — __builtin_amdgcn_sched_barrier(0) — to be correct, without knowing
      // why.  Round 5 found why: with the block as one region the register allocator put the NEXT store's address default
      // (`v_mov_b32 v96, 0x40000000`, the out-of-range mark) into the first data register of the plane store issued one
      // instruction earlier (`buffer_store_dwordx4 v[96:99], v118, s[28:31], s43 offen`).  A 16-byte store reads its data
      // registers over several cycles, four lanes of each 16-lane row per cycle; LLVM pads the documented 1 wait state only for
      // stores WITHOUT an SGPR soffset, and this one has one (the plane offset).  Result: dword 0 = 0x40000000 = bf16 (0, 2.0)
      // in lanes 12-15 / 28-31 / 44-47 / 60-63 of the second store (sites 8-15) of the middle plane — exactly the recorded
      // signature: sites 9 / 11 / 13 / 15, channels 32 + 8 g + {0, 1}, errors of 2.0, younger waves more often (issue
      // arbitration decides whether the v_mov lands in the next cycle).  buf_st16_held (igemm_shared.h) keeps the data registers
      // allocated past the store; tools/isa_store_hazard.py finds the pattern in the library's assembly: 1 site in the
      // one-region build of round 4, 0 with the per-row regions, 0 now.  profiles/r05_conv_first_root_cause.txt.)
    }
  };
  using T_ = std::true_type;
  using F_ = std::false_type;

  int t = blockIdx.x;
  bool have_prev = false;
  load_rows(t);
#pragma unroll 1
  for (; t < p.ntiles; t += gridDim.x) {
    __syncthreads();                             // every wave is done with the previous tile's rows (first pass: the filter stores)
    // (the compiler's own wait for this tile's rows counts the younger plane stores of the previous tile and lets them fly)
#pragma unroll
    for (int i = 0; i < F_NLOAD; i++)
      if (h_lds[i] >= 0) *reinterpret_cast<u32x4*>(Hl + h_lds[i]) = hl[i];
    __syncthreads();
    load_rows(t + gridDim.x);                    // the next tile's rows land under this tile's products
    if (have_prev) block(T_{}, T_{});
    else block(T_{}, F_{});
    int b, oy0, ox0;
    tile_of(t, b, oy0, ox0);
    pb = b; poy = oy0 + wid; pox0 = ox0;
#pragma unroll
    for (int c = 0; c < 2; c++) accp[c] = acc[c];
    have_prev = true;
  }
  if (have_prev) block(F_{}, T_{});
}

int first7_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

}  // namespace

// conv_planes.hip (unflow_conv2d_fwd_pl) asks here first for the rgb4 form.  Returns UNFLOW_ERR_UNSUPPORTED when the request is
// not this kernel's (then the gather kernel takes it).
int conv_first7_fwd(const unflow_planes* x_pl, const unflow_planes* w_pl, const float* bias, const unflow_planes* y_pl, int B, int H,
                    int W, int Cout, int leaky, hipStream_t st) {
  if (!unflow::options().conv1_direct) return UNFLOW_ERR_UNSUPPORTED;
  if (Cout != 64 || !x_pl || !w_pl || !y_pl || !y_pl->base || x_pl->n_planes != 3 || w_pl->n_planes != 3 || y_pl->n_planes != 3)
    return UNFLOW_ERR_UNSUPPORTED;
  if (x_pl->ld != 4 || w_pl->ld != 32 || y_pl->ld < 64 || y_pl->ld % 8 != 0 || H % 2 != 0 || W % 2 != 0) return UNFLOW_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(y_pl->base) & 15) || (y_pl->plane_stride & 7) || (reinterpret_cast<uintptr_t>(x_pl->base) & 15) ||
      (x_pl->plane_stride & 7) || (reinterpret_cast<uintptr_t>(w_pl->base) & 15) || (w_pl->plane_stride & 7))
    return UNFLOW_ERR_UNSUPPORTED;
  const float sx = x_pl->scale, sw = w_pl->scale, sy = y_pl->scale;
  if ((sx != 0.f && sx != 1.f) || (sw != 0.f && sw != 1.f) || (sy != 0.f && sy != 1.f)) return UNFLOW_ERR_UNSUPPORTED;
  const int Ho = H / 2, Wo = W / 2;
  if (x_pl->plane_stride < 0 || y_pl->plane_stride < 0 || w_pl->plane_stride < 0 ||
      (size_t)4 * x_pl->plane_stride + (size_t)B * H * W * 8 > 0x3fffffffu ||
      (size_t)4 * y_pl->plane_stride + ((size_t)B * Ho * Wo * y_pl->ld) * 2 > 0x3fffffffu || (size_t)4 * w_pl->plane_stride > 0x3ffffffu)
    return UNFLOW_ERR_UNSUPPORTED;
  First7Params p{};
  p.x = reinterpret_cast<const unsigned short*>(x_pl->base); p.x_ps = x_pl->plane_stride;
  p.w = reinterpret_cast<const unsigned short*>(w_pl->base); p.w_ps = w_pl->plane_stride;
  p.bias = bias;
  p.y = reinterpret_cast<unsigned short*>(y_pl->base); p.y_ps = y_pl->plane_stride; p.ldy = y_pl->ld;
  p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.pad = 2;      // SAME, k 7, stride 2, even sizes: 2 above / left, 3 below / right
  p.tiles_y = (Ho + F_TH - 1) / F_TH; p.tiles_x = (Wo + F_TW - 1) / F_TW;
  p.ntiles = B * p.tiles_y * p.tiles_x;
  p.leaky = leaky;
  static DynLdsBook attr_book{};
  (void)ensure_dyn_lds(reinterpret_cast<const void*>(&conv_first7_kernel), F_SMEM, attr_book);
  const int grid = p.ntiles < first7_cus() ? p.ntiles : first7_cus();
  conv_first7_kernel<<<grid, 512, F_SMEM, st>>>(p);
  return launch_status();
}
