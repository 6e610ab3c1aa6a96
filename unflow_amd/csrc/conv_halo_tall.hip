// fp16 (BASELINE configs[4]) halo gather kernel with a 256-site x 128-channel workgroup tile (8 x 32 sites; four waves of 128 x 64).
//
// Why: at ONE product per fragment pair the fp16 halo kernel is bound by the bytes it pulls through the L1 / TA path beside its
// MFMAs, and most of those are the weight tile (8 KB per K tile of a 128 x 128 block, 2 KB of halo).  Knock-outs on the step's
// layers (profiles/r06_f16_knockouts.txt): the weight tile always the same lines (L1 hits) -5 %, no weight loads at all -30 ... -35 %,
// the loads a whole K tile ahead (double-buffered LDS) +-0, a 128 x 256 tile (same weight bytes per MFMA, 0.75 LDS fragment reads per MFMA)
// +20 ... +60 %.  A tile of twice the SITES halves the weight bytes per MFMA — and with them the LDS writes — and has the 128 x 64 wave tile's
// 0.75 fragment reads per MFMA as well.
//
// Its own translation unit because of ONE compiler setting: the epilogue's sub-tile loops (planes_shared.h) are `#pragma unroll`
// loops the default unroll threshold leaves rolled at TM = 4, which makes acc[TM][TN] dynamically indexed — 128 accumulator
// registers in scratch, the layers 8x slower.  -mllvm -pragma-unroll-threshold=262144 (build.py EXTRA_FLAGS) unrolls them; given
// to conv_planes.hip the same flag rewrites seven production kernels, here it touches these instantiations only.
#include "halo_kernel.h"

namespace igemm {

template <bool DB>
static int launch_tall(const PlGatherParams& p, hipStream_t st) {
  constexpr int BM = 256, BN = 128, WM = 128, WN = 64;
  const int hp = pl_halo_pixels(p, TALL_TH);
  const int smem = pl_halo_main_bytes(BN, WN, 1, hp, DB ? 2 : 1) + BM * 4 + 16;
  static DynLdsBook book{};
  (void)ensure_dyn_lds(reinterpret_cast<const void*>(&igemm_pl_halo_kernel<BN, WM, WN, 1, true, DB, BM>), smem, book);
  PlGatherParams q = p;
  q.tiles_y = cdiv(p.Hg, TALL_TH);
  q.mt = p.B * q.tiles_y * p.tiles_x; q.nt = cdiv(p.N, BN);
  const int grid = pl_grid(q);
  igemm_pl_halo_kernel<BN, WM, WN, 1, true, DB, BM><<<grid, 256, smem, st>>>(q, hp);
  return launch_status();
}

int launch_pl_halo_f16_tall(const PlGatherParams& p, hipStream_t st) {
  return unflow::options().f16_db > 0 ? launch_tall<true>(p, st) : launch_tall<false>(p, st);      // (weight tile double-buffered, halo_kernel.h)
}

}  // namespace igemm
