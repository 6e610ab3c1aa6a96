// Process-wide options of libunflow_hip.so: ONE struct, written only through the exported unflow_set_option(name, value)
// (include/unflow_hip.h) — the library itself never reads the process environment.  The host layer (unflow_amd/_lib.py)
// sets them once, right after dlopen and before the first launch; an option changed later applies to the launches that
// follow it (options are read at launch-planning time, never inside a kernel).  Defaults are the measured-best settings
// on MI355X for FlowNetC 384x512; everything else is an A/B switch kept for the profiles under profiles/.
#pragma once

#define UNFLOW_OPTION_LIST(X)                                                                                              \
  X(conv_math_fp32, 0)    /* 1: conv / deconv data paths on v_mfma_f32_32x32x2_f32 instead of the 3 x bf16 split */       \
  X(wgrad_math_fp32, 0)   /* 1: filter gradients on the fp32 MFMA (conv_math_fp32 implies it) */                          \
  X(corr_math_fp32, 0)    /* 1: correlation on the fp32 MFMA */                                                           \
  X(gather_min_kt, 8)     /* split-K of the gather kernels: at least this many K32 tiles per split */                     \
  X(gather_max_split, 16) /* ... and at most this many splits */                                                          \
  X(wgrad_min_kt, 8)      /* split-K of the filter gradients: at least this many 32-site tiles per split */               \
  X(gather_pp, 1)         /* ping-pong gather kernel (two 128 x 128 tiles per 8-wave workgroup): 1 where it pays, 2 wherever eligible */ \
  X(gather_pp_fill, 60)   /* ... gather_pp = 1: taken when one round of 8-wave workgroups fills at least this % of the CUs (equal tap counts) */ \
  X(f16_k64, 1)           /* fp16 halo launches with K tiles of 64 channels (two chunk planes): 0 never, 1 short items (conv_transpose forward, 3x3 stride-2 forward), 2 always */ \
  X(f16_wgrad_dma, 1)     /* fp16 filter gradients on the LDS-DMA kernels (stages of three 16-site groups): 1 the 4-wave kernel, 2 also the 8-wave ping-pong one by wgrad_pp's rule (measured slower: 12-24 MFMAs per slot do not cover a barrier), 0 register-staged */ \
  X(f16_tall, 0)          /* fp16 halo layers with N > 64 on the 256-site x 128 tile of conv_halo_tall.hip (a third of the operand bytes per MFMA): 1 by rule, 2 wherever it can run, 0 off */ \
  X(f16_db, 0)            /* fp16 one-plane halo launches with the weight tile double-buffered in LDS (halo_kernel.h, DB): loads issued a whole K tile ahead, one barrier per tile (measured neutral, profiles/r06_f16_knockouts.txt); 0 the single-buffer loop */ \
  X(gather_tile2d, 1)     /* 2-D site tiles for the plain gather kernel */                                                \
  X(conv1_direct, 1)      /* FlowNetC's first layer on its own kernel (conv_first.hip: rows staged once per tile, filter resident) */ \
  X(halo, 1)              /* halo kernel for source-stride-1 layers (0: plain gather everywhere) */                       \
  X(streamk, 1)           /* persistent stream-K halo kernel (conv_streamk.hip): 1 where it pays, 2 wherever eligible, 0 off */     \
  X(streamk_groups, 8)    /* ... M groups of its item order (one per XCD); 1: tap class slowest over the whole launch */         \
  X(halo_max_split, 16)   /* split-K of the one-shot halo kernel: at most this many splits */                                      \
  X(halo_s2, 1)           /* halo kernel also for source-stride-2 layers (four accumulating parity classes) */            \
  X(xcd_swizzle, 1)       /* XCD-contiguous work order */                                                                 \
  X(wgrad_dma, 1)         /* LDS-DMA filter-gradient kernel (0: register-staged) */                                       \
  X(wgrad_pp, 1)          /* ping-pong filter-gradient kernel (8 waves): 1 where it pays, 3 wherever eligible, 0 off */                                           \
  X(corr_nb, 1)           /* narrow-band correlation forward kernel */                                                    \
  X(corr_rs, 2)           /* ... stride_2 = 1, r <= 4: 1 = rows shared by a 4-row workgroup, 2 = at r = 4 an LDS ring of rows */ \
  X(corr_wb, 1)           /* wide-band correlation forward kernel */                                                      \
  X(corr_rw, 1)           /* ... its wave-pair-per-row form (C = 128 / 256) */                                            \
  X(corr_bwd_b128, 1)     /* 16-byte band loads in the correlation backward */                                            \
  X(corr_bwd_rot, -1)     /* rotated displacement-row order in the correlation backward (-1: narrow band only) */         \
  X(corr_bwd_planes, 1)   /* correlation backward from the feature planes */                                              \
  X(corr_bwd_share, 1)    /* ... with the band operand fetched once per workgroup (C a multiple of 256) */

namespace unflow {
struct Options {
#define X(name, def) int name = def;
  UNFLOW_OPTION_LIST(X)
#undef X
};
// the one instance (options.hip); plain ints, set before the first launch
Options& options();
}  // namespace unflow
