#!/usr/bin/env python3
"""Golden vectors for BASELINE configs[3] at its benchmarked batch: FlowNetCSS 768x1024, B = 2, forward pass of the ORACLE
(oracle/model_ref.py) in fp64 AND in fp32 on the same inputs -> tests/golden/css_768x1024_b2_fp64.npz:

  loss64, loss32            the unsupervised loss of the step (default terms)
  fw64, bw64                final flows of the fp64 pass on the lattice [::8, ::8] (float32 copies of fp64 values, 96 x 128 x 2 per sample)
  epe32_fw, epe32_bw        mean end-point error of the fp32 oracle against the fp64 oracle over ALL pixels: the noise floor of an
                            fp32 evaluation of this graph — what tests/test_parity_fullsize_gpu.py compares the HIP path's error with
  epe32_lat_fw / _bw        the same over the lattice only (the test sees only the lattice)
  fmax                      max |final flow| of the fp64 pass

Weights: M.init_params_spec('CSS', seed 31) with every flow-head / flow-upsampler filter scaled by 0.3 (a trained stack refines by a
few pixels; unscaled random stacks reach ~700 px and sit on the kinks of the bilinear warp); images: tests/parity_util.images(2, 768,
1024, 32).  Both are regenerated from their seeds by the test — the fixture holds outputs only.  ~10 minutes on 8 cores.

    python tests/golden/make_css_fp64_fixture.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

B, H, W, SPEC, WSEED, ISEED, HEAD_SCALE, STEP = 2, 768, 1024, 'CSS', 31, 32, 0.3, 8


def css_params():
    from oracle import model_ref as M
    P = M.init_params_spec(SPEC, seed=WSEED)
    for k in P:
        if k.split('/')[-2].startswith('flow') and k.endswith('/weights'):
            P[k] = P[k] * HEAD_SCALE
    return P


def main():
    from oracle import model_ref as M
    from parity_util import images, oracle_step
    torch.set_num_threads(os.cpu_count() or 1)
    params = dict(flownet=SPEC, pyramid_loss=True, border_mask=True, ternary_weight=1.0, smooth_2nd_weight=3.0)
    P = css_params()
    im1, im2 = images(B, H, W, ISEED)
    t0 = time.time()
    loss32, fw32, bw32, _ = oracle_step(P, im1, im2, params, dtype=torch.float32, backward=False)
    print("fp32 oracle: loss %.6f  (%.0f s)" % (loss32, time.time() - t0), flush=True)
    t0 = time.time()
    loss64, fw64, bw64, _ = oracle_step(P, im1, im2, params, dtype=torch.float64, backward=False)
    print("fp64 oracle: loss %.6f  (%.0f s)" % (loss64, time.time() - t0), flush=True)

    def epe(a, b):
        return ((a.double() - b.double()) ** 2).sum(-1).sqrt().mean().item()
    out = dict(loss64=np.float64(loss64), loss32=np.float64(loss32),
               fw64=fw64[:, ::STEP, ::STEP].float().numpy(), bw64=bw64[:, ::STEP, ::STEP].float().numpy(),
               epe32_fw=np.float64(epe(fw32, fw64)), epe32_bw=np.float64(epe(bw32, bw64)),
               epe32_lat_fw=np.float64(epe(fw32[:, ::STEP, ::STEP], fw64[:, ::STEP, ::STEP])),
               epe32_lat_bw=np.float64(epe(bw32[:, ::STEP, ::STEP], bw64[:, ::STEP, ::STEP])),
               fmax=np.float64(max(fw64.abs().max().item(), bw64.abs().max().item())),
               meta=np.array([B, H, W, WSEED, ISEED, STEP], dtype=np.int64), head_scale=np.float64(HEAD_SCALE))
    path = os.path.join(ROOT, "tests", "golden", "css_768x1024_b2_fp64.npz")
    np.savez_compressed(path, **out)
    print({k: (float(v) if np.ndim(v) == 0 else v.shape) for k, v in out.items()}, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
