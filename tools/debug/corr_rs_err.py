"""Max error of the +-4 planes forward against the C oracle on a few shapes, per kernel choice (UNFLOW_OPT_CORR_RS)."""
import sys, zlib
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from unflow_amd import _lib
from unflow_amd._lib import check, ptr, stream
from oracle import ops_ref
from test_planes_gpu import make_pt
dev = torch.device('cuda:0')
for case in [(2, 64, 12, 40), (2, 64, 6, 131), (2, 256, 6, 70), (4, 256, 21, 200)]:
    N, C, H, W = case
    attrs = dict(kernel_size=1, max_displacement=4, pad=4, stride_1=1, stride_2=1)
    B = N // 2
    rs = np.random.RandomState(zlib.crc32(str(case).encode()))
    feat = torch.from_numpy(rs.randn(N, H, W, C).astype(np.float32))
    F = make_pt(feat, dev, 3, extra=8)
    oc, oh, ow = ops_ref.correlation_out_shape(H, W, **attrs)
    x = np.ascontiguousarray(feat.numpy().transpose(0, 3, 1, 2))
    ref = ops_ref.correlation(x, np.ascontiguousarray(np.roll(x, -B, axis=0)), **attrs)
    for extra in (3, 0):
        out = torch.full((N, oh, ow, oc + extra), float('nan'), device=dev)
        check(_lib.lib().unflow_correlation_nhwc_fwd_pl(ptr(F.t), ptr(F.t), F.t.stride(2), _lib.planes_of(F.pl), _lib.planes_of(F.pl), B, ptr(out),
                                                        oc + extra, N, C, H, W, 1, 4, 4, 1, 1, stream()), "corr")
        got = out[..., :oc].permute(0, 3, 1, 2).cpu().numpy()
        err = np.abs(got - ref)
        bad = np.argwhere(~(err <= 2e-5 * max(1.0, np.abs(ref).max())))
        print(case, 'extra', extra, 'max err', np.nanmax(err), 'nan', int(np.isnan(got).sum()), 'bad', len(bad), bad[:6].tolist())
