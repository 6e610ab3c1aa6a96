"""CPU tests of the (f2) groundwork: .npz parameter files keyed by the reference's variable names, .flo and KITTI flow-PNG
readers (fixtures under tests/golden, written by tests/golden/make_io_fixtures.py), resize_output_flow, outlier metrics,
the product learning-rate schedule."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_flo_fixture_and_roundtrip(tmp_path):
    from unflow_amd.core import input as I
    flow, mask = I.read_flo(os.path.join(GOLD, 'tiny.flo'))
    assert tuple(flow.shape) == (3, 4, 2) and tuple(mask.shape) == (3, 4, 1)
    # values written by make_io_fixtures.py: u = x - 1.5, v = 0.25 * y, one invalid pixel (1e10) at (1, 2)
    assert flow[0, 0].tolist() == [-1.5, 0.0] and flow[2, 3].tolist() == [1.5, 0.5]
    assert mask[1, 2, 0] == 0 and mask.sum() == 11
    p = tmp_path / "o.flo"
    I.write_flo(str(p), flow)
    f2, m2 = I.read_flo(str(p))
    assert torch.equal(f2, flow) and torch.equal(m2, mask)
    with pytest.raises(ValueError):
        (tmp_path / "bad.flo").write_bytes(b"\x00" * 20)
        I.read_flo(str(tmp_path / "bad.flo"))


def test_kitti_flow_png_fixture():
    from unflow_amd.core import input as I
    flow, mask = I.read_kitti_flow_png(os.path.join(GOLD, 'tiny_kitti_flow.png'))
    assert tuple(flow.shape) == (2, 3, 2)
    # uint16 payload (make_io_fixtures.py): u16 = 2^15 + 64 * u with u = [-2, 0, 0.5 / 3.25, -0.015625, 100], valid = x % 2 == 0
    assert flow[0, :, 0].tolist() == [-2.0, 0.0, 0.5] and flow[1, :, 0].tolist() == [3.25, -0.015625, 100.0]
    assert flow[..., 1].abs().max() == 1.0
    assert mask[..., 0].tolist() == [[1.0, 0.0, 1.0], [1.0, 0.0, 1.0]]


def test_png_decoder_filters_roundtrip():
    """All five PNG filter types against numpy (the encoder below re-filters rows by hand)."""
    import struct
    import zlib
    from unflow_amd.core import input as I
    rs = np.random.RandomState(0)
    arr = rs.randint(0, 65536, size=(6, 5, 3)).astype(np.uint16)
    be = np.ascontiguousarray(arr, dtype='>u2').view(np.uint8).reshape(6, -1).astype(np.int32)
    bpp, rows, prev = 6, [], np.zeros(be.shape[1], dtype=np.int32)
    for y in range(6):
        ft = y % 5
        cur = be[y]
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0:
            f = cur
        elif ft == 1:
            f = (cur - a) & 255
        elif ft == 2:
            f = (cur - prev) & 255
        elif ft == 3:
            f = (cur - ((a + prev) >> 1)) & 255
        else:
            pa = np.array([I._paeth(int(x), int(yv), int(z)) for x, yv, z in zip(a, prev, c)])
            f = (cur - pa) & 255
        rows.append(bytes([ft]) + f.astype(np.uint8).tobytes())
        prev = cur

    def chunk(t, b):
        return struct.pack('>I', len(b)) + t + b + struct.pack('>I', zlib.crc32(t + b) & 0xffffffff)
    png = b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', 5, 6, 16, 2, 0, 0, 0)) + \
        chunk(b'IDAT', zlib.compress(b''.join(rows))) + chunk(b'IEND', b'')
    assert np.array_equal(I.decode_png(png), arr)
    assert np.array_equal(I.decode_png(I.encode_png16_rgb(arr)), arr)


def test_params_npz_roundtrip_and_restore(tmp_path):
    from unflow_amd.core import input as I
    from unflow_amd.core.engine import FlowNetEngine
    from oracle import model_ref as M
    P = M.init_params_spec('CS', seed=3)
    f = str(tmp_path / "w.npz")
    I.save_params_npz(f, P)
    Q = I.load_params_npz(f)
    assert list(Q) == list(P) and all(torch.equal(Q[k], P[k]) for k in P)
    # restore_networks: first network from a file, second keeps its initialisation (train.py:23-65)
    eng = FlowNetEngine(1, 64, 64, params=dict(flownet='CS'), device='cpu', layout_only=True)
    eng.init_params(seed=11)
    before = eng.export_tf_params()
    first = {k: v for k, v in P.items() if not k.startswith('stack_')}
    f1 = str(tmp_path / "c.npz")
    I.save_params_npz(f1, first)
    after = I.restore_networks(eng, dict(flownet='CS'), [f1])
    got = eng.export_tf_params()
    for k in P:
        if k.startswith('stack_'):
            assert torch.equal(got[k], before[k])
        else:
            assert torch.equal(got[k], P[k]) and torch.equal(after[k], P[k])
    bad = dict(first)
    bad['flownet_c/conv4/weights'] = torch.zeros(3, 3, 8, 8)
    I.save_params_npz(str(tmp_path / "bad.npz"), bad)
    with pytest.raises(ValueError):
        I.restore_networks(eng, dict(flownet='CS'), [str(tmp_path / "bad.npz")])


def test_resize_output_flow_matches_oracle_resize():
    from unflow_amd.core import input as I
    from oracle import model_ref as M
    g = torch.Generator().manual_seed(2)
    t = torch.randn(2, 6, 8, 2, generator=g)
    out = I.resize_output_flow(t, 9, 20)
    ref = M.resize_bilinear_tf1(t, 9, 20)
    assert torch.allclose(out[..., 0], ref[..., 0] * (20 / 8), atol=1e-6)
    assert torch.allclose(out[..., 1], ref[..., 1] * (9 / 6), atol=1e-6)
    # identity size: unchanged
    assert torch.allclose(I.resize_output_flow(t, 6, 8), t, atol=1e-7)


def test_outlier_metrics_and_euclidean():
    """flow_util.py:106-123: a pixel is an outlier when its endpoint error is >= max(3, 5 % of the ground-truth magnitude)."""
    from unflow_amd.core.flow_util import euclidean, outlier_pct, outlier_ratio
    gt = torch.zeros(1, 1, 4, 2)
    gt[0, 0, :, 0] = torch.tensor([0.0, 10.0, 100.0, 100.0])
    fl = gt.clone()
    fl[0, 0, :, 0] += torch.tensor([2.9, 3.0, 4.9, 5.0])        # thresholds: 3, 3, 5, 5 -> outliers: no, yes, no, yes
    mask = torch.ones(1, 1, 4, 1)
    assert outlier_ratio(gt, fl, mask).item() == 0.5
    assert outlier_pct(gt, fl, mask).item() == 50.0
    assert outlier_ratio(gt, fl, mask, relative=None).item() == 0.75     # absolute threshold 3 only
    mask[0, 0, 1] = 0                                                      # masked pixels count neither way
    assert abs(outlier_ratio(gt, fl, mask).item() - 1 / 3) < 1e-7
    assert torch.equal(euclidean(torch.tensor([[[[3.0, 4.0]]]])), torch.tensor([[[[5.0]]]]))


def test_product_learning_rate_schedule():
    """core/train.py::learning_rate_at (train.py:225-244) — the PRODUCT function, incl. the manual list's priority."""
    from unflow_amd.core.train import learning_rate_at
    p = dict(learning_rate=1e-4, decay_interval=100, decay_after=200)
    assert learning_rate_at(p, 0) == 1e-4 and learning_rate_at(p, 199) == 1e-4
    assert learning_rate_at(p, 200) == 1e-4                  # decay = 200 // 100 - 200 / 100 = 0
    assert learning_rate_at(p, 300) == 0.5e-4 and learning_rate_at(p, 399) == 0.5e-4 and learning_rate_at(p, 400) == 0.25e-4
    m = dict(p, manual_decay_iters=[10, 20, 30], manual_decay_lrs=[1e-3, 1e-4, 1e-5])
    assert [learning_rate_at(m, i) for i in (0, 10, 11, 30, 31, 60)] == [1e-3, 1e-3, 1e-4, 1e-4, 1e-5, 1e-5]
    assert learning_rate_at(m, 61) == 1e-3                   # past the list: the reference's loop leaves index 0
    assert learning_rate_at(dict(learning_rate=3e-5), 12345) == 3e-5


def test_read_png_image_grey_alpha_and_16_bit(tmp_path):
    """tf.image.decode_png(channels=3) (input.py:208-218) replicates grey to RGB, drops alpha, and returns uint8 also for a
    16-bit file (the high byte)."""
    import struct
    import zlib
    from unflow_amd.core.input import read_png_image

    def png(arr, ctype, depth):
        h, w = arr.shape[:2]
        a = arr.astype('>u2' if depth == 16 else np.uint8).reshape(h, -1)
        raw = b''.join(b'\x00' + a[y].tobytes() for y in range(h))

        def chunk(t, b):
            return struct.pack('>I', len(b)) + t + b + struct.pack('>I', zlib.crc32(t + b) & 0xffffffff)
        return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, depth, ctype, 0, 0, 0)) +
                chunk(b'IDAT', zlib.compress(raw)) + chunk(b'IEND', b''))
    rs = np.random.RandomState(3)
    g8 = rs.randint(0, 256, size=(5, 7, 1))
    ga8 = rs.randint(0, 256, size=(5, 7, 2))
    rgb16 = rs.randint(0, 65536, size=(5, 7, 3))
    g16 = rs.randint(0, 65536, size=(5, 7, 1))
    for name, arr, ctype, depth, want in (("g8", g8, 0, 8, np.repeat(g8, 3, 2)), ("ga8", ga8, 4, 8, np.repeat(ga8[:, :, :1], 3, 2)),
                                          ("rgb16", rgb16, 2, 16, rgb16 >> 8), ("g16", g16, 0, 16, np.repeat(g16 >> 8, 3, 2))):
        p = tmp_path / (name + ".png")
        p.write_bytes(png(arr, ctype, depth))
        got = read_png_image(str(p))
        assert got.dtype == np.float32 and got.shape == (5, 7, 3) and np.array_equal(got, want.astype(np.float32)), name
