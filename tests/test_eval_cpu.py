"""CPU tests of the evaluation input (f2): KITTIInput's example tuples (kitti/input.py:32-82), the resize helpers of
core/input.py:10-34 and losses.occlusion's host-side contract.  reference: src/e2eflow/core/train.py:265-385."""
import numpy as np
import torch

from kitti_fixture import Data, SIZES, make_tree


def test_kitti_input_train_tuples(tmp_path):
    from unflow_amd.kitti.input import KITTIInput
    written = make_tree(tmp_path, n_pairs=3)
    kin = KITTIInput(Data(tmp_path), batch_size=1, normalize=False, dims=(384, 1280))
    batches = list(kin.input_train_2012())
    assert len(batches) == 3
    for i, (b, w) in enumerate(zip(batches, written)):
        im1, im2, shape, flow_occ, mask_occ, flow_noc, mask_noc = b
        h, wd = SIZES[i]
        assert im1.shape == (1, 384, 1280, 3) and flow_occ.shape == (1, 384, 1280, 2) and mask_noc.shape == (1, 384, 1280, 1)
        assert shape.tolist() == [[h, wd, 3]]
        top, left = (384 - h) // 2, (1280 - wd) // 2               # resize_image_with_crop_or_pad: centred, zero-padded
        assert np.array_equal(im1[0, top:top + h, left:left + wd], w[0])
        assert np.array_equal(im2[0, top:top + h, left:left + wd], w[1])
        assert np.array_equal(flow_occ[0, top:top + h, left:left + wd], w[2])
        assert np.array_equal(mask_noc[0, top:top + h, left:left + wd], w[5])
        assert im1[0, :top].sum() == 0 and mask_occ[0, :, :left].sum() == 0
    # hold_out_inv: the first n of the seed-0 shuffle, images and ground truth shuffled by the same seed over equally long lists
    held = list(kin.input_train_2012(hold_out_inv=2))
    assert len(held) == 2


def test_resize_input_and_output_crop_undo_the_pipeline_padding(tmp_path):
    from unflow_amd.core import input as I
    h, w = 370, 1226
    rs = np.random.RandomState(1)
    frame = rs.rand(h, w, 3).astype(np.float32) * 255
    padded = I.resize_image_with_crop_or_pad(frame, 384, 1280)
    t = torch.from_numpy(padded).unsqueeze(0)
    r = I.resize_input(t, h, w, 384, 1280)
    assert tuple(r.shape) == (1, 384, 1280, 3)
    want = I.resize_bilinear_tf1(torch.from_numpy(frame).unsqueeze(0), 384, 1280)
    assert torch.equal(r, want)
    # a frame larger than the dims is centre-cropped by the pipeline; resize_input then zero-pads it back (the reference does)
    big = rs.rand(400, 1300, 3).astype(np.float32)
    cropped = I.resize_image_with_crop_or_pad(big, 384, 1280)
    r2 = I.resize_input(torch.from_numpy(cropped).unsqueeze(0), 400, 1300, 384, 1280)
    back = I.resize_image_with_crop_or_pad(cropped, 400, 1300)
    assert torch.equal(r2, I.resize_bilinear_tf1(torch.from_numpy(back).unsqueeze(0), 384, 1280))
    gt = torch.from_numpy(I.resize_image_with_crop_or_pad(frame[..., :2], 384, 1280)).unsqueeze(0)
    assert torch.equal(I.resize_output_crop(gt, h, w, 2)[0], torch.from_numpy(frame[..., :2]))
    assert tuple(I.resize_output(r, h, w, 3).shape) == (1, h, w, 3)
