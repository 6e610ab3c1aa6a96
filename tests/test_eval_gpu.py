"""Trainer.eval (f2) end to end on a synthetic KITTI-2012-format tree: checkpoint on disk -> restore_networks -> batch-1
pairs through resize_input -> unsupervised_loss(augment=False, return_flow=True) at 384 x 1280 -> resize_output_flow ->
AEE / outlier % against both ground-truth maps — compared with the same chain computed by the ORACLE (oracle/model_ref.py
forward on the host, the oracle's own bilinear resize, numpy metrics) from the same checkpoint.
reference: src/e2eflow/core/train.py:265-385, core/input.py:10-34, kitti/input.py:12-82, core/flow_util.py:98-123."""
import numpy as np
import pytest
import torch

from kitti_fixture import Data, make_tree

pytestmark = pytest.mark.gpu


def _oracle_eval(tf_params, params, example):
    """One example through the reference's evaluation chain, restated with the oracle's pieces only."""
    from oracle import model_ref as M
    im1, im2, flow_occ, mask_occ, flow_noc, mask_noc = [torch.from_numpy(np.ascontiguousarray(a)) for a in example]
    h, w = im1.shape[:2]
    a = M.resize_bilinear_tf1(im1.unsqueeze(0), 384, 1280)          # resize_input: the frame itself, stretched
    b = M.resize_bilinear_tf1(im2.unsqueeze(0), 384, 1280)
    with torch.no_grad():
        loss, ffw, _, _ = M.unsupervised_loss(tf_params, a, b, params, return_flow=True)
    f = M.resize_bilinear_tf1(ffw, h, w)
    f = torch.stack([f[..., 0] * (w / 1280.0), f[..., 1] * (h / 384.0)], 3)
    vals = []
    for gt, mask in ((flow_occ, mask_occ), (flow_noc, mask_noc)):
        gt, mask = gt.unsqueeze(0), mask.unsqueeze(0)
        d = ((gt - f) ** 2).sum(3, keepdim=True).sqrt() * mask
        thr = torch.clamp(((gt ** 2).sum(3, keepdim=True)).sqrt() * 0.05, min=3.0)
        vals += [(d.sum() / mask.sum()).item(), ((d >= thr).float().sum() / mask.sum()).item() * 100]
    return vals + [loss.item()], f


def test_trainer_eval_on_kitti_format_tree_vs_oracle(dev, tmp_path):
    from unflow_amd.core.train import Trainer
    from unflow_amd.kitti.input import KITTIInput
    written = make_tree(tmp_path / "kitti", n_pairs=3)
    params = dict(flownet='C', pyramid_loss=True, border_mask=True, ternary_weight=1.0, smooth_2nd_weight=3.0,
                  learning_rate=1e-4, save_interval=1, display_interval=1)
    tr = Trainer(1, 128, 192, params, device=dev, seed=3, augment=False)
    # flow heads scaled up so that the predicted flow is a few pixels and the outlier threshold (3 px / 5 %) is in play
    tfp = tr.engine.export_tf_params()
    tfp = {k: (v * 4.0 if k.split('/')[-2] == 'flow2' and k.endswith('/weights') else v) for k, v in tfp.items()}
    tr.engine.load_tf_params(tfp)
    ckpt_dir = str(tmp_path / "ckpt")
    tr.save(ckpt_dir, 7)
    einput = KITTIInput(Data(tmp_path / "kitti"), batch_size=1, normalize=False, dims=(384, 1280))
    res = tr.eval(lambda: einput.input_train_2012(), ckpt_dir)
    assert res['global_step'] == 7 and res['num_examples'] == 3
    ref_rows = []
    for ex in written:
        row, _ = _oracle_eval({k: v.cpu() for k, v in tfp.items()}, params, ex)
        ref_rows.append(row)
    got_rows = res['per_example']
    for got, ref in zip(got_rows, ref_rows):
        assert abs(got[0] - ref[0]) < 1e-3 and abs(got[2] - ref[2]) < 1e-3, (got, ref)          # AEE: the north-star EPE bar
        assert abs(got[1] - ref[1]) < 0.05 and abs(got[3] - ref[3]) < 0.05, (got, ref)          # outlier %: threshold flips
        assert abs(got[4] - ref[4]) <= 2e-4 * abs(ref[4]), (got, ref)
    ref_avg = np.mean(np.asarray(ref_rows), axis=0)
    for k, r in zip(res['names'], ref_avg):
        assert abs(res[k] - r) <= max(1e-3, 2e-4 * abs(r)) or k.startswith('outliers'), (k, res[k], r)
    # the numbers must be live: AEE of a few pixels against the synthetic ground truth, outliers strictly between 0 and 100
    assert 0.5 < res['AEE/occluded'] < 50 and 0.0 < res['outliers/occluded'] <= 100.0
    print("eval on the synthetic KITTI tree:", {k: round(res[k], 4) for k in res['names']}, "oracle:", [round(float(r), 4) for r in ref_avg])
    # a second evaluation after training steps picks up the newer checkpoint
    tr.save(ckpt_dir, 9)
    assert tr.eval(lambda: einput.input_train_2012(hold_out_inv=1), ckpt_dir)['global_step'] == 9
    with pytest.raises(AssertionError, match="No checkpoints"):
        tr.eval(lambda: einput.input_train_2012(), str(tmp_path / "nothing_here"))
