"""Mirror of src/e2eflow/kitti/input.py for the evaluation inputs of Trainer.eval (train.py:265-385): image pairs of a
KITTI `training` directory with their occluded / non-occluded ground-truth flow maps, batch 1, one epoch, no TF queues.
The dataset downloaders (kitti/data.py) and the supervised fine-tuning input (input_train_gt :86-146) are out of scope."""
import os
import random

import numpy as np

from ..core.input import Input, read_kitti_flow_png, read_png_image, resize_image_with_crop_or_pad


class KITTIInput(Input):
    def __init__(self, data, batch_size, dims, *, num_threads=1, normalize=True, skipped_frames=False):
        super().__init__(data, batch_size, dims, num_threads=num_threads, normalize=normalize, skipped_frames=skipped_frames)

    def _preprocess_flow(self, gt):
        """kitti/input.py:32-38: flow and validity mask cropped / zero-padded to the input's dims."""
        flow, mask = gt
        h, w = self.dims
        return (resize_image_with_crop_or_pad(flow.numpy(), h, w).reshape(h, w, 2),
                resize_image_with_crop_or_pad(mask.numpy(), h, w).reshape(h, w, 1))

    def _flow_files(self, flow_dir, hold_out_inv):
        """kitti/input.py:40-66: the sorted listings of flow_occ / flow_noc, each cut to the first hold_out_inv entries of
        its own seed-0 shuffle."""
        out = []
        for sub in ('flow_occ', 'flow_noc'):
            d = os.path.join(self.data.current_dir, flow_dir, sub)
            files = sorted(os.listdir(d))
            if hold_out_inv is not None:
                random.seed(0)
                random.shuffle(files)
                files = files[:hold_out_inv]
            out.append([os.path.join(d, f) for f in files])
        assert len(out[0]) == len(out[1])
        return out

    def _input_train(self, image_dir, flow_dir, hold_out_inv=None):
        """kitti/input.py:75-82: batches of [im1, im2, input_shape, flow_occ, mask_occ, flow_noc, mask_noc] (numpy,
        NHWC), one epoch, a smaller final batch allowed.  The images come from Input.input_test's pair list (same
        hold-out shuffle), the ground truth from _flow_files, position by position as the reference's queues pair them."""
        pairs = self.test_pairs(image_dir, hold_out_inv)
        occ, noc = self._flow_files(flow_dir, hold_out_inv)
        assert len(pairs) == len(occ), (len(pairs), len(occ))
        for b0 in range(0, len(pairs), self.batch_size):
            cols = [[] for _ in range(7)]
            for (fn1, fn2), f_occ, f_noc in zip(pairs[b0:b0 + self.batch_size], occ[b0:], noc[b0:]):
                a, b = read_png_image(fn1), read_png_image(fn2)
                fo, mo = self._preprocess_flow(read_kitti_flow_png(f_occ))
                fnc, mn = self._preprocess_flow(read_kitti_flow_png(f_noc))
                for c, v in zip(cols, (self._preprocess_image(a), self._preprocess_image(b), np.asarray(a.shape, dtype=np.int32),
                                       fo, mo, fnc, mn)):
                    c.append(v)
            yield tuple(np.stack(c) for c in cols)

    def input_train_2015(self, hold_out_inv=None):
        return self._input_train('data_scene_flow/training/image_2', 'data_scene_flow/training', hold_out_inv)

    def input_test_2015(self, hold_out_inv=None):
        return self.input_test('data_scene_flow/testing/image_2', hold_out_inv)

    def input_train_2012(self, hold_out_inv=None):
        return self._input_train('data_stereo_flow/training/colored_0', 'data_stereo_flow/training', hold_out_inv)

    def input_test_2012(self, hold_out_inv=None):
        return self.input_test('data_stereo_flow/testing/colored_0', hold_out_inv)
