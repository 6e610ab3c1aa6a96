"""Host-side rows around the hot path: config.ini parsing (util.py:37-85), the raw-frame input pipeline
(core/input.py:37-218), the experiment shell (experiment.py:11-83) and Trainer.run's resume / save arithmetic
(train.py:116-145) — CPU only; the GPU step itself is covered by tests/test_train_gpu.py."""
import os
import random

import numpy as np
import pytest

from unflow_amd.core import input as I
from unflow_amd.core import util as U
from unflow_amd.core import tf_checkpoint as T


CONFIG = """
[dirs]
log = {log}
checkpoints = {ckpt}
data = /nowhere

[run]
batch_size = 4
num_iters = 600000
save_interval = 10
display_interval = 5
learning_rate = 1.0e-4
border_mask = True
flownet = CSS
manual_decay_iters = 100,50,50
manual_decay_lrs = 1e-4,5e-5,2.5e-5
finetune = exp_a,exp_b
"""


def test_config_dict_coercion_and_input_strings(tmp_path):
    log, ckpt = tmp_path / "log", tmp_path / "ckpt"
    p = tmp_path / "config.ini"
    p.write_text(CONFIG.format(log=log, ckpt=ckpt))
    d = U.config_dict(str(p))
    run = d['run']
    assert run['batch_size'] == 4 and isinstance(run['batch_size'], int)
    assert run['learning_rate'] == 1.0e-4 and isinstance(run['learning_rate'], float)
    assert run['border_mask'] is True and run['flownet'] == 'CSS'
    assert run['manual_decay_iters'] == '100,50,50'
    # finetune names resolve to the latest checkpoint prefix: checkpoints/<name> first, then log/ex/<name> (util.py:75-85)
    os.makedirs(ckpt / "exp_a")
    T.write_checkpoint(str(ckpt / "exp_a" / "model.ckpt-7"), {"w": np.ones((2, 2), np.float32)})
    os.makedirs(log / "ex" / "exp_b")
    T.write_checkpoint(str(log / "ex" / "exp_b" / "model.ckpt-9"), {"w": np.ones((2, 2), np.float32)})
    U.convert_input_strings(run, d['dirs'])
    assert run['manual_decay_iters'] == [100, 50, 50] and run['manual_decay_lrs'] == [1e-4, 5e-5, 2.5e-5]
    assert run['num_iters'] == 200
    assert run['finetune'] == [str(ckpt / "exp_a" / "model.ckpt-7"), str(log / "ex" / "exp_b" / "model.ckpt-9")]
    run2 = dict(finetune='missing')
    with pytest.raises(AssertionError):
        U.convert_input_strings(run2, d['dirs'])


class _Data:
    def __init__(self, dirs, current_dir=None):
        self._dirs, self.current_dir = dirs, current_dir

    def get_raw_dirs(self):
        return self._dirs


def _write_frames(d, names, h=12, w=16, seed=0):
    os.makedirs(d, exist_ok=True)
    rng = np.random.RandomState(seed)
    imgs = {}
    for n in names:
        a = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        with open(os.path.join(d, n), 'wb') as f:
            f.write(I.encode_png8_rgb(a))
        imgs[n] = a
    return imgs


def test_frame_name_to_num_and_png8_roundtrip(tmp_path):
    assert I.frame_name_to_num('0000012.png') == 12 and I.frame_name_to_num('000.png') == 0
    imgs = _write_frames(str(tmp_path), ['a.png'])
    np.testing.assert_array_equal(I.read_png_image(str(tmp_path / 'a.png')), imgs['a.png'].astype(np.float32))


def _reference_pairs(dirs, swap_images, sequence, shift, seed, skip, skipped_frames):
    """input_raw's list construction, transcribed from core/input.py:139-177 (np.roll on the [n, 2] array included)."""
    if not isinstance(skip, list):
        skip = [skip]
    filenames = []
    for dir_path in dirs:
        files = os.listdir(dir_path)
        files.sort()
        if sequence:
            steps = [1 + s for s in skip]
            stops = [len(files) - s for s in steps]
        else:
            steps = [2]
            stops = [len(files)]
        for step, stop in zip(steps, stops):
            for i in range(0, stop, step):
                if skipped_frames and sequence:
                    if I.frame_name_to_num(files[i]) + 1 != I.frame_name_to_num(files[i + 1]):
                        continue
                filenames.append((os.path.join(dir_path, files[i]), os.path.join(dir_path, files[i + 1])))
    random.seed(seed)
    random.shuffle(filenames)
    ext = []
    for fn1, fn2 in filenames:
        ext.append((fn1, fn2))
        if swap_images:
            ext.append((fn2, fn1))
    shift = shift % len(ext)
    return [tuple(str(x) for x in p) for p in np.roll(ext, shift)]


@pytest.mark.parametrize("swap,sequence,shift,skip,skipped", [(True, True, 0, 0, False), (True, True, 5, 0, False),
                                                             (False, False, 2, 0, False), (True, True, 3, [0, 1], False),
                                                             (True, True, 4, 0, True)])
def test_raw_pair_list_matches_the_reference_construction(tmp_path, swap, sequence, shift, skip, skipped):
    d1, d2 = str(tmp_path / "seq1"), str(tmp_path / "seq2")
    _write_frames(d1, ['%06d.png' % i for i in (0, 1, 2, 3, 5, 6)])
    _write_frames(d2, ['%06d.png' % i for i in (10, 11, 12, 13)], seed=1)
    inp = I.Input(_Data([d1, d2]), batch_size=2, dims=(8, 8), skipped_frames=skipped)
    got = inp.raw_pairs(swap_images=swap, sequence=sequence, shift=shift, seed=3, skip=skip)
    assert got == _reference_pairs([d1, d2], swap, sequence, shift, 3, skip, skipped)


def test_input_raw_batches_crop_normalise_and_cycle(tmp_path):
    d = str(tmp_path / "seq")
    imgs = _write_frames(d, ['%06d.png' % i for i in range(4)], h=12, w=16)
    inp = I.Input(_Data([d]), batch_size=4, dims=(8, 8))
    it = inp.input_raw(swap_images=False, sequence=True, needs_crop=True, shift=0, seed=0)
    pairs = inp.raw_pairs(swap_images=False, sequence=True, shift=0, seed=0)
    assert len(pairs) == 3
    im1, im2 = next(it)
    assert im1.shape == (4, 8, 8, 3) and im1.dtype == np.float32
    mean, std = np.asarray(inp.mean, np.float32), np.float32(inp.stddev)
    for k in range(4):                                   # 3 pairs, batch of 4: the list is walked cyclically
        fn1, fn2 = pairs[k % 3]
        a, b = imgs[os.path.basename(fn1)].astype(np.float32), imgs[os.path.basename(fn2)].astype(np.float32)
        # the same window in both frames: find it in the first frame, check it in the second
        x = im1[k] * std + mean
        hits = [(oy, ox) for oy in range(5) for ox in range(9) if np.allclose(a[oy:oy + 8, ox:ox + 8], x, atol=1e-3)]
        assert len(hits) >= 1
        oy, ox = hits[0]
        np.testing.assert_allclose(im2[k] * std + mean, b[oy:oy + 8, ox:ox + 8], atol=1e-3)
    # no crop: frames already have the network size
    inp2 = I.Input(_Data([d]), batch_size=1, dims=(12, 16), normalize=False)
    a1, _ = next(inp2.input_raw(swap_images=False, needs_crop=False, seed=0))
    np.testing.assert_array_equal(a1[0], imgs[os.path.basename(pairs[0][0])].astype(np.float32))


def test_input_test_pairs_center_crop_or_pad(tmp_path):
    root = str(tmp_path)
    imgs = _write_frames(os.path.join(root, "val"), ['a0.png', 'a1.png', 'b0.png', 'b1.png', 'c0.png', 'c1.png'], h=10, w=20)
    inp = I.Input(_Data([], current_dir=root), batch_size=2, dims=(12, 16), normalize=False)
    batches = list(inp.input_test("val"))
    assert [b[0].shape[0] for b in batches] == [2, 1]              # smaller final batch
    im1, im2, shp = batches[0]
    assert im1.shape == (2, 12, 16, 3) and tuple(shp[0]) == (10, 20, 3)
    a = imgs['a0.png'].astype(np.float32)
    np.testing.assert_array_equal(im1[0][1:11], a[:, 2:18])        # padded 1 row top / bottom, centre-cropped 2 columns each side
    assert not im1[0][0].any() and not im1[0][11].any()
    assert len(inp.test_pairs("val", hold_out_inv=2)) == 2


def test_experiment_shell_dirs_config_copy_and_checkpoint_bookkeeping(tmp_path):
    from unflow_amd.experiment import Experiment
    log, ckpt = tmp_path / "log", tmp_path / "ckpt"
    cfg = tmp_path / "config.ini"
    cfg.write_text(CONFIG.format(log=log, ckpt=ckpt))
    ex = Experiment("e1", config_path=str(cfg))
    for d in (ex.train_dir, ex.eval_dir, ex.save_dir):
        assert os.path.isdir(d)
    assert os.path.isfile(os.path.join(ex.log_dir, "config.ini")) and ex.config['run']['batch_size'] == 4
    assert ex.latest_checkpoint() is None
    T.write_checkpoint(os.path.join(ex.save_dir, "model.ckpt-20"), {"w": np.arange(4, dtype=np.float32)}, directory_state=True)
    assert ex.latest_checkpoint().endswith("model.ckpt-20")
    ex.conclude()                                                   # final checkpoint -> the permanent log dir
    assert T.latest_checkpoint(ex.log_dir).endswith("model.ckpt-20")
    # intermediate checkpoints deleted: a new Experiment object restores the stored one into a fresh save dir
    import shutil
    shutil.rmtree(ex.save_dir)
    ex2 = Experiment("e1", config_path=str(cfg))
    got = T.read_checkpoint(ex2.latest_checkpoint(), ["w"])
    np.testing.assert_array_equal(got["w"], np.arange(4, dtype=np.float32))
    # overwrite clears both trees
    ex3 = Experiment("e1", overwrite=True, config_path=str(cfg))
    assert ex3.latest_checkpoint() is None and not os.listdir(ex3.save_dir)


def test_trainer_run_resume_and_save_arithmetic(tmp_path):
    """Trainer.run on a stand-in for the GPU step: chunks of save_interval, iter_offset of each chunk, learning-rate index
    (decay_iters = local_i + iter_offset), checkpoint names, resume from global_step + 1, 'max_iter reached'."""
    from unflow_amd.core.train import Trainer

    class Fake(Trainer):
        def __init__(self, params):
            self.params, self.rank, self.world, self.iteration = dict(params), 0, 1, 0
            self.steps, self.saved, self.offsets = [], [], []
            self.engine = type('E', (), {'step_count': 0, 'check_device_faults': lambda self, world_sync=False: 0})()

        def train_step(self, im1, im2, augment=None):
            self.steps.append((self.iteration, int(im1)))
            return 1.0

        def save(self, ckpt_dir, global_step):
            self.saved.append(global_step)
            T.write_checkpoint(os.path.join(ckpt_dir, 'model.ckpt-%d' % global_step), {"w": np.zeros(1, np.float32)})

        def restore(self, ckpt_dir=None):       # (run() calls restore(None) on a fresh start: params['finetune'] only)
            return T.latest_checkpoint(ckpt_dir) if ckpt_dir else None

    def batches(iter_offset):
        k = iter_offset
        while True:
            yield k, k
            k += 1

    ck = str(tmp_path / "ck")
    os.makedirs(ck)
    tr = Fake(dict(save_interval=10, display_interval=5))
    log = tr.run(0, 30, lambda off: (tr.offsets.append(off), batches(off))[1], ck)
    assert tr.saved == [10, 20, 30] and tr.offsets == [0, 10, 20]
    assert tr.steps == [(i, i) for i in range(30)]                 # decay_iters and the shifted input agree
    assert [i for i, _ in log] == [1, 5, 10, 15, 20, 25, 30]
    assert tr.checkpoint_step(ck) == 30
    # resume: nothing left in [0, 30]; a later stage [30, 50] continues from 31 with iter_offset 0
    tr2 = Fake(dict(save_interval=10, display_interval=100))
    assert tr2.run(0, 30, batches, ck) == []
    tr2.run(30, 50, lambda off: (tr2.offsets.append(off), batches(off))[1], ck)
    assert tr2.saved == [40, 50] and tr2.offsets == [0, 10]
    tr3 = Fake(dict(save_interval=10))
    with pytest.raises(AssertionError):                            # checkpoint from an earlier stage than min_iter
        tr3.run(60, 70, batches, ck)
