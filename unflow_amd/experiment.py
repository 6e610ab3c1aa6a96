"""Experiment shell (src/e2eflow/experiment.py:11-83): the directory layout of a named experiment — log/ex/<name>/{train,eval},
checkpoints/<name> — its private copy of config.ini, and the latest-checkpoint bookkeeping, on top of the pure-python
checkpoint state files of core/tf_checkpoint.py (no TensorFlow)."""
import os
from shutil import copyfile, rmtree

from .core.util import config_dict
from .core import tf_checkpoint as T


class Experiment:
    def __init__(self, name, overwrite=False, config_path='../config.ini'):
        global_config = config_dict(config_path)
        dirs = global_config['dirs']
        log_dir = os.path.join(dirs['log'], 'ex', name)
        train_dir = os.path.join(log_dir, 'train')
        eval_dir = os.path.join(log_dir, 'eval')
        save_dir = os.path.join(dirs['checkpoints'], name)

        def _init_dirs():
            os.makedirs(log_dir)
            os.makedirs(save_dir)
            os.makedirs(train_dir)
            os.makedirs(eval_dir)

        if os.path.isdir(log_dir):                       # the experiment exists
            if overwrite:
                rmtree(log_dir)
                if os.path.isdir(save_dir):
                    rmtree(save_dir)
                _init_dirs()
            elif not os.path.isdir(save_dir):
                os.makedirs(save_dir)
                # the stored final checkpoint, in case the intermediate ones were deleted (experiment.py:38-46)
                ckpt = self._copy_latest_checkpoint(log_dir, save_dir)
                if not ckpt:
                    raise RuntimeError('Failed to restore "{}".Use --overwrite=True to clear.'.format(name))
                print('Warning: intermediate checkpoints could not be restored.')
        else:
            _init_dirs()

        own_config = os.path.join(log_dir, 'config.ini')
        if not os.path.isfile(own_config) or overwrite:
            copyfile(config_path, own_config)
        self.config = config_dict(own_config)
        self.train_dir, self.eval_dir, self.save_dir, self.log_dir, self.name = train_dir, eval_dir, save_dir, log_dir, name

    def latest_checkpoint(self):
        return T.latest_checkpoint(self.save_dir)

    def _copy_latest_checkpoint(self, src, dst, reset_global_step=False):
        ckpt = T.latest_checkpoint(src)
        if ckpt:
            ckpt_base = os.path.basename(ckpt)
            new_base = 'model.ckpt-0' if reset_global_step else ckpt_base
            with open(os.path.join(dst, 'checkpoint'), 'w') as f:
                f.write('model_checkpoint_path: "' + new_base + '"\n')
                f.write('all_model_checkpoint_paths: "' + new_base + '"\n')
            for filename in os.listdir(src):
                if ckpt_base in filename:
                    copyfile(os.path.join(src, filename), os.path.join(dst, filename.replace(ckpt_base, new_base)))
        return ckpt

    def conclude(self):
        """Move the final checkpoint to the permanent log dir (experiment.py:78-82)."""
        ckpt = self._copy_latest_checkpoint(self.save_dir, self.log_dir)
        if not ckpt:
            print('Warning: no checkpoints written')
