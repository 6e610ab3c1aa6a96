"""Experiment shell: what src/e2eflow/experiment.py:11-83 provides, rebuilt on the pure-python checkpoint state files of
core/tf_checkpoint.py (no TensorFlow).

Contract.  An experiment `name` owns
    <dirs.log>/ex/<name>/            its private config.ini, the FINAL checkpoint after conclude()
    <dirs.log>/ex/<name>/train|eval  summaries
    <dirs.checkpoints>/<name>/       the working checkpoints of a run
Opening an unknown name creates the tree; opening a known one keeps it (overwrite=True wipes logs and checkpoints first).
When the logs exist but the working checkpoints are gone, they are re-seeded from the final checkpoint kept with the logs;
without one the experiment cannot be continued and the constructor says so."""
import os
import shutil

from .core import tf_checkpoint as T
from .core.util import config_dict


def carry_checkpoint(src_dir, dst_dir, rename_to=None):
    """Copy the latest checkpoint of src_dir (every file of the bundle) into dst_dir and point dst_dir's `checkpoint` state
    file at it; rename_to gives the copy another prefix (e.g. 'model.ckpt-0' to restart the step count).  Returns the source
    prefix, or None when src_dir has no checkpoint."""
    prefix = T.latest_checkpoint(src_dir)
    if prefix is None:
        return None
    stem = os.path.basename(prefix)
    target = rename_to or stem
    for entry in os.listdir(src_dir):
        if stem in entry:
            shutil.copyfile(os.path.join(src_dir, entry), os.path.join(dst_dir, entry.replace(stem, target)))
    with open(os.path.join(dst_dir, 'checkpoint'), 'w') as state:
        state.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (target, target))
    return prefix


class Experiment:
    def __init__(self, name, overwrite=False, config_path='../config.ini'):
        dirs = config_dict(config_path)['dirs']
        self.name = name
        self.log_dir = os.path.join(dirs['log'], 'ex', name)
        self.train_dir = os.path.join(self.log_dir, 'train')
        self.eval_dir = os.path.join(self.log_dir, 'eval')
        self.save_dir = os.path.join(dirs['checkpoints'], name)

        known = os.path.isdir(self.log_dir)
        if known and overwrite:
            for folder in (self.log_dir, self.save_dir):
                shutil.rmtree(folder, ignore_errors=True)
            known = False
        if not known:
            for folder in (self.log_dir, self.save_dir, self.train_dir, self.eval_dir):
                os.makedirs(folder)                      # (an orphaned checkpoint folder of that name is an error, as in the reference)
        elif not os.path.isdir(self.save_dir):
            os.makedirs(self.save_dir)
            if carry_checkpoint(self.log_dir, self.save_dir) is None:
                raise RuntimeError('Failed to restore "{}".Use --overwrite=True to clear.'.format(name))
            print('Warning: intermediate checkpoints could not be restored.')

        own = os.path.join(self.log_dir, 'config.ini')
        if overwrite or not os.path.isfile(own):
            shutil.copyfile(config_path, own)
        self.config = config_dict(own)

    def latest_checkpoint(self):
        return T.latest_checkpoint(self.save_dir)

    def conclude(self):
        """Keep the final checkpoint with the logs (experiment.py:78-82)."""
        if carry_checkpoint(self.save_dir, self.log_dir) is None:
            print('Warning: no checkpoints written')
