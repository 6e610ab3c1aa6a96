"""Mirror of src/e2eflow/core/util.py:12-26."""
import torch

from ..ops import downsample as downsample_ops


def resize_area(tensor, like):
    """tf.image.resize_area(tensor, like.shape[1:3]) (core/util.py:12-14; stop_gradient there — no gradient here either):
    one HIP kernel (csrc/ops_warp.hip resize_area_kernel), GPU tensors only like every op of the package."""
    from .. import _lib
    from .._lib import check, ptr, stream
    from ..ops import _dev
    _, h, w, _ = like.shape
    t = _dev(tensor.detach(), 'tensor')
    B, H, W, C = t.shape
    out = torch.empty(B, int(h), int(w), C, dtype=torch.float32, device=t.device)
    check(_lib.lib().unflow_resize_area(ptr(t), ptr(out), B, H, W, C, int(h), int(w), stream(t.device)), "resize_area")
    return out


def downsample(tensor, num):
    """core/util.py:21-26: the downsample op when H and W are even (the reference tests % 2; the op itself requires
    divisibility by num, downsample_op.cc:37-40), else tf.image.resize_area to (int(H / num), int(W / num))."""
    _, height, width, _ = tensor.shape
    if height % 2 == 0 and width % 2 == 0:
        return downsample_ops(tensor, num)
    like = torch.empty(1, int(height / num), int(width / num), 1)
    return resize_area(tensor, like)


# ------------------------------------------------------------------------------------------------- config.ini (util.py:37-85)
def _ini_value(text):
    """One ini value under the reference's coercion ladder (util.py:45-59): an int if it parses as one, else a float, else a
    ConfigParser boolean word ('yes' / 'no' / 'true' / 'false' / 'on' / 'off', any case; '0' / '1' were ints already), else
    the string itself."""
    import configparser
    for number in (int, float):
        try:
            return number(text)
        except ValueError:
            pass
    return configparser.ConfigParser.BOOLEAN_STATES.get(text.lower(), text)


def config_dict(config_path='../config.ini'):
    """config_dict (src/e2eflow/util.py:37-62): {section: {key: coerced value}} of an ini file (a missing file gives {}, as
    ConfigParser.read does); keys of [DEFAULT] appear in every section."""
    import configparser
    ini = configparser.ConfigParser()
    ini.read(config_path)
    return {name: {key: _ini_value(raw) for key, raw in ini[name].items()} for name in ini.sections()}


def experiment_checkpoint(name, dirs):
    """Latest checkpoint prefix of experiment `name`: its working checkpoints (dirs['checkpoints']/<name>) first, then the
    final one kept with its logs (dirs['log']/ex/<name>) — the two places util.py:75-83 asks tf.train.get_checkpoint_state."""
    import os
    from . import tf_checkpoint as T
    for folder in (os.path.join(dirs['checkpoints'], name), os.path.join(dirs['log'], 'ex', name)):
        found = T.latest_checkpoint(folder)
        if found:
            return found
    raise AssertionError("Could not load experiment " + name)


def convert_input_strings(config_dct, dirs):
    """convert_input_strings (util.py:65-85), in place.  A manual learning-rate schedule — both comma lists
    'manual_decay_iters' and 'manual_decay_lrs' present — is parsed into lists and fixes 'num_iters' to the sum of its
    stage lengths; 'finetune' (comma-separated experiment names) becomes the list of their latest checkpoint prefixes."""
    def parts(key, cast):
        return [cast(tok) for tok in str(config_dct[key]).split(',')]
    if {'manual_decay_iters', 'manual_decay_lrs'} <= config_dct.keys():
        config_dct['manual_decay_iters'] = parts('manual_decay_iters', int)
        config_dct['manual_decay_lrs'] = parts('manual_decay_lrs', float)
        config_dct['num_iters'] = sum(config_dct['manual_decay_iters'])
    if 'finetune' in config_dct:
        config_dct['finetune'] = [experiment_checkpoint(name, dirs) for name in parts('finetune', str)]
