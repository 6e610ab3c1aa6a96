"""Mirror of src/e2eflow/core/util.py:12-26."""
import math

import torch

from ..ops import downsample as downsample_ops


def _area_weights(n_in, n_out, device):
    """[n_out, n_in] weights of tf.image.resize_area along one axis: output i averages the source interval
    [i*s, (i+1)*s), s = n_in / n_out, every source pixel weighted by the covered fraction (TF's ResizeAreaOp)."""
    s = n_in / n_out
    w = torch.zeros(n_out, n_in, dtype=torch.float64)
    for i in range(n_out):
        lo, hi = i * s, (i + 1) * s
        for j in range(int(math.floor(lo)), min(n_in, int(math.ceil(hi)))):
            w[i, j] = (min(hi, j + 1) - max(lo, j)) / s
    return w.to(device=device, dtype=torch.float32)


def resize_area(tensor, like):
    """tf.image.resize_area(tensor, like.shape[1:3]) (core/util.py:12-14; stop_gradient there — no gradient here either)."""
    _, h, w, _ = like.shape
    B, H, W, C = tensor.shape
    wy, wx = _area_weights(H, h, tensor.device), _area_weights(W, w, tensor.device)
    return torch.einsum('yh,bhwc,xw->byxc', wy, tensor.detach(), wx)


def downsample(tensor, num):
    """core/util.py:21-26: the downsample op when H and W are even (the reference tests % 2; the op itself requires
    divisibility by num, downsample_op.cc:37-40), else tf.image.resize_area to (int(H / num), int(W / num))."""
    _, height, width, _ = tensor.shape
    if height % 2 == 0 and width % 2 == 0:
        return downsample_ops(tensor, num)
    like = torch.empty(1, int(height / num), int(width / num), 1)
    return resize_area(tensor, like)


# ------------------------------------------------------------------------------------------------- config.ini (util.py:37-85)
def config_dict(config_path='../config.ini'):
    """config_dict (src/e2eflow/util.py:37-62): the ini file as {section: {key: value}} with the reference's coercion order —
    int, then float, then ConfigParser boolean ('yes' / 'true' / 'on' / '1' ...), else the string."""
    import configparser
    config = configparser.ConfigParser()
    config.read(config_path)
    d = {}
    for section_key in config.sections():
        section = config[section_key]
        sd = {}
        for key in section:
            val = section[key]
            try:
                sd[key] = int(val)
            except ValueError:
                try:
                    sd[key] = float(val)
                except ValueError:
                    try:
                        sd[key] = section.getboolean(key)
                    except ValueError:
                        sd[key] = val
        d[section_key] = sd
    return d


def convert_input_strings(config_dct, dirs):
    """convert_input_strings (util.py:65-85), in place: the comma lists 'manual_decay_iters' / 'manual_decay_lrs' become lists
    (and fix num_iters to their sum); 'finetune' = comma-separated experiment names becomes the list of their latest
    checkpoint prefixes — looked up under dirs['checkpoints']/<name>, then dirs['log']/ex/<name>, as the reference does with
    tf.train.get_checkpoint_state (here: the 'checkpoint' state file read by core/tf_checkpoint.py)."""
    import os
    from . import tf_checkpoint as T
    if 'manual_decay_iters' in config_dct and 'manual_decay_lrs' in config_dct:
        iters_lst = [int(i) for i in str(config_dct['manual_decay_iters']).split(',')]
        lrs_lst = [float(x) for x in str(config_dct['manual_decay_lrs']).split(',')]
        config_dct['manual_decay_iters'] = iters_lst
        config_dct['manual_decay_lrs'] = lrs_lst
        config_dct['num_iters'] = sum(iters_lst)
    if 'finetune' in config_dct:
        finetune = []
        for name in str(config_dct['finetune']).split(','):
            ckpt = T.latest_checkpoint(os.path.join(dirs['checkpoints'], name))
            if ckpt is None:
                ckpt = T.latest_checkpoint(os.path.join(dirs['log'], 'ex', name))
            assert ckpt, "Could not load experiment " + name
            finetune.append(ckpt)
        config_dct['finetune'] = finetune
