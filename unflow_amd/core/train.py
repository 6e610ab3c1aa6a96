"""Mirror of the hot-path parts of src/e2eflow/core/train.py: optimizer + tower/gradient averaging
(get_train_and_loss_ops :147-185, average_gradients :388-422) and the learning-rate schedule (:225-244).
Checkpoint restore, evaluation and TF summaries (:23-65, :265-385) are out of scope."""
import torch
import torch.distributed as dist

from .data_parallel import GradAllReducer
from .engine import FlowNetCEngine


def learning_rate_at(params, decay_iters):
    """train.py:225-244."""
    if 'manual_decay_lrs' in params and 'manual_decay_iters' in params:
        decay_index, iter_counter = 0, 0
        for decay_i, manual_decay_iter in enumerate(params['manual_decay_iters']):
            iter_counter += manual_decay_iter
            if decay_iters <= iter_counter:
                decay_index = decay_i
                break
        return params['manual_decay_lrs'][decay_index]
    decay_interval = params['decay_interval']
    decay_after = params.get('decay_after', 0)
    if decay_iters >= decay_after:
        decay = (decay_iters // decay_interval) - decay_after / decay_interval
        return params['learning_rate'] / (2 ** decay)
    return params['learning_rate']


class Trainer:
    """One process per GPU.  `params` carries the reference's [train] keys (learning_rate, decay_interval,
    decay_after, loss weights ...).  With torch.distributed initialised, every rank trains on its own shard of the
    minibatch and the gradients are averaged with one RCCL all-reduce (average_gradients semantics)."""

    def __init__(self, batch_size, height, width, params, device=None, seed=0):
        self.params = dict(params)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        loss_params = {k: v for k, v in params.items()
                       if k.endswith('_weight') or k in ('flownet', 'pyramid_loss', 'border_mask', 'mask_occlusion')}
        self.engine = FlowNetCEngine(batch_size, height, width, params=loss_params or None, device=device, seed=seed)
        self.reducer = GradAllReducer(self.engine.G, self.world) if self.world > 1 else None
        self.iteration = 0

    def train_step(self, im1, im2):
        """sess.run([train_op, loss_]) (train.py:247-251): returns the loss tensor (device, no sync)."""
        lr = learning_rate_at(self.params, self.iteration) if 'decay_interval' in self.params else self.params['learning_rate']
        loss = self.engine.fwd_bwd(im1, im2)
        if self.reducer is not None:
            self.reducer.all_reduce()
        self.engine.adam_step(lr, grad_scale=1.0 / self.world)
        self.iteration += 1
        return loss
