"""Mirror of the hot-path parts of src/e2eflow/core/train.py: the training op (get_train_and_loss_ops :147-185:
AdamOptimizer + towers + average_gradients :388-422), the learning-rate schedule (:225-244) and the loop body
(:247-251), Trainer.run / train with checkpoint save and resume (:116-145, :186-262; TF checkpoint-V2 bundles written and
read by core/tf_checkpoint.py).  Evaluation and TF summaries (:265-385) are out of scope.

One process per GPU.  A training step is

    set_input (+ augmentation)                        eager launches
    forward + losses + backward part 0                hipGraph replay
      -> all-reduce of part 0's gradients, then their fused L2 + Adam update      } on the communication stream,
    backward part 1 ...                               hipGraph replay             } under the remaining backward
      -> all-reduce + Adam of part 1 ...
    join

so the optimizer update of a bucket overlaps the exchange and the backward pass of the earlier layers (a layer's weights are
only read by its own forward / data-gradient launches, which are complete when its bucket is released).  The re-split of
the updated weights into their operand planes (csrc/conv_planes.hip) follows each bucket's update on the same stream
instead of opening the next forward pass.  With one rank there is no exchange: one graph, one Adam launch.  (Keeping the cut
on one rank — StepRunner(local_overlap=True): the HBM-bound Adam + re-split of the deep layers beside the backward pass of the shallow
ones — measured 2.7 % SLOWER on MI355X, 544 vs 559 pairs/s: the streaming blocks take CU slots from conv launches that are
sized to fill the chip in exactly one round.)"""
import os

import torch
import torch.distributed as dist

from .data_parallel import GradAllReducer
from .engine import FlowNetEngine


def learning_rate_at(params, decay_iters):
    """train.py:225-244: the manual piecewise schedule has priority; else halve every decay_interval after decay_after."""
    if 'manual_decay_lrs' in params and 'manual_decay_iters' in params:
        decay_index, iter_counter = 0, 0
        for decay_i, manual_decay_iter in enumerate(params['manual_decay_iters']):
            iter_counter += manual_decay_iter
            if decay_iters <= iter_counter:
                decay_index = decay_i
                break
        return params['manual_decay_lrs'][decay_index]
    if 'decay_interval' not in params:
        return params['learning_rate']
    decay_interval = params['decay_interval']
    decay_after = params.get('decay_after', 0)
    if decay_iters >= decay_after:
        decay = (decay_iters // decay_interval) - decay_after / decay_interval
        return params['learning_rate'] / (2 ** decay)
    return params['learning_rate']


# default cuts of the backward pass for the bucketed exchange (layers after which a bucket is released): decoder + conv6_1 +
# conv6 (118 MB of FlowNetC's 157 MB of gradients), conv5_1 .. conv4 (33 MB), the rest + all biases
DEFAULT_BUCKET_CUTS = ('conv6', 'conv4')


class StepRunner:
    """Runs training steps of an engine: hipGraph replay of forward + loss + backward (cut into parts when gradients are
    exchanged), bucketed RCCL all-reduce and bucketed Adam on a communication stream.  Used by bench.py and Trainer."""

    def __init__(self, engine, world=1, use_graph=True, force_reducer=False, bucket_cuts=DEFAULT_BUCKET_CUTS,
                 bucket_bytes=64 << 20, group=None, local_overlap=False, transport=None):
        self.eng = engine
        self.world = world
        local = world == 1 and not force_reducer and engine.dev.type == 'cuda' and local_overlap
        self.dist = world > 1 or force_reducer or local
        self.reducer = GradAllReducer(engine.G, world, bucket_bytes=bucket_bytes, group=group, force=force_reducer,
                                      local_overlap=local, transport=transport) if self.dist else None
        self.nparts = engine.set_backward_parts(bucket_cuts if (self.dist and not engine.train_all) else ())
        self.buckets = engine.part_buckets()
        self.frozen = engine.frozen_ranges()
        self.use_graph = use_graph
        self.graphs = None
        self._captured = False
        engine.defer_l2 = True      # the L2 term of the loss rides on the pass Adam makes over the parameters

    # ---- the pieces of a step
    def _part(self, k):
        e = self.eng
        if k == 0:
            e.forward_net()
            e.forward_loss(with_grad=True)
        e.backward_net(k)

    def capture(self):
        """One eager pass (grows the workspaces), then one hipGraph per backward part."""
        e = self.eng
        e.refresh_weight_planes(force=True)
        e.planes_external = self.reducer is not None     # from here on: re-split per bucket, after its update (_update)
        for k in range(self.nparts):
            self._part(k)
        torch.cuda.synchronize(e.dev)
        if not self.use_graph:
            return
        s = torch.cuda.Stream(e.dev)
        s.wait_stream(torch.cuda.current_stream(e.dev))
        with torch.cuda.stream(s):
            for k in range(self.nparts):
                self._part(k)
        torch.cuda.current_stream(e.dev).wait_stream(s)
        graphs, pool = [], None
        for k in range(self.nparts):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                self._part(k)
            pool = g.pool()
            graphs.append(g)
        self.graphs = graphs

    def step(self, im1, im2, lr, augment=None):
        """One optimisation step on the minibatch (im1, im2) [B,H,W,3] in [0,255]; returns the loss tensor [1] (complete
        when the stream has drained)."""
        e = self.eng
        if not self._captured:
            e.set_input(im1, im2, augment=augment)
            self.capture()
            self._captured = True
        e.set_input(im1, im2, augment=augment)
        if e.planes_external and e._wplanes_version != e.P._version:
            e.refresh_weight_planes(force=True)          # the parameters were written behind the runner's back (a restore)
        lr_t = e.adam_begin(lr)
        scale = 1.0 / self.world

        def update(ranges):
            for lo, hi in ranges:
                e.adam_range(lo, hi, lr_t, scale)
            if e.planes_external:
                e.refresh_weight_planes_ranges(ranges)

        for k in range(self.nparts):
            if self.graphs is not None:
                self.graphs[k].replay()
            else:
                self._part(k)
            if self.reducer is not None:
                ranges = self.buckets[k]
                # Frozen networks (stacked spec without train_all): zero data gradient everywhere, nothing to exchange, but
                # the optimizer still applies the L2 term to them.  Their update rides behind bucket 0 — i.e. behind part 0,
                # which holds the forward pass of those very networks and the loss_acc.zero_() of forward_loss: issued before
                # the replay it would overwrite P / the weight planes under the frozen stages' forward kernels and lose its
                # L2 loss term to the zeroing.
                upd = ranges + self.frozen if k == 0 else ranges
                self.reducer.reduce_then(ranges, lambda r=upd: update(r))
        if self.reducer is not None:
            self.reducer.finish()
        else:
            e.adam_range(0, e.n_params, lr_t, scale)
        if e.planes_external:
            e._wplanes_version = e.P._version
        return e.loss_acc


class Trainer:
    """One process per GPU.  `params` carries the reference's [train] keys (learning_rate, decay_interval, decay_after,
    manual_decay_*, flownet, train_all, full_res, loss weights, mask modes ...).  With torch.distributed initialised,
    every rank trains on its own shard of the minibatch and the gradients are averaged with a bucketed RCCL all-reduce
    (average_gradients semantics).  The training step augments like the reference's (unsupervised_loss(augment=True),
    train.py:160,170); `augment=False` switches that off."""

    ENGINE_KEYS = ('flownet', 'train_all', 'full_res', 'pyramid_loss', 'border_mask', 'mask_occlusion')

    def __init__(self, batch_size, height, width, params, device=None, seed=0, augment=True, use_graph=True):
        self.params = dict(params)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        eng_params = {k: v for k, v in params.items() if k.endswith('_weight') or k in self.ENGINE_KEYS}
        self.engine = FlowNetEngine(batch_size, height, width, params=eng_params or None, device=device, seed=seed)
        self.runner = StepRunner(self.engine, self.world, use_graph=use_graph)
        self.augment = augment
        self.generator = torch.Generator().manual_seed(1000003 * (seed + 1) + self.rank)   # per-rank augmentation draws
        self.iteration = 0

    # ---------------------------------------------------------------------------------------------- train.py:116-145, 247-262
    def checkpoint_step(self, ckpt_dir):
        """global_step of the latest checkpoint of a directory ('model.ckpt-<step>', train.py:124-127), or None."""
        from . import tf_checkpoint as T
        ckpt = T.latest_checkpoint(ckpt_dir)
        return None if ckpt is None else int(os.path.basename(ckpt).split('-')[-1])

    def _saved_networks(self):
        """Indices of the networks in the Saver's scope (train.py:32-37): every network with train_all, else the last one."""
        n = len(self.engine.spec)
        return list(range(n)) if self.params.get('train_all') else [n - 1]

    def _in_saved_scope(self, name):
        from .input import network_scope
        return any(name.startswith(sc) for i in self._saved_networks() for sc in network_scope(i))

    def save(self, ckpt_dir, global_step):
        """saver.save(sess, ckpt_dir/model.ckpt, global_step) (train.py:260-261): a TF checkpoint-V2 bundle under the reference's
        variable names + the directory's `checkpoint` state file, holding what the reference's scoped Saver holds — the trained
        networks (all with train_all, else the last one, train.py:32-37) and their Adam slots.  Rank 0 writes; every rank
        waits for the file before it goes on (a resume right after must find it)."""
        from .input import save_checkpoint
        if self.rank == 0:
            os.makedirs(ckpt_dir, exist_ok=True)
            tensors = self.engine.export_tf_params()
            tensors.update(self.engine.export_tf_adam_slots())      # '<var>/Adam', '<var>/Adam_1': in the Saver's scope too
            tensors = {k: v for k, v in tensors.items() if self._in_saved_scope(k)}
            save_checkpoint(os.path.join(ckpt_dir, 'model.ckpt-%d' % global_step), tensors)
        if self.world > 1:
            dist.barrier()

    def restore(self, ckpt_dir=None):
        """restore_networks (train.py:23-65).  Without a checkpoint in ckpt_dir: the networks named by params['finetune'] (in
        network order) are loaded, the rest keep their initialisation.  With one (continue training): the networks of the
        Saver's scope and their Adam slots come from the checkpoint, and — unless train_all — the frozen networks in front
        again from finetune[:n-1] (the reference's checkpoint of a stacked run does not contain them).  Returns the checkpoint
        prefix or None."""
        from . import tf_checkpoint as T
        from .input import restore_networks
        n = len(self.engine.spec)
        finetune = list(self.params.get('finetune') or [])
        if len(finetune) > n:
            raise ValueError("%d finetune entries for the %d networks of spec %r (train.py:31)" % (len(finetune), n, self.engine.spec))
        ckpt = T.latest_checkpoint(ckpt_dir) if ckpt_dir is not None else None
        files = [None] * n
        if ckpt is not None:
            have = set(T.checkpoint_entries(ckpt)[1])
            for i in self._saved_networks():
                files[i] = ckpt
            external = [] if self.params.get('train_all') else finetune[:n - 1]
            if not self.params.get('train_all') and not external:
                # a checkpoint written with every network in it (this trainer before round 4, or a train_all run continued
                # frozen): take the frozen networks from it rather than leaving them at their initialisation
                from .input import network_scope
                for i in range(n - 1):
                    if any(k.startswith(sc) for k in have for sc in network_scope(i)):
                        files[i] = ckpt
        else:
            external = finetune
        for i, f in enumerate(external):       # restored after the checkpoint, like the reference's second loop (:46-63)
            files[i] = f
        if any(f is not None for f in files):
            restore_networks(self.engine, self.params, files)
        if ckpt is not None:
            names = [k for k in T.checkpoint_entries(ckpt)[1] if (k.endswith('/Adam') or k.endswith('/Adam_1')) and self._in_saved_scope(k)]
            slots = {k: torch.from_numpy(v) for k, v in T.read_checkpoint(ckpt, names).items()}
            self.engine.load_tf_adam_slots(slots)
        return ckpt

    def run(self, min_iter, max_iter, train_batch_fn, ckpt_dir, eval_fn=None):
        """Trainer.run (train.py:116-145): train (at most) from min_iter + 1 to max_iter in chunks of params['save_interval']
        steps, a checkpoint after every chunk.  A checkpoint found in ckpt_dir must carry a global_step within [min_iter, max_iter];
        training then continues from global_step + 1.  train_batch_fn(iter_offset) returns an iterator of (im1, im2) batches
        already shifted by iter_offset steps (the reference builds its input queue with shift = batch_size * iter_offset);
        eval_fn(i), if given, runs after each chunk (self.eval(1) of the reference).  Returns the list of (iteration, loss) at the
        display interval."""
        save_interval = self.params['save_interval']
        global_step = self.checkpoint_step(ckpt_dir)
        if global_step is not None:
            assert global_step >= min_iter, 'training stage not reached'
            start_iter = global_step + 1
            if start_iter > max_iter:
                print('-- train: max_iter reached')
                return []
        else:
            start_iter = min_iter + 1
        print('-- training from i = {} to {}'.format(start_iter, max_iter))
        assert (max_iter - start_iter + 1) % save_interval == 0
        log = []
        for i in range(start_iter, max_iter + 1, save_interval):
            # every chunk of the reference builds a fresh graph and calls restore_networks (train.py:186-218): the networks of the
            # Saver's scope and their Adam slots continue from the checkpoint just written (or start from params['finetune'] /
            # their initialisation), frozen networks come from finetune again and every other optimizer slot restarts at zero
            self.restore(ckpt_dir)
            log += self.train(i, i + save_interval - 1, i - (min_iter + 1), train_batch_fn, ckpt_dir)
            if eval_fn is not None:
                eval_fn(i + save_interval - 1)
        return log

    def train(self, start_iter, max_iter, iter_offset, train_batch_fn, ckpt_dir):
        """Trainer.train (train.py:186-262): steps start_iter .. max_iter with the learning rate of decay_iters = local_i +
        iter_offset (learning_rate_at), the loss printed at i == 1 and every display_interval, one checkpoint at the end."""
        batches = iter(train_batch_fn(iter_offset))
        display = self.params.get('display_interval', 100)
        # every train() call of the reference builds a fresh graph and runs global_variables_initializer before the restore
        # (train.py:186-218, 40): beta1_power / beta2_power are not in the Saver's scope, so Adam's bias correction restarts at
        # t = 1 with every chunk while the moments continue — reproduced
        self.engine.step_count = 0
        log = []
        for local_i, i in enumerate(range(start_iter, max_iter + 1)):
            self.iteration = local_i + iter_offset
            im1, im2 = next(batches)
            loss = self.train_step(im1, im2)
            if i == 1 or i % display == 0:
                loss = float(loss)
                log.append((i, loss))
                print("-- train: i = {}, loss = {}".format(i, loss))
        self.save(ckpt_dir, max_iter)
        return log

    def train_step(self, im1, im2, augment=None):
        """sess.run([train_op, loss_]) (train.py:247-251): returns the loss tensor (device, no sync).  `augment`: None = the
        trainer's setting (random draws per step), False = off, or a dict of draws to replay."""
        lr = learning_rate_at(self.params, self.iteration)
        aug = self.augment if augment is None else augment
        if aug is True:
            from .augment import draw_training_augmentation
            aug = draw_training_augmentation(self.engine.B, self.generator)
        loss = self.runner.step(im1, im2, lr, augment=aug or None)
        self.iteration += 1
        return loss
