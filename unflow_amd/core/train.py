"""Mirror of the hot-path parts of src/e2eflow/core/train.py: the training op (get_train_and_loss_ops :147-185:
AdamOptimizer + towers + average_gradients :388-422), the learning-rate schedule (:225-244) and the loop body
(:247-251), Trainer.run / train with checkpoint save and resume (:116-145, :186-262; TF checkpoint-V2 bundles written and
read by core/tf_checkpoint.py), and Trainer.eval (:265-385: batch-1 evaluation against KITTI-format ground truth; the TF
summaries and the plot process around it are out of scope).

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
    """The schedule of train.py:225-244 as a function of the step index within the run.
    Manual form (both manual_decay_* keys given) wins: stage k lasts manual_decay_iters[k] steps and uses manual_decay_lrs[k];
    a step belongs to the first stage whose cumulative end it does not exceed, and — a quirk of the reference's loop kept
    on purpose — a step past the last stage falls back to stage 0.  Otherwise: constant until decay_after, then divided by
    2^(floor(step / decay_interval) - decay_after / decay_interval) (a true division: fractional exponents occur)."""
    from itertools import accumulate
    base = params.get('learning_rate')
    if 'manual_decay_lrs' in params and 'manual_decay_iters' in params:
        stage = next((k for k, end in enumerate(accumulate(params['manual_decay_iters'])) if decay_iters <= end), 0)
        return params['manual_decay_lrs'][stage]
    every = params.get('decay_interval')
    if every is None:
        return base
    start = params.get('decay_after', 0)
    if decay_iters < start:
        return base
    return base / 2 ** (decay_iters // every - start / every)


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
        # from here on the captured forward does not re-split the weights: the optimizer update does (fused kernel), or — two
        # launches — the runner right after each bucket's update (_update)
        e.planes_external = self.reducer is not None or (e.fused_adam and e.n_planes > 0)
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
            if e.fused_adam:
                e.adam_ranges_fused(ranges, lr_t, scale)       # L2 + Adam + the planes of these ranges, one launch
                return
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
        elif e.fused_adam:
            e.adam_ranges_fused([(0, e.n_params)], lr_t, scale)
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
            if self.runner.reducer is not None:
                self.runner.reducer.quiesce()       # the library's own communicator drains before the process group's barrier
            dist.barrier()

    def restore(self, ckpt_dir=None, engine=None):
        """restore_networks (train.py:23-65).  Without a checkpoint in ckpt_dir: the networks named by params['finetune'] (in
        network order) are loaded, the rest keep their initialisation.  With one (continue training): the networks of the
        Saver's scope and their Adam slots come from the checkpoint, and — unless train_all — the frozen networks in front
        again from finetune[:n-1] (the reference's checkpoint of a stacked run does not contain them).  Returns the checkpoint
        prefix or None."""
        from . import tf_checkpoint as T
        from .input import restore_networks, network_scope
        engine = self.engine if engine is None else engine
        n = len(engine.spec)
        finetune = list(self.params.get('finetune') or [])
        if len(finetune) > n:
            raise ValueError("%d finetune entries for the %d networks of spec %r (train.py:31)" % (len(finetune), n, engine.spec))
        ckpt = T.latest_checkpoint(ckpt_dir) if ckpt_dir is not None else None
        files = [None] * n
        if ckpt is not None:
            have = set(T.checkpoint_entries(ckpt)[1])
            for i in self._saved_networks():
                files[i] = ckpt
            external = [] if self.params.get('train_all') else finetune[:n - 1]
            if not self.params.get('train_all'):
                # network by network: a frozen network without a finetune entry comes from the checkpoint when the checkpoint
                # holds it (written with every network in it: this trainer before round 4, or a train_all run continued
                # frozen); one that is in neither place would silently run on its random initialisation — say so
                for i in range(n - 1):
                    if i < len(external) and external[i] is not None:
                        continue
                    if any(k.startswith(sc) for k in have for sc in network_scope(i)):
                        files[i] = ckpt
                    else:
                        import warnings
                        warnings.warn("frozen network %d of spec %r is neither in params['finetune'] nor in %s: it keeps its "
                                      "random initialisation" % (i, engine.spec, ckpt))
        else:
            external = finetune
        for i, f in enumerate(external):       # restored after the checkpoint, like the reference's second loop (:46-63)
            if f is not None:
                files[i] = f
        if any(f is not None for f in files):
            restore_networks(engine, self.params, files)
        if ckpt is not None and engine is self.engine:
            names = [k for k in T.checkpoint_entries(ckpt)[1] if (k.endswith('/Adam') or k.endswith('/Adam_1')) and self._in_saved_scope(k)]
            slots = {k: torch.from_numpy(v) for k, v in T.read_checkpoint(ckpt, names).items()}
            engine.load_tf_adam_slots(slots)
        return ckpt

    # ---------------------------------------------------------------------------------------------- train.py:265-385
    def eval(self, eval_batch_fn, ckpt_dir, num=1, resized=(384, 1280)):
        """Trainer.eval (train.py:265-385) without the TF summaries / plot process: every batch-1 example of eval_batch_fn()
        — (im1, im2, input_shape, flow_occ, mask_occ, flow_noc, mask_noc), kitti/input.py:75-82 — is brought from the
        input pipeline's crop / pad back to its own size and stretched onto the 384 x 1280 network input (resize_input),
        run through unsupervised_loss(augment=False, return_flow=True) with the networks of the latest checkpoint of
        ckpt_dir (restore_networks), the final flow is resized to the frame with per-axis rescaling (resize_output_flow) and
        compared with both ground-truth maps.  Returns what the reference writes to its 'eval_avg' summaries: the averages
        over the examples of AEE/<name>, outliers/<name> (name = occluded, non-occluded) and the loss; plus global_step
        and the example count.  `per_example` (list of the per-example values) is kept for tests."""
        from . import tf_checkpoint as T
        from .flow_util import flow_error_avg, outlier_pct
        from .input import resize_input, resize_output_crop, resize_output_flow
        from .unsupervised import unsupervised_loss
        assert num == 1                                                       # train.py:266
        ckpt = T.latest_checkpoint(ckpt_dir)
        assert ckpt is not None, "No checkpoints to evaluate"                 # train.py:330
        rh, rw = resized
        dev = self.engine.dev
        if getattr(self, '_eval_engine', None) is None or (self._eval_engine.H, self._eval_engine.W) != (rh, rw):
            eng_params = {k: v for k, v in self.params.items() if k.endswith('_weight') or k in self.ENGINE_KEYS}
            self._eval_engine = FlowNetEngine(1, rh, rw, params=eng_params or None, device=dev, seed=0)
        eng = self._eval_engine
        self.restore(ckpt_dir, engine=eng)
        names = ['AEE/occluded', 'outliers/occluded', 'AEE/non-occluded', 'outliers/non-occluded', 'loss']
        sums = [0.0] * len(names)
        per_example = []
        for batch in eval_batch_fn():
            im1, im2, input_shape = (torch.as_tensor(t) for t in batch[:3])
            truths = [torch.as_tensor(t).float().to(dev) for t in batch[3:]]
            if len(truths) != 4:
                raise NotImplementedError()                                   # train.py:306-307
            height, width = int(input_shape.reshape(-1)[0]), int(input_shape.reshape(-1)[1])
            a = resize_input(im1.float().to(dev), height, width, rh, rw)
            b = resize_input(im2.float().to(dev), height, width, rh, rw)
            loss, flow, flow_bw = unsupervised_loss((a, b), self.params, augment=False, return_flow=True, engine=eng)
            flow = resize_output_flow(flow, height, width, 2)
            flow_occ, mask_occ, flow_noc, mask_noc = truths
            vals = []
            for gt, mask in ((resize_output_crop(flow_occ, height, width, 2), resize_output_crop(mask_occ, height, width, 1)),
                             (resize_output_crop(flow_noc, height, width, 2), resize_output_crop(mask_noc, height, width, 1))):
                vals += [float(flow_error_avg(gt, flow, mask)), float(outlier_pct(gt, flow, mask))]
            vals.append(float(loss))
            per_example.append(vals)
            sums = [s0 + v for s0, v in zip(sums, vals)]
        n_ex = len(per_example)
        assert n_ex > 0, "eval_batch_fn() yielded no examples"
        global_step = int(os.path.basename(ckpt).split('-')[-1])
        print("-- eval: i = {}".format(global_step))
        out = {k: s0 / n_ex for k, s0 in zip(names, sums)}
        out.update(global_step=global_step, num_examples=n_ex, per_example=per_example, names=names)
        return out

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
                self._check_faults()                    # the host has just synchronised on the loss: read the fault counters
                log.append((i, loss))
                print("-- train: i = {}, loss = {}".format(i, loss))
        self._check_faults()                            # never checkpoint a run whose kernels reported a fault
        self.save(ckpt_dir, max_iter)
        return log

    def _check_faults(self):
        """Device fault counters, agreed over the ranks (every rank reaches this at the same iterations): with more than one
        rank the library's own communicator is drained first, then one MAX all-reduce — all ranks raise, or none."""
        if self.world > 1 and self.runner.reducer is not None:
            self.runner.reducer.quiesce()
        self.engine.check_device_faults(world_sync=self.world > 1)

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
