"""Data-parallel gradient exchange — replaces the reference's in-graph towers + CPU-side
average_gradients (src/e2eflow/core/train.py:163-183,388-422) with one process per GPU and an RCCL
all-reduce of the flat fp32 gradient buffer over xGMI.

Semantics kept: mean over replicas of the per-replica mean-loss gradient (average_gradients =
concat + reduce_mean).  Deviation: the reference feeds the SAME minibatch to every tower;
here every rank gets its own shard.  The 1/world scaling is fused into the Adam kernel (grad_scale).

The gradient buffer is reduced in a few large buckets (xGMI is point-to-point, 7 links x ~153 GB/s per
GPU: few, large messages), issued on a side stream so the exchange overlaps with whatever the compute
stream still has queued.  Works with any torch.distributed backend (gloo on CPU for the tests).
"""
import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, flat_grad, world_size, bucket_bytes=64 << 20, group=None, force=False, local_overlap=False):
        self.g = flat_grad
        self.world = world_size
        self.force = force      # issue the collectives even for world_size 1 (exercises the stream logic on one GPU)
        # one rank, no process group: reduce_then() still moves fn() to the side stream, so the bucketed optimizer update
        # runs beside the rest of the backward pass exactly as it does behind an exchange
        self.local_overlap = local_overlap and world_size <= 1 and not force
        self.group = group
        n = flat_grad.numel()
        per = max(1, bucket_bytes // 4)
        self.per = per
        self._pending = []
        self.bounds = [(i, min(n, i + per)) for i in range(0, n, per)]
        self.cuda = flat_grad.is_cuda
        self.stream = torch.cuda.Stream(device=flat_grad.device) if self.cuda else None
        self.dry = False        # measurement switch (bench.py comm record): reduce_then keeps its stream order but skips the collectives

    def all_reduce(self):
        """SUM over ranks, in place (scale by 1/world in the optimizer).  Returns after enqueueing; the
        compute stream is made to wait for the exchange."""
        if self.world <= 1:
            return
        if self.cuda:
            cur = torch.cuda.current_stream(self.g.device)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                works = [dist.all_reduce(self.g[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                         for a, b in self.bounds]
                for w in works:
                    w.wait()
            cur.wait_stream(self.stream)
        else:
            for a, b in self.bounds:
                dist.all_reduce(self.g[a:b], op=dist.ReduceOp.SUM, group=self.group)

    def start_ranges(self, ranges):
        """Enqueue the SUM all-reduce of flat ranges [(lo, hi), ...] on the side stream, ordered after everything the
        current stream has queued so far; returns immediately.  finish() makes the current stream wait for all
        exchanges started since the last finish()."""
        if self.world <= 1 and not self.force:
            return
        for lo, hi in ranges:
            for a in range(lo, hi, self.per):
                b = min(hi, a + self.per)
                if self.cuda:
                    cur = torch.cuda.current_stream(self.g.device)
                    self.stream.wait_stream(cur)
                    with torch.cuda.stream(self.stream):
                        self._pending.append(dist.all_reduce(self.g[a:b], op=dist.ReduceOp.SUM, group=self.group,
                                                             async_op=True))
                else:
                    dist.all_reduce(self.g[a:b], op=dist.ReduceOp.SUM, group=self.group)

    def reduce_then(self, ranges, fn):
        """Enqueue on the side stream: the SUM all-reduce of flat ranges [(lo, hi), ...] (ordered after everything the
        current stream has queued so far), then fn() — launched with the side stream current, i.e. ordered after the
        exchange and running concurrently with whatever the compute stream does next (the bucketed optimizer update:
        train.py StepRunner).  Returns immediately; finish() joins."""
        if self.world <= 1 and not self.force:
            if self.local_overlap and self.cuda:
                self.stream.wait_stream(torch.cuda.current_stream(self.g.device))
                with torch.cuda.stream(self.stream):
                    fn()
            else:
                fn()
            return
        if not self.cuda:
            for lo, hi in ranges:
                for a in range(lo, hi, self.per):
                    dist.all_reduce(self.g[a:min(hi, a + self.per)], op=dist.ReduceOp.SUM, group=self.group)
            fn()
            return
        cur = torch.cuda.current_stream(self.g.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            works = [] if self.dry else [
                dist.all_reduce(self.g[a:min(hi, a + self.per)], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                for lo, hi in ranges for a in range(lo, hi, self.per)]
            for w in works:
                w.wait()          # stream-level: the side stream waits for the collective, the host does not
            fn()

    def finish(self):
        if self.world <= 1 and not self.force:
            if self.local_overlap and self.cuda:
                torch.cuda.current_stream(self.g.device).wait_stream(self.stream)
            return
        if self.cuda:
            with torch.cuda.stream(self.stream):
                for w in self._pending:
                    w.wait()
            torch.cuda.current_stream(self.g.device).wait_stream(self.stream)
        self._pending = []

    def mean_(self):
        """all_reduce + divide: the semantics of train.py:388-422 in one call (used when the optimizer does not
        fuse the scaling)."""
        self.all_reduce()
        self.g.mul_(1.0 / self.world)
        return self.g
