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
import ctypes
import os

import torch
import torch.distributed as dist


class RcclComm:
    """An RCCL communicator behind the C ABI (csrc/comm_rccl.hip: unflow_comm_* / unflow_allreduce_sum_f32) — the exchange
    itself without torch.distributed in the data path: ncclAllReduce is enqueued on the stream the caller names, nothing else.
    Bootstrap: rank 0's 128-byte unique id travels over the process group that torch.distributed already has (any backend,
    gloo included); every rank must have its GPU current (torch.cuda.set_device) when this is constructed."""

    def __init__(self, world, rank, group=None):
        from .. import _lib
        self.lib = _lib.lib()
        self.version = self.lib.unflow_comm_available()
        if not self.version:
            raise RuntimeError("RCCL is not available to libunflow_hip.so (dlopen of librccl.so failed)")
        uid = ctypes.create_string_buffer(128)
        if rank == 0:
            _lib.check(self.lib.unflow_comm_unique_id(uid), "comm_unique_id")
        if world > 1:
            box = [bytes(uid.raw)]
            # `src` is a GLOBAL rank: the group's own rank 0 (a sub-group need not contain global rank 0)
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            uid = ctypes.create_string_buffer(box[0], 128)
        self.comm = ctypes.c_void_p()
        _lib.check(self.lib.unflow_comm_init(uid, int(world), int(rank), ctypes.byref(self.comm)), "comm_init")
        n, r = ctypes.c_int(), ctypes.c_int()
        _lib.check(self.lib.unflow_comm_info(self.comm, ctypes.byref(n), ctypes.byref(r)), "comm_info")
        self.world, self.rank = n.value, r.value
        assert (self.world, self.rank) == (world, rank), "RCCL reports rank %d of %d" % (self.rank, self.world)

    def all_reduce_sum_(self, t, stream):
        """In-place SUM of a contiguous fp32 CUDA tensor, enqueued on `stream` (a torch.cuda.Stream)."""
        from .. import _lib
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        _lib.check(self.lib.unflow_allreduce_sum_f32(ctypes.c_void_p(t.data_ptr()), ctypes.c_long(t.numel()), self.comm,
                                                     ctypes.c_void_p(stream.cuda_stream)), "allreduce_sum_f32")

    def close(self):
        if self.comm:
            self.lib.unflow_comm_destroy(self.comm)
            self.comm = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Enqueued:
    """What the direct transport returns in place of a torch.distributed work object: the collective is already ordered on
    the communication stream."""

    def wait(self):
        pass


class GradAllReducer:
    def __init__(self, flat_grad, world_size, bucket_bytes=64 << 20, group=None, force=False, local_overlap=False, transport=None):
        """transport: 'torch' — dist.all_reduce of the process group (RCCL when its backend is nccl; gloo for the CPU tests);
        'rccl' — ncclAllReduce through the library's own C ABI (RcclComm) on the communication stream.  Default: the
        UNFLOW_COMM environment variable, else 'torch'."""
        self.g = flat_grad
        self.transport = transport or os.environ.get("UNFLOW_COMM", "torch")
        if self.transport not in ("torch", "rccl"):
            raise ValueError("transport must be 'torch' or 'rccl'")
        self.rccl = None
        if self.transport == "rccl" and flat_grad.is_cuda and (world_size > 1 or force):
            rank = dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0
            self.rccl = RcclComm(world_size, rank, group)
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

    def quiesce(self):
        """Host-side join of the communication stream.  With transport 'rccl' this process holds TWO RCCL communicators on
        the same GPU (this one and the process group's); collectives of two communicators in flight at once can deadlock
        unless their order is the same on every rank — so anything that is about to call a torch.distributed collective
        (Trainer.save's barrier, bench.py's checksums) drains this one first."""
        if self.cuda and self.rccl is not None:
            self.stream.synchronize()

    def close(self):
        """Release the library's communicator (idempotent; the process group is the caller's)."""
        self.quiesce()
        if self.rccl is not None:
            self.rccl.close()
            self.rccl = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _issue(self, a, b):
        """One SUM all-reduce of g[a:b], issued with the communication stream current; returns something with .wait()."""
        if self.rccl is not None:
            self.rccl.all_reduce_sum_(self.g[a:b], self.stream)
            return _Enqueued()
        return dist.all_reduce(self.g[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def all_reduce(self):
        """SUM over ranks, in place (scale by 1/world in the optimizer).  Returns after enqueueing; the
        compute stream is made to wait for the exchange."""
        if self.world <= 1 and self.rccl is None:
            return
        if self.cuda:
            cur = torch.cuda.current_stream(self.g.device)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                works = [self._issue(a, b) for a, b in self.bounds]
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
                        self._pending.append(self._issue(a, b))
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
            works = [] if self.dry else [self._issue(a, min(hi, a + self.per)) for lo, hi in ranges for a in range(lo, hi, self.per)]
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
