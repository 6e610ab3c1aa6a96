"""Helpers shared by the parity tests: error metrics, the oracle step, and the bench-identical launch path
(one eager step to grow the workspaces, then fwd + loss + bwd captured into the two hipGraphs bench.py replays)."""
import torch


def max_rel(a, b):
    """max|a-b| / max|b|: bound on the worst element in units of the tensor's scale."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def mean_rel(a, b):
    """mean|a-b| / mean|b|: the average element's relative error (does not hide behind one large element)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().mean() / (b.abs().mean() + 1e-300)).item()


def elem_rel(a, b, floor_frac=1e-3):
    """Element-wise relative error max_i |a_i-b_i| / (|b_i| + floor), floor = floor_frac * rms(b): every element is
    judged against its OWN magnitude down to a floor three decades under the tensor's rms."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    floor = floor_frac * b.pow(2).mean().sqrt() + 1e-300
    return ((a - b).abs() / (b.abs() + floor)).max().item()


def images(B, H, W, seed):
    """Second frame = first frame shifted + attenuated + noise, so the warps leave the trivial regime."""
    g = torch.Generator().manual_seed(seed)
    im1 = torch.rand(B, H, W, 3, generator=g) * 255
    im2 = torch.roll(im1, shifts=(2, -3), dims=(1, 2)) * 0.9 + torch.rand(B, H, W, 3, generator=g) * 25
    return im1, im2


def oracle_step(tf_params, im1, im2, params=None, dtype=torch.float64, backward=True):
    """(loss, final_flow_fw, final_flow_bw, grads or None) of the oracle's unsupervised step in `dtype`."""
    import gc
    from oracle import model_ref as M
    gc.collect()
    P = {k: v.clone().to(dtype) for k, v in tf_params.items()}
    if backward:
        for v in P.values():
            v.requires_grad_()
    with torch.set_grad_enabled(backward):
        loss, ffw, fbw, _ = M.unsupervised_loss(P, im1.to(dtype), im2.to(dtype), params, return_flow=True)
    grads = None
    if backward:
        loss.backward()
        grads = {k: v.grad for k, v in P.items()}
    return loss.item(), ffw.detach(), fbw.detach(), grads


def graph_step(eng, im1, im2):
    """Run one fwd + loss + bwd of `eng` the way bench.py does: an eager step first (grows the split-K workspaces),
    then the two captured hipGraphs (part a: forward, losses, deep backward; part b: shallow backward + bias
    gradients), replayed once.  Returns the loss value."""
    eng.set_input(im1, im2)
    eng.forward_net()
    eng.forward_loss(with_grad=True)
    eng.backward_net()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.forward_net()
        eng.forward_loss(with_grad=True)
        eng.backward_net(0)
        eng.backward_net(1)
    torch.cuda.current_stream().wait_stream(s)
    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga):
        eng.forward_net()
        eng.forward_loss(with_grad=True)
        eng.backward_net(0)
    with torch.cuda.graph(gb, pool=ga.pool()):
        eng.backward_net(1)
    eng.G.zero_()                      # the replay must produce every gradient itself
    ga.replay()
    gb.replay()
    torch.cuda.synchronize()
    return eng.loss_acc.item()


def check_grads(got, grads_ref, tf_params, max_tol, mean_tol, skip=None, min_mean_numel=1024, label="", small_tol=None):
    """Every parameter gradient of the engine (data loss only: the L2 gradient 0.0004*w is fused into Adam) against the
    oracle's: max-normalised bound on every tensor, mean-relative bound on every tensor with >= min_mean_numel elements.
    small_tol (default max_tol): max-normalised bound for tensors with <= 64 elements (the 2 -> 2 deconvs and 2-element
    biases, whose few-hundred-term sums cancel heavily).  Prints the table of the worst tensors, then asserts.
    Returns (worst max_rel, worst mean_rel)."""
    small_tol = max_tol if small_tol is None else small_tol
    rows = []
    for k, gr in grads_ref.items():
        if skip is not None and skip(k):
            continue
        l2 = 0.0004 * tf_params[k].double() if k.endswith('/weights') else 0.0
        ref = gr.double() - l2
        e = max_rel(got[k], ref)
        m = mean_rel(got[k], ref) if ref.numel() >= min_mean_numel else float('nan')
        rows.append((e, m, k, ref.numel()))
    rows.sort(reverse=True)
    print("%s gradient errors vs oracle (worst first): max|d|/max|ref|, mean|d|/mean|ref|" % label)
    for e, m, k, n in rows[:10]:
        print("   %-48s n=%-9d max-rel %.2e  mean-rel %.2e" % (k, n, e, m))
    worst_max = rows[0][0]
    worst_mean = max((m for _, m, _, _ in rows if m == m), default=0.0)
    bad = [(k, e, m) for e, m, k, n in rows if e >= (small_tol if n <= 64 else max_tol) or (m == m and m >= mean_tol)]
    assert not bad, bad
    return worst_max, worst_mean


class _LeakyWithBranch(torch.autograd.Function):
    """max(0.1 x, x) with the derivative branch (1 or 0.1) taken from a given mask instead of from sign(x)."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return torch.maximum(0.1 * x, x)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * torch.where(mask, 1.0, 0.1).to(g.dtype), None


# (engine buffer, channel lo, channel hi) of every leaky-ReLU layer of FlowNetC in the oracle's call order
_FEATURE_ACTS = [('c1', 0, 64), ('cat2', 0, 128), ('c3', 0, 256)]
_FLOWNETC_ACTS = [('catc', 0, 32), ('cat3', 0, 256), ('c4', 0, 512), ('cat4', 0, 512), ('c5', 0, 512), ('cat5', 0, 512),
                  ('c6', 0, 1024), ('c6_1', 0, 1024), ('cat5', 512, 1024), ('cat4', 512, 768), ('cat3', 256, 384),
                  ('cat2', 128, 192)]


_FLOWNETS_ACTS = [('c1', 0, 64), ('cat2', 0, 128), ('c3', 0, 256)] + [('cat3', 0, 256)] + _FLOWNETC_ACTS[2:]


def flownet_c_order(B, act=None):
    """flownet.py:30-44: features of im1, features of im2, flownet_c forward direction, flownet_c backward direction.
    Entries: (buffer spec, sample slice, activation dict of the stage or None = the one given to BranchAligned)."""
    return [(a, slice(0, B), act) for a in _FEATURE_ACTS] + [(a, slice(B, 2 * B), act) for a in _FEATURE_ACTS] + \
           [(a, slice(0, B), act) for a in _FLOWNETC_ACTS] + [(a, slice(B, 2 * B), act) for a in _FLOWNETC_ACTS]


def flownet_s_order(B, act=None):
    """flownet.py:58-67: flownet_s on the forward inputs, then on the backward inputs."""
    return [(a, slice(0, B), act) for a in _FLOWNETS_ACTS] + [(a, slice(B, 2 * B), act) for a in _FLOWNETS_ACTS]


def engine_order(eng):
    """Leaky-ReLU call order of the oracle's flownet() for the engine's whole (possibly stacked) spec."""
    out = []
    for st in eng.stages:
        assert st.kind in 'CS' and not st.full_res, "full-width nets without full_res only"
        out += flownet_c_order(eng.B, st.act) if st.is_c else flownet_s_order(eng.B, st.act)
    return out


class BranchAligned:
    """Context manager: while active, the oracle's leaky-ReLUs (in call order `order`, each mapped to a channel slice of an
    engine activation buffer in `act`) differentiate along the branch the ENGINE took (its stored activation > 0), and
    the units where fp64 and the engine disagree are counted.

    Why: tf.maximum(0.1 x, x) has a kink at 0.  A pre-activation within fp32 noise of 0 lands on different sides in
    the fp32 HIP path and the fp64 oracle; the VALUE is continuous (difference ~1e-7), the derivative jumps 0.1 <-> 1
    for that unit.  At 384x512 there are 2e7 such units per sample pair, so a handful flip, and each flip moves a conv1
    filter-gradient element (a sum over ~1e5 pixels with random signs) by ~1e-3 of the tensor's max.  Both are valid
    sub-gradients of the same function; aligning the branch isolates everything ELSE."""

    def __init__(self, act, order):
        self.act, self.order, self.calls, self.flips, self.units = act, order, 0, 0, 0

    def _hook(self, x):
        (name, lo, hi), smp, act = self.order[self.calls]
        self.calls += 1
        y = (self.act if act is None else act)[name][smp, :, :, lo:hi]
        mask = (y > 0).permute(0, 3, 1, 2).cpu()
        assert mask.shape == x.shape, (name, mask.shape, x.shape)
        self.flips += int(((x.detach() > 0) != mask).sum())
        self.units += x.numel()
        return _LeakyWithBranch.apply(x, mask)

    def __enter__(self):
        from oracle import model_ref as M
        M.LEAKY_HOOK = self._hook
        return self

    def __exit__(self, *exc):
        from oracle import model_ref as M
        M.LEAKY_HOOK = None
        if exc[0] is None:
            assert self.calls == len(self.order), (self.calls, len(self.order))
