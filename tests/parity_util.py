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
    from oracle import model_ref as M
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


def check_grads(got, grads_ref, tf_params, max_tol, mean_tol, skip=None, min_mean_numel=1024):
    """Every parameter gradient of the engine (data loss only: the L2 gradient 0.0004*w is fused into Adam) against the
    oracle's: max-normalised bound on every tensor, mean-relative bound on every tensor with >= min_mean_numel elements.
    Returns (worst max_rel, worst mean_rel)."""
    worst = [0.0, 0.0]
    for k, gr in grads_ref.items():
        if skip is not None and skip(k):
            continue
        l2 = 0.0004 * tf_params[k].double() if k.endswith('/weights') else 0.0
        ref = gr.double() - l2
        e = max_rel(got[k], ref)
        worst[0] = max(worst[0], e)
        assert e < max_tol, (k, 'max_rel', e)
        if ref.numel() >= min_mean_numel:
            m = mean_rel(got[k], ref)
            worst[1] = max(worst[1], m)
            assert m < mean_tol, (k, 'mean_rel', m)
    return tuple(worst)
