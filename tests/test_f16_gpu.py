"""-m gpu: BASELINE configs[4] — FlowNetC with fp16 activations and weights into the fp16 MFMA (fp32 accumulate).

The reference ops are float-only (REGISTER_OP(... ": float"), ops/correlation_op.cc:134-135), so there is no reference
result for this mode; its tolerance against the fp32 oracle is STATED here: loss within 1e-2 relative, final
flow EPE within 5e-2 px.  The correlation, the warps, the loss pyramid, the flow heads and Adam stay fp32; the master
weights are fp32 (their fp16 planes are refreshed after every update)."""
import pytest
import torch

from parity_util import images, oracle_step

pytestmark = pytest.mark.gpu

F16_TENSOR_TOL = 3e-2     # stated: per-tensor gradient error of the fp16-operand mode vs the fp32 oracle, max-normalised


# (8, 384, 512) is BASELINE configs[4] itself — the shape bench.py reports as its fp16 secondary line
@pytest.mark.parametrize("shape", [(1, 128, 192), (2, 384, 512), (8, 384, 512)])
def test_f16_step_vs_fp32_oracle(shape, dev, monkeypatch):
    from unflow_amd.core.engine import FlowNetCEngine, flow_error_avg
    monkeypatch.setenv("UNFLOW_CONV_MATH", "f16")
    B, H, W = shape
    eng = FlowNetCEngine(B, H, W, device=dev, seed=None)
    assert eng.math == "f16" and eng.n_planes == 1
    tf_params = eng.init_params(seed=5)
    im1, im2 = images(B, H, W, 6)
    loss = eng.fwd_bwd(im1.to(dev), im2.to(dev)).item()
    fw, bw = eng.final_flows()
    got = eng.export_tf_grads()
    loss_ref, ffw, fbw, grads = oracle_step(tf_params, im1, im2, dtype=torch.float32)
    e_loss = abs(loss - loss_ref) / abs(loss_ref)
    e_fw, e_bw = flow_error_avg(fw, ffw.to(dev)).item(), flow_error_avg(bw, fbw.to(dev)).item()
    # gradients: cosine similarity of the whole flat gradient (the quantity an optimizer step follows)
    a = torch.cat([got[k].flatten().double() for k in grads])
    b = torch.cat([(grads[k].double() - (0.0004 * tf_params[k].double() if k.endswith('/weights') else 0.0)).flatten()
                   for k in grads])
    cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
    print("f16 %s: loss rel %.2e, EPE fw %.2e bw %.2e px, gradient cosine %.6f" % (shape, e_loss, e_fw, e_bw, cos))
    assert e_loss <= 1e-2 and e_fw <= 5e-2 and e_bw <= 5e-2
    assert cos > 0.99
    # per tensor (a cosine over the flat gradient says nothing about a small tensor).  fp16 rounding moves a few leaky-ReLU
    # units across the kink (derivative 1 <-> 0.1), which changes single gradient entries by up to a third of the tensor's
    # maximum on the small deep layers whatever the arithmetic precision — so the bound is asserted against the oracle
    # differentiated along the ENGINE's branches (parity_util.BranchAligned, as in the fp32 parity tests), and the plain
    # comparison is printed.
    from parity_util import BranchAligned, flownet_c_order
    with BranchAligned(eng.act, flownet_c_order(B)) as al:
        _, _, _, grads_al = oracle_step(tf_params, im1, im2, dtype=torch.float32)

    def table(ref_grads):
        rows = []
        for k in ref_grads:
            ref = ref_grads[k].double() - (0.0004 * tf_params[k].double() if k.endswith('/weights') else 0.0)
            d = (got[k].double() - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
            rows.append((d, k, ref.numel()))
        rows.sort(reverse=True)
        return rows
    plain, aligned = table(grads), table(grads_al)
    print("   f16 %s: %d of %d leaky units flipped; worst tensor plain %.2e (%s), branch-aligned %.2e (%s)"
          % (shape, al.flips, al.units, plain[0][0], plain[0][1], aligned[0][0], aligned[0][1]))
    for d, k, n in aligned[:6]:
        print("   f16 gradient (aligned) %-48s n=%-9d max-rel %.2e" % (k, n, d))
    # stated bound at the benchmarked resolution (measured 2.0e-2); the 128 x 192 smoke shape sums 9x fewer pixels per
    # gradient element, so fp16 rounding averages less there (measured 4.6e-2 on tensors >= 1024 elements): twice the bound;
    # tensors below 1024 elements (2-element flow biases, 2 -> 2 upsamplers): 5x
    tol = F16_TENSOR_TOL * (1.0 if H >= 384 else 2.0)
    bad = [(k, d) for d, k, n in aligned if d > (tol if n >= 1024 else 5 * tol)]
    assert not bad, bad


def test_f16_training_tracks_fp32(dev, monkeypatch):
    """30 Adam steps on one batch from the same initialisation: the fp16-operand run follows the fp32-equivalent one."""
    from unflow_amd.core.engine import FlowNetCEngine
    B, H, W = 1, 128, 192
    im1, im2 = images(B, H, W, 9)
    curves = {}
    for mode in ("bf16x3", "f16"):
        monkeypatch.setenv("UNFLOW_CONV_MATH", mode)
        eng = FlowNetCEngine(B, H, W, device=dev, seed=4)
        ls = []
        for _ in range(30):
            ls.append(eng.train_step(im1.to(dev), im2.to(dev), 1e-4).item())
        curves[mode] = ls
    a, b = curves["bf16x3"], curves["f16"]
    assert b[-1] < b[0]                                                   # it trains
    worst = max(abs(x - y) / abs(x) for x, y in zip(a, b))
    print("f16 vs fp32-equivalent loss curves, 30 steps: %.4f -> %.4f vs %.4f -> %.4f, worst rel diff %.2e"
          % (a[0], a[-1], b[0], b[-1], worst))
    assert worst < 2e-2
