"""-m gpu: BASELINE configs[4] — FlowNetC with fp16 activations and weights into the fp16 MFMA (fp32 accumulate).

The reference ops are float-only (REGISTER_OP(... ": float"), ops/correlation_op.cc:134-135), so there is no reference
result for this mode; its tolerance against the fp32 oracle is STATED here: loss within 1e-2 relative, final
flow EPE within 5e-2 px.  The correlation, the warps, the loss pyramid, the flow heads and Adam stay fp32; the master
weights are fp32 (their fp16 planes are refreshed after every update)."""
import pytest
import torch

from parity_util import images, oracle_step

pytestmark = pytest.mark.gpu

F16_TENSOR_TOL = 0.35     # PROVISIONAL (measured 0.31 on conv6_1 at 128x192 without gradient scaling): per-tensor gradient error of the fp16-operand mode vs the fp32 oracle, max-normalised


@pytest.mark.parametrize("shape", [(1, 128, 192), (2, 384, 512)])
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
    # per tensor (a cosine over the flat gradient says nothing about a small tensor): max |d| / max |ref| of every tensor with
    # >= 1024 elements within F16_TENSOR_TOL, the 2-channel flow heads / biases within 3x that
    rows = []
    for k in grads:
        ref = grads[k].double() - (0.0004 * tf_params[k].double() if k.endswith('/weights') else 0.0)
        d = (got[k].double() - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
        rows.append((d, k, ref.numel()))
    rows.sort(reverse=True)
    for d, k, n in rows[:8]:
        print("   f16 gradient %-48s n=%-9d max-rel %.2e" % (k, n, d))
    bad = [(k, d) for d, k, n in rows if d > (F16_TENSOR_TOL if n >= 1024 else 3 * F16_TENSOR_TOL)]
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
