"""-m gpu: the training-step runner and the API mirrors.

  * StepRunner with the nccl backend at world size 1 (bucketed all-reduce + bucketed Adam on the communication stream, three
    hipGraph parts) leaves bit-identical parameters / moments as the single-graph, single-Adam path;
  * Trainer forwards train_all / full_res to the engine, augments by default, follows the manual learning-rate list;
  * core.flownet.flownet / core.unsupervised.unsupervised_loss / core.losses.* (the API mirrors) agree with the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from parity_util import images

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_NCCL_CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from unflow_amd.core.engine import FlowNetEngine, DEFAULT_PARAMS
from unflow_amd.core.train import StepRunner
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_world_size() == 1
B, H, W = 2, 128, 192
g = torch.Generator().manual_seed(3)
batches = [((torch.rand(B, H, W, 3, generator=g) * 255).to(dev), (torch.rand(B, H, W, 3, generator=g) * 255).to(dev)) for _ in range(3)]
# 'CS': a frozen first network (flownet.py:51-54) whose L2-only update must not race with its forward pass (ADVICE r2)
for spec in ('C', 'CS'):
    res = []
    # plain path; bucketed path through torch.distributed (backend nccl); bucketed path through the library's own C ABI
    # (csrc/comm_rccl.hip: a one-rank RCCL communicator from unflow_comm_init, ncclAllReduce on the communication stream)
    for force, transport in ((False, None), (True, 'torch'), (True, 'rccl')):
        eng = FlowNetEngine(B, H, W, params=dict(DEFAULT_PARAMS, flownet=spec), device=dev, seed=7)
        run = StepRunner(eng, 1, use_graph=True, force_reducer=force, transport=transport)
        assert run.nparts == (3 if force else 1)
        if transport == 'rccl':
            assert run.reducer.rccl is not None and (run.reducer.rccl.world, run.reducer.rccl.rank) == (1, 0)
        assert bool(run.frozen) == (spec == 'CS')
        losses = []
        for i in range(4):
            losses.append(run.step(*batches[i %% 3], 1e-4).item())
        torch.cuda.synchronize()
        res.append((eng.P.clone(), eng.M.clone(), eng.V.clone(), losses))
    (p0, m0, v0, l0) = res[0]
    for (p1, m1, v1, l1) in res[1:]:
        assert torch.equal(p0, p1) and torch.equal(m0, m1) and torch.equal(v0, v1), "bucketed path differs (%%s)" %% spec
        assert all(abs(a - b) <= 1e-4 * abs(a) for a, b in zip(l0, l1)), (spec, l0, l1)    # (the L2 loss term is summed per bucket: float-atomic order)
    assert l0[-1] != l0[0]
dist.barrier(); dist.destroy_process_group()
print("NCCL_WORLD1_OK", l0)
'''


def test_bucketed_nccl_world1_step_is_bit_identical_to_the_plain_step(dev):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _NCCL_CHILD % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


_GLOO2_CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from unflow_amd.core.engine import FlowNetEngine, DEFAULT_PARAMS
from unflow_amd.core.train import StepRunner
dev = torch.device("cuda:0")            # both ranks share the one GPU of the box: RCCL refuses that, gloo moves CUDA tensors
torch.cuda.set_device(dev)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
assert world == 2
H, W = 128, 192
g = torch.Generator().manual_seed(11)
# global minibatch of 2 pairs per step; rank r trains on pair r
steps = [((torch.rand(2, H, W, 3, generator=g) * 255), (torch.rand(2, H, W, 3, generator=g) * 255)) for _ in range(3)]
eng = FlowNetEngine(1, H, W, params=dict(DEFAULT_PARAMS, flownet='C'), device=dev, seed=7)
run = StepRunner(eng, world, use_graph=True)
assert run.nparts == 3 and run.reducer.world == 2
# rank 0 also holds the reference: ONE process on the concatenated minibatch (mean over replicas of the per-replica gradient
# = gradient of the mean loss: average_gradients, train.py:388-422)
if rank == 0:
    one = FlowNetEngine(2, H, W, params=dict(DEFAULT_PARAMS, flownet='C'), device=dev, seed=7)
    start = one.P.clone()
    r1 = StepRunner(one, 1, use_graph=True)
losses, l1, closeness = [], [], []
for a, b in steps:
    losses.append(run.step(a[rank:rank + 1].to(dev), b[rank:rank + 1].to(dev), 1e-4).item())
    torch.cuda.synchronize()
    # (1) every rank holds the same parameters and moments, bit for bit — after every step
    for t in (eng.P, eng.M, eng.V):
        ref = t.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(ref, t), "rank %%d diverged" %% rank
    # (2) the step equals the single-process step FROM THE SAME STATE up to fp32 summation order.  Per step, not per
    #     trajectory: the loss has hard occlusion / border masks (losses.py:19-45), so two runs that differ in the last bit part
    #     ways at the first mask pixel that flips (measured: one conv3 weight moved by 3.6e-6 changes 2 M gradient entries by
    #     > 1e-6 of the largest), and Adam turns every sign change of a near-zero gradient into 2 lr.  After the comparison the
    #     reference takes over the ranks' state, so every step is compared from identical parameters and moments.
    if rank == 0:
        l1.append(r1.step(a.to(dev), b.to(dev), 1e-4).item())
        torch.cuda.synchronize()
        if len(l1) == 1:
            # (2a) the exchanged gradient itself, where no optimizer step sits in between (ADVICE r4): the all-reduced flat
            #      buffer of the ranks x 1/world against the one-process gradient of the concatenated minibatch, first step
            #      (identical parameters).  Adam's 1/world is fused into the update, so eng.G holds the SUM over the ranks.
            gsum, gone = eng.G.double() / world, one.G.double()
            gd = (gsum - gone).abs()
            gmax = gone.abs().max().item()
            grad_close = (gd.max().item() / gmax, (gd.mean() / gone.abs().mean()).item())
            assert grad_close[0] <= 1e-4 and grad_close[1] <= 2e-5, grad_close
        d = (one.P - eng.P).abs()
        frac_close = (d <= 2e-6).float().mean().item()
        assert frac_close > 0.999 and d.max().item() <= 3.1e-4, (len(l1), frac_close, d.max().item())
        closeness.append((frac_close, d.max().item()))
        for dst, src in ((one.P, eng.P), (one.M, eng.M), (one.V, eng.V)):
            dst.copy_(src)
if rank == 0:
    moved = (start - eng.P).abs().max().item()
    assert moved > 1e-4                                            # three Adam steps of 1e-4 did move the weights
    print("GLOO2_OK", closeness, losses, l1, "reduced gradient vs one process (max-rel, mean-rel):", grad_close)
dist.barrier(); dist.destroy_process_group()
'''


def test_two_ranks_on_one_gpu_over_gloo_match_each_other_and_the_single_process_step(dev, tmp_path):
    """The N > 1 path of StepRunner (three backward parts, bucketed all-reduce + Adam + weight re-split on the communication
    stream) with TWO real ranks.  The build session has one GPU and RCCL refuses two ranks on one device, so the ranks share
    cuda:0 and exchange over gloo: same host code, same kernels, same stream choreography; only the transport differs."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "child.py"
    script.write_text(_GLOO2_CHILD % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, o[-2000:] + e[-3000:]
    assert "GLOO2_OK" in outs[0][0], outs[0][0][-2000:] + outs[0][1][-2000:]


def test_bench_launches_its_own_ranks(dev):
    """`python bench.py --gpus 2` with NO launcher around it (the command shape the driver uses) starts its two ranks
    itself and rank 0 prints the one JSON line (reference: one invocation builds all towers from the GPU list, run.py:40-49,
    train.py:163-183).  One GPU here, so the ranks share it over the gloo test transport."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["UNFLOW_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-secondary",
                        "--no-cpu-baseline", "--no-alt", "--no-parity", "--no-roofline", "--sustain-seconds", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_world_size"] == 2 and d["config"]["global_batch"] == 8
    assert d["comm"]["params_identical_across_ranks"] is True
    assert d["value"] > 0 and d["steps"] == 3


def test_trainer_forwards_train_all_and_trains_the_first_network(dev):
    """Trainer dropped train_all (stacks silently trained only the last network) and full_res."""
    from unflow_amd.core.train import Trainer
    B, H, W = 1, 128, 128
    params = dict(flownet='CS', train_all=True, learning_rate=1e-4, pyramid_loss=True, border_mask=True,
                  ternary_weight=1.0, smooth_2nd_weight=3.0, finetune=['a', 'b'], manual_decay_iters=[2, 2],
                  manual_decay_lrs=[1e-4, 1e-5])
    tr = Trainer(B, H, W, params, device=dev, seed=3, augment=False)
    assert tr.engine.train_all and tr.engine.spec == 'CS'
    im1, im2 = images(B, H, W, 5)
    p0 = tr.engine.export_tf_params()
    tr.train_step(im1.to(dev), im2.to(dev))
    torch.cuda.synchronize()
    g = tr.engine.export_tf_grads()
    first = [k for k in g if not k.startswith('stack_') and k.endswith('/weights')]
    assert first and all(g[k].abs().max().item() > 0 for k in first)       # the first network receives data gradients
    p1 = tr.engine.export_tf_params()
    assert any(not torch.equal(p0[k], p1[k]) for k in first)
    with pytest.raises(ValueError):
        Trainer(B, H, W, dict(params, flownet='C', full_res=True), device=dev)   # full_res reaches the engine (C: invalid)
    tr2 = Trainer(B, H, W, dict(flownet='S', full_res=True, learning_rate=1e-4, ternary_weight=1.0, smooth_2nd_weight=3.0,
                                pyramid_loss=True, border_mask=True), device=dev, seed=1, augment=False)
    assert tr2.engine.full_res and len(tr2.engine.lv) == 7


def test_trainer_augments_by_default_and_loss_decreases(dev):
    from unflow_amd.core.train import Trainer
    B, H, W = 2, 128, 192
    params = dict(flownet='C', learning_rate=1e-4, pyramid_loss=True, border_mask=True, ternary_weight=1.0,
                  smooth_2nd_weight=3.0)
    im1, im2 = images(B, H, W, 6)
    im1, im2 = im1.to(dev), im2.to(dev)
    tr = Trainer(B, H, W, params, device=dev, seed=2)           # augment=True like unsupervised_loss's default
    l_aug = [tr.train_step(im1, im2).item() for _ in range(3)]
    assert tr.engine._mask_aug                                     # the per-sample warped border masks are in use
    assert len(set(round(x, 3) for x in l_aug)) == 3              # fresh draws every step
    tr0 = Trainer(B, H, W, params, device=dev, seed=2, augment=False)
    ls = [tr0.train_step(im1, im2).item() for _ in range(25)]
    assert ls[-1] < ls[0]
    assert not tr0.engine._mask_aug and tr0.iteration == 25


@pytest.mark.parametrize("spec,full_res", [("c", False), ("s", False), ("cs", False), ("S", True), ("s", True), ("Cs", True)])
def test_small_and_full_res_nets_vs_oracle(spec, full_res, dev):
    """3/8-width networks (flownet.py:22-23) and the full_res decoder (flownet.py:133-153; 7-level loss pyramid,
    unsupervised.py:89-97): loss, final flows and the trained network's gradients vs the fp64 oracle."""
    from unflow_amd.core.engine import FlowNetEngine, flow_error_avg
    from oracle import model_ref as M
    B, H, W = 1, 128, 128
    params = dict(flownet=spec, full_res=full_res, pyramid_loss=True, border_mask=True, ternary_weight=1.0,
                  smooth_2nd_weight=3.0)
    eng = FlowNetEngine(B, H, W, params=params, device=dev, seed=None)
    tf_params = M.init_params_spec(spec, seed=13, full_res=full_res)
    assert [l.name for l in eng.layers] == [k[:-8] for k in tf_params if k.endswith('/weights')]
    if len(spec) > 1:
        for k in tf_params:
            if k.split('/')[-2].startswith('flow') and k.endswith('/weights'):
                tf_params[k] = tf_params[k] * 0.3                  # keep stacked flows in a trained network's regime
    eng.load_tf_params(tf_params)
    exp = eng.export_tf_params()
    assert all(torch.equal(exp[k], tf_params[k]) for k in tf_params)       # load / export round trip through the padded layout
    im1, im2 = images(B, H, W, 14)
    loss = eng.fwd_bwd(im1.to(dev), im2.to(dev)).item()
    P64 = {k: v.clone().double().requires_grad_() for k, v in tf_params.items()}
    loss_ref, ffw, fbw, _ = M.unsupervised_loss(P64, im1.double(), im2.double(), params, return_flow=True)
    loss_ref.backward()
    assert abs(loss - loss_ref.item()) <= 1e-4 * abs(loss_ref.item()), (loss, loss_ref.item())
    fw, bw = eng.final_flows()
    assert flow_error_avg(fw, ffw.float().to(dev)).item() < 1e-3
    assert flow_error_avg(bw, fbw.float().to(dev)).item() < 1e-3
    got = eng.export_tf_grads()
    last = '' if len(spec) == 1 else 'stack_%d_flownet/' % (len(spec) - 1)
    worst = 0.0
    for k, v in P64.items():
        if not k.startswith(last):
            continue
        l2 = 0.0004 * tf_params[k].double() if k.endswith('/weights') else 0.0
        ref = v.grad - l2
        a = got[k].double()
        e = ((a - ref).abs().max() / (ref.abs().max() + 1e-30)).item()
        worst = max(worst, e)
        # plain oracle at 128x128: a leaky-ReLU unit on the other side of the kink moves a deep gradient by ~1e-3
        assert e < (2e-2 if a.numel() <= 64 else 5e-3), (k, e)
        if a.numel() >= 1024:
            assert ((a - ref).abs().mean() / (ref.abs().mean() + 1e-300)).item() < 2e-3, k
    print("%s full_res=%s: loss rel %.1e, worst gradient max-rel %.1e" % (spec, full_res, abs(loss - loss_ref.item()) / abs(loss_ref.item()), worst))


def test_flownet_mirror_and_unsupervised_loss_mirror(dev):
    """core/flownet.py::flownet for a stacked spec and core/unsupervised.py::unsupervised_loss vs the oracle."""
    from unflow_amd.core import flownet as F
    from unflow_amd.core.unsupervised import unsupervised_loss
    from oracle import model_ref as M
    B, H, W = 1, 128, 128
    im1, im2 = images(B, H, W, 21)
    mean = torch.tensor(M.CHANNEL_MEAN) / 255.0
    a, b = (im1 / 255.0 - mean).to(dev), (im2 / 255.0 - mean).to(dev)
    fw, bw = F.flownet(a, b, flownet_spec='Cs', backward_flow=True)
    eng = F.get_engine(B, H, W, params=dict(flownet='Cs', full_res=False, train_all=False), device=dev)
    P = eng.export_tf_params()
    rfw, rbw = M.flownet({k: v.double() for k, v in P.items()}, (im1 / 255.0 - mean).double(), (im2 / 255.0 - mean).double(),
                         'Cs', backward_flow=True)
    assert len(fw) == 2 and len(fw[0]) == 5
    for net in range(2):
        for lvl in range(5):
            assert (fw[net][lvl].cpu().double() - rfw[net][lvl]).abs().max().item() < 1e-4
            assert (bw[net][lvl].cpu().double() - rbw[net][lvl]).abs().max().item() < 1e-4
    # unsupervised_loss mirror: params may carry unhashable reference keys (get_engine hashed them)
    params = dict(flownet='C', pyramid_loss=True, border_mask=True, ternary_weight=1.0, smooth_2nd_weight=3.0,
                  finetune=['x'], manual_decay_iters=[1, 2])
    loss, f1, f2 = unsupervised_loss((im1.to(dev), im2.to(dev)), params, normalization=[M.CHANNEL_MEAN], augment=False,
                                     return_flow=True)
    eng2 = F.get_engine(B, H, W, params=params, device=dev)
    ref, rf1, rf2, _ = M.unsupervised_loss({k: v.double() for k, v in eng2.export_tf_params().items()}, im1.double(),
                                           im2.double(), dict(params), return_flow=True)
    assert abs(loss.item() - ref.item()) <= 1e-4 * abs(ref.item())
    assert (f1.cpu().double() - rf1).abs().max().item() < 1e-3


def test_losses_mirrors_vs_oracle(dev):
    """core/losses.py::ternary_loss / second_order_loss / create_*_mask (value-level API mirrors)."""
    from unflow_amd.core import losses as LS
    from oracle import model_ref as M
    g = torch.Generator().manual_seed(4)
    B, H, W = 2, 24, 40
    im1 = torch.rand(B, H, W, 3, generator=g)
    im2 = torch.rand(B, H, W, 3, generator=g)
    mask = LS.create_border_mask(im1, 0.1)
    assert torch.equal(mask, M.create_border_mask(im1, 0.1))
    for D in (1, 2, 3):
        got = LS.ternary_loss(im1.to(dev), im2.to(dev), mask.to(dev), max_distance=D).item()
        ref = M.ternary_loss(im1.double(), im2.double(), mask.double(), max_distance=D).item()
        assert abs(got - ref) <= 2e-5 * abs(ref), (D, got, ref)
    flow = torch.randn(B, H, W, 2, generator=g) * 2
    got = LS.second_order_loss(flow.to(dev)).item()
    ref = M.second_order_loss(flow.double()).item()
    assert abs(got - ref) <= 2e-5 * abs(ref)
    assert torch.equal(LS.create_outgoing_mask(flow), M.create_outgoing_mask(flow))


@pytest.mark.gpu
def test_trainer_run_saves_checkpoints_and_resumes_identically(tmp_path):
    """Trainer.run (train.py:116-145) end to end: 4 steps in chunks of 2 with a TF checkpoint-V2 bundle after each chunk; a second
    trainer that starts from the first chunk's checkpoint (weights + Adam slots restored, bias correction restarted, input shifted)
    ends with bit-identical parameters."""
    import shutil
    from unflow_amd.core.train import Trainer
    from unflow_amd.core import tf_checkpoint as T
    dev = torch.device("cuda:0")
    params = dict(flownet='S', learning_rate=1e-4, decay_interval=100000, save_interval=2, display_interval=1)
    g = torch.Generator().manual_seed(5)
    frames = [(torch.rand(1, 64, 64, 3, generator=g) * 255, torch.rand(1, 64, 64, 3, generator=g) * 255) for _ in range(4)]

    def batches(iter_offset):
        k = iter_offset
        while True:
            yield frames[k % 4][0].to(dev), frames[k % 4][1].to(dev)
            k += 1

    ck_a, ck_b = str(tmp_path / "a"), str(tmp_path / "b")
    tr = Trainer(1, 64, 64, params, device=dev, seed=3, augment=False, use_graph=False)
    log = tr.run(0, 4, batches, ck_a)
    assert [i for i, _ in log] == [1, 2, 3, 4] and all(np.isfinite(l) for _, l in log)
    assert tr.checkpoint_step(ck_a) == 4
    ent = T.checkpoint_entries(os.path.join(ck_a, 'model.ckpt-4'))[1]
    assert 'flownet_s/conv1/weights' in ent and 'flownet_s/conv1/weights/Adam_1' in ent
    assert 'global_step' not in ent          # the reference's scoped Saver holds network variables and their slots only
    final = tr.engine.export_tf_params()
    # resume from the checkpoint of the first chunk
    os.makedirs(ck_b)
    for f in os.listdir(ck_a):
        if 'model.ckpt-2' in f:
            shutil.copy(os.path.join(ck_a, f), ck_b)
    with open(os.path.join(ck_b, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "model.ckpt-2"\n')
    tr2 = Trainer(1, 64, 64, params, device=dev, seed=99, augment=False, use_graph=False)     # different initialisation
    tr2.run(0, 4, batches, ck_b)
    assert tr2.checkpoint_step(ck_b) == 4
    got = tr2.engine.export_tf_params()
    for k in final:
        assert torch.equal(final[k], got[k]), k


@pytest.mark.gpu
def test_trainer_stacked_restore_follows_restore_networks(tmp_path):
    """restore_networks (train.py:23-65) for a CS run: without a checkpoint the frozen FlowNetC comes from params['finetune'];
    the run's checkpoints hold only the trained network (the Saver's scope) and its Adam slots; a resumed trainer takes the
    frozen network from finetune[:n-1] again and ends bit-identical to the uninterrupted run."""
    import shutil
    from unflow_amd.core.train import Trainer
    from unflow_amd.core import tf_checkpoint as T
    dev = torch.device("cuda:0")
    H = W = 64
    base = dict(learning_rate=1e-4, decay_interval=100000, save_interval=2, display_interval=1)
    c_dir = str(tmp_path / "C")
    trc = Trainer(1, H, W, dict(base, flownet='C'), device=dev, seed=11, augment=False, use_graph=False)
    trc.save(c_dir, 7)
    c_params = trc.engine.export_tf_params()
    del trc
    g = torch.Generator().manual_seed(8)
    frames = [(torch.rand(1, H, W, 3, generator=g) * 255, torch.rand(1, H, W, 3, generator=g) * 255) for _ in range(4)]

    def batches(iter_offset):
        k = iter_offset
        while True:
            yield frames[k % 4]                       # host tensors: set_input brings them to the device
            k += 1

    params = dict(base, flownet='CS', finetune=[T.latest_checkpoint(c_dir)])
    ck_a, ck_b = str(tmp_path / "a"), str(tmp_path / "b")
    os.makedirs(ck_a)
    tr = Trainer(1, H, W, params, device=dev, seed=3, augment=False, use_graph=False)
    assert tr.restore(ck_a) is None                                   # no checkpoint yet: finetune only
    got = tr.engine.export_tf_params()
    for k, v in c_params.items():
        assert torch.equal(got[k], v), k                              # the frozen FlowNetC is the finetune checkpoint's
    tr.run(0, 4, batches, ck_a)
    ent = T.checkpoint_entries(os.path.join(ck_a, 'model.ckpt-4'))[1]
    assert any(k.startswith('stack_1_flownet/') and k.endswith('/Adam_1') for k in ent)
    assert not any(k.startswith('flownet_c') for k in ent)            # non-train_all: the last network only (train.py:35-37)
    final = tr.engine.export_tf_params()
    os.makedirs(ck_b)
    for f in os.listdir(ck_a):
        if 'model.ckpt-2' in f:
            shutil.copy(os.path.join(ck_a, f), ck_b)
    with open(os.path.join(ck_b, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "model.ckpt-2"\n')
    tr2 = Trainer(1, H, W, params, device=dev, seed=99, augment=False, use_graph=False)
    tr2.run(0, 4, batches, ck_b)
    got = tr2.engine.export_tf_params()
    for k in final:
        assert torch.equal(final[k], got[k]), k
    # a checkpoint without the frozen network and no finetune entry for it: the trained network alone is restored, no KeyError
    tr3 = Trainer(1, H, W, dict(base, flownet='CS'), device=dev, seed=5, augment=False, use_graph=False)
    assert tr3.restore(ck_a) is not None
    got3 = tr3.engine.export_tf_params()
    assert all(torch.equal(got3[k], final[k]) for k in final if k.startswith('stack_1_flownet/'))


@pytest.mark.gpu
def test_input_raw_feeds_trainer_run(tmp_path):
    """The input pipeline and the trainer connected end to end (run.py: Input.input_raw -> Trainer.run): PNG frames on disk ->
    pair list -> random-crop numpy batches -> train steps -> checkpoints; resuming with shift = batch_size * iter_offset."""
    from unflow_amd.core.input import Input, encode_png8_rgb
    from unflow_amd.core.train import Trainer
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    d = tmp_path / "frames"
    d.mkdir()
    for i in range(6):
        (d / ("%06d.png" % i)).write_bytes(encode_png8_rgb(rs.randint(0, 256, size=(72, 80, 3)).astype(np.uint8)))

    class Data:
        def get_raw_dirs(self):
            return [str(d)]
    inp = Input(Data(), 2, (64, 64), normalize=False)
    params = dict(flownet='S', learning_rate=1e-4, decay_interval=100000, save_interval=2, display_interval=1)
    tr = Trainer(2, 64, 64, params, device=dev, seed=1, augment=True, use_graph=False)
    ck = str(tmp_path / "ck")
    log = tr.run(0, 4, lambda off: inp.input_raw(shift=2 * off, seed=0), ck)
    assert [i for i, _ in log] == [1, 2, 3, 4] and all(np.isfinite(l) for _, l in log)
    assert tr.checkpoint_step(ck) == 4
