"""CPU (-m "not gpu"): the data-parallel exchange with gloo, world_size 2.

(1) GradAllReducer: bucketed in-place SUM == sum of the ranks' buffers; mean_() == the reference's
    average_gradients (src/e2eflow/core/train.py:388-422: concat + reduce_mean per variable).
(2) N-rank sharded step == 1-rank step on the concatenated batch: each rank computes the gradient of
    its shard's mean loss (here with the CPU oracle as the compute, the GPU engine being unavailable), the reducer
    averages, and the result equals the single-process gradient of the full batch's loss.
"""
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker_reducer(rank, world, port, out_dir):
    from unflow_amd.core.data_parallel import GradAllReducer
    _init(rank, world, port)
    g = torch.Generator().manual_seed(100 + rank)
    n = 100003                                    # not a multiple of the bucket size
    buf = torch.randn(n, generator=g)
    mine = buf.clone()
    red = GradAllReducer(buf, world, bucket_bytes=64 * 1024)
    assert len(red.bounds) > 1
    red.all_reduce()
    torch.save((mine, buf.clone()), os.path.join(out_dir, "sum%d.pt" % rank))
    buf3 = mine.clone()                            # two-phase (overlap) API: early range, then the rest
    r3 = GradAllReducer(buf3, world, bucket_bytes=32 * 1024)
    r3.start_ranges([(40000, n)])
    r3.start_ranges([(0, 100), (100, 40000)])
    r3.finish()
    assert torch.equal(buf3, buf)
    buf2 = mine.clone()
    GradAllReducer(buf2, world, bucket_bytes=1 << 20).mean_()
    torch.save(buf2, os.path.join(out_dir, "mean%d.pt" % rank))
    dist.destroy_process_group()


def test_grad_all_reducer_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker_reducer, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    mine0, sum0 = torch.load(tmp_path / "sum0.pt")
    mine1, sum1 = torch.load(tmp_path / "sum1.pt")
    assert torch.equal(sum0, sum1)
    assert torch.allclose(sum0, mine0 + mine1, atol=1e-6)
    m0, m1 = torch.load(tmp_path / "mean0.pt"), torch.load(tmp_path / "mean1.pt")
    assert torch.equal(m0, m1)
    # average_gradients: expand_dims + concat + reduce_mean over towers
    assert torch.allclose(m0, torch.stack([mine0, mine1], 0).mean(0), atol=1e-6)


H, W = 128, 128


def _data(i):
    g = torch.Generator().manual_seed(1234 + i)
    return torch.rand(1, H, W, 3, generator=g) * 255, torch.rand(1, H, W, 3, generator=g) * 255


def _flat_grad(P, im1, im2):
    """Flat gradient of the oracle's step in fp64: the statement under test is an identity of the MATHS (mean of the shard
    gradients = gradient of the full-batch loss), so it is checked where rounding cannot blur it."""
    from oracle import model_ref as M
    Pg = {k: v.clone().double().requires_grad_() for k, v in P.items()}
    M.unsupervised_loss(Pg, im1.double(), im2.double()).backward()
    return torch.cat([Pg[k].grad.reshape(-1) for k in Pg])


def _worker_step(rank, world, port, out_dir):
    from oracle import model_ref as M
    from unflow_amd.core.data_parallel import GradAllReducer
    torch.set_num_threads(2)
    _init(rank, world, port)
    P = M.init_params('C', 0)                     # same weights on every rank
    im1, im2 = _data(rank)                        # distinct shard per rank
    flat = _flat_grad(P, im1, im2)
    GradAllReducer(flat, world).mean_()
    if rank == 0:
        torch.save(flat, os.path.join(out_dir, "dp.pt"))
    dist.destroy_process_group()


def test_two_rank_sharded_step_equals_one_rank_full_batch(tmp_path):
    from oracle import model_ref as M
    world, port = 2, _free_port()
    mp.spawn(_worker_step, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    dp = torch.load(tmp_path / "dp.pt")
    P = M.init_params('C', 0)
    a, b = _data(0), _data(1)
    full = _flat_grad(P, torch.cat([a[0], b[0]]), torch.cat([a[1], b[1]]))
    # every loss term is a mean over the replica batch (charbonnier_loss normaliser, losses.py:311-312), the L2
    # term is data independent: mean of the shard gradients == gradient of the full-batch loss
    assert dp.dtype == torch.float64
    rel = ((dp - full).abs().max() / full.abs().max()).item()
    assert rel < 1e-10, rel    # (in fp32 the same comparison sits at ~1e-3: torch-CPU conv gradients change their summation with the batch size)


def test_engine_layer_table_matches_reference_variables():
    """Host logic of the engine that needs no GPU: for every network kind (full / 3/8 width, first / refinement stage,
    with / without the full_res decoder) the layer list equals the oracle's variable list, every input channel of the
    reference has exactly one physical slot, and the backward plan applies each leaky-ReLU derivative exactly once."""
    from unflow_amd.core import engine
    from oracle import model_ref as M

    class FakeEng:
        pass
    for kind, index, full_res in [('C', 0, False), ('c', 0, False), ('S', 0, False), ('S', 1, False), ('s', 0, False),
                                  ('S', 0, True), ('S', 1, True), ('s', 1, True)]:
        st = engine._Stage(FakeEng(), kind, index, full_res)
        spec = M.flownet_layer_specs(kind, 14 if index else 6, full_res)
        scope = '' if index == 0 else 'stack_%d_flownet/' % index
        assert [l.name for l in st.layers] == [scope + s[0] for s in spec], (kind, index, full_res)
        for l, sp in zip(st.layers, spec):
            assert (l.kind, l.k, l.cin, l.cout, l.stride, l.act) == sp[1:], (l.name, sp)
            assert l.cin_p >= l.cin and (l.cin_p % 4 == 0 or l.cin == 2)
            slots = sorted((plo, plo + n) for plo, _, n in l.in_map)
            assert sum(n for _, _, n in l.in_map) == l.cin and sorted(t for _, t, _ in l.in_map)[0] == 0
            assert all(a[1] <= b[0] for a, b in zip(slots, slots[1:])) and slots[-1][1] <= l.cin_p
        # every activated layer's output segment gets its derivative from exactly one data-gradient call
        applied = {}
        for op, first, lo, hi in st.bwd:
            if hi > lo:
                key = (op.src[0], op.src[1] + lo, op.src[1] + hi)
                applied[key] = applied.get(key, 0) + 1
        covered = set()
        for (b, lo, hi), cnt in applied.items():
            assert cnt == 1
            covered.update((b, ch) for ch in range(lo, hi))
        for op in st.ops:
            if op.kind == 'layer' and op.l.act:
                b, lo, hi = op.dst
                consumers = [o for o in st.ops if o.src[0] == b]
                if consumers:
                    assert all((b, ch) in covered for ch in range(lo, hi)), (op.l.name, op.dst)
    layers = engine._Stage(FakeEng(), 'C', 0).layers
    n_logical = sum(l.k * l.k * l.cin * l.cout + l.cout for l in layers)
    assert n_logical == 39175298
    assert engine.LAYER_WEIGHTS == [12.7, 4.35, 3.9, 3.4, 1.1] and engine.LAYER_PATCH_DISTANCES == [3, 2, 2, 1, 1]
    assert engine.LAYER_WEIGHTS_FULL_RES == [12.7, 5.5, 5.0, 4.35, 3.9, 3.4, 1.1]


def _worker_buckets(rank, world, port, out_dir):
    """The bucketed exchange over the ENGINE's flat gradient layout: every part's ranges are all-reduced, then the part's
    optimizer callback runs (StepRunner's order), on gloo."""
    _init(rank, world, port)
    from unflow_amd.core.data_parallel import GradAllReducer
    from unflow_amd.core.engine import FlowNetEngine
    from unflow_amd.core.train import DEFAULT_BUCKET_CUTS
    eng = FlowNetEngine(1, 64, 64, params=dict(flownet='CS'), device='cpu', layout_only=True)
    nparts = eng.set_backward_parts(DEFAULT_BUCKET_CUTS)
    buckets, frozen = eng.part_buckets(), eng.frozen_ranges()
    g = torch.Generator().manual_seed(100 + rank)
    eng.G.copy_(torch.randn(eng.n_params, generator=g))
    for lo, hi in frozen:
        eng.G[lo:hi] = 0                                  # frozen networks: zero data gradient on every rank
    local = eng.G.clone()
    red = GradAllReducer(eng.G, world, bucket_bytes=1 << 20)
    seen = torch.zeros(eng.n_params, dtype=torch.int32)
    order = []
    for k in range(nparts):
        def fn(r=buckets[k], k=k):
            for lo, hi in r:
                seen[lo:hi] += 1
                order.append((k, lo, hi, eng.G[lo:hi].clone()))
        red.reduce_then(buckets[k], fn)
    red.finish()
    for lo, hi in frozen:
        seen[lo:hi] += 1
    torch.save(dict(local=local, reduced=eng.G.clone(), seen=seen, nparts=nparts, buckets=buckets, frozen=frozen,
                    at_callback=[(k, lo, hi, t) for k, lo, hi, t in order]), os.path.join(out_dir, "b%d.pt" % rank))
    dist.destroy_process_group()


def test_engine_buckets_cover_the_gradient_once_and_reduce_to_the_sum(tmp_path):
    port = _free_port()
    mp.spawn(_worker_buckets, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "b%d.pt" % i)) for i in range(2)]
    want = r[0]['local'] + r[1]['local']
    for i in range(2):
        assert r[i]['nparts'] == 3
        assert torch.all(r[i]['seen'] == 1)               # buckets + frozen ranges: a partition of the flat buffer
        assert torch.equal(r[i]['reduced'], want)         # gloo SUM is exact: same order on both ranks
        for k, lo, hi, t in r[i]['at_callback']:          # the optimizer callback of a bucket saw the REDUCED gradient
            assert torch.equal(t, want[lo:hi])
    # the first bucket is the big one (decoder + conv6_1 + conv6): it leaves while two thirds of the backward pass remain
    (lo, hi), = r[0]['buckets'][0]
    assert (hi - lo) * 4 > 100e6


@pytest.mark.parametrize("spec, extra, cuts", [
    ('C', {}, ('conv6', 'conv4')), ('C', {}, ()), ('CS', {}, ('conv6', 'conv4')), ('CSS', {}, ('conv6', 'conv4')),
    ('CS', dict(train_all=True), ()), ('S', dict(full_res=True), ('conv6', 'conv4')), ('cs', {}, ('conv4',)),
])
def test_part_buckets_and_frozen_ranges_partition_the_parameters(spec, extra, cuts):
    """part_buckets() (what is all-reduced and updated after each backward part) + frozen_ranges() (updated with the L2 term
    only, nothing to exchange) cover [0, n_params) exactly once, for every way StepRunner cuts a spec (train.py:163-183:
    every variable gets exactly one averaged gradient and one optimizer update)."""
    from unflow_amd.core.engine import FlowNetEngine
    eng = FlowNetEngine(1, 64, 64, params=dict(flownet=spec, **extra), device='cpu', layout_only=True)
    nparts = eng.set_backward_parts(() if eng.train_all else cuts)
    buckets, frozen = eng.part_buckets(), eng.frozen_ranges()
    assert len(buckets) == nparts == len(cuts if not eng.train_all else ()) + 1
    seen = torch.zeros(eng.n_params, dtype=torch.int32)
    for part in buckets:
        for lo, hi in part:
            assert 0 <= lo < hi <= eng.n_params
            seen[lo:hi] += 1
    for lo, hi in frozen:
        seen[lo:hi] += 1
    assert torch.all(seen == 1)
    assert bool(frozen) == (len(spec) > 1 and not eng.train_all)
    # every weight tensor lies wholly inside ONE range (a bucket never cuts a tensor: its Adam update is one launch range)
    ranges = [r for part in buckets for r in part] + frozen
    for l in eng.layers:
        lo = (l.dw.data_ptr() - eng.G.data_ptr()) // 4
        assert any(a <= lo and lo + l.dw.numel() <= b for a, b in ranges), l.name
    # the gradients of a part are those of the layers its backward slice runs (final when the part returns)
    st = eng.stages[-1]
    if not eng.train_all:
        for k in range(nparts):
            lo, hi = st.part_weight_range(k)
            names = [op.l.name for op, _, _, _ in st.bwd[st.part_bounds[k]:st.part_bounds[k + 1]] if op.kind == 'layer']
            for l in st.layers:
                inside = lo <= (l.dw.data_ptr() - eng.G.data_ptr()) // 4 < hi
                assert inside == (l.name in names), (k, l.name)


def test_bench_self_launch_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus 2` with no launcher around it starts its own ranks (bench.py::self_launch); on a box with fewer
    GPUs than ranks every rank must refuse loudly and the command must fail — not hang, not fall back to fewer ranks."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs here: the refusal path does not apply")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                               "UNFLOW_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs 2 visible GPUs" in (r.stdout + r.stderr)
    assert '{"metric"' not in r.stdout


def test_bench_clock_sampler_parses_rocm_smi_json_and_degrades_to_none():
    """bench.py's ClockSampler (sclk / board power beside the sustained pass; context for roofline.frac): the parser on rocm-smi's JSON as
    the MI355X boxes print it (a warning line in front, values like "(2117Mhz)"), the busiest card of several, and no samples -> None."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    txt = ('WARNING: AMD GPU device(s) is/are in a low-power state.\n'
           '{"card0": {"Temperature (Sensor junction) (C)": "46.0", "fclk clock speed:": "(1250Mhz)", "mclk clock speed:": "(2000Mhz)", '
           '"sclk clock speed:": "(115Mhz)", "sclk clock level:": "S", "Current Socket Graphics Package Power (W)": "243.0"}, '
           '"card1": {"sclk clock speed:": "(2117Mhz)", "sclk clock level:": "S", "Current Socket Graphics Package Power (W)": "1265.0"}, '
           '"system": "x"}')
    assert bench.ClockSampler.parse(txt) == (1265.0, 2117.0)
    assert bench.ClockSampler.parse('{"card0": {"mclk clock speed:": "(2000Mhz)"}}') is None
    s = bench.ClockSampler(period=0.05)
    s.SMI = "/nonexistent/rocm-smi"
    time.sleep(0.2)
    assert s.stop() is None
