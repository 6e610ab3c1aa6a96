#!/usr/bin/env python3
"""UnFlow training-step benchmark (BASELINE.json metric: image-pairs/s, FlowNetC 384x512).

One "step" = one unsupervised training step of FlowNetC on a synthetic minibatch of 4 image pairs per
GPU (BASELINE.json configs[1]; configs[2] = the same per-GPU work on 8 ranks): bidirectional forward
(both feature towers, both flownet_c passes, 441-channel correlation), census + second-order loss
pyramid, full backward, gradient all-reduce over RCCL when N > 1, fused L2 + Adam update.
Raw input minibatches (4, rotated) are resident in HBM before the timed region; their preparation (/255, mean
subtraction) is part of every timed step.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W          # N > 1 without a launcher: spawns its own N ranks (self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # or under the launcher: ranks read RANK / WORLD_SIZE / MASTER_*
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# CPU-baseline leg: bound the host thread count BEFORE torch / libgomp initialise (a 256-thread oneDNN +
# OpenMP run of this small problem is ~100x slower than 32 threads on the GPU host).
CPU_THREADS = min(os.cpu_count() or 1, 32)
os.environ.setdefault("OMP_NUM_THREADS", str(CPU_THREADS))
os.environ.setdefault("MKL_NUM_THREADS", str(CPU_THREADS))

FWD_BWD_GFLOP_PER_PAIR = 204.9      # algorithmic 2*MAC, FlowNetC 384x512 bidirectional
FP32_MFMA_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
BF16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 (v_mfma_f32_32x32x16_bf16)
MFMA_SUSTAINED_FRACTION = 0.761       # measured: profiles/r03_mfma_power_scaling.txt (all CUs, random operand bits)
BF16X3_TERMS = 6                    # product terms of the fp32-equivalent 3-way bf16 split (csrc/conv_igemm.hip)


CONV_MATH_TEXT = {
    "bf16x3": "conv fwd/dgrad/wgrad: fp32-equivalent 3-way bf16 split (6 product terms, fp32 accumulate) on the bf16 MFMA, "
              "operands pre-split into planes by their producers (csrc/conv_planes.hip) — same error class as the fp32 MFMA",
    "fp32": "fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere",
    "f16": "fp16 activations and weights into v_mfma_f32_32x32x16_f16, fp32 accumulate (NOT fp32-equivalent: BASELINE "
           "configs[4]; tolerance vs the fp32 oracle stated in tests/test_f16_gpu.py)",
}


KITTI_FLOW_HEAD_SCALE = {'flow2': 0.15, 'flow3': 1.0, 'flow4': 3.5, 'flow5': 10.0, 'flow6': 15.0}


def kitti_variant_params():
    """The KITTI training loss of the reference (config_template/config.ini:172-174 over losses.py:43-56): forward-backward
    consistency + occlusion penalty, the occlusion mask thresholded from the flows."""
    return dict(flownet='C', pyramid_loss=True, border_mask=True, ternary_weight=1.0, smooth_2nd_weight=3.0, fb_weight=0.2,
                occ_weight=12.4, mask_occlusion='fb')


def kitti_variant_weights(tf_params):
    """Random-initialised flow heads put out ~2 px at flow2 and ~0.02 px at flow6: the fb mask is then all-occluded on top and
    empty below.  Rescaled per level (calibrated on the oracle) every level sees 0.3-0.7 px and a mixed mask."""
    return {k: (v * KITTI_FLOW_HEAD_SCALE[k.split('/')[-2]] if k.endswith('/weights') and k.split('/')[-2] in KITTI_FLOW_HEAD_SCALE else v)
            for k, v in tf_params.items()}


def conv_family_gflop(eng):
    """Algorithmic GFLOP (2*MAC, true channel counts) of all conv/deconv fwd + dgrad + wgrad launches of one step."""
    N = eng.N
    sizes = {}
    H, W = eng.H, eng.W
    tot = 0.0
    res = {'conv1': 2, 'conv2': 4, 'conv3': 8, 'conv_redir': 8, 'conv3_1': 8, 'conv4': 16, 'conv4_1': 16, 'conv5': 32,
           'conv5_1': 32, 'conv6': 64, 'conv6_1': 64, 'flow6': 64, 'deconv5': 32, 'flow6_up5': 32, 'flow5': 32,
           'deconv4': 16, 'flow5_up4': 16, 'flow4': 16, 'deconv3': 8, 'flow4_up3': 8, 'flow3': 8, 'deconv2': 4,
           'flow3_up2': 4, 'flow2': 4, 'deconv1': 2, 'flow2_up1': 2, 'flow1': 2, 'deconv0': 1, 'flow1_up0': 1, 'flow0': 1}
    for st in eng.stages:
        for l in st.layers:
            nm = l.name.split('/')[-1]
            d = res[nm]
            opix = N * (H // d) * (W // d)
            taps = l.k * l.k if l.kind == 'conv' else 4      # conv_transpose k4 s2: 4 taps reach each output pixel
            f = 2.0 * opix * taps * l.cin * l.cout / 1e9
            sizes[l.name] = f
            if not st.trainable:
                passes = 1                                   # behind stop_gradient: forward only
            elif nm == 'conv1' and not (st.need_in_grad and not st.is_c):
                passes = 2                                   # first layer: fwd + wgrad, its input is data
            else:
                passes = 3
            tot += f * passes
    return tot, sizes


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (one process per GPU, the job
    the reference does in-process from its GPU list, run.py:40-49 -> train.py:163-183) by re-executing this script under
    torch.distributed.run on a free loopback port; the ranks inherit stdout, rank 0 prints the one JSON line.  Returns
    the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["UNFLOW_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class ClockSampler:
    """Polls `rocm-smi --showpower --showclocks --json` from a thread while the sustained pass runs: the shader clock and board power the
    step actually gets (profiles/r06_clock_power_under_load.txt: ~2.1 GHz of the 2.4 GHz the MFMA peak is quoted at).  Context only — `peak`
    and `frac` stay the nominal ones; every failure (no rocm-smi, no permission, no samples) yields None."""

    SMI = "/opt/rocm/bin/rocm-smi"

    def __init__(self, period=0.4):
        import threading
        self.rows, self.period, self._stop = [], period, threading.Event()
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    @staticmethod
    def parse(txt):
        """(board power W, sclk MHz) of the busiest card in rocm-smi's JSON, or None."""
        import re
        best = None
        for card, c in json.loads(txt[txt.index("{"):]).items():
            if not isinstance(c, dict):
                continue

            def num(pat):
                for k, v in c.items():
                    if re.search(pat, k, re.I):
                        m = re.search(r"\d+(\.\d+)?", str(v))
                        if m:
                            return float(m.group(0))
                return None
            row = (num(r"power"), num(r"sclk clock speed"))
            if row[0] is not None and row[1] is not None and (best is None or row[0] > best[0]):
                best = row                              # the busiest card (N > 1: any rank's)
        return best

    def _run(self):
        import subprocess
        while not self._stop.is_set():
            try:
                row = self.parse(subprocess.run([self.SMI, "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout)
                if row is not None:
                    self.rows.append(row)
            except Exception:
                pass
            self._stop.wait(self.period)

    def stop(self):
        self._stop.set()
        self.th.join(timeout=10)
        rows = self.rows[1:] if len(self.rows) > 2 else self.rows        # (the first sample may precede the ramp)
        if not rows:
            return None
        med = lambda v: sorted(v)[len(v) // 2]                           # noqa: E731
        return {"sclk_mhz_median": med([r[1] for r in rows]), "power_w_median": med([r[0] for r in rows]), "samples": len(rows),
                "source": "rocm-smi --showpower --showclocks polled every %.1f s beside the sustained pass" % self.period}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4, help="image pairs per GPU")
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--flownet", default="C", help="network spec: C (the benchmarked config), S, CS, CSS (BASELINE configs[3] "
                    "with --batch 2 --height 768 --width 1024); non-default specs print the line without roofline/cpu_baseline")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the UNFLOW_CONV_MATH=fp32 re-measurement (a sub-process)")
    ap.add_argument("--no-parity", action="store_true", help="skip the step-1 loss/flow comparison with the CPU oracle")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the two secondary configs (BASELINE configs[3] CSS 768x1024 B=2 and configs[4] fp16 B=8: sub-processes)")
    ap.add_argument("--no-comm", action="store_true", help="N > 1: skip the second timing pass without the all-reduce")
    ap.add_argument("--comm-ab", action="store_true",
                    help="N > 1: also time K steps on the OTHER transport (a second RCCL communicator is created for it); always on "
                         "in the forced one-rank record (UNFLOW_FORCE_REDUCER=1)")
    ap.add_argument("--comm", default=None, choices=["torch", "rccl"],
                    help="N > 1: gradient exchange through torch.distributed (backend nccl = RCCL; default) or through the library's "
                         "own C ABI (ncclAllReduce behind unflow_allreduce_sum_f32); the other one is timed as well and reported "
                         "in the comm record")
    ap.add_argument("--overlap-adam", action="store_true",
                    help="one rank: keep the backward cuts and run each part's L2/Adam + weight re-split on a second stream "
                         "under the remaining backward pass (what N > 1 does behind its all-reduce); default off (measured slower)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16"],
                    help="f32 (default): fp32-equivalent arithmetic (UNFLOW_CONV_MATH picks the kernels); f16: fp16 activations "
                         "and weights into the fp16 MFMA with fp32 accumulation (BASELINE configs[4], use --batch 8)")
    ap.add_argument("--loss-variant", default="default", choices=["default", "kitti"],
                    help="kitti: the reference's KITTI training loss (fb_weight 0.2, occ_weight 12.4, mask_occlusion fb) with flow "
                         "heads rescaled so that the occlusion masks are mixed at every level (SURVEY 8(d) config 2)")
    ap.add_argument("--sustain-seconds", type=float, default=5.0,
                    help="after the K timed steps, keep stepping this long and report it as sustained_value (0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    if args.dtype == "f16":
        os.environ["UNFLOW_CONV_MATH"] = "f16"
    import torch
    import torch.distributed as dist
    from unflow_amd.core.engine import FlowNetCEngine
    from unflow_amd.core.train import StepRunner

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("UNFLOW_FORCE_REDUCER") == "1" and "RANK" in os.environ   # test knob
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s); run `python bench.py --gpus %d` (it starts its own "
                 "ranks) or torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus, args.gpus))
    # UNFLOW_DIST_BACKEND=gloo: test knob — several ranks on ONE GPU (RCCL refuses that); same code path, other transport
    backend = os.environ.get("UNFLOW_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and ndev < world:
        sys.exit("bench.py: --gpus %d needs %d visible GPUs, this node shows %d (RCCL wants one GPU per rank)" % (world, world, ndev))
    if backend != "nccl" and ndev:
        local_rank %= ndev                            # the test transport may stack ranks on the GPUs there are
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the comm record quotes RCCL's algorithm / protocol / channel lines: INFO logging (init-time lines only) into a file
        # per process, unless the caller chose otherwise
        if os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION", "WARN"):     # (the image exports NCCL_DEBUG=VERSION)
            os.environ["NCCL_DEBUG"] = "INFO"
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,COLL,GRAPH,TUNING")
        if os.environ.get("NCCL_DEBUG", "").upper() == "INFO":
            os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/unflow_rccl_%h_%p.log")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    B, H, W = args.batch, args.height, args.width
    from unflow_amd.core.engine import DEFAULT_PARAMS
    net_params = dict(DEFAULT_PARAMS, flownet=args.flownet) if args.loss_variant == "default" else dict(kitti_variant_params(), flownet=args.flownet)
    eng = FlowNetCEngine(B, H, W, params=net_params, device=dev, seed=0)   # same weights on every rank
    if args.loss_variant == "kitti":
        eng.load_tf_params(kitti_variant_weights(eng.export_tf_params()))
    head_scale = None
    if len(args.flownet) > 1:
        # A randomly initialised STACK multiplies the flow up to ~700 px by the third network (every stage adds its own random
        # field to the upsampled one): warp sample points then sit within fp32 noise of pixel boundaries and the end-to-end
        # error of ANY fp32 evaluation of the graph leaves the north star's 1e-3 px (round 5's line: 2.9e-3).  A trained stack
        # refines by a few pixels: the flow heads and flow upsamplers are scaled by 0.3 so that the flows stay at tens of
        # pixels — the regime tests/test_parity_fullsize_gpu.py pins against the fp64 oracle at this very batch.  Same kernels,
        # same launch list, same FLOPs; only the values differ.
        head_scale = 0.3
        eng.load_tf_params({k: (v * head_scale if k.split('/')[-2].startswith('flow') and k.endswith('/weights') else v)
                            for k, v in eng.export_tf_params().items()})
    g = torch.Generator().manual_seed(1234 + rank)               # distinct shard per rank
    NBATCH = 4                                                   # raw minibatches resident in HBM, rotated through
    batches = [((torch.rand(B, H, W, 3, generator=g) * 255).to(dev), (torch.rand(B, H, W, 3, generator=g) * 255).to(dev))
               for _ in range(NBATCH)]
    lr = 1e-4
    parity = None
    if rank == 0 and world == 1 and not args.no_parity:
        parity = measure_parity(eng, batches[0])      # step-1 loss and flows vs the CPU oracle, BEFORE any timing
    step_no = [0]
    # forward + loss + backward as hipGraph replays; with more than one rank the backward pass is cut into parts, and each
    # part's gradients are all-reduced and Adam-updated on the communication stream under the rest of the backward pass
    # (unflow_amd/core/train.py).  One rank: one graph, one Adam launch.
    runner = StepRunner(eng, world, use_graph=not args.no_graph, force_reducer=force_dist, local_overlap=args.overlap_adam,
                        transport=args.comm)

    def step():
        # input preparation is part of the step (unsupervised.py:29-31,67-68): next raw minibatch -> /255, mean
        # subtraction (eager launches in front of the graph replay)
        im1, im2 = batches[step_no[0] % NBATCH]
        step_no[0] += 1
        runner.step(im1, im2, lr)

    step()      # captures the graphs (an eager pass first grows the workspaces), then runs the first step
    torch.cuda.synchronize()
    graphs = runner.graphs
    for _ in range(args.warmup):
        step()

    def barrier():
        torch.cuda.synchronize()    # drains the communication stream too: the library's own RCCL communicator is idle before the
        if world > 1:               # process group's collective starts (two communicators on one GPU)
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    loss = eng.loss_acc.item()
    eng.check_device_faults()       # raises if a stream-K fix-up wait timed out during the timed steps (a wrong tile)
    ms = dt / args.steps * 1e3
    pairs_per_s = world * B * args.steps / dt
    sustained = None
    if args.sustain_seconds > 0:
        # the K-step figure above is a near-cold number (0.15 s of work); a training run sits at the clocks the chip
        # sustains: same step, >= sustain_seconds of it (step count from the all-reduced time: identical on every rank)
        n_sus = max(args.steps, int(args.sustain_seconds / (dt / args.steps)) + 1)
        barrier()
        sampler = ClockSampler() if rank == 0 else None      # sclk / board power beside the run (rocm-smi; context for roofline.frac)
        t0 = time.perf_counter()
        for _ in range(n_sus):
            step()
        barrier()
        ds = time.perf_counter() - t0
        clock = sampler.stop() if sampler is not None else None
        if world > 1:
            tmax = torch.tensor([ds], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            ds = tmax.item()
        sustained = {"value": round(world * B * n_sus / ds, 3), "steps": n_sus, "seconds": round(ds, 2), "clock": clock}

    out = {
        "metric": "image-pairs/s (fwd+bwd) FlowNet%s %dx%d" % (args.flownet, H, W), "value": round(pairs_per_s, 3), "unit": "image-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "FlowNet%s unsupervised step: bidirectional fwd + census/2nd-order loss pyramid + bwd + "
                               "L2/Adam%s, %d pairs/GPU, %dx%d, 441-ch correlation (BASELINE configs[%d])"
                               % (args.flownet, " + RCCL grad all-reduce" if world > 1 else "", B, H, W,
                                  3 if args.flownet != "C" else (2 if world > 1 else 1)),
                   "global_batch": world * B, "height": H, "width": W, "parallelism": "dp%d" % world,
                   "hipgraph": graphs is not None, "final_loss": round(loss, 4),
                   "conv_math": CONV_MATH_TEXT[eng.math],
                   **({"flow_head_scale": head_scale} if head_scale else {})},
        "model_tflops_per_gpu": round(FWD_BWD_GFLOP_PER_PAIR * B / ms, 2) if (H, W, args.flownet) == (384, 512, "C") else None,
        "sustained_value": None if sustained is None else sustained["value"],
        "sustained": sustained,
        "input_prep_in_step": "set_input (raw batch %d-way rotation -> /255, mean subtraction) runs inside every timed step" % NBATCH,
    }
    if parity is not None:
        out["parity"] = parity
    if world > 1 or force_dist:
        out["rccl_world_size"] = dist.get_world_size()      # the rank count the RCCL communicator reports
        out["comm"] = measure_comm(runner, eng, step, barrier, args, world, ms, dev, dist)

    if rank == 0 and world == 1 and not args.no_roofline:
        out["roofline"] = measure_roofline(eng, args)
        clk = (sustained or {}).get("clock")
        if clk and out["roofline"].get("frac") and clk["sclk_mhz_median"] > 0:
            # context, never the judged fraction: `peak` is quoted at the 2400 MHz peak engine clock; under this step the part holds less
            out["roofline"]["clock_context"] = {
                "nominal_mhz": 2400, "sampled_sclk_mhz": clk["sclk_mhz_median"], "sampled_power_w": clk["power_w_median"],
                "frac_at_sampled_clock": round(out["roofline"]["frac"] * 2400.0 / clk["sclk_mhz_median"], 4),
                "note": "frac x 2400 / sclk sampled beside the sustained pass (profiles/r06_clock_power_under_load.txt); `peak` and `frac` are the nominal ones"}
    if rank == 0 and world == 1 and not args.no_alt and args.flownet == 'C' and eng.math == "bf16x3":
        out["value_fp32_mfma_only"] = measure_alt_fp32(args)      # same step with every conv kernel on the fp32 MFMA
    if rank == 0 and world == 1 and not args.no_secondary and (args.flownet, args.dtype, H, W) == ('C', 'f32', 384, 512):
        out["secondary"] = measure_secondary(args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.flownet == 'C':
        out["cpu_baseline"] = measure_cpu_baseline(H, W)
        if out["cpu_baseline"]["value"]:
            out["speedup_vs_cpu_baseline"] = round(pairs_per_s / out["cpu_baseline"]["value"], 1)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    # The JSON line is the LAST thing on stdout: RCCL writes its version banner through C stdio, which is block-buffered
    # when stdout is a pipe/file and would otherwise be flushed after Python's line at exit.
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        print(json.dumps(out), flush=True)


def measure_roofline(eng, args):
    """Dominant kernel class = the conv / conv_transpose layers (implicit-GEMM kernels of csrc/conv_planes.hip — gather, halo,
    LDS-DMA filter gradients — or of csrc/conv_igemm.hip with UNFLOW_CONV_MATH=fp32, + their split-K reduces + the Cout=2
    flow-head kernels): algorithmic FLOPs of those layers per step / the GPU time of exactly those launches.  The launches of one step are recorded, captured alone into a
    hipGraph (so the measurement has the same back-to-back dispatch as the benchmarked step, no Python launch gaps)
    and its replay is timed live with HIP events on the replay stream."""
    import torch
    from unflow_amd.core import layers as L
    gflop, _ = conv_family_gflop(eng)
    names = ["conv_fwd", "conv_bwd_data", "conv_bwd_filter", "deconv_fwd", "deconv_bwd_data", "deconv_bwd_filter",
             "flow_wgrad_batched"]
    orig = {n: getattr(L, n) for n in names}
    calls = []

    def wrap(fn):
        def inner(*a, **k):
            calls.append((fn, a, k))
            return fn(*a, **k)
        return inner

    try:
        for n in names:
            setattr(L, n, wrap(orig[n]))
        eng._bias_grads = wrap(type(eng)._bias_grads.__get__(eng))     # the bias-gradient pass (column sums of every dz) belongs
        eng.fwd_bwd()              # records the conv-family calls of one step                    # to conv backward: timed too
        torch.cuda.synchronize()
    finally:
        for n in names:
            setattr(L, n, orig[n])
        eng.__dict__.pop("_bias_grads", None)
    launches = len(calls)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for f, a, k in calls:      # warm on the capture stream
            f(*a, **k)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for f, a, k in calls:
                f(*a, **k)
        graph.replay()
        torch.cuda.synchronize()
        reps = 5
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
    torch.cuda.current_stream().wait_stream(side)
    achieved = gflop / ms            # GFLOP / ms == TFLOP/s
    # Peak of the class = its FLOPs / the time its kernels need at their own matrix-core peaks.  Default: gather kernels
    # (conv fwd / dgrad, 2/3 of the FLOPs) and filter-gradient kernels (1/3) both run the fp32-equivalent 3-way bf16 split =
    # bf16 peak / 6 product terms; UNFLOW_WGRAD_MATH=fp32 / UNFLOW_CONV_MATH=fp32 put the latter / both on
    # v_mfma_f32_32x32x2_f32.
    f16 = eng.math == "f16"
    bf16x3 = eng.math == "bf16x3"
    wg_b3 = bf16x3
    g_peak = BF16_MFMA_PEAK_TFLOPS if f16 else BF16_MFMA_PEAK_TFLOPS / BF16X3_TERMS if bf16x3 else FP32_MFMA_PEAK_TFLOPS
    w_peak = BF16_MFMA_PEAK_TFLOPS if f16 else BF16_MFMA_PEAK_TFLOPS / BF16X3_TERMS if wg_b3 else FP32_MFMA_PEAK_TFLOPS
    peak = 1.0 / ((2.0 / 3.0) / g_peak + (1.0 / 3.0) / w_peak)
    b3name, f32name = "fp32-equivalent 3xbf16 split, 6 terms on v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x2_f32"
    if f16:
        b3name = "fp16 operands on v_mfma_f32_32x32x16_f16, fp32 accumulate"
    kname = {"bf16x3": "igemm_pl_gather_kernel / igemm_pl_halo_kernel / igemm_pl_wgrad_dma_kernel (operand planes, csrc/conv_planes.hip)",
             "f16": "igemm_pl_gather_kernel / igemm_pl_wgrad_kernel (fp16 planes, csrc/conv_planes.hip)"}.get(
                 eng.math, "igemm_gather_kernel / igemm_wgrad kernel (csrc/conv_igemm.hip)")
    return {"bound": "mfma", "kernel": "%s: gather (%s) + filter gradients (%s) incl. their split-K reduces and "
                                       "the Cout=2 flow-head kernels and the batched bias-gradient column sums: %d layer launches/step"
                                       % (kname, b3name if (bf16x3 or f16) else f32name, b3name if (wg_b3 or f16) else f32name, launches),
            "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "frac_of_fp32_mfma_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
            "peak_note": "FLOP-weighted: 2/3 of the class at %.1f (gather kernels), 1/3 at %.1f TFLOP/s (filter gradients); "
                         "%s" % (g_peak, w_peak, "fp16 operands: one product term, the dense 16-bit MFMA peak" if f16 else
                                 "bf16x3 peak = 2500 / 6 product terms"),
            # context, not the judged fraction: what the board sustains on the same MFMA instruction with NO operand traffic at
            # all when every CU runs it on random operand bits (tools/microbench/mfma_scaling.hip: 76 % of nominal — the power limit)
            **({"sustained_mfma_fraction_of_nominal": MFMA_SUSTAINED_FRACTION,
                "frac_of_sustained_mfma_rate": round(achieved / (peak * MFMA_SUSTAINED_FRACTION), 4),
                "sustained_note": "profiles/r03_mfma_power_scaling.txt: pure v_mfma_f32_32x32x16_bf16 loops on all 256 CUs hold 98 % "
                                  "of nominal with constant operands and 76 % with random operand bits"}
               if (bf16x3 and wg_b3) or f16 else {}),
            **(_pmc_traffic() if (eng.B, eng.H, eng.W, eng.spec) == (4, 384, 512, 'C') else {"traffic": None}),
            "algorithmic_gflop_per_step": round(gflop, 1), "ms_per_step_in_kernel_class": round(ms, 3)}


def measure_comm(runner, eng, step, barrier, args, world, ms_with, dev, dist):
    """What the gradient exchange costs, measured inside this run: the same K steps once more with the collectives skipped
    (GradAllReducer.dry: identical stream choreography, bucketed Adam + weight re-split still on the communication stream), so
    allreduce_ms_exposed = ms/step with the exchange - ms/step without = the part of the all-reduce the backward pass and the
    optimizer did not hide.  bytes_per_step = the fp32 gradient bytes every rank contributes (frozen stages excluded:
    train.py:388-422 skips None gradients).  With NCCL_DEBUG=INFO the RCCL log lines naming algorithm / protocol are attached
    (a ring over xGMI is per-link bound, ~1.8 ms for 157 MB at 8 ranks; direct reduce-scatter + all-gather ~0.26)."""
    import torch
    red = runner.reducer
    ranges = [r for part in runner.buckets for r in part]
    nbytes = 4 * sum(hi - lo for lo, hi in ranges)
    sub = sum((hi - lo + red.per - 1) // red.per for lo, hi in ranges)
    out = {"transport": "%s (%s)" % (red.transport, "ncclAllReduce behind unflow_allreduce_sum_f32, csrc/comm_rccl.hip" if red.rccl is not None
                                     else "torch.distributed all_reduce, backend %s" % dist.get_backend()),
           "bytes_per_step": nbytes, "buckets": len(runner.buckets), "collectives_per_step": sub,
           "bucket_bytes": [4 * sum(hi - lo for lo, hi in part) for part in runner.buckets],
           "sub_bucket_bytes_max": red.per * 4, "backward_parts": runner.nparts,
           "overlap": "all-reduce of a part's gradients + their fused L2/Adam + weight re-split run on a communication stream "
                      "under the remaining backward parts (unflow_amd/core/train.py)",
           "env": {k: os.environ[k] for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_DEBUG", "RCCL_MSCCL_ENABLE") if k in os.environ}}
    # The replicas must still be bit-identical after the timed steps (same initialisation, same summed gradients, same
    # update): a bit-level checksum of the flat parameter buffer, min and max over the ranks.  Checked BEFORE the dry pass
    # below lets them diverge.
    def replicas_identical():
        chk = eng.P.view(torch.int32).sum(dtype=torch.int64).reshape(1)
        lo, hi = chk.clone(), chk.clone()
        if world > 1:
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return bool((lo == hi).item())
    out["params_identical_across_ranks"] = replicas_identical()
    # the other transport on the same runner (same streams, buckets, graphs): K more steps
    try:
        if world > 1 and not args.comm_ab:
            raise RuntimeError("not requested (--comm-ab)")
        if world > 1 and dist.get_backend() != "nccl":
            raise RuntimeError("ranks share a GPU under the %s test backend: RCCL needs one GPU per rank" % dist.get_backend())
        from unflow_amd.core.data_parallel import RcclComm
        prev = red.rccl
        other = None if prev is not None else RcclComm(world, dist.get_rank() if world > 1 else 0)
        red.rccl = other
        try:
            for _ in range(2):
                step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            dd = time.perf_counter() - t0
        finally:
            red.rccl = prev
        if world > 1:
            tmax = torch.tensor([dd], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dd = tmax.item()
        out["other_transport"] = {"transport": "torch.distributed" if other is None else "rccl through the C ABI (RCCL %d)" % other.version,
                                  "ms_per_step": round(dd / args.steps * 1e3, 4), "params_identical_across_ranks": replicas_identical()}
        if other is not None:
            other.close()
    except Exception as e:
        out["other_transport"] = "not measured: %s" % (e,)
    if not args.no_comm:
        red.dry = True
        try:
            for _ in range(2):
                step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            dd = time.perf_counter() - t0
            if world > 1:
                tmax = torch.tensor([dd], dtype=torch.float64, device=dev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dd = tmax.item()
        finally:
            red.dry = False
        ms_dry = dd / args.steps * 1e3
        out.update(ms_per_step_without_allreduce=round(ms_dry, 4), allreduce_ms_exposed=round(ms_with - ms_dry, 4),
                   note="the parameters of the ranks diverge during the dry pass (no exchange); it runs after every timed figure")
    out["predicted"] = predict_exposed(runner, eng, barrier)
    log = os.environ.get("NCCL_DEBUG_FILE")
    if os.environ.get("NCCL_DEBUG", "").upper() == "INFO" and log:
        try:
            import glob
            import re
            lines = []
            for f in glob.glob(re.sub(r"%[hp]", "*", log)):
                for ln in open(f, errors="ignore"):
                    if re.search(r"[Aa]lgo|[Pp]roto|Ring|Tree|Direct|channels", ln) and len(lines) < 12:
                        lines.append(ln.strip()[:240])
            out["rccl_info"] = lines
        except Exception as e:
            out["rccl_info"] = "failed: %r" % (e,)
    return out


def predict_exposed(runner, eng, barrier):
    """A PREDICTION, not a measurement (no multi-GPU node has been available in six rounds): what the first real SCALE run can be
    read against.  Measured here: the GPU time of every captured backward part (HIP events around its replay, this rank, this
    box).  Model: bucket k is released when part k ends; the communication stream then runs its all-reduce, 2 (W - 1) / W x bytes /
    bus bandwidth, and the fused L2 + Adam + weight re-split of its range (11 B moved per parameter byte... 7 + 3 x 1.5 streams at 6
    TB/s) in order; what is still running when the last part ends is exposed.  Two bus-bandwidth assumptions: 300 GB/s (what RCCL's
    multi-ring all-reduce reaches on an 8-GPU xGMI node for messages of tens of MB) and 153 GB/s (one xGMI link, a single ring:
    SURVEY 8e's per-link bound)."""
    import torch
    if runner.graphs is None:
        return None
    parts = []
    for g in runner.graphs:
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        parts.append(e0.elapsed_time(e1) / 3)
    nbytes = [4 * sum(hi - lo for lo, hi in part) for part in runner.buckets]
    ready = [sum(parts[:k + 1]) for k in range(len(parts))]
    rows = {}
    for W in (2, 4, 8):
        for name, bw in (("busbw_300GBs", 300e9), ("one_link_ring_153GBs", 153e9)):
            t, per = 0.0, []
            for k, b in enumerate(nbytes):
                ar = 2.0 * (W - 1) / W * b / bw * 1e3
                upd = b * (7 + 4.5) / 4 / 6.0e12 * 1e3 * 4      # Adam: 7 streams of 4 B; re-split: 1 read + 3 planes x 2 copies x 2 B per 4 B
                start = max(ready[k], t)
                t = start + ar + upd
                per.append({"bucket": k, "bytes": b, "released_at_ms": round(ready[k], 3), "allreduce_ms": round(ar, 3),
                            "update_ms": round(upd, 3), "done_at_ms": round(t, 3)})
            rows["world_%d/%s" % (W, name)] = {"buckets": per, "exposed_ms": round(max(0.0, t - ready[-1]), 3),
                                                "efficiency_vs_this_step": round(ready[-1] / max(t, ready[-1]), 3)}
    return {"what": "PREDICTION from this rank's measured backward parts and two bandwidth assumptions; NOT a measurement",
            "backward_part_ms": [round(x, 3) for x in parts], "scenarios": rows}


def measure_secondary(args):
    """BASELINE configs[3] and configs[4] in the driver's own line: each the same script in a sub-process (own roofline leg,
    no CPU baseline / fp32 re-measurement / sustained pass), reduced to the fields a reader needs."""
    import subprocess
    base = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup),
            "--no-cpu-baseline", "--no-alt", "--no-secondary", "--sustain-seconds", "0"]
    cfgs = [("FlowNetCSS 768x1024 B=2 (BASELINE configs[3]; last network trained, the two in front frozen; flow heads x 0.3: flows of tens of pixels, "
             "the regime pinned against the fp64 oracle at this batch)",
             ["--flownet", "CSS", "--batch", "2", "--height", "768", "--width", "1024"], None),
            ("FlowNetC f16 B=8 384x512 (BASELINE configs[4])", ["--dtype", "f16", "--batch", "8"],
             "vs the fp32 oracle: loss rel <= 1e-2, final-flow EPE <= 5e-2 px, per-tensor gradient max-rel <= 3e-2 "
             "(tests/test_f16_gpu.py); the reference has no fp16 path (ops are float-only, correlation_op.cc:134-135)"),
            ("FlowNetC B=4 384x512, KITTI training loss: fb_weight 0.2, occ_weight 12.4, mask_occlusion 'fb' (SURVEY 8(d) config 2; "
             "flow heads rescaled for mixed occlusion masks)", ["--loss-variant", "kitti", "--no-roofline"],
             "loss rel <= 2e-4, final-flow EPE <= 1e-3 px vs the fp64 oracle (tests/test_parity_fullsize_gpu.py::"
             "test_flownetc_b4_384x512_kitti_loss_variant_vs_fp64_oracle)")]
    out = []
    out.append(measure_ops())
    for name, extra, tol in cfgs:
        try:
            r = subprocess.run(base + extra, capture_output=True, text=True, timeout=900)
            d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
            rf = d.get("roofline") or {}
            e = {"config": name, "dtype": d["dtype"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                 "roofline": {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "algorithmic_gflop_per_step",
                                                     "ms_per_step_in_kernel_class")}}
            if tol:
                e["tolerance"] = tol
            if "parity" in d:
                e["parity"] = d["parity"]
            out.append(e)
        except Exception as ex:
            out.append({"config": name, "value": None, "error": repr(ex)[:300]})
    return out


def measure_ops():
    """The HBM-family roofline in the driver's own line: `python bench_ops.py --driver` (a sub-process; <= 3 s of GPU time) — the
    warps, downsample and Adam at 16 x 768 x 1024 / 39.2 M parameters, both correlation points (FlowNetC's 441 channels at the
    step's shape; the north star's +-4 / 81 channels at 16 x 96 x 128 x 256), each with SURVEY 8(d)'s fp32 algorithmic bytes, the
    median of 20 HIP-event timings and the fraction of 8 TB/s (correlation rows also TFLOP/s; their plane-format bytes separately)."""
    import subprocess
    name = "op-level rooflines (bench_ops.py --driver): algorithmic bytes per SURVEY 8(d) / median time, fraction of the 8 TB/s HBM peak"
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_ops.py"), "--driver"], capture_output=True, text=True, timeout=600)
        rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"op"')]
        if not rows:
            raise RuntimeError("no rows: " + r.stderr[-300:])
        return {"config": name, "ops": rows}
    except Exception as ex:
        return {"config": name, "ops": None, "error": repr(ex)[:300]}


def measure_alt_fp32(args):
    """image-pairs/s of the same step with UNFLOW_CONV_MATH=fp32 (the library reads the knob once per process, so this
    is a sub-process of this script)."""
    import subprocess
    env = dict(os.environ, UNFLOW_CONV_MATH="fp32")
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--batch",
           str(args.batch), "--height", str(args.height), "--width", str(args.width), "--no-cpu-baseline", "--no-roofline",
           "--no-alt", "--no-parity", "--no-secondary", "--sustain-seconds", "0"] + (["--no-graph"] if args.no_graph else [])
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1]
        return json.loads(line)["value"]
    except Exception as e:
        return "failed: %r" % (e,)


def _pmc_traffic():
    """HBM-side bytes of the conv family per step from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
    correction, WRITE_SIZE; both calibrated to 1.00 on the Adam kernel in the same trace, tools/pmc_traffic.py).
    PMC needs its own rocprofv3 runs, so this is read from profiles/, not collected inside bench.py."""
    import glob
    # the latest round's file; within a round the end-of-round record (rNN_end_*) over the mid-round one
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r*_pmc_traffic.json')),
                   key=lambda f: (os.path.basename(f)[:3], '_end_' in os.path.basename(f)))
    if not files:
        return {"traffic": None}
    d = json.load(open(files[-1]))
    return {"traffic": d['conv_family']['total_bytes'], "traffic_unit": "bytes/step over the kernel class "
            "(B=4 384x512 only)", "traffic_source": 'profiles/' + os.path.basename(files[-1])}


def measure_parity(eng, batch):
    """Loss and final flows of the FIRST step (initial weights, first minibatch) from the HIP path vs the CPU oracle
    (oracle/model_ref.py, fp32 forward), computed before any timing.  The checker never enters the timed region."""
    import torch
    try:
        from oracle import model_ref as M
        from unflow_amd.core.engine import flow_error_avg
        torch.set_num_threads(CPU_THREADS)
        prev, eng.defer_l2 = eng.defer_l2, False
        eng.set_input(*batch)
        eng.forward_net()
        loss = eng.forward_loss(with_grad=False).item()
        fw, bw = eng.final_flows()
        eng.defer_l2 = prev
        tfp = eng.export_tf_params()
        with torch.no_grad():
            ref, ffw, fbw, _ = M.unsupervised_loss(tfp, batch[0].cpu(), batch[1].cpu(), dict(eng.params), return_flow=True)
        ref = ref.item()
        return {"loss_step1": round(loss, 5), "oracle_loss_step1": round(ref, 5), "rel": float("%.3e" % (abs(loss - ref) / abs(ref))),
                "final_flow_epe_fw_px": float("%.3e" % flow_error_avg(fw, ffw.to(fw.device)).item()),
                "final_flow_epe_bw_px": float("%.3e" % flow_error_avg(bw, fbw.to(bw.device)).item()),
                "oracle": "oracle/model_ref.py fp32 forward on the host, same weights and first minibatch (B=%d)" % eng.B}
    except Exception as e:
        return {"loss_step1": None, "error": repr(e)}


def measure_cpu_baseline(H, W):
    """The reference's step on the host cores: the literal TF graph cannot run (no TensorFlow, GPU-only ops),
    so this times the CPU oracle restatement (oracle/model_ref.py: torch-CPU convs, C ops) — kind "port".
    Bounded sample: 1 image pair per step, 2 warm-up steps + the MEDIAN of 5 timed fwd+bwd steps, hard stop
    at 60 s of CPU work."""
    import torch
    try:
        from oracle import model_ref as M
        cores = CPU_THREADS
        torch.set_num_threads(cores)
        P = M.init_params('C', 0)
        for v in P.values():
            v.requires_grad_()
        g = torch.Generator().manual_seed(1234)
        im1 = torch.rand(1, H, W, 3, generator=g) * 255
        im2 = torch.rand(1, H, W, 3, generator=g) * 255
        times = []
        WARM, TIMED = 2, 5
        for it in range(WARM + TIMED):
            for v in P.values():
                v.grad = None
            t0 = time.perf_counter()
            loss = M.unsupervised_loss(P, im1, im2)
            loss.backward()
            times.append(time.perf_counter() - t0)
            if it >= WARM and sum(times) + times[-1] > 60.0:   # bounded CPU leg
                break
        timed = sorted(times[WARM:] or times[-1:])
        t = timed[len(timed) // 2]
        return {"value": round(1.0 / t, 4), "unit": "image-pairs/s", "cores": cores, "host_cores": os.cpu_count(),
                "kind": "port",
                "sample": "1 image pair %dx%d per step, %d warm-up + median of %d timed fwd+bwd steps of the torch-CPU/C "
                          "oracle on %d threads of a %d-core host (%.1f s of CPU work)"
                          % (H, W, WARM, len(timed), cores, os.cpu_count() or 0, sum(times))}
    except Exception as e:  # the oracle is a checker, never a dependency of the measured path
        return {"value": None, "unit": "image-pairs/s", "cores": CPU_THREADS, "host_cores": os.cpu_count(), "kind": "port",
                "sample": "failed: %r" % (e,)}


if __name__ == "__main__":
    main()
