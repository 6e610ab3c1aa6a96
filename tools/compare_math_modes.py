"""Train the same FlowNetC (same init, same synthetic frame pairs, Adam lr 1e-4) for N steps with the implicit-GEMM kernels
on (a) the fp32-equivalent 3-way bf16 split (default) and (b) the fp32 MFMA, each in its own process (the library reads
UNFLOW_CONV_MATH once), and print both loss curves and their relative difference.

    python tools/compare_math_modes.py [steps]  >  profiles/rNN_math_mode_training_compare.txt
"""
import json
import os
import subprocess
import sys


def worker(steps):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from unflow_amd.core.engine import FlowNetCEngine
    dev = torch.device('cuda:0')
    B, H, W = 4, 384, 512
    eng = FlowNetCEngine(B, H, W, device=dev, seed=0)
    g = torch.Generator().manual_seed(77)
    im1 = torch.rand(B, H, W, 3, generator=g) * 255
    im2 = torch.roll(im1, shifts=(3, -4), dims=(1, 2)) * 0.92 + torch.rand(B, H, W, 3, generator=g) * 20   # a real motion signal
    im1, im2 = im1.to(dev), im2.to(dev)
    losses = []
    for _ in range(steps):
        loss = eng.train_step(im1, im2, 1e-4)
        torch.cuda.synchronize()
        losses.append(loss.item())
    print("LOSSES " + json.dumps(losses))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--worker':
        worker(int(sys.argv[2]))
        return
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    curves = {}
    for mode in ('bf16x3', 'fp32'):
        env = dict(os.environ, UNFLOW_CONV_MATH=mode)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', str(steps)], env=env, capture_output=True,
                           text=True, timeout=1200)
        line = [l for l in r.stdout.splitlines() if l.startswith('LOSSES ')]
        if not line:
            print(r.stdout[-2000:], r.stderr[-2000:])
            sys.exit(1)
        curves[mode] = json.loads(line[-1][7:])
    a, b = curves['bf16x3'], curves['fp32']
    print("FlowNetC B=4 384x512, %d Adam steps (lr 1e-4) on one fixed batch; loss per step" % steps)
    print("%5s %14s %14s %10s" % ("step", "bf16x3 split", "fp32 MFMA", "rel diff"))
    for i in list(range(0, min(10, steps))) + list(range(10, steps, max(1, steps // 20))) + [steps - 1]:
        print("%5d %14.5f %14.5f %10.2e" % (i + 1, a[i], b[i], abs(a[i] - b[i]) / abs(b[i])))
    worst = max(abs(x - y) / abs(y) for x, y in zip(a, b))
    print("max relative difference over all steps: %.2e; loss %0.3f -> %0.3f (bf16x3), %0.3f -> %0.3f (fp32)"
          % (worst, a[0], a[-1], b[0], b[-1]))


if __name__ == '__main__':
    main()
