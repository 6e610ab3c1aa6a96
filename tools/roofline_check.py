"""Cross-check of bench.py's `roofline` object against the committed rocprofv3 summary.

    python tools/roofline_check.py profiles/rNN_bench_kernel_stats.csv <one-step kernel_trace.csv> [gflop_per_step]

`kernel_stats.csv` (rocprofv3 --kernel-trace --stats of the bench command) gives every kernel's AVERAGE duration; the
one-step trace (bench.py --no-graph) gives how many times each kernel runs per step.  The conv-family time per step is
sum(avg * launches_per_step) over the kernels bench.py's roofline class contains (implicit-GEMM, split-K reduces,
flow-head and 2->2 deconv kernels); algorithmic GFLOP / that time must agree with roofline.achieved."""
import csv
import sys
from collections import defaultdict

FAMILY = ('igemm_', 'conv_first7', 'splitk_reduce', 'sum_partials', 'head3_', 'skinny', 'tiny_deconv', 'pointwise32',
          'flow_wgrad', 'colsum', 'deconv_sib', 'fillBuffer')


def short(name):
    return name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]


def main():
    stats, trace = sys.argv[1], sys.argv[2]
    gflop = float(sys.argv[3]) if len(sys.argv) > 3 else 803.1
    avg = {}
    for r in csv.DictReader(open(stats)):
        avg[short(r['Name'])] = float(r['AverageNs'])
    rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
    per_step = defaultdict(int)
    for r in rows[idx[-2] + 1: idx[-1] + 1]:
        per_step[short(r['Kernel_Name'])] += 1
    total = 0.0
    mfma = 0.0
    print("%-62s %5s %10s %10s" % ("kernel", "n/step", "avg us", "us/step"))
    for k in sorted(per_step, key=lambda k: -per_step[k] * avg.get(k, 0)):
        if not any(f in k for f in FAMILY):
            continue
        t = per_step[k] * avg.get(k, 0.0) / 1e3
        total += t
        if 'igemm' in k or 'conv_first7' in k:
            mfma += t
        print("%-62s %5d %10.1f %10.1f" % (k[:62], per_step[k], avg.get(k, 0.0) / 1e3, t))
    print("conv family: %.3f ms/step -> %.1f TFLOP/s for %.1f algorithmic GFLOP (compare with roofline.achieved of the "
          "bench line; %.1f %% of the 157.3 TFLOP/s fp32-MFMA peak)"
          % (total / 1e3, gflop / total * 1e3, gflop, gflop / total * 1e3 / 157.3 * 100))
    print("implicit-GEMM kernels alone: %.3f ms/step -> %.1f TFLOP/s" % (mfma / 1e3, gflop / mfma * 1e3))


if __name__ == '__main__':
    main()
