"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, as MI355X_MICROARCH.md prescribes)
of `bench.py --no-graph` into per-step HBM traffic of the conv-kernel family.

    python tools/pmc_traffic.py <FETCH_SIZE pass dir> <WRITE_SIZE pass dir> [n_params] > profiles/rNN_pmc_traffic.json

Units / corrections: rocprofv3 reports both counters in KiB-like units of 1024 B ("KB"); on gfx950 FETCH_SIZE tallies
128-byte requests at 64 B, so wide coalesced reads show exactly half their bytes (guide, HBM section) -> x2.  Both
corrections are CHECKED here against a kernel of known traffic that runs in the same trace: the fused Adam kernel
reads 4 and writes 3 streams of n_params fp32 (adam_calibration in the output; 1.0 = the corrected counter matches).
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def load(d):
    """Counter sums and dispatch counts per kernel over ONE training step: the dispatches between the last two Adam launches."""
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Dispatch_Id']))
    adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
    rows = rows[adam[-2] + 1: adam[-1] + 1]
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in rows:
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0]
        tot[k] += float(r['Counter_Value'])
        cnt[k] += 1
    return tot, cnt


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    nsteps = 1          # load() keeps exactly one step
    n_params = 39175298 if len(sys.argv) < 4 else int(sys.argv[3])
    f, fc = load(fd)
    w, wc = load(wd)
    adam = [k for k in f if 'adam' in k][0]
    rd_adam = f[adam] / fc[adam] * 1024 * 2          # bytes per launch, x2 gfx950 correction
    wr_adam = w[adam] / wc[adam] * 1024
    cal_r = rd_adam / (4.0 * 4 * n_params)
    cal_w = wr_adam / (3.0 * 4 * n_params)
    out = dict(unit="bytes per step", steps_in_trace=nsteps,
               adam_calibration=dict(read=round(cal_r, 3), write=round(cal_w, 3)), kernels={})
    fam_r = fam_w = 0.0
    fam_n = 0
    for k in sorted(f, key=lambda k: -f[k]):
        r = f[k] * 1024 * 2 / nsteps
        ww = w.get(k, 0.0) * 1024 / nsteps
        if r + ww < 1e6:
            continue
        out['kernels'][k] = dict(launches_per_step=round(fc[k] / nsteps, 2), read_MB=round(r / 1e6, 1),
                                 write_MB=round(ww / 1e6, 1))
        if any(f in k for f in ('igemm', 'splitk', 'sum_partials', 'head3_', 'skinny_conv', 'tiny_deconv')):   # bench.py's roofline class
            fam_r += r
            fam_w += ww
            fam_n += fc[k]
    out['conv_family'] = dict(read_MB=round(fam_r / 1e6, 1), write_MB=round(fam_w / 1e6, 1),
                              total_bytes=int(fam_r + fam_w), launches_per_step=round(fam_n / nsteps, 1))
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
