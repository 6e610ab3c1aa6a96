#!/usr/bin/env python3
"""Timeline of the benchmarked step from tools/step_trace.sh's csv (name, queue, start_us, end_us of every dispatch of the last two
replays): per step — wall time, union busy time (any kernel running), time with two or more kernels running (the filter-gradient
stream beside the data-gradient chain), idle time, the largest gaps with the kernels on either side, and the kernels' summed
durations per queue.

    python tools/step_timeline.py gpurun_out/<tag>_step_trace.csv"""
import csv
import sys


def main():
    rows = [dict(r, s=float(r['start_us']), e=float(r['end_us'])) for r in csv.DictReader(open(sys.argv[1]))]
    ends = [i for i, r in enumerate(rows) if 'adam_kernel' in r['name']]
    steps, lo = [], 0
    for i in ends:
        steps.append(rows[lo:i + 1])
        lo = i + 1
    for k, st in enumerate(steps):
        t0, t1 = min(r['s'] for r in st), max(r['e'] for r in st)
        ev = sorted([(r['s'], 1) for r in st] + [(r['e'], -1) for r in st])
        busy = over = 0.0
        depth, prev = 0, t0
        for t, d in ev:
            if depth >= 1:
                busy += t - prev
            if depth >= 2:
                over += t - prev
            depth += d
            prev = t
        print("step %d: %d dispatches, wall %.1f us, busy %.1f us (idle %.1f), >= 2 kernels in flight %.1f us, sum of durations %.1f us"
              % (k, len(st), t1 - t0, busy, (t1 - t0) - busy, over, sum(r['e'] - r['s'] for r in st)))
        qs = {}
        for r in st:
            qs.setdefault(r['queue'], []).append(r)
        for q, rs in qs.items():
            print("   queue %s: %d dispatches, %.1f us of kernels, first start %.1f, last end %.1f"
                  % (q, len(rs), sum(r['e'] - r['s'] for r in rs), min(r['s'] for r in rs) - t0, max(r['e'] for r in rs) - t0))
        # gaps of the union timeline
        gaps = []
        st2 = sorted(st, key=lambda r: r['s'])
        cur_end, last = st2[0]['e'], st2[0]
        for r in st2[1:]:
            if r['s'] > cur_end:
                gaps.append((r['s'] - cur_end, last['name'], r['name']))
            if r['e'] > cur_end:
                cur_end, last = r['e'], r
        gaps.sort(reverse=True)
        print("   %d gaps, total %.1f us; largest:" % (len(gaps), sum(g[0] for g in gaps)))
        for g in gaps[:8]:
            print("      %6.1f us  after %-50s before %s" % (g[0], g[1][:50], g[2][:50]))


if __name__ == '__main__':
    main()
