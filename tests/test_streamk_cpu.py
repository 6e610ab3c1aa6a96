"""Work partition of the persistent stream-K halo kernel (csrc/conv_streamk.hip: sk_unit) through its host-side test hook —
no GPU.  The kernel's correctness argument needs: the G ranges tile the unit list exactly (monotone, first = 0, last = all),
every range is cut at a chunk boundary, ranges carry the same number of K tiles up to one chunk of the heaviest class, and a
workgroup has at most one segment that does not start its item (its first) and at most one that starts but does not end one
(its last)."""
import ctypes

import numpy as np
import pytest


def units(G, ipc, nchunk, ntaps):
    from unflow_amd import _lib
    L = _lib.lib()
    out = (ctypes.c_int * (G + 1))()
    rc = L.unflow_debug_streamk_units(G, ipc, nchunk, len(ntaps), (ctypes.c_int * len(ntaps))(*ntaps), out)
    assert rc == 0
    return np.array(out[:])


# (G, items per class, chunks, taps per class): conv3_1 fwd, conv4 dgrad (4/2/2/1), conv3 dgrad (9/6/6/4), deconv3 fwd,
# a launch with fewer units than workgroups, a single item, primes
CASES = [(256, 192, 15, [9]), (256, 48, 16, [4, 2, 2, 1]), (256, 96, 8, [9, 6, 6, 4]), (256, 24, 25, [4, 4, 4, 4]),
         (256, 4, 8, [9]), (256, 1, 3, [1]), (7, 13, 5, [3, 1, 2]), (256, 1000, 1, [1, 7]), (304, 17, 11, [5, 3])]


@pytest.mark.parametrize("case", CASES)
def test_streamk_ranges_tile_the_work(case):
    G, ipc, nchunk, ntaps = case
    U = units(G, ipc, nchunk, ntaps)
    total = len(ntaps) * ipc * nchunk
    assert U[0] == 0 and U[G] == total and np.all(np.diff(U) >= 0)
    # K tiles per range: equal up to one chunk of the heaviest class on either side
    w_of_unit = np.repeat(np.repeat(np.array(ntaps), ipc), nchunk)
    cum = np.concatenate([[0], np.cumsum(w_of_unit)])
    work = cum[U[1:]] - cum[U[:-1]]
    ideal = cum[-1] / G
    assert work.max() <= ideal + max(ntaps) and work.min() >= ideal - max(ntaps) - 1, (work.min(), work.max(), ideal)
    for w in range(G):
        a, b = U[w], U[w + 1]
        if a == b:
            continue
        segs = []
        u = a
        while u < b:
            item, c0 = divmod(u, nchunk)
            c1 = min(nchunk, c0 + (b - u))
            segs.append((item, c0, c1))
            u += c1 - c0
        not_start = [s for s in segs if s[1] > 0]
        start_not_end = [s for s in segs if s[1] == 0 and s[2] < nchunk]
        assert len(not_start) <= 1 and (not not_start or not_start[0] == segs[0])
        assert len(start_not_end) <= 1 and (not start_not_end or start_not_end[0] == segs[-1])
        # the workgroups an item-starting, unfinished segment waits for all have a HIGHER id and hold that item FIRST
        for item, c0, c1 in start_not_end:
            end = (item + 1) * nchunk
            w2 = w + 1
            covered = c1 + item * nchunk
            while w2 < G and U[w2] < end:
                if U[w2 + 1] > U[w2]:
                    assert U[w2] == covered and U[w2] // nchunk == item and U[w2] % nchunk > 0
                    covered = min(end, U[w2 + 1])
                w2 += 1
            assert covered == end


def test_streamk_hook_rejects_bad_arguments():
    from unflow_amd import _lib
    L = _lib.lib()
    out = (ctypes.c_int * 9)()
    assert L.unflow_debug_streamk_units(8, 4, 4, 5, (ctypes.c_int * 5)(1, 1, 1, 1, 1), out) != 0
    assert L.unflow_debug_streamk_units(8, 4, 4, 1, (ctypes.c_int * 1)(0), out) != 0
    assert L.unflow_debug_streamk_units(256, 1 << 20, 64, 1, (ctypes.c_int * 1)(9), out) != 0      # K tiles x G overflows 31 bits
