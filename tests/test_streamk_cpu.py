"""Work partition of the persistent stream-K halo kernel (csrc/conv_streamk.hip: sk_pos / sk_decode) through its host-side
test hook — no GPU.  The kernel's correctness argument needs: decoding the K-tile positions 0 .. all-1 enumerates every
(class, N tile, M pair, K tile) exactly once, items contiguous and their K tiles in order; the G ranges tile the list exactly
and differ by at most one K tile; a workgroup has at most one segment that does not start its item (its first) and at most one
that starts but does not end one (its last); and the workgroups such a segment waits for have HIGHER ids and hold the rest of
that item as their first segment."""
import ctypes

import numpy as np
import pytest


def plan(G, mtp, nt, nchunk, ntaps, ngroups):
    from unflow_amd import _lib
    L = _lib.lib()
    total = mtp * nt * nchunk * sum(ntaps)
    rng = (ctypes.c_int * (G + 1))()
    pos = (ctypes.c_int * total)(*range(total))
    dec = (ctypes.c_int * (4 * total))()
    rc = L.unflow_debug_streamk_plan(G, mtp, nt, nchunk, len(ntaps), (ctypes.c_int * len(ntaps))(*ntaps), ngroups, rng, total, pos, dec)
    assert rc == 0
    return np.array(rng[:]), np.array(dec[:]).reshape(total, 4)


# (G, M pairs, N tiles, chunks, taps per class, M groups): conv3_1 fwd, conv4 dgrad (4/2/2/1), conv3 dgrad (9/6/6/4), deconv3 fwd,
# launches with fewer K tiles than workgroups, a single item, ragged last groups, class-major order (1 group)
CASES = [(256, 96, 2, 15, [9], 8), (256, 24, 2, 16, [4, 2, 2, 1], 8), (256, 96, 1, 8, [9, 6, 6, 4], 8), (256, 24, 1, 25, [4, 4, 4, 4], 8),
         (256, 2, 2, 8, [9], 8), (256, 1, 1, 3, [1], 8), (7, 13, 1, 5, [3, 1, 2], 4), (256, 500, 2, 1, [1, 7], 8), (304, 17, 1, 11, [5, 3], 8),
         (256, 24, 2, 16, [4, 2, 2, 1], 1), (256, 21, 3, 4, [2, 2, 2, 2], 8), (256, 9, 1, 7, [4, 2], 5)]


@pytest.mark.parametrize("case", CASES)
def test_streamk_ranges_tile_the_work(case):
    G, mtp, nt, nchunk, ntaps, ngroups = case
    P, D = plan(G, mtp, nt, nchunk, ntaps, ngroups)
    total = len(D)
    # every (class, N tile, pair, K tile) exactly once; an item's K tiles are consecutive positions, in order
    nk = np.array(ntaps)[D[:, 0]] * nchunk
    assert len({tuple(r) for r in D}) == total
    assert np.all((D[:, 3] >= 0) & (D[:, 3] < nk)) and D[:, 1].max() == nt - 1 and D[:, 2].max() == mtp - 1 and D[:, 0].max() == len(ntaps) - 1
    same = np.all(D[1:, :3] == D[:-1, :3], axis=1)
    assert np.all(D[1:, 3][same] == D[:-1, 3][same] + 1) and np.all(D[1:, 3][~same] == 0) and np.all(D[:-1, 3][~same] == nk[:-1][~same] - 1)
    assert P[0] == 0 and P[G] == total and np.all(np.diff(P) >= 0)
    assert np.diff(P).max() - np.diff(P).min() <= 1                    # balanced to one K tile
    item_start = np.arange(total) - D[:, 3]
    for w in range(G):
        a, b = P[w], P[w + 1]
        if a == b:
            continue
        segs = []
        pos = a
        while pos < b:
            k0 = D[pos, 3]
            k1 = min(nk[pos], k0 + (b - pos))
            segs.append((item_start[pos], k0, k1, nk[pos]))
            pos += k1 - k0
        not_start = [s for s in segs if s[1] > 0]
        start_not_end = [s for s in segs if s[1] == 0 and s[2] < s[3]]
        assert len(not_start) <= 1 and (not not_start or not_start[0] == segs[0])
        assert len(start_not_end) <= 1 and (not start_not_end or start_not_end[0] == segs[-1])
        for start, k0, k1, n in start_not_end:
            end = start + n
            covered = start + k1
            w2 = w + 1
            while w2 < G and P[w2] < end:
                if P[w2 + 1] > P[w2]:
                    assert P[w2] == covered and item_start[P[w2]] == start and D[P[w2], 3] > 0
                    covered = min(end, P[w2 + 1])
                w2 += 1
            assert covered == end


def test_streamk_groups_keep_the_classes_of_a_tile_together():
    """8 M groups: the eighth of the list an XCD's workgroups walk holds every class and N tile of ITS M pairs only."""
    G, mtp, nt, nchunk, ntaps = 256, 24, 2, 16, [4, 2, 2, 1]
    P, D = plan(G, mtp, nt, nchunk, ntaps, 8)
    for x in range(8):
        its = D[P[32 * x]:P[32 * (x + 1)]]
        assert set(its[:, 2]) == {3 * x, 3 * x + 1, 3 * x + 2}
        assert set(its[:, 0]) == {0, 1, 2, 3} and set(its[:, 1]) == {0, 1}


def test_streamk_hook_rejects_bad_arguments():
    from unflow_amd import _lib
    L = _lib.lib()
    out = (ctypes.c_int * 9)()
    assert L.unflow_debug_streamk_plan(8, 4, 1, 4, 5, (ctypes.c_int * 5)(1, 1, 1, 1, 1), 1, out, 0, None, None) != 0
    assert L.unflow_debug_streamk_plan(8, 4, 1, 4, 1, (ctypes.c_int * 1)(0), 1, out, 0, None, None) != 0
    assert L.unflow_debug_streamk_plan(256, 1 << 20, 1, 64, 1, (ctypes.c_int * 1)(9), 8, out, 0, None, None) != 0      # overflows 31 bits
