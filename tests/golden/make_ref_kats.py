#!/usr/bin/env python3
"""Writes tests/golden/ref_kats.json — the known-answer vectors the reference's
own unit tests hold for the hot path.

The reference cannot be imported here (every module imports TensorFlow, which is
not installed and cannot be; the ops additionally need nvcc), so these vectors
are TRANSCRIBED from the reference's test files — inputs and expected outputs
exactly as written there — and each entry cites its source lines.  Derived
entries (marked "derived") follow from a closed form stated in the entry.
"""
import json
import math
import os

Z = [0, 0]

kats = {}

_grid = [[1, 1, 2, 2], [0, 0, 2, 2], [3, 3, 4, 4], [3, 3, 2, 2]]

kats["correlation_trivial"] = {
    "source": "src/e2eflow/test/ops/correlation.py:30-47 (expected = first**2)",
    "first": [[_grid]], "second": [[_grid]],          # [1,1,4,4] NCHW
    "attrs": {"kernel_size": 1, "stride_2": 1, "max_displacement": 0, "pad": 0},
    "expected": [[[[v * v for v in r] for r in _grid]]],
    "tol": "assertAllClose default (rtol 1e-6, atol 1e-6)",
}
kats["correlation_batch"] = {
    "source": "src/e2eflow/test/ops/correlation.py:49-69",
    "first": [[_grid], [_grid]], "second": [[_grid], [_grid]],
    "attrs": {"kernel_size": 1, "stride_2": 1, "max_displacement": 0, "pad": 0},
    "expected": [[[[v * v for v in r] for r in _grid]]] * 2,
}
kats["correlation_3x3_inputs"] = {
    "source": "src/e2eflow/test/ops/correlation.py:74-89 (test returns early: inputs only, no expected values; "
              "used for the Jacobian check recipe of :21-28, rtol/atol 1e-3)",
    "first": [[[[1, 1, 3], [0, 0, 1], [2, 2, 0.2]]]],
    "second": [[[[1, 2, 0.1], [3, 4, 2.2], [4, 5, 1.6]]]],
    "attrs": {"kernel_size": 3, "stride_2": 1, "max_displacement": 1, "pad": 2},
}

_move_first = [[0, 0, 0, 0], [0, 1, 0.5, 0], [0, 0.3, 0.4, 0], [0, 0, 0, 0]]
_move_second = [[0, 1, 0, 0], [0, 0, 0, 0.5], [0.3, 0, 0, 0], [0, 0, 0.4, 0]]
_move_flow = [[Z, [-1, 0], Z, Z], [Z, [0, -1], [1, 0], [0, -1]], [[0, -1], [-1, 0], [0, 1], Z], [Z, Z, [0, -1], Z]]
_zeros44 = [[0] * 4 for _ in range(4)]

kats["warp_move"] = {
    "source": "src/e2eflow/test/ops/backward_warp.py:27-48 and src/e2eflow/test/test_image_warp.py:23-44 "
              "(same vectors for ops.backward_warp and image_warp)",
    "expected": [_move_first], "image": [_move_second], "flow": [_move_flow],   # [1,4,4] -> reshape [1,4,4,1]/[1,4,4,2]
}
kats["backward_warp_batches"] = {
    "source": "src/e2eflow/test/ops/backward_warp.py:50-78 (middle sample: zero image, zero flow)",
    "expected": [_move_first, _zeros44, _move_first],
    "image": [_move_second, _zeros44, _move_second],
    "flow": [_move_flow, [[Z] * 4 for _ in range(4)], _move_flow],
}
kats["image_warp_batches"] = {
    "source": "src/e2eflow/test/test_image_warp.py:46-74 (middle sample: zero image, flow of ones :66)",
    "expected": [_move_first, _zeros44, _move_first],
    "image": [_move_second, _zeros44, _move_second],
    "flow": [_move_flow, [[[1, 1]] * 4 for _ in range(4)], _move_flow],
}
kats["warp_interpolate"] = {
    "source": "src/e2eflow/test/ops/backward_warp.py:80-101 and src/e2eflow/test/test_image_warp.py:76-97 (expects 2.1)",
    "expected": [[[0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 2.1]]],
    "image": [[[0, 0, 0, 0], [0, 1, 2, 0], [0, 3, 4, 0], [0, 0, 0, 0]]],
    "flow": [[[Z, Z, Z, Z], [Z, [-2, -2], [-2, -2], Z], [Z, [-2, -2], [-2, -2], Z], [Z, Z, Z, [-1.7, -1.6]]]],
}
kats["downsample"] = {
    "source": "src/e2eflow/test/ops/downsample.py:8-22",
    "image": [_grid], "scale": 2, "expected": [[[0.5, 2], [3, 3]]],
}

kats["smoothness_deltas"] = {
    "source": "src/e2eflow/test/test_losses.py:11-41 (assertAllEqual; deltas are multiplied by the mask)",
    "flow_u": [[0, 0, 0], [0, 8, 3], [0, 1, 0]], "flow_v": [[0, 0, 0], [0, 8, 3], [0, 1, 0]],
    "mask_x": [[1, 1, 0], [1, 1, 0], [1, 1, 0]], "mask_y": [[1, 1, 1], [1, 1, 1], [0, 0, 0]],
    "delta_x": [[0, 0, 0], [-8, 5, 0], [-1, 1, 0]], "delta_y": [[0, -8, -3], [0, 7, 3], [0, 0, 0]],
}
kats["outgoing_mask_all_directions"] = {
    "source": "src/e2eflow/test/test_losses.py:43-55",
    "flow_u": [[0, 0, 1], [-1, 3, 0], [0, 1, 0]], "flow_v": [[-1, 0, 0], [0, 0, 0], [1, -1, 0]],
    "expected": [[0, 1, 0], [0, 0, 1], [0, 1, 1]],
}
kats["outgoing_mask_large_movement"] = {
    "source": "src/e2eflow/test/test_losses.py:56-68",
    "flow_u": [[3, 2, 1], [2, 1, 0], [0, -2, -1]], "flow_v": [[0, 0, 0], [0, 0, 0], [0, 0, 0]],
    "expected": [[0, 0, 0], [1, 1, 1], [1, 0, 1]],
}
_g1 = [[0, 1, 0], [0, 2, 0], [0, 3, 4]]
kats["gradient_loss_constant_offset"] = {
    "source": "src/e2eflow/test/test_losses.py:97-121 (gradient_loss(im, im+1, ones) ~ 0, atol 1e-2)",
    "im1_channel": _g1, "im2_channel": [[v + 1 for v in r] for r in _g1], "expected": 0.0, "atol": 1e-2,
}

kats["forward_warp_zero_flow_interior"] = {
    "source": "derived: ops/forward_warp_op.cu.cc:38-61 with zero flow, interior pixel = "
              "sum_{|dx|,|dy|<=4} exp(-(dx^2+dy^2)/2)",
    "expected": sum(math.exp(-(dx * dx + dy * dy) / 2.0) for dx in range(-4, 5) for dy in range(-4, 5)),
}
kats["flownetc_correlation_shape"] = {
    "source": "derived: src/e2eflow/core/flownet.py:221-222 attrs through ops/correlation_op.h:36-51",
    "in_hw": [48, 64], "attrs": {"pad": 20, "kernel_size": 1, "max_displacement": 20, "stride_1": 1, "stride_2": 2},
    "expected_chw": [441, 48, 64],
}

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_kats.json")
with open(out, "w") as f:
    json.dump(kats, f, indent=1)
print("wrote", out, len(kats), "entries")
