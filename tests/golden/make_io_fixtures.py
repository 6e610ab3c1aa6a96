#!/usr/bin/env python3
"""Writes the small IO fixtures the CPU tests read (tests/test_input_cpu.py):
  tiny.flo              3 x 4 Middlebury flow, u = x - 1.5, v = 0.25 y, pixel (1, 2) invalid (1e10) — format of
                        middlebury/input.py:10-29
  tiny_kitti_flow.png   2 x 3 KITTI flow PNG, uint16 RGB = (2^15 + 64 u, 2^15 + 64 v, valid) — kitti/input.py:12-22
Run from the repo root: python tests/golden/make_io_fixtures.py"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from unflow_amd.core import input as I  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
flow = np.zeros((3, 4, 2), dtype=np.float32)
flow[..., 0] = np.arange(4, dtype=np.float32)[None, :] - 1.5
flow[..., 1] = 0.25 * np.arange(3, dtype=np.float32)[:, None]
flow[1, 2] = 1e10
with open(os.path.join(here, 'tiny.flo'), 'wb') as f:
    f.write(struct.pack('<f', I.FLO_TAG) + struct.pack('<ii', 4, 3) + flow.astype('<f4').tobytes())

u = np.array([[-2.0, 0.0, 0.5], [3.25, -0.015625, 100.0]])
v = np.array([[1.0, -1.0, 0.0], [0.5, 0.25, -1.0]])
valid = np.array([[1, 0, 1], [1, 0, 1]])
png = np.stack([2 ** 15 + 64 * u, 2 ** 15 + 64 * v, valid], axis=2).astype(np.uint16)
with open(os.path.join(here, 'tiny_kitti_flow.png'), 'wb') as f:
    f.write(I.encode_png16_rgb(png))
print("wrote tiny.flo, tiny_kitti_flow.png")
