#!/usr/bin/env python3
"""Writes tests/golden/ckpt_fixture/ — a small TensorFlow checkpoint-V2 bundle (model.ckpt-42.index + .data-00000-of-00001
+ the `checkpoint` state file) with the kinds of entries tf.train.Saver leaves behind for the reference's graphs
(src/e2eflow/core/train.py:29-44): conv / conv_transpose weights and biases under the reference's scopes, Adam slot
variables, beta powers, a scalar int64 global_step — enough keys (and long shared prefixes) to span several table blocks.
TensorFlow cannot be installed here, so the bundle is written by unflow_amd/core/tf_checkpoint.py itself; the test
(tests/test_tf_checkpoint_cpu.py) reads these COMMITTED bytes and checks them against the values this script's seed
defines, which pins the format against accidental changes on either side (a reader/writer pair that drifts together
still fails on the committed files).

    python tests/golden/make_ckpt_fixture.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def fixture_tensors():
    rs = np.random.RandomState(20260926)
    t = {}
    for scope in ("flownet_c_features/conv1", "flownet_c/conv_redir", "flownet_c/flow6", "flownet_c/flow6_up5",
                  "stack_1_flownet/flownet_s/conv1", "stack_1_flownet/flownet_s/full_res/flow0"):
        shape = {"conv1": (7, 7, 3, 8), "conv_redir": (1, 1, 16, 4), "flow6": (3, 3, 8, 2), "flow6_up5": (4, 4, 2, 2),
                 "flow0": (3, 3, 5, 2)}[scope.split('/')[-1]]
        t[scope + "/weights"] = rs.randn(*shape).astype(np.float32)
        t[scope + "/biases"] = rs.randn(shape[3] if 'up' not in scope else shape[2]).astype(np.float32)
        for slot in ("Adam", "Adam_1"):
            t[scope + "/weights/" + slot] = rs.randn(*shape).astype(np.float32)
            t[scope + "/biases/" + slot] = np.zeros_like(t[scope + "/biases"])
    for i in range(40):                    # many keys with a long common prefix: several 4 KB table blocks
        t["stack_2_flownet/flownet_s/filler_%02d/weights" % i] = rs.randn(2, 3).astype(np.float32)
    t["beta1_power"] = np.asarray(0.9 ** 42, dtype=np.float32)
    t["beta2_power"] = np.asarray(0.999 ** 42, dtype=np.float32)
    t["global_step"] = np.asarray(42, dtype=np.int64)
    return t


if __name__ == "__main__":
    from unflow_amd.core import tf_checkpoint as T
    d = os.path.join(HERE, "ckpt_fixture")
    os.makedirs(d, exist_ok=True)
    T.write_checkpoint(os.path.join(d, "model.ckpt-42"), fixture_tensors(), block_size=512)     # several table blocks
    print(sorted(os.listdir(d)), sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)), "bytes")
