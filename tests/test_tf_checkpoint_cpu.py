"""TF checkpoint-V2 bundles without TensorFlow (unflow_amd/core/tf_checkpoint.py) and restore_networks on top of them —
the (f2) row of SURVEY 8f: src/e2eflow/core/train.py:23-65 (tf.train.Saver restore of the released C ... CSS_ft models,
README.md:116-128).  CPU only."""
import os
import struct
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from unflow_amd.core import tf_checkpoint as T        # noqa: E402
from unflow_amd.core import input as I                # noqa: E402

FIX = os.path.join(HERE, "golden", "ckpt_fixture")


def test_crc32c_known_answers_and_chunked_path():
    assert T.crc32c(b"") == 0
    assert T.crc32c(b"123456789") == 0xe3069283                      # the CRC-32C check value (RFC 3720 B.4)
    assert T.crc32c(bytes(32)) == 0x8a9136aa and T.crc32c(b"\xff" * 32) == 0x62a8ab43     # RFC 3720 B.4 test patterns
    rs = np.random.RandomState(1)
    big = rs.randint(0, 256, size=(1 << 18) + 37, dtype=np.uint8).tobytes()
    c = 0xffffffff
    for b in big:
        c = T._CRC_T0[(c ^ b) & 0xff] ^ (c >> 8)
    assert T.crc32c(big) == c ^ 0xffffffff                            # vectorised chunks == the byte loop
    assert T.crc32c(big[70001:], T.crc32c(big[:70001])) == T.crc32c(big)
    assert T._unmask(T._mask(0xdeadbeef)) == 0xdeadbeef


def test_committed_fixture_reads_back_the_seeded_values():
    from make_ckpt_fixture import fixture_tensors
    want = fixture_tensors()
    prefix = T.latest_checkpoint(FIX)
    assert prefix.endswith("model.ckpt-42")
    header, entries = T.checkpoint_entries(prefix)
    assert header["num_shards"] == 1 and list(entries) == sorted(want, key=lambda s: s.encode())
    got = T.read_checkpoint(prefix, verify_data=True)
    assert set(got) == set(want)
    for k, v in want.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    assert got["global_step"].shape == () and int(got["global_step"]) == 42
    # the index really spans several data blocks (prefix-compressed keys, restart points, index block)
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack_from("<Q", raw, len(raw) - 8)[0] == T.TABLE_MAGIC
    foot = raw[-48:]
    pos = 0
    for _ in range(2):
        _, pos = T._get_varint(foot, pos)
    ioff, pos = T._get_varint(foot, pos)
    isize, pos = T._get_varint(foot, pos)
    assert len(list(T._block_entries(T._read_block(raw, ioff, isize)))) >= 4      # data blocks listed by the index block
    sel = T.read_checkpoint(prefix, names=["flownet_c/flow6/weights"])
    assert list(sel) == ["flownet_c/flow6/weights"]
    with pytest.raises(KeyError):
        T.read_checkpoint(prefix, names=["flownet_c/flow7/weights"])


def test_corruption_is_detected(tmp_path):
    from make_ckpt_fixture import fixture_tensors
    p = str(tmp_path / "m.ckpt-1")
    T.write_checkpoint(p, fixture_tensors())
    idx = bytearray(open(p + ".index", "rb").read())
    idx[100] ^= 0x40
    open(p + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError, match="checksum"):
        T.read_checkpoint(p)
    idx[100] ^= 0x40
    open(p + ".index", "wb").write(bytes(idx))
    dat = bytearray(open(p + ".data-00000-of-00001", "rb").read())
    dat[10] ^= 1
    open(p + ".data-00000-of-00001", "wb").write(bytes(dat))
    with pytest.raises(ValueError, match="tensor checksum"):
        T.read_checkpoint(p, verify_data=True)
    T.read_checkpoint(p, verify_data=False)                          # (the index alone is still consistent)
    open(p + ".index", "wb").write(bytes(idx[:-8]) + b"\0" * 8)
    with pytest.raises(ValueError, match="magic"):
        T.read_checkpoint(p)


def _engine(spec, **extra):
    from unflow_amd.core.engine import FlowNetEngine
    return FlowNetEngine(1, 64, 64, params=dict(flownet=spec, **extra), device="cpu", layout_only=True)


def _random_params(eng, seed):
    g = torch.Generator().manual_seed(seed)
    return {k: torch.randn(v.shape, generator=g) for k, v in eng.export_tf_params().items()}


def test_css_full_res_every_variable_round_trips_through_a_checkpoint(tmp_path):
    """Every variable name and shape of 'css' + full_res (the 3/8-width stack: the same names and layer list as 'CSS',
    flownet.py:22-23, a seventh of the bytes) written as ONE checkpoint per network — what the reference's experiments
    leave on disk — and restored with restore_networks; then the train.py:56-63 fallback: a last-network checkpoint that
    predates full_res restores everything else and leaves the full_res variables at their initialisation."""
    eng = _engine("css", full_res=True)
    names = list(eng.export_tf_params())
    assert any(k.startswith("flownet_c_features/") for k in names) and any("stack_2_flownet/flownet_s/full_res/" in k for k in names)
    src = _random_params(eng, 5)
    files = []
    for i in range(3):
        scope = I.network_scope(i)
        d = tmp_path / ("net%d" % i)
        d.mkdir()
        part = {k: v for k, v in src.items() if k.startswith(scope)}
        # optimizer slots ride along like in a real Saver checkpoint and must be ignored
        part.update({k + "/Adam": torch.zeros_like(v) for k, v in list(part.items())[:3]})
        I.save_checkpoint(str(d / ("model.ckpt-%d" % (1000 * (i + 1)))), part, global_step=1000 * (i + 1))
        files.append(str(d))                                      # a directory: resolved through its `checkpoint` file
    eng.load_tf_params(_random_params(eng, 6))
    I.restore_networks(eng, eng.params, files)
    got = eng.export_tf_params()
    assert list(got) == names
    for k in names:
        assert torch.equal(got[k], src[k]), k
    # --- a checkpoint of the last network written before full_res existed
    old = {k: v for k, v in src.items() if k.startswith(I.network_scope(2)) and "full_res" not in k}
    I.save_checkpoint(str(tmp_path / "old.ckpt-7"), old)
    init = _random_params(eng, 7)
    eng.load_tf_params(init)
    I.restore_networks(eng, eng.params, [None, None, str(tmp_path / "old.ckpt-7")])
    got = eng.export_tf_params()
    for k in names:
        if k.startswith(I.network_scope(2)) and "full_res" not in k:
            assert torch.equal(got[k], src[k]), k
        else:
            assert torch.equal(got[k], init[k]), k                # untouched: other networks, and the missing full_res variables
    # --- anything else missing is an error, as with tf.train.Saver.restore
    broken = dict(old)
    del broken["stack_2_flownet/flownet_s/conv3_1/weights"]
    I.save_checkpoint(str(tmp_path / "broken.ckpt-1"), broken)
    with pytest.raises(KeyError, match="conv3_1"):
        I.restore_networks(eng, eng.params, [None, None, str(tmp_path / "broken.ckpt-1")])
    wrong = dict(old)
    wrong["stack_2_flownet/flownet_s/conv1/biases"] = torch.zeros(5)
    I.save_checkpoint(str(tmp_path / "wrong.ckpt-1"), wrong)
    with pytest.raises(ValueError, match="shape"):
        I.restore_networks(eng, eng.params, [None, None, str(tmp_path / "wrong.ckpt-1")])
    with pytest.raises(ValueError):
        I.restore_networks(eng, eng.params, [None, None, None, None])


def test_full_width_names_equal_the_reference_scopes():
    """The variable names a 'CSS' checkpoint must carry (flownet.py:72-77,166-237): spot checks of both scopes and TF layouts."""
    eng = _engine("CS")
    p = eng.export_tf_params()
    assert tuple(p["flownet_c_features/conv1/weights"].shape) == (7, 7, 3, 64)
    assert tuple(p["flownet_c/conv3_1/weights"].shape) == (3, 3, 473, 256)
    assert tuple(p["flownet_c/deconv5/weights"].shape) == (4, 4, 512, 1024)              # conv2d_transpose: [k, k, out, in]
    assert tuple(p["stack_1_flownet/flownet_s/conv1/weights"].shape) == (7, 7, 14, 64)
    assert tuple(p["stack_1_flownet/flownet_s/flow2/biases"].shape) == (2,)


# ---- bytes this repo's writer never touched: tests/golden/ckpt_golden/ is assembled from the LevelDB table + tensor_bundle.proto
# specifications by tests/golden/make_ckpt_golden_independent.py, which does not import unflow_amd (own varint / protobuf / block
# builder, a bit-serial CRC-32C) and follows TensorFlow's writer settings (restart interval 16 / 1, last-key index entries)
GOLD = os.path.join(HERE, "golden", "ckpt_golden")


def test_reader_against_independently_assembled_bundle():
    import make_ckpt_golden_independent as G
    src = open(os.path.join(HERE, "golden", "make_ckpt_golden_independent.py")).read()
    assert "import unflow" not in src and "from unflow" not in src and "import tf_checkpoint" not in src
    table, data = G.build()                                           # the committed bytes are what the script builds
    assert open(os.path.join(GOLD, G.STEM + ".index"), "rb").read() == table
    assert open(os.path.join(GOLD, G.STEM + ".data-00000-of-00001"), "rb").read() == data
    prefix = T.latest_checkpoint(GOLD)
    assert os.path.basename(prefix) == G.STEM
    assert T.all_checkpoint_paths(GOLD) == ["model.ckpt-400", "model.ckpt-800", G.STEM]
    header, entries = T.checkpoint_entries(prefix)
    assert header["num_shards"] == 1
    assert list(entries) == sorted(G.TENSORS, key=lambda s: s.encode())
    got = T.read_checkpoint(prefix, verify_data=True)                 # checks every block CRC and every tensor CRC
    for name, (dtype, shape) in G.TENSORS.items():
        want = G.value_of(name, dtype, shape)
        assert got[name].dtype == want.dtype and got[name].shape == want.shape and np.array_equal(got[name], want), name
    assert int(got["global_step"]) == 1234 and got["global_step"].shape == ()
    # several data blocks with prefix-compressed keys behind restart points: the index block lists >= 3 of them
    raw = open(prefix + ".index", "rb").read()
    foot, pos = raw[-48:], 0
    for _ in range(2):
        _, pos = T._get_varint(foot, pos)
    ioff, pos = T._get_varint(foot, pos)
    isize, pos = T._get_varint(foot, pos)
    assert len(list(T._block_entries(T._read_block(raw, ioff, isize)))) >= 3
    # and the loader on top of it: network variables only, optimizer slots / counters dropped (input.py load_params)
    params = I.load_params(GOLD)
    assert "flownet_c/conv3_1/weights" in params and not any(k.endswith("/Adam") or k in ("global_step", "beta1_power") for k in params)
    # a flipped payload byte is caught by the entry CRC
    bad = bytearray(data)
    bad[17] ^= 0x40
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        for fn in os.listdir(GOLD):
            shutil.copy(os.path.join(GOLD, fn), d)
        with open(os.path.join(d, G.STEM + ".data-00000-of-00001"), "wb") as f:
            f.write(bytes(bad))
        with pytest.raises(Exception):
            T.read_checkpoint(os.path.join(d, G.STEM), verify_data=True)


def test_state_file_keeps_the_checkpoint_history(tmp_path):
    """Saver(max_to_keep=1000) + recover_last_checkpoints (train.py:38-44): all_model_checkpoint_paths grows."""
    for step in (10, 20, 30, 20):
        T.write_checkpoint(str(tmp_path / ("model.ckpt-%d" % step)), {"w": np.full(2, step, np.float32)})
    assert T.all_checkpoint_paths(str(tmp_path)) == ["model.ckpt-10", "model.ckpt-30", "model.ckpt-20"]
    assert os.path.basename(T.latest_checkpoint(str(tmp_path))) == "model.ckpt-20"
