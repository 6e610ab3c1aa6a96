#!/usr/bin/env python3
"""A TensorFlow checkpoint-V2 bundle assembled byte by byte from the two format specifications, WITHOUT importing
unflow_amd (in particular not core/tf_checkpoint.py, whose reader this fixture pins):

  * the LevelDB table format (leveldb/doc/table_format.md): data blocks of prefix-compressed entries
    [shared varint32][non_shared varint32][value_len varint32][key suffix][value], a restart array of uint32 offsets +
    uint32 count, the 5-byte block trailer (compression type 0 + masked CRC-32C over contents + type), a meta-index block,
    an index block whose values are BlockHandles (offset varint64, size varint64), and the 48-byte footer
    (both handles, zero padding to 40 bytes, magic 0xdb4775248b80fb57 little endian);
  * tensorflow/core/protobuf/tensor_bundle.proto: key "" -> BundleHeaderProto {1: num_shards, 2: endianness, 3: version
    {1: producer}}, every other key -> BundleEntryProto {1: dtype, 2: shape {2: dim {1: size}}, 3: shard_id, 4: offset,
    5: size, 6: crc32c fixed32}; tensors lie back to back, in key order, in <prefix>.data-00000-of-00001.

Choices that follow TensorFlow's own writer (BundleWriter / table::TableBuilder): restart interval 16 in data blocks and
1 in the index block, no compression, proto3 default-valued fields omitted (shard_id 0, offset 0, endianness LITTLE),
index keys = the last key of each block, masked CRC = rotate-right-15 + 0xa282ead8.  The data-block size is set to 160 bytes
so that these few tensors span several blocks.

Tensor VALUES are closed-form (value_of below), so the test recomputes what it expects without any file.

    python tests/golden/make_ckpt_golden_independent.py      # rewrites tests/golden/ckpt_golden/
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "ckpt_golden")
STEM = "model.ckpt-1234"

# name -> (numpy dtype, shape); the names are the reference's (flownet.py scopes + Adam slots + a scalar)
TENSORS = {
    "flownet_c/conv3_1/biases": ("<f4", (7,)),
    "flownet_c/conv3_1/weights": ("<f4", (3, 3, 5, 7)),
    "flownet_c/conv3_1/weights/Adam": ("<f4", (3, 3, 5, 7)),
    "flownet_c/conv3_1/weights/Adam_1": ("<f4", (3, 3, 5, 7)),
    "flownet_c/flow2/biases": ("<f4", (2,)),
    "flownet_c/flow2/weights": ("<f4", (3, 3, 6, 2)),
    "flownet_c_features/conv1/biases": ("<f4", (4,)),
    "flownet_c_features/conv1/weights": ("<f4", (7, 7, 3, 4)),
    "global_step": ("<i8", ()),
    "stack_1_flownet/flownet_s/deconv2/weights": ("<f4", (4, 4, 3, 5)),
    "stack_1_flownet/flownet_s/flow2_up1_full_res/weights": ("<f4", (4, 4, 2, 2)),
    "beta1_power": ("<f4", ()),
}
DT_ENUM = {"<f4": 1, "<i8": 9}      # tensorflow/core/framework/types.proto: DT_FLOAT = 1, DT_INT64 = 9


def value_of(name, dtype, shape):
    """Closed-form contents: element i of tensor `name` = ((i * 37 + sum(name bytes)) % 1009 - 504) / 64 (floats: exactly
    representable), global_step = 1234, beta1_power = 0.9 ** 3 in float32."""
    if name == "global_step":
        return np.asarray(1234, dtype=dtype)
    if name == "beta1_power":
        return np.asarray(np.float32(0.9) * np.float32(0.9) * np.float32(0.9), dtype=dtype)
    n = int(np.prod(shape)) if shape else 1
    salt = sum(name.encode())
    v = ((np.arange(n, dtype=np.int64) * 37 + salt) % 1009 - 504).astype(np.float64) / 64.0
    return v.astype(dtype).reshape(shape)


# ---- CRC-32C (Castagnoli, reflected polynomial 0x82f63b78), bit by bit — deliberately not the table-driven form
def crc32c(data):
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def masked(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def pb_varint_field(field, n):
    return varint(field << 3) + varint(n)


def pb_bytes_field(field, payload):
    return varint((field << 3) | 2) + varint(len(payload)) + payload


def header_proto():
    version = pb_varint_field(1, 1)                          # VersionDef.producer = 1
    return pb_varint_field(1, 1) + pb_bytes_field(3, version)   # num_shards = 1; endianness LITTLE (0) omitted


def entry_proto(dtype, shape, offset, size, crc):
    dims = b"".join(pb_bytes_field(2, pb_varint_field(1, d)) for d in shape)      # TensorShapeProto.dim[i].size
    out = pb_varint_field(1, DT_ENUM[dtype]) + pb_bytes_field(2, dims)
    if offset:
        out += pb_varint_field(4, offset)
    out += pb_varint_field(5, size)
    out += varint((6 << 3) | 5) + struct.pack("<I", masked(crc))                   # fixed32
    return out


class Block:
    def __init__(self, restart_every):
        self.every, self.body, self.restarts, self.n, self.prev = restart_every, bytearray(), [], 0, b""

    def add(self, key, value):
        if self.n % self.every == 0:
            self.restarts.append(len(self.body))
            shared = 0
        else:
            shared = 0
            while shared < min(len(key), len(self.prev)) and key[shared] == self.prev[shared]:
                shared += 1
        self.body += varint(shared) + varint(len(key) - shared) + varint(len(value)) + key[shared:] + value
        self.prev, self.n = key, self.n + 1

    def finish(self):
        rs = self.restarts or [0]
        return bytes(self.body) + b"".join(struct.pack("<I", r) for r in rs) + struct.pack("<I", len(rs))


def append_block(file_bytes, contents):
    """-> BlockHandle bytes of the block just appended (offset, size exclude the trailer)."""
    handle = varint(len(file_bytes)) + varint(len(contents))
    file_bytes += contents + b"\x00" + struct.pack("<I", masked(crc32c(contents + b"\x00")))
    return handle


def build():
    assert crc32c(b"123456789") == 0xE3069283                  # RFC 3720 B.4 check value
    names = sorted(TENSORS, key=lambda s: s.encode())
    data = bytearray()
    records = [(b"", header_proto())]
    for name in names:
        dtype, shape = TENSORS[name]
        raw = value_of(name, dtype, shape).tobytes()
        records.append((name.encode(), entry_proto(dtype, shape, len(data), len(raw), crc32c(raw))))
        data += raw
    table = bytearray()
    index = Block(1)
    cur = Block(16)
    for key, value in records:
        cur.add(key, value)
        if len(cur.body) >= 160:
            index.add(cur.prev, append_block(table, cur.finish()))
            cur = Block(16)
    if cur.n:
        index.add(cur.prev, append_block(table, cur.finish()))
    meta_handle = append_block(table, Block(16).finish())
    index_handle = append_block(table, index.finish())
    footer = meta_handle + index_handle
    table += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    return bytes(table), bytes(data)


def main():
    os.makedirs(OUT, exist_ok=True)
    table, data = build()
    with open(os.path.join(OUT, STEM + ".index"), "wb") as f:
        f.write(table)
    with open(os.path.join(OUT, STEM + ".data-00000-of-00001"), "wb") as f:
        f.write(data)
    with open(os.path.join(OUT, "checkpoint"), "w") as f:       # text-format CheckpointState, two earlier entries in the history
        f.write('model_checkpoint_path: "%s"\n' % STEM)
        for s in ("model.ckpt-400", "model.ckpt-800", STEM):
            f.write('all_model_checkpoint_paths: "%s"\n' % s)
    with open(os.path.join(OUT, STEM + ".index.hex"), "w") as f:      # the same bytes, reviewable in a diff
        for i in range(0, len(table), 32):
            f.write(table[i:i + 32].hex() + "\n")
    print("wrote %s: index %d bytes, data %d bytes" % (OUT, len(table), len(data)))


if __name__ == "__main__":
    main()
