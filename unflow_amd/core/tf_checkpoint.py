"""TensorFlow checkpoint-V2 ("tensor bundle") files without TensorFlow — what tf.train.Saver writes and restores in
src/e2eflow/core/train.py:23-65 (`saver.restore(sess, ckpt.model_checkpoint_path)`) and what the authors' released models
(README.md:116-128: the C ... CSS_ft experiments) are stored as:

    <prefix>.index                   a LevelDB-format sorted string table: key "" -> BundleHeaderProto,
                                     key <variable name> -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}
    <prefix>.data-0000S-of-0000N     the raw little-endian tensor bytes of shard S

This module reads such a bundle into {variable name: numpy array} (read_checkpoint) and writes one (write_checkpoint: the
fixture side of the tests, and a way to hand weights trained here back to the reference).  Formats followed:

  * table: data blocks of prefix-compressed entries (varint32 shared / non_shared / value_length, key delta, value) + the
    restart array + uint32 restart count; per block a 1-byte compression type (0 = none, 1 = snappy — TF writes the index
    uncompressed) and a masked CRC32C; metaindex block; index block (separator key -> BlockHandle varint64 offset, size);
    48-byte footer = two BlockHandles padded to 40 bytes + magic 0xdb4775248b80fb57 (little endian);
  * protos: hand-decoded varint / length-delimited / fixed32 fields of BundleHeaderProto (num_shards = 1, endianness = 2,
    version = 3) and BundleEntryProto (dtype = 1, shape = 2 {dim = 2 {size = 1}}, shard_id = 3, offset = 4, size = 5,
    crc32c = 6 fixed32, slices = 7).

Partitioned variables (`slices`) are not supported (the reference has none).  Host-side I/O: no GPU, no torch needed."""
import os
import struct
from collections import OrderedDict

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8

# TensorFlow DataType enum values <-> numpy
_DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 4: np.dtype('u1'), 5: np.dtype('<i2'), 6: np.dtype('i1'),
           9: np.dtype('<i8'), 10: np.dtype('bool'), 17: np.dtype('<u2'), 19: np.dtype('<f2'), 22: np.dtype('<u4'),
           23: np.dtype('<u8')}
_DTYPE_ENUM = {v: k for k, v in _DTYPES.items()}


# ------------------------------------------------------------------------------------------------------------ CRC32C
def _make_crc_tables():
    poly = 0x82f63b78
    t0 = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (poly if c & 1 else 0)
        t0[i] = c
    return [t0]


_CRC_T = _make_crc_tables()
_CRC_T0 = [int(x) for x in _CRC_T[0]]


_CHUNK = 4096
_ZTAB = None


def _advance_tables():
    """Z[k][b] = CRC register after _CHUNK zero bytes, started from b << 8k: the (linear) map that carries a register across
    one chunk, as four byte-indexed tables."""
    global _ZTAB
    if _ZTAB is None:
        t0 = _CRC_T[0]
        c = (np.arange(256, dtype=np.uint32)[None, :] << (np.arange(4, dtype=np.uint32) * 8)[:, None]).reshape(-1)
        for _ in range(_CHUNK):
            c = t0[c & np.uint32(0xff)] ^ (c >> np.uint32(8))
        _ZTAB = [[int(x) for x in row] for row in c.reshape(4, 256)]
    return _ZTAB


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), the checksum of LevelDB blocks and of every tensor in a bundle.  Small inputs: a byte loop.  Large
    inputs: the register update is linear, so all 4 KB chunks are run from a zero register AT ONCE (one numpy step per byte
    position across every chunk) and then folded left to right through the 'advance by one chunk of zeros' map — ~60 MB/s,
    enough to verify the data of a full FlowNetCSS checkpoint (0.5 GB) in seconds."""
    buf = np.frombuffer(bytes(data) if not isinstance(data, (bytes, bytearray, memoryview, np.ndarray)) else data, dtype=np.uint8)
    c = (crc ^ 0xffffffff) & 0xffffffff
    t0 = _CRC_T0
    n = len(buf)
    nch = n // _CHUNK if n >= 16 * _CHUNK else 0
    if nch:
        cols = np.ascontiguousarray(buf[:nch * _CHUNK].reshape(nch, _CHUNK).T)        # [byte position][chunk]
        T0 = _CRC_T[0]
        r = np.zeros(nch, dtype=np.uint32)
        m8, s8 = np.uint32(0xff), np.uint32(8)
        for j in range(_CHUNK):
            r = T0[(r ^ cols[j]) & m8] ^ (r >> s8)
        Z0, Z1, Z2, Z3 = _advance_tables()
        for x in r.tolist():
            c = Z0[c & 0xff] ^ Z1[(c >> 8) & 0xff] ^ Z2[(c >> 16) & 0xff] ^ Z3[c >> 24] ^ x
    for bt in buf[nch * _CHUNK:].tolist():
        c = t0[(c ^ bt) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xffffffff


def _unmask(m):
    rot = (m - _MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ------------------------------------------------------------------------------------------------------------ varints
def _get_varint(buf, pos):
    shift = val = 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        val |= (b & 0x7f) << shift
        if not b & 0x80:
            return val, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _proto_fields(buf):
    """Yield (field number, wire type, value) of a serialized message: varints as ints, length-delimited as bytes,
    fixed32 / fixed64 as ints."""
    pos = 0
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            if len(v) != n:
                raise ValueError("truncated field")
            pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield fn, wt, v


def _field(fn, wt, payload):
    key = _put_varint((fn << 3) | wt)
    if wt == 0:
        return key + _put_varint(payload)
    if wt == 2:
        return key + _put_varint(len(payload)) + payload
    if wt == 5:
        return key + struct.pack('<I', payload)
    raise ValueError(wt)


# ------------------------------------------------------------------------------------------------------------ table (read)
def _read_block(data, offset, size, verify=True):
    """Contents of the block at (offset, size): checks the trailer (compression type + masked CRC32C)."""
    if offset + size + 5 > len(data):
        raise ValueError("block handle (%d, %d) outside the file" % (offset, size))
    raw = data[offset:offset + size]
    ctype = data[offset + size]
    if verify:
        want = _unmask(struct.unpack_from('<I', data, offset + size + 1)[0])
        if crc32c(data[offset:offset + size + 1]) != want:
            raise ValueError("block checksum mismatch at offset %d" % offset)
    if ctype == 1:
        return _snappy_decompress(raw)
    if ctype != 0:
        raise ValueError("unknown block compression %d" % ctype)
    return raw


def _block_entries(block):
    """(key, value) pairs of a table block (prefix-compressed keys; the restart array at the end is not needed for a scan)."""
    if len(block) < 4:
        raise ValueError("block too small")
    nrestarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestarts
    if end < 0:
        raise ValueError("bad restart array")
    pos, key = 0, b''
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > end:
            raise ValueError("corrupt block entry")
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def _snappy_decompress(buf):
    """Raw snappy (for tables written with compression; TF's bundle writer does not compress)."""
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        t = tag & 3
        if t == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if t == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif t == 2:
            ln = (tag >> 2) + 1
            off = struct.unpack_from('<H', buf, pos)[0]
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("bad snappy copy")
        for _ in range(ln):
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy length mismatch")
    return bytes(out)


def read_table(path, verify=True):
    """All (key, value) pairs of a LevelDB-format table file, in key order."""
    with open(path, 'rb') as f:
        data = f.read()
    if len(data) < 48:
        raise ValueError("%s: too small for a table" % path)
    footer = data[-48:]
    if struct.unpack_from('<Q', footer, 40)[0] != TABLE_MAGIC:
        raise ValueError("%s: not a checkpoint index (bad table magic)" % path)
    pos = 0
    _, pos = _get_varint(footer, pos)        # metaindex handle
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p = _get_varint(handle, 0)
        bsize, p = _get_varint(handle, p)
        out.extend(_block_entries(_read_block(data, boff, bsize, verify)))
    return out


# ------------------------------------------------------------------------------------------------------------ bundle (read)
def _parse_entry(buf):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for fn, wt, v in _proto_fields(buf):
        if fn == 1:
            e['dtype'] = v
        elif fn == 2:
            for f2, _, dim in _proto_fields(v):
                if f2 == 2:          # TensorShapeProto.dim
                    size = 0
                    for f3, _, dv in _proto_fields(dim):
                        if f3 == 1:
                            size = dv
                    e['shape'].append(size)
                elif f2 == 3 and dim:   # unknown_rank
                    raise ValueError("tensor of unknown rank in a checkpoint")
        elif fn == 3:
            e['shard_id'] = v
        elif fn == 4:
            e['offset'] = v
        elif fn == 5:
            e['size'] = v
        elif fn == 6:
            e['crc32c'] = v
        elif fn == 7:
            e['sliced'] = True
    return e


def checkpoint_entries(prefix, verify=True):
    """(header, {name: entry}) of the bundle <prefix>.index; entry = dict(dtype, shape, shard_id, offset, size, crc32c)."""
    header, entries = dict(num_shards=1, endianness=0, version=None), OrderedDict()
    for k, v in read_table(prefix + '.index', verify):
        if k == b'':
            for fn, _, val in _proto_fields(v):
                if fn == 1:
                    header['num_shards'] = val
                elif fn == 2:
                    header['endianness'] = val
            continue
        entries[k.decode('utf-8')] = _parse_entry(v)
    if header['endianness'] != 0:
        raise ValueError("big-endian checkpoint")
    return header, entries


def read_checkpoint(prefix, names=None, verify_data=True):
    """{variable name: numpy array} of the bundle at `prefix` (the path tf.train.Saver reports, without .index / .data-*).
    names: restrict to these variables (KeyError if one is missing).  verify_data: also check every tensor's CRC32C."""
    header, entries = checkpoint_entries(prefix)
    if names is not None:
        missing = [n for n in names if n not in entries]
        if missing:
            raise KeyError("not in checkpoint %s: %s" % (prefix, ", ".join(missing[:5])))
    shards = {}
    out = OrderedDict()
    for name, e in entries.items():
        if names is not None and name not in names:
            continue
        if e['sliced']:
            raise NotImplementedError("%s: partitioned variable" % name)
        if e['dtype'] not in _DTYPES:
            raise NotImplementedError("%s: dtype enum %d" % (name, e['dtype']))
        dt = _DTYPES[e['dtype']]
        n = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
        if n * dt.itemsize != e['size']:
            raise ValueError("%s: %d bytes for shape %s of %s" % (name, e['size'], e['shape'], dt))
        sid = e['shard_id']
        if sid not in shards:
            shards[sid] = np.memmap("%s.data-%05d-of-%05d" % (prefix, sid, header['num_shards']), dtype=np.uint8, mode='r')
        raw = shards[sid][e['offset']:e['offset'] + e['size']]
        if len(raw) != e['size']:
            raise ValueError("%s: data shard %d is truncated" % (name, sid))
        if verify_data and e['crc32c'] is not None and _unmask(e['crc32c']) != crc32c(np.asarray(raw)):
            raise ValueError("%s: tensor checksum mismatch" % name)
        out[name] = np.frombuffer(raw.tobytes(), dtype=dt).reshape(e['shape']).copy()
    return out


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: the prefix named by the text-proto `checkpoint` file of a directory (the reference finds
    its restore path this way: experiment.py, util.py:75-85), or None."""
    p = os.path.join(directory, 'checkpoint')
    if not os.path.exists(p):
        return None
    with open(p) as f:
        for line in f:
            line = line.strip()
            if line.startswith('model_checkpoint_path:'):
                name = line.split(':', 1)[1].strip().strip('"')
                return name if os.path.isabs(name) else os.path.join(directory, name)
    return None


def all_checkpoint_paths(directory):
    """all_model_checkpoint_paths of a directory's `checkpoint` state file, oldest first (what
    Saver.recover_last_checkpoints is fed, train.py:43-44); [] without a state file."""
    p = os.path.join(directory, 'checkpoint')
    out = []
    if os.path.exists(p):
        with open(p) as f:
            for line in f:
                line = line.strip()
                if line.startswith('all_model_checkpoint_paths:'):
                    out.append(line.split(':', 1)[1].strip().strip('"'))
    return out


# ------------------------------------------------------------------------------------------------------------ bundle (write)
class _BlockBuilder:
    def __init__(self, restart_interval=16):
        self.buf, self.restarts, self.count, self.last, self.ri = bytearray(), [0], 0, b'', restart_interval

    def add(self, key, value):
        shared = 0
        if self.count % self.ri == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def finish(self):
        out = bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))
        return out


def _emit_block(out, block):
    """Append block + trailer to the bytearray `out`; returns the BlockHandle bytes."""
    off = len(out)
    out += block + b'\x00'
    out += struct.pack('<I', _mask(crc32c(block + b'\x00')))
    return _put_varint(off) + _put_varint(len(block))


def write_table(path, items, block_size=4096):
    """items: [(key bytes, value bytes)] in strictly increasing key order -> LevelDB-format table (uncompressed)."""
    out = bytearray()
    index = _BlockBuilder(restart_interval=1)
    cur, last_key = _BlockBuilder(), None
    prev = None
    for k, v in items:
        if prev is not None and not prev < k:
            raise ValueError("table keys must be strictly increasing")
        prev = k
        cur.add(k, v)
        last_key = k
        if len(cur.buf) >= block_size:
            index.add(last_key, _emit_block(out, cur.finish()))
            cur, last_key = _BlockBuilder(), None
    if cur.count:
        index.add(last_key, _emit_block(out, cur.finish()))
    meta = _emit_block(out, _BlockBuilder().finish())
    idx = _emit_block(out, index.finish())
    footer = meta + idx
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    out += footer
    with open(path, 'wb') as f:
        f.write(bytes(out))


def write_checkpoint(prefix, tensors, directory_state=True, block_size=4096):
    """Write {variable name: array} as a single-shard V2 bundle (prefix.index + prefix.data-00000-of-00001) and, like
    tf.train.Saver, the `checkpoint` state file next to it."""
    names = sorted(tensors, key=lambda s: s.encode('utf-8'))
    data = bytearray()
    items = [(b'', _field(1, 0, 1) + _field(3, 2, _field(1, 0, 1)))]       # num_shards = 1, version.producer = 1
    for name in names:
        a = np.asarray(tensors[name])
        if not a.flags.c_contiguous:
            a = np.ascontiguousarray(a)      # (never for 0-d: ascontiguousarray would make it 1-d)
        if a.dtype.byteorder == '>':
            a = a.astype(a.dtype.newbyteorder('<'))
        if a.dtype not in _DTYPE_ENUM:
            raise NotImplementedError("%s: dtype %s" % (name, a.dtype))
        raw = a.tobytes()
        shape = b''.join(_field(2, 2, _field(1, 0, int(d))) for d in a.shape)
        entry = _field(1, 0, _DTYPE_ENUM[a.dtype]) + (_field(2, 2, shape) if a.shape else _field(2, 2, b''))
        if len(data):
            entry += _field(4, 0, len(data))
        entry += _field(5, 0, len(raw)) + _field(6, 5, _mask(crc32c(raw)))
        items.append((name.encode('utf-8'), entry))
        data += raw
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(bytes(data))
    write_table(prefix + '.index', items, block_size)
    if directory_state:
        d, base = os.path.split(prefix)
        # the Saver of the reference keeps up to 1000 checkpoints and recovers the list on resume (max_to_keep=1000,
        # recover_last_checkpoints, train.py:38-44): the history is appended to, never rewritten
        hist = [h for h in all_checkpoint_paths(d) if h != base] + [base]
        with open(os.path.join(d, 'checkpoint'), 'w') as f:
            f.write('model_checkpoint_path: "%s"\n' % base)
            for h in hist[-1000:]:
                f.write('all_model_checkpoint_paths: "%s"\n' % h)
