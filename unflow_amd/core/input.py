"""Groundwork for importing the authors' weights and evaluating on real data — host-side mirrors of

  * tf.train.Saver variable naming (core/train.py:23-65): parameters travel as an .npz keyed by the reference's variable
    names ('flownet_c/conv3_1/weights', 'stack_1_flownet/flownet_s/full_res/flow0/biases', ...) in TF layouts (HWIO conv,
    [k,k,out,in] conv_transpose) — what `tf.train.load_checkpoint(...).get_tensor(name)` yields on a machine that has
    TensorFlow; no TensorFlow is needed (or available) here;
  * the ground-truth readers: KITTI 16-bit flow PNGs (kitti/input.py:12-22: flow = (uint16[..., :2] - 2^15) / 64, mask =
    channel 2) and Middlebury / Sintel .flo files (middlebury/input.py:10-29: 'PIEH' tag, width, height, float32 u,v;
    mask = both components < 1e9);
  * resize_output_flow (core/input.py:28-34): TF1 bilinear resize of a flow field to the ground-truth size with the
    vectors scaled by the size ratio — what eval uses before flow_error_avg / outlier_pct (eval_gui.py:68-93).

PNG decoding uses only the standard library (zlib): 16-bit RGB, non-interlaced, the five PNG filters."""
import struct
import zlib
from collections import OrderedDict

import numpy as np
import torch

FLO_TAG = 202021.25      # 'PIEH' as float32


# ----------------------------------------------------------------------------------------------------------- parameters
def save_params_npz(path, tf_params):
    """{variable name: tensor} -> .npz (the names contain '/', which np.savez keeps verbatim)."""
    np.savez(path, **{k: v.detach().cpu().numpy() for k, v in tf_params.items()})


def load_params_npz(path):
    """.npz keyed by the reference's variable names -> OrderedDict of float32 tensors (pass to engine.load_tf_params).
    A trailing ':0' (tensor names instead of variable names) is stripped."""
    out = OrderedDict()
    with np.load(path) as z:
        for k in z.files:
            name = k[:-2] if k.endswith(':0') else k
            out[name] = torch.from_numpy(np.ascontiguousarray(z[k], dtype=np.float32))
    return out


def load_params(path):
    """Variables of one saved network as {name: float32 tensor}: `path` is an .npz (save_params_npz), a TF checkpoint-V2
    prefix (`model.ckpt-70000`, read by core/tf_checkpoint.py without TensorFlow) or an experiment directory holding a
    `checkpoint` state file (what util.py:75-85 / experiment.py hand to train.py:48-64).  Optimizer slots
    ('.../Adam', '.../Adam_1'), beta powers and global_step are dropped: only conv weights and biases are returned."""
    import os
    from . import tf_checkpoint as T
    if path.endswith('.npz'):
        return load_params_npz(path)
    prefix = path
    if os.path.isdir(path):
        prefix = T.latest_checkpoint(path)
        if prefix is None:
            raise FileNotFoundError("no `checkpoint` state file in %s" % path)
    if not os.path.exists(prefix + '.index'):
        raise FileNotFoundError("%s: neither an .npz nor a checkpoint prefix (no %s.index)" % (path, prefix))
    out = OrderedDict()
    for k, v in T.read_checkpoint(prefix).items():
        if k.endswith('/weights') or k.endswith('/biases'):
            out[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return out


def network_scope(i):
    """Variable-name prefixes of network i of a spec (flownet.py:72-77, train.py:29): the first network's variables live
    under 'flownet_c' / 'flownet_c_features' (a FlowNetC) or 'flownet_s' (a FlowNetS), the others under 'stack_<i>_flownet/'."""
    return ('flownet_c', 'flownet_s') if i == 0 else ('stack_%d_flownet/' % i,)


def restore_networks(engine, params, net_files):
    """restore_networks (core/train.py:23-65): net_files[i] holds the variables of network i of the engine's spec — an .npz,
    a TF checkpoint prefix or an experiment directory (load_params); networks without a file keep their initialisation.
    Every variable of a restored network must be in its file with the network's shape — except, exactly like the reference's
    second attempt (:56-63: `if not 'full_res' in v.name`), the 'full_res' variables, which keep their initialisation when
    the file predates them.  Anything else missing raises, as tf.train.Saver.restore does."""
    cur = engine.export_tf_params()
    spec = engine.spec
    if len(net_files) > len(spec):
        raise ValueError("%d files for the %d networks of spec %r (train.py:31: len(finetune) <= flownet_num)"
                         % (len(net_files), len(spec), spec))
    for i, f in enumerate(net_files):
        if f is None:
            continue
        scope = network_scope(i)
        loaded = load_params(f)
        want = [k for k in cur if k.startswith(scope)]
        missing = [k for k in want if k not in loaded]
        hard = [k for k in missing if 'full_res' not in k]
        if hard:
            raise KeyError("%s lacks %d variable(s) of network %d of spec %r, e.g. %s" % (f, len(hard), i, spec, hard[0]))
        for k in want:
            if k in loaded:
                v = loaded[k]
                if tuple(v.shape) != tuple(cur[k].shape):
                    raise ValueError("%s: shape %s in %s, %s in the network" % (k, tuple(v.shape), f, tuple(cur[k].shape)))
                cur[k] = v
    engine.load_tf_params(cur)
    return cur


def save_checkpoint(prefix, tf_params, global_step=None):
    """Write parameters as a TF checkpoint-V2 bundle under the reference's variable names (readable by tf.train.Saver.restore
    of the reference graph, train.py:40-44)."""
    from . import tf_checkpoint as T
    t = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in tf_params.items()}
    if global_step is not None:
        t['global_step'] = np.asarray(global_step, dtype=np.int64)
    T.write_checkpoint(prefix, t)


# ----------------------------------------------------------------------------------------------------------- .flo
def read_flo(path):
    """middlebury/input.py:10-29 -> (flow [H,W,2] float32, mask [H,W,1] float32)."""
    with open(path, 'rb') as f:
        data = f.read()
    tag, = struct.unpack('<f', data[0:4])
    if tag != FLO_TAG:
        raise ValueError("%s: not a .flo file (tag %r)" % (path, tag))
    w, h = struct.unpack('<ii', data[4:12])
    flow = np.frombuffer(data, dtype='<f4', count=2 * w * h, offset=12).reshape(h, w, 2).copy()
    mask = np.logical_and(flow[:, :, 0] < 1e9, flow[:, :, 1] < 1e9).astype(np.float32)[:, :, None]
    return torch.from_numpy(flow), torch.from_numpy(mask)


def write_flo(path, flow):
    flow = np.ascontiguousarray(flow.detach().cpu().numpy() if isinstance(flow, torch.Tensor) else flow, dtype='<f4')
    h, w, _ = flow.shape
    with open(path, 'wb') as f:
        f.write(struct.pack('<f', FLO_TAG))
        f.write(struct.pack('<ii', w, h))
        f.write(flow.tobytes())


# ----------------------------------------------------------------------------------------------------------- PNG (16 bit)
def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def decode_png(data):
    """Minimal PNG decoder (8 / 16 bit, grey / RGB / RGBA, non-interlaced) -> uint8 / uint16 array [H,W,C]."""
    if data[:8] != b'\x89PNG\r\n\x1a\n':
        raise ValueError("not a PNG")
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, typ = struct.unpack('>I4s', data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b'IHDR':
            hdr = struct.unpack('>IIBBBBB', body)
        elif typ == b'IDAT':
            idat.append(body)
        elif typ == b'IEND':
            break
    if hdr is None:
        raise ValueError("PNG without an IHDR chunk")
    w, h, depth, ctype, _, _, interlace = hdr
    if interlace or depth not in (8, 16) or ctype not in (0, 2, 6):
        raise NotImplementedError("PNG variant (depth %d, colour type %d, interlace %d)" % (depth, ctype, interlace))
    ch = {0: 1, 2: 3, 6: 4}[ctype]
    bpp = ch * depth // 8
    stride = w * bpp
    raw = zlib.decompress(b''.join(idat))
    if len(raw) < h * (stride + 1):
        raise ValueError("truncated PNG image data")
    out = np.zeros((h, w, bpp), dtype=np.uint8)
    prev = np.zeros((w, bpp), dtype=np.int32)
    zero = np.zeros(bpp, dtype=np.int32)
    for y in range(h):
        ft = raw[y * (stride + 1)]
        line = np.frombuffer(raw, dtype=np.uint8, count=stride, offset=y * (stride + 1) + 1).astype(np.int32).reshape(w, bpp)
        if ft == 0:
            cur = line
        elif ft == 1:      # Sub: every byte lane of a pixel is a running sum along the row (mod 256)
            cur = np.cumsum(line, axis=0) & 255
        elif ft == 2:      # Up
            cur = (line + prev) & 255
        elif ft == 3 or ft == 4:
            # Average / Paeth depend on the pixel to the left: one step per PIXEL, all bpp byte lanes of it at once (a KITTI
            # flow map is 1242 x 6 bytes per row: 1242 short vector steps instead of 7452 interpreted byte steps)
            cur = np.empty((w, bpp), dtype=np.int32)
            a, c = zero, zero
            for x in range(w):
                b_ = prev[x]
                if ft == 3:
                    pred = (a + b_) >> 1
                else:
                    pa, pb, pc = np.abs(b_ - c), np.abs(a - c), np.abs(a + b_ - 2 * c)
                    pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b_, c))
                a = (line[x] + pred) & 255
                cur[x] = a
                c = b_
        else:
            raise ValueError("bad PNG filter %d" % ft)
        out[y] = cur
        prev = cur
    out = out.reshape(h, stride)
    if depth == 16:
        return out.reshape(h, w, ch, 2).astype(np.uint16).dot(np.array([256, 1], dtype=np.uint16)).astype(np.uint16)
    return out.reshape(h, w, ch)


def encode_png16_rgb(arr):
    """uint16 [H,W,3] -> PNG bytes (filter 0; for fixtures and round-trip tests)."""
    arr = np.ascontiguousarray(arr, dtype='>u2')
    h, w, _ = arr.shape
    raw = b''.join(b'\x00' + arr[y].tobytes() for y in range(h))

    def chunk(t, b):
        return struct.pack('>I', len(b)) + t + b + struct.pack('>I', zlib.crc32(t + b) & 0xffffffff)
    return b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 16, 2, 0, 0, 0)) + \
        chunk(b'IDAT', zlib.compress(raw)) + chunk(b'IEND', b'')


def read_kitti_flow_png(path):
    """kitti/input.py:12-22 -> (flow [H,W,2] float32 = (uint16 - 2^15) / 64, mask [H,W,1] float32 = channel 2)."""
    with open(path, 'rb') as f:
        gt = decode_png(f.read()).astype(np.float32)
    flow = (gt[:, :, 0:2] - 2 ** 15) / 64.0
    return torch.from_numpy(flow.copy()), torch.from_numpy(gt[:, :, 2:3].copy())


# ----------------------------------------------------------------------------------------------------------- resizing
def resize_bilinear_tf1(x, out_h, out_w):
    """tf.image.resize_bilinear, TF1 legacy (align_corners=False, no half-pixel centres): src = dst * in / out."""
    B, H, W, C = x.shape
    dev = x.device
    ys = torch.arange(out_h, device=dev, dtype=torch.float32) * (H / out_h)
    xs = torch.arange(out_w, device=dev, dtype=torch.float32) * (W / out_w)
    y0, x0 = ys.floor().long(), xs.floor().long()
    y1, x1 = (y0 + 1).clamp(max=H - 1), (x0 + 1).clamp(max=W - 1)
    fy, fx = (ys - y0.float()).view(1, -1, 1, 1), (xs - x0.float()).view(1, 1, -1, 1)
    top = x[:, y0][:, :, x0] * (1 - fx) + x[:, y0][:, :, x1] * fx
    bot = x[:, y1][:, :, x0] * (1 - fx) + x[:, y1][:, :, x1] * fx
    return top * (1 - fy) + bot * fy


def resize_output_flow(t, height, width, channels=2):
    """core/input.py:28-34: bilinear resize of a flow field to (height, width), u scaled by width / old_width, v by
    height / old_height."""
    _, old_h, old_w, _ = t.shape
    r = resize_bilinear_tf1(t, height, width)
    return torch.stack([r[..., 0] * (width / old_w), r[..., 1] * (height / old_h)], dim=3)
