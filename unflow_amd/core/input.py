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
import os
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
    """Minimal PNG decoder (8 / 16 bit, grey / grey+alpha / RGB / RGBA, non-interlaced) -> uint8 / uint16 array [H,W,C]."""
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
    if interlace or depth not in (8, 16) or ctype not in (0, 2, 4, 6):
        raise NotImplementedError("PNG variant (depth %d, colour type %d, interlace %d)" % (depth, ctype, interlace))
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
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


def resize_input(t, height, width, resized_h, resized_w):
    """core/input.py:10-14 (evaluation): the batch-1 image that the input pipeline cropped / zero-padded to (resized_h,
    resized_w) goes back to its own (height, width) and is then stretched bilinearly onto the network's size."""
    import torch
    dev = t.device
    a = resize_image_with_crop_or_pad(t.reshape(resized_h, resized_w, 3).cpu().numpy(), int(height), int(width))
    return resize_bilinear_tf1(torch.from_numpy(np.ascontiguousarray(a)).unsqueeze(0).to(dev), resized_h, resized_w)


def resize_output_crop(t, height, width, channels):
    """core/input.py:17-21: a batch-1 ground-truth map cropped / zero-padded back to the frame's own size."""
    import torch
    _, oldh, oldw, c = t.shape
    a = resize_image_with_crop_or_pad(t.reshape(oldh, oldw, c).cpu().numpy(), int(height), int(width))
    return torch.from_numpy(np.ascontiguousarray(a)).reshape(1, int(height), int(width), channels).to(t.device)


def resize_output(t, height, width, channels):
    """core/input.py:24-25."""
    return resize_bilinear_tf1(t, int(height), int(width))


# ------------------------------------------------------------------------------------- raw-frame input (core/input.py:37-218)
def frame_name_to_num(name):
    """frame_name_to_num (input.py:37-41): '0000012.png' -> 12, '000.png' -> 0."""
    stripped = name.split('.')[0].lstrip('0')
    return 0 if stripped == '' else int(stripped)


def encode_png8_rgb(arr):
    """uint8 [H,W,3] -> PNG bytes (filter 0, one IDAT): the writer for the input-pipeline fixtures and tests."""
    import struct
    import zlib
    a = np.ascontiguousarray(arr, dtype=np.uint8)
    h, w, c = a.shape
    assert c == 3
    raw = b''.join(b'\x00' + a[y].tobytes() for y in range(h))

    def chunk(t, b):
        return struct.pack('>I', len(b)) + t + b + struct.pack('>I', zlib.crc32(t + b) & 0xffffffff)
    return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0)) +
            chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def read_png_image(path):
    """read_png_image (input.py:208-218) for one file: decode_png(channels=3) cast to float32, [H,W,3]."""
    with open(path, 'rb') as f:
        a = decode_png(f.read())
    if a.dtype == np.uint16:          # decode_png(dtype=uint8) of a 16-bit file keeps the high byte
        a = (a >> 8).astype(np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.shape[2] in (1, 2):          # grey (+ alpha): replicated to RGB, as channels=3 does
        a = np.repeat(a[:, :, :1], 3, axis=2)
    return a[:, :, :3].astype(np.float32)


def resize_image_with_crop_or_pad(a, height, width):
    """tf.image.resize_image_with_crop_or_pad on [H,W,C]: central crop and / or symmetric zero padding (the extra row /
    column of an odd difference goes to the bottom / right, as in TF)."""
    h, w = a.shape[:2]
    if h > height:
        o = (h - height) // 2
        a = a[o:o + height]
    if w > width:
        o = (w - width) // 2
        a = a[:, o:o + width]
    h, w = a.shape[:2]
    if h < height or w < width:
        top, left = (height - h) // 2, (width - w) // 2
        out = np.zeros((height, width) + a.shape[2:], dtype=a.dtype)
        out[top:top + h, left:left + w] = a
        a = out
    return a


class RawPairBatches:
    """What tf.train.batch over a string_input_producer(shuffle=False, num_epochs=None) yields (input.py:186-205): the pair
    list is walked in order, cyclically, `batch_size` examples per call; every example is read, cropped with ONE random
    window for both frames (augment.random_crop, augment.py:113-134) or reshaped, and normalised.  next() returns
    (image_1, image_2) as float32 arrays [B,H,W,3]."""

    def __init__(self, pairs, batch_size, dims, needs_crop, normalize, mean, stddev, seed):
        self.pairs, self.batch_size, self.dims = list(pairs), batch_size, tuple(dims)
        self.needs_crop, self.normalize = needs_crop, normalize
        self.mean, self.stddev = np.asarray(mean, dtype=np.float32), np.float32(stddev)
        self.pos = 0
        self.rng = np.random.RandomState(seed)

    def __iter__(self):
        return self

    def _example(self, fn1, fn2):
        a, b = read_png_image(fn1), read_png_image(fn2)
        h, w = self.dims
        if self.needs_crop:
            lim_h, lim_w = min(a.shape[0], b.shape[0]), min(a.shape[1], b.shape[1])
            oy = int(self.rng.randint(0, lim_h - h + 1))
            ox = int(self.rng.randint(0, lim_w - w + 1))
            a, b = a[oy:oy + h, ox:ox + w], b[oy:oy + h, ox:ox + w]
        else:
            a, b = a.reshape(h, w, 3), b.reshape(h, w, 3)
        if self.normalize:
            a, b = (a - self.mean) / self.stddev, (b - self.mean) / self.stddev
        return a, b

    def __next__(self):
        im1, im2 = [], []
        for _ in range(self.batch_size):
            a, b = self._example(*self.pairs[self.pos % len(self.pairs)])
            self.pos += 1
            im1.append(a)
            im2.append(b)
        return np.stack(im1).astype(np.float32), np.stack(im2).astype(np.float32)


class Input:
    """core/input.py:44-205 of the reference without the TF queue runners: the same pair lists (sorted listing, sequence /
    pair stepping, skipped-frame filter, seeded shuffle, swapped pairs, `shift` for resuming), the same preprocessing, batches as
    numpy arrays.  `data` needs get_raw_dirs() and, for the test inputs, current_dir (the reference's Data classes)."""
    mean = [104.920005, 110.1753, 114.785955]
    stddev = 1 / 0.0039216

    def __init__(self, data, batch_size, dims, *, num_threads=1, normalize=True, skipped_frames=False):
        assert len(dims) == 2
        self.data = data
        self.dims = dims
        self.batch_size = batch_size
        self.num_threads = num_threads          # kept for signature parity; reading is synchronous here
        self.normalize = normalize
        self.skipped_frames = skipped_frames

    def get_normalization(self):
        return self.mean, self.stddev

    def _normalize_image(self, image):
        return (image - np.asarray(self.mean, dtype=np.float32)) / np.float32(self.stddev)

    def _preprocess_image(self, image):
        h, w = self.dims
        image = resize_image_with_crop_or_pad(image, h, w).reshape(h, w, 3)
        return self._normalize_image(image) if self.normalize else image

    @staticmethod
    def _pair_indices(n_files, sequence, skip):
        """Index pairs (i, i + 1) into one directory's sorted listing, in the order input_raw emits them (input.py:142-160).
        sequence: one pass per entry g of `skip` (frames to jump over), first files i = 0, 1 + g, 2 (1 + g), ... while
        i < n - (1 + g); the partner is ALWAYS the next file in the listing.  Not a sequence: the listing is pairs (0,1), (2,3), ..."""
        if not sequence:
            if n_files % 2:
                raise AssertionError("an uncorrelated-pairs directory must hold an even number of images")
            return [(i, i + 1) for i in range(0, n_files, 2)]
        out = []
        for gap in skip:
            stride = gap + 1
            out += [(i, i + 1) for i in range(0, n_files - stride, stride)]
        return out

    def raw_pairs(self, swap_images=True, sequence=True, shift=0, seed=0, skip=0):
        """The ordered example list of input_raw (input.py:121-184): [(first file, second file), ...]."""
        import random
        skip = skip if isinstance(skip, list) else [skip]
        pairs = []
        for folder in self.data.get_raw_dirs():
            listing = sorted(os.listdir(folder))
            for i, j in self._pair_indices(len(listing), sequence, skip):
                if sequence and self.skipped_frames:
                    # datasets with dropped frames: keep a pair only when the frame numbers are consecutive (input.py:153-158)
                    assert len(skip) == 1 and skip[0] == 0
                    if frame_name_to_num(listing[j]) - frame_name_to_num(listing[i]) != 1:
                        continue
                pairs.append((os.path.join(folder, listing[i]), os.path.join(folder, listing[j])))
        random.Random(seed).shuffle(pairs)       # == random.seed(seed); random.shuffle(...) of the reference (same generator, same state)
        if swap_images:
            pairs = [q for a, b in pairs for q in ((a, b), (b, a))]
        # the reference rolls the FLATTENED [n, 2] name array by `shift` (np.roll without an axis, input.py:173): an odd shift
        # therefore also exchanges the roles of first and second file — kept
        names = [f for pr in pairs for f in pr]
        k = shift % len(pairs)
        names = names[len(names) - k:] + names[:len(names) - k]
        return list(zip(names[0::2], names[1::2]))

    def input_raw(self, swap_images=True, sequence=True, needs_crop=True, shift=0, seed=0, center_crop=False, skip=0):
        """input_raw (input.py:121-205): an iterator of (image_1, image_2) batches [B,H,W,3] float32.  `shift`: examples to skip
        at the start — the reference resumes training with shift = batch_size * iterations done (train.py / run.py)."""
        pairs = self.raw_pairs(swap_images=swap_images, sequence=sequence, shift=shift, seed=seed, skip=skip)
        print("Training on {} frame pairs.".format(len(pairs) // (2 if swap_images else 1)))
        return RawPairBatches(pairs, self.batch_size, self.dims, needs_crop, self.normalize, self.mean, self.stddev, seed)

    def test_pairs(self, image_dir, hold_out_inv=None):
        """_input_images (input.py:76-108): consecutive files of a directory are pairs; hold_out_inv keeps the first n pairs
        of the seed-0 shuffle."""
        import random
        image_dir = os.path.join(self.data.current_dir, image_dir)
        files = sorted(os.listdir(image_dir))
        assert len(files) % 2 == 0, 'expected pairs of images'
        pairs = [(os.path.join(image_dir, files[2 * i]), os.path.join(image_dir, files[2 * i + 1])) for i in range(len(files) // 2)]
        if hold_out_inv is not None:
            random.seed(0)
            random.shuffle(pairs)
            pairs = pairs[:hold_out_inv]
        return pairs

    def input_test(self, image_dir, hold_out_inv=None):
        """_input_test (input.py:110-116): batches of (image_1, image_2, input_shape), one epoch, a smaller final batch allowed."""
        pairs = self.test_pairs(image_dir, hold_out_inv)
        for b0 in range(0, len(pairs), self.batch_size):
            im1, im2, shp = [], [], []
            for fn1, fn2 in pairs[b0:b0 + self.batch_size]:
                a, b = read_png_image(fn1), read_png_image(fn2)
                shp.append(np.asarray(a.shape, dtype=np.int32))
                im1.append(self._preprocess_image(a))
                im2.append(self._preprocess_image(b))
            yield np.stack(im1).astype(np.float32), np.stack(im2).astype(np.float32), np.stack(shp)
