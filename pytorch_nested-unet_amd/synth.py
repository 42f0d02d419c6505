"""Seeded synthetic inputs and closed-form parameters.

Everything here is regenerable from integers on any box (numpy only), so golden
fixtures under tests/golden/ store *outputs* only, never inputs or weights.

Data model follows SURVEY.md §8(d):
  image : uint8 U{0..255} [N,H,W,3] -> (x/255 - mean)/std (albumentations
          Normalize() defaults used at reference trains.py:266) -> /255 again
          (reference dataset.py:71 quirk) -> CHW float32
  mask  : K in [5,30] random filled ellipses, 255 -> /255 -> {0,1}
          (reference dataset.py:73); one independent field per class
          (reference dataset.py:60-64).
"""
import math

import numpy as np

_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float64)
_STD = np.array([0.229, 0.224, 0.225], dtype=np.float64)

NB_FILTER = (32, 64, 128, 256, 512)  # reference finished/archs1.py:78


def synth_images(n, h, w, cin=3, seed=1234):
    """[n,cin,h,w] float32, distribution of a reference Dataset sample."""
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 256, size=(n, h, w, cin), dtype=np.uint8).astype(np.float64)
    mean = np.resize(_MEAN, cin)
    std = np.resize(_STD, cin)
    x = (raw / 255.0 - mean) / std
    x = x / 255.0
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2)).astype(np.float32)


def synth_masks(n, h, w, ncls=1, seed=1234):
    """[n,ncls,h,w] float32 in {0,1}: nuclei-like blob fields."""
    rng = np.random.default_rng(seed + 7919)
    yy, xx = np.mgrid[0:h, 0:w]
    out = np.zeros((n, ncls, h, w), dtype=np.float32)
    for i in range(n):
        for c in range(ncls):
            k = int(rng.integers(5, 31))
            m = np.zeros((h, w), dtype=bool)
            for _ in range(k):
                cy = rng.uniform(0, h)
                cx = rng.uniform(0, w)
                ry = rng.uniform(0.03, 0.10) * h + 1.0
                rx = rng.uniform(0.03, 0.10) * w + 1.0
                m |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
            out[i, c] = m
    return out


def synth_batch(n, h, w, cin=3, ncls=1, seed=1234):
    return synth_images(n, h, w, cin, seed), synth_masks(n, h, w, ncls, seed)


def synth_blob_pairs(n, h, w, seed=1234, cin=3):
    """A LEARNABLE segmentation set (the "val IoU vs ref" fixture): the image is rendered from
    the blob mask - background pixels U{0..199}, nuclei pixels U{56..255} per channel, so a single pixel says
    little (72 % of either range is shared) and the network needs spatial context - then goes through the same uint8 -> Normalize -> /255
    pipeline as synth_images (reference dataset.py:66-74). Returns (image [n,cin,h,w] f32, mask [n,1,h,w] f32)."""
    raw8, m8 = synth_blob_pairs_u8(n, h, w, seed, cin)
    msk = np.ascontiguousarray((m8 > 127).astype(np.float32).transpose(0, 3, 1, 2))
    raw = raw8.astype(np.float64)
    mean = np.resize(_MEAN, cin)
    std = np.resize(_STD, cin)
    x = ((raw / 255.0 - mean) / std) / 255.0
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2)).astype(np.float32), msk


def synth_blob_pairs_u8(n, h, w, seed=1234, cin=3):
    """The same set as synth_blob_pairs BEFORE the sample pipeline: decoded uint8 images [n,h,w,cin] and uint8 masks
    [n,h,w,1] in {0,255} - what a reference Dataset holds after cv2.imread (dataset.py:56-64), and what the device-side
    pipeline (dataset.py of this package, TrainStep(input_u8=True)) takes."""
    msk = synth_masks(n, h, w, 1, seed)
    rng = np.random.default_rng(seed + 104729)
    bg = rng.integers(0, 200, size=(n, h, w, cin), dtype=np.int32)
    fg = rng.integers(56, 256, size=(n, h, w, cin), dtype=np.int32)
    raw = np.where(msk[:, 0, :, :, None] > 0.5, fg, bg).astype(np.uint8)
    m8 = (msk[:, 0, :, :, None] > 0.5).astype(np.uint8) * 255
    return np.ascontiguousarray(raw), np.ascontiguousarray(m8)


def synth_split(n, h, w, cin=3, ncls=1, seed=1000):
    """Seeded dataset split of the drivers (train.py seed 1000, val.py / validation seed 2000). One class and three
    channels: the learnable blob set (synth_blob_pairs, the set behind tests/golden/train_log_blobs.npz); other shapes:
    independent noise images and blob masks (synth_batch)."""
    if cin == 3 and ncls == 1:
        return synth_blob_pairs(n, h, w, seed=seed)
    return synth_batch(n, h, w, cin, ncls, seed=seed)


# ---------------------------------------------------------------------------
# model topology (reference finished/archs1.py:85-111,113-143)
# ---------------------------------------------------------------------------

def grid_nodes():
    """(i, j) nodes of the x_{i,j} grid in forward/registration-independent
    execution order: anti-diagonals, deepest first within a diagonal
    (reference finished/archs1.py:114-131)."""
    out = []
    for s in range(5):
        for j in range(s + 1):
            out.append((s - j, j))
    return out


def registration_order():
    """Module registration order (reference finished/archs1.py:85-103): by
    column j, then row i. Drives parameters()/state_dict order."""
    return [(i, j) for j in range(5) for i in range(5 - j)]


def block_channels(i, j, cin=3):
    """(in, mid, out) of VGGBlock conv{i}_{j} (reference finished/archs1.py:85-103)."""
    f = NB_FILTER
    if j == 0:
        inc = cin if i == 0 else f[i - 1]
    else:
        inc = f[i] * j + f[i + 1]
    return inc, f[i], f[i]


def head_names(deep_supervision):
    return ["final1", "final2", "final3", "final4"] if deep_supervision else ["final"]


def state_dict_spec(ncls=1, cin=3, deep_supervision=False):
    """Ordered [(name, shape, kind)] of the reference state_dict
    (reference finished/archs1.py:15-21,85-111)."""
    spec = []
    for (i, j) in registration_order():
        ci, cm, co = block_channels(i, j, cin)
        p = "conv%d_%d." % (i, j)
        for k, (a, b) in (("1", (ci, cm)), ("2", (cm, co))):
            spec.append((p + "conv%s.weight" % k, (b, a, 3, 3), "conv_w"))
            spec.append((p + "conv%s.bias" % k, (b,), "conv_b"))
            spec.append((p + "bn%s.weight" % k, (b,), "bn_w"))
            spec.append((p + "bn%s.bias" % k, (b,), "bn_b"))
            spec.append((p + "bn%s.running_mean" % k, (b,), "bn_rm"))
            spec.append((p + "bn%s.running_var" % k, (b,), "bn_rv"))
            spec.append((p + "bn%s.num_batches_tracked" % k, (), "bn_nbt"))
    for hn in head_names(deep_supervision):
        spec.append((hn + ".weight", (ncls, NB_FILTER[0], 1, 1), "conv_w"))
        spec.append((hn + ".bias", (ncls,), "conv_b"))
    return spec


def _hash_uniform(n, stream):
    """n float64 in [-1, 1): splitmix64 of (stream, index). Pure integer
    arithmetic, so identical on every box and numpy version."""
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = z + np.uint64(stream) * np.uint64(0xD1B54A32D192ED03)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0


def closed_form_state(ncls=1, cin=3, deep_supervision=False, fresh_bn=True, salt=0):
    """Closed-form (hash-generated) parameter set as {name: np.ndarray}, float32
    (int64 for num_batches_tracked). Conv weights/biases follow the reference's
    default init distribution U(+-1/sqrt(fan_in)) (nn.Conv2d defaults behind
    reference finished/archs1.py:18,20); BN affine is perturbed off (1, 0) so
    that gamma/beta paths are exercised. fresh_bn=True gives running stats 0/1
    as after construction; False gives non-trivial running stats (eval tests)."""
    out = {}
    fan_in = 1
    for t, (name, shape, kind) in enumerate(state_dict_spec(ncls, cin, deep_supervision)):
        n = int(np.prod(shape)) if len(shape) else 1
        u = _hash_uniform(n, 1000 * salt + t + 1)
        if kind == "conv_w":
            fan_in = shape[1] * shape[2] * shape[3]
            v = u / math.sqrt(fan_in)
        elif kind == "conv_b":
            v = u / math.sqrt(fan_in)          # fan_in of the weight just before
        elif kind == "bn_w":
            v = 1.0 + 0.1 * u
        elif kind == "bn_b":
            v = 0.1 * u
        elif kind == "bn_rm":
            v = np.zeros(n) if fresh_bn else 0.05 * u
        elif kind == "bn_rv":
            v = np.ones(n) if fresh_bn else 0.02 + 0.05 * (u + 1.0)
        elif kind == "bn_nbt":
            out[name] = np.array(0 if fresh_bn else 3, dtype=np.int64)
            continue
        out[name] = v.reshape(shape).astype(np.float32)
    return out


def unet_state_dict_spec(ncls=1, cin=3):
    """Ordered [(name, shape, kind)] of the reference plain U-Net's state_dict (reference finished/archs1.py:35-71):
    encoder column conv{i}_0, then decoder conv3_1, conv2_2, conv1_3, conv0_4, then `final`."""
    f = NB_FILTER
    blocks = [("conv%d_0" % i, (cin if i == 0 else f[i - 1]), f[i]) for i in range(5)]
    blocks += [("conv%d_%d" % (i, 4 - i), f[i] + f[i + 1], f[i]) for i in (3, 2, 1, 0)]
    spec = []
    for name, ci, co in blocks:
        for k, (a, b) in (("1", (ci, co)), ("2", (co, co))):
            spec.append((name + ".conv%s.weight" % k, (b, a, 3, 3), "conv_w"))
            spec.append((name + ".conv%s.bias" % k, (b,), "conv_b"))
            spec.append((name + ".bn%s.weight" % k, (b,), "bn_w"))
            spec.append((name + ".bn%s.bias" % k, (b,), "bn_b"))
            spec.append((name + ".bn%s.running_mean" % k, (b,), "bn_rm"))
            spec.append((name + ".bn%s.running_var" % k, (b,), "bn_rv"))
            spec.append((name + ".bn%s.num_batches_tracked" % k, (), "bn_nbt"))
    spec.append(("final.weight", (ncls, f[0], 1, 1), "conv_w"))
    spec.append(("final.bias", (ncls,), "conv_b"))
    return spec


def closed_form_state_unet(ncls=1, cin=3):
    """closed_form_state() for the plain U-Net (fresh BatchNorm buffers), hash streams 5000+."""
    out = {}
    fan_in = 1
    for t, (name, shape, kind) in enumerate(unet_state_dict_spec(ncls, cin)):
        n = int(np.prod(shape)) if len(shape) else 1
        u = _hash_uniform(n, 5000 + t)
        if kind == "conv_w":
            fan_in = shape[1] * shape[2] * shape[3]
            v = u / math.sqrt(fan_in)
        elif kind == "conv_b":
            v = u / math.sqrt(fan_in)
        elif kind == "bn_w":
            v = 1.0 + 0.1 * u
        elif kind == "bn_b":
            v = 0.1 * u
        elif kind == "bn_rm":
            v = np.zeros(n)
        elif kind == "bn_rv":
            v = np.ones(n)
        else:
            out[name] = np.array(0, dtype=np.int64)
            continue
        out[name] = v.reshape(shape).astype(np.float32)
    return out
