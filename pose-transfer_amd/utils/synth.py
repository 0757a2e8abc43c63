"""Counter-based synthetic data + weight generator.

Everything the tests, the golden-fixture generator and ``bench.py`` feed to the hot path is
derived from ``(seed, stream, index)`` through a stateless 64-bit mixer, so the same tensors
can be regenerated anywhere (this container, the GPU box) without shipping them.  Only numpy
is used; callers convert to torch.

Value distributions follow SURVEY.md §8d:
  * images / targets ~ U(-1, 1)
  * heat-maps: the ``cords_to_map`` Gaussian, sigma 6 (reference src_deformable/utils/pose_utils.py:79-86)
  * warps rows ``[s cos, -s sin, tx, s sin, s cos, ty, 0, 0]`` or the reference's "no point"
    transform ``[1,0,1000,0,1,1000,0,0]`` (reference src_deformable/utils/pose_transform.py:221)
  * masks: row 0 all ones (pose_transform.py:149), rows 1..9 filled rectangles
  * weights: Xavier-uniform conv weights, zero biases (reference models/networks.py:26-31)
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    """splitmix64 finaliser on a uint64 array."""
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def _stream_key(seed, stream):
    with np.errstate(over="ignore"):
        h = np.uint64(seed & 0xFFFFFFFF) * np.uint64(0x100000001B3)
        for ch in str(stream).encode():
            h = ((h ^ np.uint64(ch)) * np.uint64(0x100000001B3)) & _M64
        return _mix(np.array([h], dtype=np.uint64))[0]


def uniform(seed, stream, shape, lo=0.0, hi=1.0, dtype=np.float32):
    """U[lo,hi) tensor; element i = f(seed, stream, i) independent of shape."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        key = _stream_key(seed, stream)
        idx = np.arange(n, dtype=np.uint64)
        bits = _mix((idx * np.uint64(0xD1342543DE82EF95) + key) & _M64)
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (lo + (hi - lo) * u).astype(dtype).reshape(shape)


def normal(seed, stream, shape, dtype=np.float32):
    u1 = uniform(seed, str(stream) + "/a", shape, dtype=np.float64)
    u2 = uniform(seed, str(stream) + "/b", shape, dtype=np.float64)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return z.astype(dtype)


def xavier_uniform(seed, stream, shape):
    """Glorot-uniform for a conv / conv-transpose weight of the given 4-D shape.
    fan_in + fan_out is symmetric in dims 0/1, so the bound is the same for both kinds."""
    rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    bound = np.sqrt(6.0 / ((shape[0] + shape[1]) * rf))
    return uniform(seed, stream, shape, -bound, bound)


def heatmaps(seed, stream, n, pose_dim, h, w, sigma=6.0, p_missing=0.15):
    """(n, pose_dim, h, w) Gaussian key-point maps; missing key-points give a zero map."""
    ky = np.floor(uniform(seed, stream + "/y", (n, pose_dim)) * h)
    kx = np.floor(uniform(seed, stream + "/x", (n, pose_dim)) * w)
    miss = uniform(seed, stream + "/m", (n, pose_dim)) < p_missing
    yy = np.arange(h, dtype=np.float32)[None, None, :, None]
    xx = np.arange(w, dtype=np.float32)[None, None, None, :]
    g = np.exp(-((yy - ky[..., None, None]) ** 2 + (xx - kx[..., None, None]) ** 2) / (2.0 * sigma ** 2))
    g[miss] = 0.0
    return g.astype(np.float32)


def warps_and_masks(seed, stream, n, h, w, t=10, p_nopoint=0.25):
    """warps (n,t,8) float32 and masks (n,t,h,w) float32 as the reference Dataset emits them."""
    s = uniform(seed, stream + "/s", (n, t), 0.8, 1.25)
    phi = uniform(seed, stream + "/phi", (n, t), -np.pi / 6, np.pi / 6)
    tx = uniform(seed, stream + "/tx", (n, t), -0.2, 0.2) * h
    ty = uniform(seed, stream + "/ty", (n, t), -0.2, 0.2) * w
    wr = np.zeros((n, t, 8), dtype=np.float32)
    wr[..., 0] = s * np.cos(phi)
    wr[..., 1] = -s * np.sin(phi)
    wr[..., 2] = tx
    wr[..., 3] = s * np.sin(phi)
    wr[..., 4] = s * np.cos(phi)
    wr[..., 5] = ty
    nop = uniform(seed, stream + "/nop", (n, t)) < p_nopoint
    nop[:, 0] = False
    wr[nop] = np.array([1, 0, 1000, 0, 1, 1000, 0, 0], dtype=np.float32)
    masks = np.zeros((n, t, h, w), dtype=np.float32)
    masks[:, 0] = 1.0
    frac = uniform(seed, stream + "/area", (n, t), 0.02, 0.08)
    asp = uniform(seed, stream + "/asp", (n, t), 0.5, 2.0)
    cy = uniform(seed, stream + "/cy", (n, t))
    cx = uniform(seed, stream + "/cx", (n, t))
    for i in range(n):
        for j in range(1, t):
            if nop[i, j]:
                continue
            area = frac[i, j] * h * w
            rh = int(max(1, min(h, round(np.sqrt(area * asp[i, j])))))
            rw = int(max(1, min(w, round(np.sqrt(area / asp[i, j])))))
            y0 = int(cy[i, j] * (h - rh + 1))
            x0 = int(cx[i, j] * (w - rw + 1))
            masks[i, j, y0:y0 + rh, x0:x0 + rw] = 1.0
    return wr, masks


def batch(seed, tag, n, pose_dim, h, w):
    """One Dataset-shaped batch: input (n,3+2P,h,w), target (n,3,h,w), warps (n,10,8), masks (n,10,h,w).
    Channel order [img3, src_pose P, tgt_pose P] (reference datasets/PoseTransfer_Dataset.py:168-173)."""
    img = uniform(seed, tag + "/img", (n, 3, h, w), -1.0, 1.0)
    tgt = uniform(seed, tag + "/tgt", (n, 3, h, w), -1.0, 1.0)
    p_src = heatmaps(seed, tag + "/psrc", n, pose_dim, h, w)
    p_tgt = heatmaps(seed, tag + "/ptgt", n, pose_dim, h, w)
    wr, mk = warps_and_masks(seed, tag + "/wm", n, h, w)
    inp = np.concatenate([img, p_src, p_tgt], axis=1)
    return inp, tgt, wr, mk


def dropout_masks(seed, tag, n, channels=(512, 512, 512), p=0.5):
    """Channel-dropout multipliers in {0, 1/(1-p)} for the first three decoder blocks
    (reference models/networks.py:161,222,225: nn.Dropout2d() in train mode)."""
    out = []
    for i, c in enumerate(channels):
        keep = uniform(seed, "%s/drop%d" % (tag, i), (n, c)) >= p
        out.append((keep.astype(np.float32) / (1.0 - p)).astype(np.float32))
    return out


# ----------------------------------------------------------------------------- parameter specs

def generator_spec(pose_dim, nfilters_enc, nfilters_dec, num_skips=2, deformable=True):
    """[(state_dict key, shape)] in the reference's module order
    (reference models/networks.py:175-250; key names probed in SURVEY.md §8b)."""
    spec = []

    def enc(prefix, cin):
        for i, nf in enumerate(nfilters_enc):
            if i == 0:
                spec.append((prefix + ".net.0.weight", (nf, cin, 3, 3)))
                spec.append((prefix + ".net.0.bias", (nf,)))
            else:
                spec.append(("%s.net.%d.net.1.weight" % (prefix, i), (nf, nfilters_enc[i - 1], 4, 4)))
                if i != len(nfilters_enc) - 1:
                    spec.append(("%s.net.%d.net.2.weight" % (prefix, i), (1,)))
                    spec.append(("%s.net.%d.net.2.bias" % (prefix, i), (1,)))

    if deformable:
        enc("encoder_app", 3 + pose_dim)
        enc("encoder_pose", pose_dim)
    else:
        enc("encoder", 3 + 2 * pose_dim)
    nd = len(nfilters_dec)
    for i, nf in enumerate(nfilters_dec):
        if i == 0:
            cin = num_skips * nfilters_enc[-1]
        else:
            cin = num_skips * nfilters_enc[-(i + 1)] + nfilters_dec[i - 1]
        if i == nd - 1:
            spec.append(("decoder.net.%d.weight" % (i + 1), (nf, cin, 3, 3)))
            spec.append(("decoder.net.%d.bias" % (i + 1), (nf,)))
        else:
            spec.append(("decoder.net.%d.net.1.weight" % i, (cin, nf, 4, 4)))
            spec.append(("decoder.net.%d.net.3.weight" % i, (1,)))
            spec.append(("decoder.net.%d.net.3.bias" % i, (1,)))
    return spec


def discriminator_spec(input_nc, check_mode=0):
    """reference models/networks.py:337-353."""
    spec = [("net.0.weight", (64, input_nc, 4, 4)), ("net.0.bias", (64,))]
    chans = [64, 128, 256, 512] if check_mode == 0 else [64, 128, 256]
    for i in range(1, len(chans)):
        spec.append(("net.%d.net.1.weight" % i, (chans[i], chans[i - 1], 4, 4)))
        spec.append(("net.%d.net.2.weight" % i, (1,)))
        spec.append(("net.%d.net.2.bias" % i, (1,)))
    spec.append(("net.%d.net.1.weight" % len(chans), (1, chans[-1], 4, 4)))
    return spec


def init_params(seed, tag, spec, norm_jitter=0.0):
    """Xavier-uniform conv weights, zero conv biases, norm gamma=1 / beta=0 (PyTorch default).
    ``norm_jitter`` perturbs gamma/beta/biases so that parity tests exercise them."""
    out = {}
    for key, shape in spec:
        if len(shape) == 4:
            out[key] = xavier_uniform(seed, tag + "/" + key, shape)
        elif shape == (1,) and key.endswith("weight"):
            out[key] = (1.0 + norm_jitter * uniform(seed, tag + "/" + key, shape, -1, 1)).astype(np.float32)
        else:
            out[key] = (norm_jitter * uniform(seed, tag + "/" + key, shape, -1, 1)).astype(np.float32)
    return out


def nfilters(image_size):
    """reference models/pose_gan.py:17-18."""
    if max(image_size) < 256:
        return (64, 128, 256, 512, 512, 512), (512, 512, 512, 256, 128, 3)
    return (64, 128, 256, 512, 512, 512, 512), (512, 512, 512, 512, 256, 128, 3)
