"""Tensor helpers on the hot path (reference src_deformable/utils/pose_utils.py:45-54,79-86,227-233,312-338)."""
import os

import numpy as np
import torch

from ..runtime import lib as L

MISSING_VALUE = -1


def get_model_list(dirname, key):
    """Lexicographically last checkpoint whose name contains `key` (reference pose_utils.py:45-54)."""
    if not os.path.exists(dirname):
        return None
    models = [os.path.join(dirname, f) for f in os.listdir(dirname)
              if os.path.isfile(os.path.join(dirname, f)) and key in f and "pkl" in f]
    if not models:
        return None
    models.sort()
    return models[-1]


def cords_to_map(cords, img_size, sigma=6):
    """Key-point coordinates (y,x) -> Gaussian heat-maps, channels last (reference pose_utils.py:79-86)."""
    result = np.zeros(tuple(img_size) + cords.shape[0:1], dtype="float32")
    yy, xx = np.meshgrid(np.arange(img_size[0]), np.arange(img_size[1]), indexing="ij")
    for i, point in enumerate(cords):
        if point[0] == MISSING_VALUE or point[1] == MISSING_VALUE:
            continue
        result[..., i] = np.exp(-((yy - point[0]) ** 2 + (xx - point[1]) ** 2) / (2 * sigma ** 2))
    return result


def cords_to_map_device(cords, img_size, sigma=6, out=None):
    """Batched cords_to_map on the GPU: cords (N, P, 2) of (y, x) key-points (-1 = missing) -> (N, P, H, W) float32
    heat-maps, or written into `out` — any (N, P, H, W) strided view, e.g. the pose channel slice of the NCHW network
    input (reference Dataset.py:87,168-173 builds an HWC array on the host, transposes and uploads it)."""
    import torch
    from ..runtime import lib as L
    cords = torch.as_tensor(cords)
    assert cords.dim() == 3 and cords.shape[-1] == 2
    n, p = cords.shape[:2]
    h, w = img_size
    dev = out.device if out is not None else "cuda"
    cd = cords.to(device=dev, dtype=torch.float32).contiguous()
    if out is None:
        out = torch.empty(n, p, h, w, dtype=torch.float32, device=dev)
    assert out.shape == (n, p, h, w) and out.dtype == torch.float32 and out.is_cuda
    L.call("pg_cords_to_map", L.ptr(cd), n, p, h, w, float(sigma), L.ptr(out), *out.stride(), L.stream())
    return out


def get_imgpose(input, use_input_pose, pose_dim):
    """Channel slices [img | src_pose | tgt_pose] (reference pose_utils.py:227-233).  Views only."""
    inp_img = input[:, :3]
    inp_pose = input[:, 3:3 + pose_dim] if use_input_pose else None
    tg_pose_index = 3 + pose_dim if use_input_pose else 6
    return inp_img, inp_pose, input[:, tg_pose_index:]


def get_layer_ind(layer_name):
    """'block1_conv2' -> 1, 'block4_conv1' -> 19: index into vgg19.features (reference pose_utils.py:312-317)."""
    block, conv = layer_name.split("_")
    blocks = [0, 5, 10, 19, 28]
    return blocks[int(block[-1]) - 1] + int(conv[-1]) - 1


def Feature_Extractor(model, input=None, layer_name=None):
    """vgg19.features[:layer+1] on the reference's view-not-permute pre-processed input
    (reference pose_utils.py:320-338).  Only 'block1_conv2' (conv1_1 + ReLU) is implemented; `model` is
    (weight (64,3,3,3), bias (64,)).  Returns NCHW like the reference."""
    if get_layer_ind(layer_name) != 1:
        raise Exception("only block1_conv2 is implemented")
    w, b = model
    n, c, h, wd = input.shape
    feat = torch.empty(n, h, wd, 64, dtype=torch.float32, device=input.device)
    L.call("pg_vgg_conv1_relu_fwd", L.ptr(input.contiguous()), L.ptr(w.contiguous()), L.ptr(b.contiguous()), n, h, wd,
           L.ptr(feat), L.stream())
    out = torch.empty(n, 64, h, wd, dtype=torch.float32, device=input.device)
    L.call("pg_nhwc_to_nchw", L.ptr(feat), L.ptr(out), n, 64, h, wd, L.stream())
    return out


# ----------------------------------------------------------------------------------------------------------------------
# Data-pipeline helpers (host side; a few dozen numbers per sample)
def load_pose_cords_from_strings(y_str, x_str):
    """Annotation CSV cell pair -> (P, 2) integer array of (y, x) (reference pose_utils.py:160-163)."""
    import json
    return np.stack([np.asarray(json.loads(y_str)), np.asarray(json.loads(x_str))], axis=1)


def peak_cords(cords, img_size=None, threshold=0.1, sigma=6):
    """What the reference obtains by rendering key-points to heat-maps and reading the peaks back (cords_to_map ->
    map_to_cord, pose_utils.py:57-86): the integer pixel nearest to each (possibly fractional) key-point — ties go to the
    smaller coordinate because np.where lists the equal maxima in row-major order and the first one is kept; missing (-1)
    stays missing.  A key-point outside the frame peaks at the nearest in-image pixel with the value
    exp(-d^2 / (2 sigma^2)); map_to_cord keeps a peak only above `threshold`, so a key-point about 12.9 px or more outside
    (d^2 >= 165.8) becomes MISSING.  Integer in-image key-points are returned unchanged."""
    c = np.asarray(cords, dtype=np.float64)
    missing = (c[:, 0] == MISSING_VALUE) | (c[:, 1] == MISSING_VALUE)
    out = np.ceil(c - 0.5).astype(np.int64)
    if img_size is not None:
        out[:, 0] = np.clip(out[:, 0], 0, img_size[0] - 1)
        out[:, 1] = np.clip(out[:, 1], 0, img_size[1] - 1)
        d2 = ((out - c) ** 2).sum(axis=1)
        peak = np.exp(-d2 / (2 * sigma ** 2)).astype(np.float32)          # the heat-map is float32 (pose_utils.py:80)
        missing |= ~(peak > np.float32(threshold))
    out[missing] = MISSING_VALUE
    return out


def compute_interpol_pose(inp_pos, tg_pos, index, num_stacks, pose_dim):
    """Key-points of interpolation stage `index` of `num_stacks` (reference pose_utils.py:89-118): linear for the
    16-joint skeleton; for 18 joints a joint missing in one pose is taken from the other during its half of the sequence."""
    assert index <= num_stacks
    inp_pos, tg_pos = np.asarray(inp_pos), np.asarray(tg_pos)
    if pose_dim == 16:
        return inp_pos + (tg_pos - inp_pos) * index / num_stacks
    out = np.zeros([pose_dim, 2], dtype="float32")
    first_half = index <= num_stacks // 2
    for i in range(pose_dim):
        miss_in = inp_pos[i, 0] == MISSING_VALUE or inp_pos[i, 1] == MISSING_VALUE
        miss_tg = tg_pos[i, 0] == MISSING_VALUE or tg_pos[i, 1] == MISSING_VALUE
        if miss_in and miss_tg:
            out[i] = MISSING_VALUE
        elif miss_in:
            out[i] = MISSING_VALUE if first_half else tg_pos[i]
        elif miss_tg:
            out[i] = inp_pos[i] if first_half else MISSING_VALUE
        else:
            out[i] = inp_pos[i] + (tg_pos[i] - inp_pos[i]) * index / num_stacks
    return out


def _deprocess_image(image):
    """[-1, 1] float -> uint8 (reference pose_utils.py:219-220)."""
    return (255 * (image + 1) / 2).byte()


def make_grid(batch, row, col, order=0):
    """(B, h, w, c) numpy -> one (row*h, col*w, c) image, columns first for order 0 (reference pose_utils.py:290-307)."""
    batch = np.asarray(batch)
    h, w = batch.shape[1], batch.shape[2]
    out = np.empty((h * row, w * col, batch.shape[3]), dtype=batch.dtype)
    k = 0
    cells = [(j, i) for i in range(col) for j in range(row)] if order == 0 else [(i, j) for i in range(row) for j in range(col)]
    for r, c in cells:
        out[r * h:(r + 1) * h, c * w:(c + 1) * w] = batch[k]
        k += 1
    return out


COLORS = [[255, 0, 0], [255, 85, 0], [255, 170, 0], [255, 255, 0], [170, 255, 0], [85, 255, 0], [0, 255, 0], [0, 255, 85],
          [0, 255, 170], [0, 255, 255], [0, 170, 255], [0, 85, 255], [0, 0, 255], [85, 0, 255], [170, 0, 255],
          [255, 0, 255], [255, 0, 170], [255, 0, 85]]
LIMB_SEQ = [[0, 1], [1, 2], [2, 6], [6, 3], [3, 4], [4, 5], [10, 11], [11, 12], [12, 8], [8, 13], [13, 14], [14, 15], [6, 8], [8, 9]]
LIMB_SEQ_PAF = [[1, 2], [1, 5], [2, 3], [3, 4], [5, 6], [6, 7], [1, 8], [8, 9], [9, 10], [1, 11], [11, 12], [12, 13], [1, 0],
                [0, 14], [14, 16], [0, 15], [15, 17], [2, 16], [5, 17]]


def draw_pose_from_cords(cords, pose_dim, img_size, radius=2):
    """Skeleton picture of (y, x) key-points: white limb segments, coloured joint discs (the picture the reference's
    display() shows for the target pose, pose_utils.py:120-158; drawn with PIL instead of scikit-image — qualitative
    output, not a parity surface)."""
    from PIL import Image, ImageDraw
    h, w = img_size
    im = Image.new("RGB", (w, h))
    d = ImageDraw.Draw(im)
    present = lambda p: p[0] != MISSING_VALUE and p[1] != MISSING_VALUE
    for f, t in (LIMB_SEQ if pose_dim == 16 else LIMB_SEQ_PAF):
        if f < len(cords) and t < len(cords) and present(cords[f]) and present(cords[t]):
            d.line([(int(cords[f][1]), int(cords[f][0])), (int(cords[t][1]), int(cords[t][0]))], fill=(255, 255, 255))
    for i, p in enumerate(cords):
        if present(p):
            d.ellipse([int(p[1]) - radius, int(p[0]) - radius, int(p[1]) + radius, int(p[0]) + radius],
                      fill=tuple(COLORS[i % len(COLORS)]))
    return np.asarray(im)


def map_to_cord(pose_map, pose_dim, threshold=0.1):
    """(H, W, P) heat-maps -> (P, 2) integer (y, x) peaks, -1 where the map never exceeds the threshold
    (reference pose_utils.py:57-76; ties: first maximum in row-major order)."""
    pose_map = np.asarray(pose_map)[..., :pose_dim]
    out = np.full((pose_dim, 2), MISSING_VALUE, dtype=np.int64)
    flat = pose_map.reshape(-1, pose_dim)
    idx = flat.argmax(axis=0)
    for i in range(pose_dim):
        if flat[idx[i], i] > threshold:
            out[i] = (idx[i] // pose_map.shape[1], idx[i] % pose_map.shape[1])
    return out


def display(input_batch, target_batch, output_batch, use_input_pose, pose_dim):
    """Image grid [input | target pose | target | generated], one row per sample (reference pose_utils.py:235-255).
    Tensors are NCHW in [-1, 1] on any device; returns an (N*H, 4*W, 3) uint8 array."""
    n = input_batch.shape[0]
    img, _, tg_pose = get_imgpose(input_batch.detach().cpu(), use_input_pose, pose_dim)
    hwc = lambda x: _deprocess_image(x.detach().cpu()).permute(0, 2, 3, 1).numpy()
    poses = np.stack([draw_pose_from_cords(map_to_cord(p.permute(1, 2, 0).numpy(), pose_dim), pose_dim, p.shape[1:])
                      for p in tg_pose])
    cols = [make_grid(hwc(img), n, 1), make_grid(poses, n, 1), make_grid(hwc(target_batch), n, 1),
            make_grid(hwc(output_batch), n, 1)]
    return np.concatenate(cols, axis=1)


def display_stacked(input_batch, interpol_batch, target_batch, output_batch, num_stacks, use_input_pose, pose_dim):
    """[input | num_stacks interpolated poses | target | num_stacks stage outputs] (reference pose_utils.py:258-286)."""
    n = input_batch.shape[0]
    img, _, _ = get_imgpose(input_batch.detach().cpu(), use_input_pose, pose_dim)
    hwc = lambda x: _deprocess_image(x.detach().cpu()).permute(0, 2, 3, 1).numpy()
    ip = interpol_batch.detach().cpu()
    poses, outs = [], []
    for s in range(num_stacks):
        ps = ip[:, s * pose_dim:(s + 1) * pose_dim]
        poses.append(make_grid(np.stack([draw_pose_from_cords(map_to_cord(p.permute(1, 2, 0).numpy(), pose_dim), pose_dim,
                                                              p.shape[1:]) for p in ps]), n, 1))
        outs.append(make_grid(hwc(output_batch[s]), n, 1))
    return np.concatenate([make_grid(hwc(img), n, 1)] + poses + [make_grid(hwc(target_batch), n, 1)] + outs, axis=1)
