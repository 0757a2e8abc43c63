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
