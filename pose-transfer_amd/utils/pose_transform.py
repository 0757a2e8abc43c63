"""Deformable skip layer with the reference's module surface
(reference src_deformable/utils/pose_transform.py:60-92): ``AffineTransformLayer(T, init_image_size,
warp_skip).forward(feat, warps, masks)`` on NCHW tensors, executed by the fused HIP kernels
(pg_mask_pyramid + pg_warp_mask_max_fwd/bwd).  Inside the generator engine the same kernels run on the
engine's native NHWC buffers without the layout conversions done here."""
import numpy as np
import torch
import torch.nn as nn

from ..runtime import lib as L


class _WarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, warps, masks, init_size, align):
        n, c, h, w = feat.shape
        t = warps.shape[1]
        dev = feat.device
        x = torch.empty(n, h, w, c, dtype=torch.float32, device=dev)
        L.call("pg_nchw_to_nhwc", L.ptr(feat.contiguous()), L.ptr(x), n, c, h, w, L.stream())
        lvl = torch.empty(n, h, w, t, dtype=torch.float32, device=dev)
        masks = masks.contiguous()
        L.call("pg_mask_pyramid", L.ptr(masks), 1 if masks.dtype == torch.float64 else 0, n, t, init_size[0],
               init_size[1], h, w, L.ptr(lvl), L.stream())
        wr = warps.to(torch.float32).contiguous()
        out = torch.empty_like(x)
        arg = torch.empty(n, h, w, c, dtype=torch.uint8, device=dev)
        L.call("pg_warp_mask_max_fwd", L.ptr(x), None, L.ptr(wr), L.ptr(lvl), n, t, c, h, w, init_size[0], init_size[1],
               align, L.ptr(out), L.ptr(arg), L.stream())
        res = torch.empty(n, c, h, w, dtype=torch.float32, device=dev)
        L.call("pg_nhwc_to_nchw", L.ptr(out), L.ptr(res), n, c, h, w, L.stream())
        ctx.save_for_backward(arg, wr, lvl)
        ctx.meta = (n, t, c, h, w, init_size, align)
        return res

    @staticmethod
    def backward(ctx, gout):
        arg, wr, lvl = ctx.saved_tensors
        n, t, c, h, w, init_size, align = ctx.meta
        g = torch.empty(n, h, w, c, dtype=torch.float32, device=gout.device)
        L.call("pg_nchw_to_nhwc", L.ptr(gout.contiguous()), L.ptr(g), n, c, h, w, L.stream())
        d = torch.zeros(n, h, w, c, dtype=torch.float32, device=gout.device)
        L.call("pg_warp_mask_max_bwd", L.ptr(g), L.ptr(arg), L.ptr(wr), L.ptr(lvl), n, t, c, h, w, init_size[0],
               init_size[1], align, L.ptr(d), L.stream())
        gin = torch.empty(n, c, h, w, dtype=torch.float32, device=gout.device)
        L.call("pg_nhwc_to_nchw", L.ptr(d), L.ptr(gin), n, c, h, w, L.stream())
        return gin, None, None, None, None


class AffineTransformLayer(nn.Module):
    def __init__(self, number_of_transforms, init_image_size, warp_skip, align_corners=False):
        super().__init__()
        if warp_skip != "mask":
            raise Exception("only warp_skip='mask' is implemented")
        self.number_of_transforms = number_of_transforms
        self.init_image_size = tuple(init_image_size)
        self.warp_skip = warp_skip
        self.align_corners = 1 if align_corners else 0

    def forward(self, input, warps, masks):
        assert warps.shape[1] == self.number_of_transforms
        return _WarpFn.apply(input, warps, masks, self.init_image_size, self.align_corners)


# ----------------------------------------------------------------------------------------------------------------------
# Key-point geometry on the device (reference utils/pose_transform.py:143-326)
def _kp_tensor(kp, device):
    k = torch.as_tensor(np.asarray(kp) if not torch.is_tensor(kp) else kp)
    if k.dim() == 2:
        k = k.unsqueeze(0)
    assert k.dim() == 3 and k.shape[-1] == 2, "key-points must be (N, P, 2) as (y, x)"
    return k.to(device=device, dtype=torch.float32).contiguous()


_TORSO = {16: (2, 3, 12, 13), 18: (8, 11, 2, 5)}       # Rhip, Lhip, Rsho, Lsho in LABELS / LABELS_PAF


def _check_torso(kp, pose_dim):
    """compute_st_distance (reference pose_transform.py:119-122) raises KeyError without the four torso joints."""
    k = np.asarray(kp.cpu() if torch.is_tensor(kp) else kp).reshape(-1, pose_dim, 2)
    bad = (k[:, list(_TORSO[pose_dim])] == -1).any(axis=(1, 2))
    if bad.any():
        raise KeyError("sample %d lacks a hip / shoulder key-point (reference: KeyError in compute_st_distance)"
                       % int(np.nonzero(bad)[0][0]))


def affine_transforms(array1, array2, pose_dim, device="cuda", out=None, check=True):
    """Batched affine_transforms (reference pose_transform.py:213-289): key-points (N, P, 2) [or (P, 2)] as (y, x),
    -1 = missing -> (N, 10, 8) float32 inverse maps target -> source, computed by pg_affine_transforms."""
    if check:
        _check_torso(array1, pose_dim), _check_torso(array2, pose_dim)
    k1, k2 = _kp_tensor(array1, device), _kp_tensor(array2, device)
    n = k1.shape[0]
    if out is None:
        out = torch.empty(n, 10, 8, dtype=torch.float32, device=k1.device)
    assert out.is_contiguous() and tuple(out.shape) == (n, 10, 8)
    L.call("pg_affine_transforms", L.ptr(k1), L.ptr(k2), n, pose_dim, L.ptr(out), L.stream())
    return out


def pose_masks(array2, img_size, pose_dim, device="cuda", out=None, check=True):
    """Batched pose_masks (reference pose_transform.py:143-184) -> (N, 10, H, W) float32 in {0, 1} (pg_pose_masks)."""
    if check:
        _check_torso(array2, pose_dim)
    k2 = _kp_tensor(array2, device)
    n, (h, w) = k2.shape[0], img_size
    if out is None:
        out = torch.empty(n, 10, h, w, dtype=torch.float32, device=k2.device)
    assert out.is_contiguous() and tuple(out.shape) == (n, 10, h, w)
    L.call("pg_pose_masks", L.ptr(k2), n, pose_dim, h, w, L.ptr(out), L.stream())
    return out


def estimate_uniform_transform(array1, array2, pose_dim, device="cuda", check=True):
    """warp_skip='full' (reference pose_transform.py:293-326): ONE fit over the torso joints plus the knees present in
    both poses -> (N, 1, 8) float32 (the reference hands over 9 numbers per row when the fit is invertible and 8
    otherwise; only the first six are ever read, pose_transform.py:28)."""
    if check:
        _check_torso(array1, pose_dim), _check_torso(array2, pose_dim)
    k1, k2 = _kp_tensor(array1, device), _kp_tensor(array2, device)
    out = torch.empty(k1.shape[0], 1, 8, dtype=torch.float32, device=k1.device)
    L.call("pg_uniform_transform", L.ptr(k1), L.ptr(k2), k1.shape[0], pose_dim, L.ptr(out), L.stream())
    return out
