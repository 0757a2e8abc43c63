"""Deformable skip layer with the reference's module surface
(reference src_deformable/utils/pose_transform.py:60-92): ``AffineTransformLayer(T, init_image_size,
warp_skip).forward(feat, warps, masks)`` on NCHW tensors, executed by the fused HIP kernels
(pg_mask_pyramid + pg_warp_mask_max_fwd/bwd).  Inside the generator engine the same kernels run on the
engine's native NHWC buffers without the layout conversions done here."""
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
