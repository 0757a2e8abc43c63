"""Helpers for the GPU parity tests: torch-CPU references for single kernels and a generic conv harness."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_cpu as R  # noqa: E402  (oracle: test infrastructure only)
from pose_transfer_amd.runtime import engine as E  # noqa: E402
from pose_transfer_amd.runtime import lib as L  # noqa: E402
from pose_transfer_amd.utils import synth  # noqa: E402

DEV = "cuda"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def act_fn(z, act):
    if act == L.ACT_RELU:
        return F.relu(z)
    if act == L.ACT_LEAKY:
        return F.leaky_relu(z, 0.2)
    return z


def maxdiff(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


class ConvCase:
    """One convolution (conv or conv-transpose+crop) with a virtual-concat, prologue-carrying input.

    kind: 'conv' (nn.Conv2d, weight (Cout,Cin,K,K)) or 'convT' (nn.ConvTranspose2d k4 s2 + crop 1, weight (Cin,Cout,4,4)).
    Sources are given as (C, has_aff, has_mask); `scalar` = sources are NCHW slices (small C, no prologue)."""

    def __init__(self, name, kind, srcs, cout, N, H, W, K, stride, pad, act, bias=False, tanh=False, scalar=False,
                 nchw_out=False, seed=7):
        self.name, self.kind, self.srcs, self.cout = name, kind, srcs, cout
        self.N, self.H, self.W, self.K, self.stride, self.pad, self.act = N, H, W, K, stride, pad, act
        self.bias, self.tanh, self.scalar, self.nchw_out = bias, tanh, scalar, nchw_out
        self.cin = sum(s[0] for s in srcs)
        g = lambda tag, shape: t(synth.normal(seed, name + "/" + tag, shape))
        self.raw = [g("raw%d" % j, (N, s[0], H, W)) for j, s in enumerate(srcs)]           # NCHW on the host
        self.aff = [t(np.stack([synth.uniform(seed, name + "/a%d" % j, (N,), 0.5, 1.5),
                                synth.uniform(seed, name + "/b%d" % j, (N,), -0.5, 0.5)], 1)) if s[1] else None
                    for j, s in enumerate(srcs)]
        self.mask = [t(synth.dropout_masks(seed, name + "/m%d" % j, N, (s[0],))[0]) if s[2] else None
                     for j, s in enumerate(srcs)]
        wshape = (cout, self.cin, K, K) if kind == "conv" else (self.cin, cout, K, K)
        self.w = t(synth.xavier_uniform(seed, name + "/w", wshape))
        self.b = t(synth.uniform(seed, name + "/bias", (cout,), -0.3, 0.3)) if bias else None
        if kind == "conv":
            self.Ho, self.Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
        else:
            self.Ho, self.Wo = 2 * H, 2 * W
        self.gout = g("gout", (N, cout, self.Ho, self.Wo))

    # ---------------------------------------------------------------- torch CPU reference
    def reference(self):
        zs = []
        for j in range(len(self.srcs)):
            z = self.raw[j]
            if self.aff[j] is not None:
                z = z * self.aff[j][:, 0].view(-1, 1, 1, 1) + self.aff[j][:, 1].view(-1, 1, 1, 1)
            zs.append(z.detach().clone().requires_grad_(True))
        xs = []
        for j, z in enumerate(zs):
            v = z
            if self.mask[j] is not None:
                v = v * self.mask[j].view(self.N, -1, 1, 1)
            xs.append(act_fn(v, self.act))
        x = torch.cat(xs, 1)
        w = self.w.clone().requires_grad_(True)
        b = self.b.clone().requires_grad_(True) if self.b is not None else None
        if self.kind == "conv":
            y = F.conv2d(x, w, b, stride=self.stride, padding=self.pad)
        else:
            y = F.conv_transpose2d(x, w, None, stride=2)[:, :, 1:-1, 1:-1]
        pre = y
        out = torch.tanh(y) if self.tanh else y
        grads = torch.autograd.grad((pre * self.gout).sum(), zs + [w] + ([b] if b is not None else []))
        return out.detach(), [g for g in grads[:len(zs)]], grads[len(zs)], (grads[-1] if b is not None else None)

    # ---------------------------------------------------------------- device side
    def device_sources(self):
        acts = []
        self._keep = []
        if self.scalar:
            big = torch.cat(self.raw, 1).to(DEV).contiguous()          # one NCHW tensor, sources = channel slices
            self._keep.append(big)
            ctot, hw = big.shape[1], self.H * self.W
            c0 = 0
            for j, s in enumerate(self.srcs):
                acts.append(E.Act(big, s[0], strides=(ctot * hw, hw, self.W, 1), base_ptr=big.data_ptr() + 4 * c0 * hw))
                c0 += s[0]
        else:
            for j, s in enumerate(self.srcs):
                r = nhwc(self.raw[j]).to(DEV)
                a = self.aff[j].to(DEV).contiguous() if self.aff[j] is not None else None
                m = self.mask[j].to(DEV).contiguous() if self.mask[j] is not None else None
                self._keep += [r, a, m]
                acts.append(E.Act(r, s[0], aff=a, mask=m))
        return acts

    def packed_weight(self):
        key = "decoder.net.0.net.1.weight" if self.kind == "convT" else "w"
        return E._pack(key, self.w).to(DEV)

    def run_forward(self, ksplit=0, stats=None):
        acts = self.device_sources()
        wp = self.packed_weight()
        N = self.N
        if self.nchw_out:
            out = torch.full((N, self.cout, self.Ho, self.Wo), float("nan"), device=DEV)
            ostr = (self.cout * self.Ho * self.Wo, self.Ho * self.Wo, self.Wo, 1)
        else:
            out = torch.full((N, self.Ho, self.Wo, self.cout), float("nan"), device=DEV)
            ostr = None
        b = self.b.to(DEV) if self.b is not None else None
        mode = 0 if self.kind == "conv" else 1
        E._conv([a.src() for a in acts], N, self.H, self.W, self.act, mode, self.K, self.stride, self.pad, self.Ho,
                self.Wo, wp, self.cout, self.cin, scalar_in=self.scalar, out=out, out_strides=ostr, bias=b,
                out_act=L.OUT_TANH if self.tanh else L.OUT_NONE, ksplit=ksplit, stats=stats)
        torch.cuda.synchronize()
        return out.cpu() if self.nchw_out else nchw(out.cpu())

    def run_dgrad(self, ksplit=0, accumulate=False):
        """data-gradient wrt every source's post-norm value (epilogue 1)."""
        acts = self.device_sources()
        wp = self.packed_weight()
        N = self.N
        gy = nhwc(self.gout).to(DEV)
        grads = [torch.full((N, self.H, self.W, s[0]), 0.5 if accumulate else float("nan"), device=DEV) for s in self.srcs]
        dsts = [L.make_dst(grads[j], a.C, fwd=a.t, aff=a.aff, mask=a.mask, act=self.act, accumulate=accumulate)
                for j, a in enumerate(acts)]
        mode = 1 if self.kind == "conv" else 0          # the data-gradient runs the opposite geometry
        E._conv_dgrad(E.Act(gy, self.cout).src(), N, self.Ho, self.Wo, mode, self.K, self.stride, self.pad,
                      self.H, self.W, wp, self.cout, self.cin, dsts, ksplit=ksplit)
        torch.cuda.synchronize()
        return [nchw(g.cpu()) - (0.5 if accumulate else 0.0) for g in grads]

    def run_wgrad(self, ksplit=0, scalar_y=False):
        acts = self.device_sources()
        N = self.N
        dW = torch.zeros(self.K, self.K, self.cout, self.cin, device=DEV)
        conv = self.kind == "conv"
        Hs, Ws, Hl, Wl = (self.Ho, self.Wo, self.H, self.W) if conv else (self.H, self.W, self.Ho, self.Wo)
        if scalar_y:
            gy = self.gout.to(DEV).contiguous()
            ystr = (self.cout * self.Ho * self.Wo, self.Ho * self.Wo, self.Wo, 1)
        else:
            gy = nhwc(self.gout).to(DEV)
            ystr = None
        E._wgrad([a.src() for a in acts], N, self.act, gy, self.cout, self.cin, conv, Hs, Ws, Hl, Wl, self.K, self.stride,
                 self.pad, dW, scalar_x=self.scalar, y_strides=ystr, ksplit=ksplit)
        torch.cuda.synchronize()
        key = "decoder.net.0.net.1.weight" if self.kind == "convT" else "w"
        return E._unpack(key, dW).cpu()
