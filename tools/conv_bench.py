"""Micro-benchmark of the contraction kernels on the real layer shapes of the 256x256 generator (batch N).
    gpurun -- python tools/conv_bench.py [N]
Prints per-layer time and TFLOP/s (fp32 MFMA peak 157.3) for forward / data-gradient / weight-gradient."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pta_bootstrap  # noqa: E402

pta_bootstrap.load()
from pose_transfer_amd.runtime import engine as E  # noqa: E402
from pose_transfer_amd.runtime import lib as L  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def layer(name, kind, N, h, w, srcC, cout, K=4, stride=2, pad=1):
    """kind 'conv': input (h,w) -> (h/2,w/2); 'convT': input (h,w) -> (2h,2w).  srcC: list of source channel counts."""
    cin = sum(srcC)
    ho, wo = (h // stride, w // stride) if kind == "conv" else (2 * h, 2 * w)
    srcs = [torch.randn(N, h, w, c, device=DEV) for c in srcC]
    affs = [torch.rand(N, 2, device=DEV) + 0.5 for _ in srcC]
    acts = [E.Act(s, c, aff=a) for s, c, a in zip(srcs, srcC, affs)]
    W = torch.randn(K, K, cout, cin, device=DEV) * 0.05
    out = torch.empty(N, ho, wo, cout, device=DEV)
    gy = torch.randn(N, ho, wo, cout, device=DEV)
    dz = [torch.empty_like(s) for s in srcs]
    dW = torch.zeros_like(W)
    flops = 2.0 * N * min(h * w, ho * wo) * K * K * cin * cout
    act = L.ACT_LEAKY if kind == "conv" else L.ACT_RELU
    mode_f = 0 if kind == "conv" else 1

    def fwd():
        E._conv([a.src() for a in acts], N, h, w, act, mode_f, K, stride, pad, ho, wo, W, cout, cin, out=out)

    def dgrad():
        dsts = [L.make_dst(d, a.C, fwd=a.t, aff=a.aff, act=act) for d, a in zip(dz, acts)]
        E._conv_dgrad(E.Act(gy, cout).src(), N, ho, wo, 1 - mode_f, K, stride, pad, h, w, W, cout, cin, dsts)

    def wgrad():
        conv = kind == "conv"
        Hs, Ws, Hl, Wl = (ho, wo, h, w) if conv else (h, w, ho, wo)
        E._wgrad([a.src() for a in acts], N, act, gy, cout, cin, conv, Hs, Ws, Hl, Wl, K, stride, pad, dW)

    res = []
    for tag, fn in (("fwd", fwd), ("dgrad", dgrad), ("wgrad", wgrad)):
        ms = timeit(fn)
        res.append("%s %7.1f us %6.1f TF" % (tag, ms * 1e3, flops / ms / 1e9))
    print("%-8s %5.1f GF | %s" % (name, flops / 1e9, " | ".join(res)), flush=True)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
    layers = [
        ("enc1", "conv", 256, 256, [64], 128), ("enc2", "conv", 128, 128, [128], 256),
        ("enc3", "conv", 64, 64, [256], 512), ("enc4", "conv", 32, 32, [512], 512),
        ("enc5", "conv", 16, 16, [512], 512), ("enc6", "conv", 8, 8, [512], 512),
        ("dec0", "convT", 4, 4, [512, 512], 512), ("dec1", "convT", 8, 8, [512, 512, 512], 512),
        ("dec2", "convT", 16, 16, [512, 512, 512], 512), ("dec3", "convT", 32, 32, [512, 512, 512], 512),
        ("dec4", "convT", 64, 64, [512, 256, 256], 256), ("dec5", "convT", 128, 128, [256, 128, 128], 128),
    ]
    for name, kind, h, w, srcC, cout in layers:
        if only and name not in only:
            continue
        layer(name, kind, N, h, w, srcC, cout)


if __name__ == "__main__":
    main()
