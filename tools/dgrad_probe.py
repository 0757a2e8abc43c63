"""Probe: data-gradient time of one decoder/encoder layer for forced split-K values and for the plain (epilogue 0)
store of the same contraction.   gpurun -- python tools/dgrad_probe.py dec3"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import conv_bench as cb
from pose_transfer_amd.runtime import engine as E, lib as L
N = 4
SHAPES = {"dec3": ("convT", 32, 32, [512, 512, 512], 512), "dec4": ("convT", 64, 64, [512, 256, 256], 256),
          "dec5": ("convT", 128, 128, [256, 128, 128], 128), "enc1": ("conv", 256, 256, [64], 128),
          "enc3": ("conv", 64, 64, [256], 512), "dec2": ("convT", 16, 16, [512, 512, 512], 512)}
for name in sys.argv[1:]:
    kind, h, w, srcC, cout = SHAPES[name]
    cin = sum(srcC); K = 4
    ho, wo = (h // 2, w // 2) if kind == "conv" else (2 * h, 2 * w)
    srcs = [torch.randn(N, h, w, c, device="cuda") for c in srcC]
    acts = [E.Act(s, c, aff=torch.rand(N, 2, device="cuda") + 0.5) for s, c in zip(srcs, srcC)]
    W = torch.randn(K, K, cout, cin, device="cuda") * 0.05
    gy = torch.randn(N, ho, wo, cout, device="cuda")
    dz = [torch.empty_like(s) for s in srcs]
    dense = torch.empty(N, h, w, cin, device="cuda")
    act = L.ACT_LEAKY if kind == "conv" else L.ACT_RELU
    mode = 1 if kind == "conv" else 0
    flops = 2.0 * N * min(h * w, ho * wo) * K * K * cin * cout
    for ks in (0, 1, 2, 3, 4, 6):
        def f():
            dsts = [L.make_dst(d, a.C, fwd=a.t, aff=a.aff, act=act) for d, a in zip(dz, acts)]
            E._conv([E.Act(gy, cout).src()], N, ho, wo, L.ACT_NONE, mode, K, 2, 1, h, w, W, cout, cin, transposed=True,
                    dsts=dsts, ksplit=ks)
        def g():
            E._conv([E.Act(gy, cout).src()], N, ho, wo, L.ACT_NONE, mode, K, 2, 1, h, w, W, cout, cin, transposed=True,
                    out=dense, ksplit=ks)
        t1 = cb.timeit(f); i1 = L.load().pg_last_launch_info()
        t2 = cb.timeit(g); i2 = L.load().pg_last_launch_info()
        print("%s ks=%d | epilogue1 %7.1f us %6.1f TF (cfg %d ks %d) | epilogue0 %7.1f us %6.1f TF (cfg %d ks %d)" % (
            name, ks, t1 * 1e3, flops / t1 / 1e9, i1 & 15, i1 >> 16, t2 * 1e3, flops / t2 / 1e9, i2 & 15, i2 >> 16), flush=True)
