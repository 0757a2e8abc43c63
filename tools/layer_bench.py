"""Per-layer timing of the generator's convolutions at batch N on the bf16 data path (bf16 STORAGE), operands pre-materialised:
    gpurun -- python tools/layer_bench.py 32 [layer ...]           # default: every layer, fwd / dgrad / wgrad
Prints microseconds and TFLOP/s of each contraction launch alone (HIP events around 5 repeats, best of 3)."""
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
LAYERS = {
    "enc1": ("conv", 256, 256, [64], 128), "enc2": ("conv", 128, 128, [128], 256), "enc3": ("conv", 64, 64, [256], 512),
    "enc4": ("conv", 32, 32, [512], 512), "enc5": ("conv", 16, 16, [512], 512),
    "dec1": ("convT", 8, 8, [512, 512, 512], 512), "dec2": ("convT", 16, 16, [512, 512, 512], 512),
    "dec3": ("convT", 32, 32, [512, 512, 512], 512), "dec4": ("convT", 64, 64, [512, 256, 256], 256),
    "dec5": ("convT", 128, 128, [256, 128, 128], 128),
}


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    names = sys.argv[2:] or list(LAYERS)
    F32 = os.environ.get("PG_LB_F32") == "1"      # the fp32 path (v_mfma_f32_32x32x2_f32 kernels, deferred-norm prologues)
    E.PRECISION = 0 if F32 else 3
    K, stride, pad = 4, 2, 1
    bf = torch.float32 if F32 else torch.bfloat16
    for name in names:
        kind, h, w, srcC, cout = LAYERS[name]
        cin = sum(srcC)
        ho, wo = (h // 2, w // 2) if kind == "conv" else (2 * h, 2 * w)
        zero = os.environ.get("PG_LB_ZERO") == "1"       # all-zero operands: same instruction stream, far less switching power
        rnd = (lambda *sh: torch.zeros(*sh, device=DEV)) if zero else (lambda *sh: torch.randn(*sh, device=DEV))
        srcs = [E._reg_bf16(rnd(N, h, w, c).to(bf)) for c in srcC]
        acts = [E.Act(s, c) for s, c in zip(srcs, srcC)]
        W = rnd(K, K, cout, cin) * 0.05
        dW = torch.zeros_like(W)
        out = E._reg_bf16(torch.empty(N, ho, wo, cout, device=DEV, dtype=bf))
        gy = E._reg_bf16(rnd(N, ho, wo, cout).to(bf))
        dz = [E._reg_bf16(torch.empty_like(s)) for s in srcs]
        stats = torch.zeros(N * 64, dtype=torch.float64, device=DEV)
        act = L.ACT_NONE       # the operands ARE the activated bf16 tensors: no materialisation pass inside the timed call
        mode_f = 0 if kind == "conv" else 1
        flops = 2.0 * N * min(h * w, ho * wo) * K * K * cin * cout
        cache = E.BfCache()
        E._BF_CTX = None if F32 else cache
        E._BF_CTX_X = None if F32 else cache

        def fwd():
            E._conv([a.src() for a in acts], N, h, w, act, mode_f, K, stride, pad, ho, wo, W, cout, cin, out=out, stats=stats)

        def dgrad():
            dsts = [L.make_dst(d, a.C, fwd=a.t, aff=None, act=L.ACT_RELU) for d, a in zip(dz, acts)]
            E._conv_dgrad(E.Act(gy, cout).src(), N, ho, wo, 1 - mode_f, K, stride, pad, h, w, W, cout, cin, dsts)

        def wgrad():
            hs, ws, hl, wl = (ho, wo, h, w) if kind == "conv" else (h, w, ho, wo)
            E._wgrad_main([a.src() for a in acts], N, act, gy, cout, cin, kind == "conv", hs, ws, hl, wl, K, stride, pad, dW)

        res = []
        for tag, fn in (("fwd", fwd), ("dgrad", dgrad), ("wgrad", wgrad)):
            try:
                us = timed(fn)
                info = L.load().pg_last_launch_info()
                res.append("%s %7.1f us %6.0f TF" % (tag, us, flops / us / 1e6))
            except Exception as ex:      # noqa: BLE001
                res.append("%s failed: %s" % (tag, str(ex)[:60]))
        print("%-5s N=%d  %s" % (name, N, " | ".join(res)), flush=True)
        E._BF_CTX = E._BF_CTX_X = None
        del srcs, acts, W, dW, out, gy, dz
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
