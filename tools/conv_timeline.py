"""Per-workgroup phase timeline of the 256-row bf16 convolution kernel on a generator layer at batch N (bf16 STORAGE).
    PG_TIMING_EXPERIMENTS=1 python -m pose_transfer_amd.runtime.build   # here: builds lib/libposegan_hip_timing.so (-DPG_TIMING_EXPERIMENTS)
    gpurun -- env PG_TIMING_EXPERIMENTS=1 PG_DEBUG_CONV_TIMELINE=1 python tools/conv_timeline.py 32 dec5 fwd
(the production library carries neither the stamps nor the K-loop experiments; with PG_TIMING_EXPERIMENTS=1 in the environment
runtime/build.py and runtime/lib.py select the timing library)
Prints, per phase, the median / mean microseconds a workgroup spends (shader clock calibrated against the 100 MHz wall
clock), and the launch's duration."""
import ctypes
import os
import sys

import numpy as np
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
    "dec3": ("convT", 32, 32, [512, 512, 512], 512), "dec4": ("convT", 64, 64, [512, 256, 256], 256),
    "dec5": ("convT", 128, 128, [256, 128, 128], 128),
}


def main():
    N, name, what = int(sys.argv[1]), sys.argv[2], sys.argv[3]
    kind, h, w, srcC, cout = LAYERS[name]
    E.PRECISION = 3
    K, stride, pad = 4, 2, 1
    cin = sum(srcC)
    ho, wo = (h // 2, w // 2) if kind == "conv" else (2 * h, 2 * w)
    bf = torch.bfloat16
    srcs = [E._reg_bf16(torch.randn(N, h, w, c, device=DEV).to(bf)) for c in srcC]
    acts = [E.Act(s, c) for s, c in zip(srcs, srcC)]
    W = torch.randn(K, K, cout, cin, device=DEV) * 0.05
    out = E._reg_bf16(torch.empty(N, ho, wo, cout, device=DEV, dtype=bf))
    gy = E._reg_bf16(torch.randn(N, ho, wo, cout, device=DEV).to(bf))
    dz = [E._reg_bf16(torch.empty_like(s)) for s in srcs]
    stats = torch.zeros(N * 64, dtype=torch.float64, device=DEV)
    act = L.ACT_LEAKY if kind == "conv" else L.ACT_RELU
    mode_f = 0 if kind == "conv" else 1
    flops = 2.0 * N * min(h * w, ho * wo) * K * K * cin * cout

    def fwd():
        E._conv([a.src() for a in acts], N, h, w, act, mode_f, K, stride, pad, ho, wo, W, cout, cin, out=out, stats=stats)

    def dgrad():
        dsts = [L.make_dst(d, a.C, fwd=a.t, aff=None, act=act) for d, a in zip(dz, acts)]
        E._conv_dgrad(E.Act(gy, cout).src(), N, ho, wo, 1 - mode_f, K, stride, pad, h, w, W, cout, cin, dsts)

    fn = fwd if what == "fwd" else dgrad
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    print("%s %s batch %d: call %.1f us (incl. operand materialisation), %.1f GFLOP, launch info %s" % (name, what, N, e0.elapsed_time(e1) * 1e3, flops / 1e9, hex(L.load().pg_last_conv_info()) if hasattr(L.load(), "pg_last_conv_info") else "-"))
    nw = 16384
    buf = np.zeros((nw, 16), dtype=np.uint64)
    code = L.load().pg_last_launch_info() & 0xF
    if code in (13, 14):        # the tap-quad kernel (igemm_bf16_quad.hip) keeps its own stamps
        fnq = L.load().pg_debug_conv_timeline_quad
        fnq.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        fnq.restype = ctypes.c_int
        L.check(fnq(buf.ctypes.data_as(ctypes.c_void_p), nw), "timeline (quad)")
    else:
        L.check(L.load().pg_debug_conv_timeline(buf.ctypes.data_as(ctypes.c_void_p), nw), "timeline")
    print("kernel code", code)
    t = buf.astype(np.int64)
    live = t[:, 4] > t[:, 0]
    t = t[live]
    print("workgroups stamped:", len(t))
    wall = (t[:, 6].max() - t[:, 5].min()) / 100.0
    print("launch wall (first start -> last end): %.1f us" % wall)
    cyc = (t[:, 4] - t[:, 0]).astype(np.float64)
    us = (t[:, 6] - t[:, 5]) / 100.0
    mhz = np.median(cyc[us > 5] / us[us > 5])
    print("shader clock ~ %.0f MHz" % mhz)
    names = ["row table", "first tile", "K loop", "epilogue"]
    for i, nm in enumerate(names):
        d = (t[:, i + 1] - t[:, i]) / mhz
        print("  %-10s median %7.2f us  mean %7.2f  p90 %7.2f" % (nm, np.median(d), d.mean(), np.percentile(d, 90)))
    if what != "fwd":
        pass
    elif (t[:, 8] > t[:, 3]).all():
        print("  epilogue split: stores issued + wave sums %.2f us | drain + barrier %.2f | merge + atomics + end %.2f" % (
            np.median(t[:, 8] - t[:, 3]) / mhz, np.median(t[:, 9] - t[:, 8]) / mhz, np.median(t[:, 4] - t[:, 9]) / mhz))
    if what == "fwd" and (t[:, 15] > t[:, 14]).all():
        print("  first 64-row call (stores + wave sums + LDS atomics): %.2f us" % (np.median(t[:, 15] - t[:, 14]) / mhz))
    if code in (13, 14) and os.environ.get("PG_TL_DISPATCH"):
        # dispatch pattern: which workgroups (linear launch index) shared a CU, in start order (HW_ID bits 8..15 + XCC id)
        ids = np.nonzero(live)[0]
        key = ((t[:, 10] >> 8) & 0xff) | (t[:, 7] << 8)
        order = np.argsort(t[:, 5], kind="stable")
        per = {}
        for o in order:
            per.setdefault(int(key[o]), []).append((int(ids[o]), float((t[o, 5] - t[:, 5].min()) / 100.0), float((t[o, 6] - t[:, 5].min()) / 100.0)))
        print("  distinct CU keys:", len(per))
        for kk in sorted(per)[:6]:
            print("  cu %04x: %s" % (kk, " ".join("%d[%.0f-%.0f]" % e for e in per[kk][:10])))
    tot = (t[:, 4] - t[:, 0]) / mhz
    print("  %-10s median %7.2f us  mean %7.2f ; sum over WGs / 256 CUs = %.1f us" % ("total", np.median(tot), tot.mean(), tot.sum() / 256))
    # per-XCD workgroup counts and the idle tail
    for x in range(8):
        m = t[:, 7] == x
        if m.any():
            print("  xcd %d: %5d WGs, last end %.1f us" % (x, m.sum(), (t[m, 6].max() - t[:, 5].min()) / 100.0))


if __name__ == "__main__":
    main()
