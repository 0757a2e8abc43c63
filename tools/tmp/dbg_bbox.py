import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, torch
from gpu_util import DEV, L, nhwc, synth, t
for io in (0, 3):
    N, C, H0, W0, s = 2, 8, 48, 40, 2
    wr, mk = synth.warps_and_masks(33, "x", N, H0, W0)
    h, w = H0 // s, W0 // s
    mkd, wrd = t(mk).to(DEV), t(wr).to(DEV)
    lvl = torch.empty(N, h, w, 10, device=DEV)
    L.call("pg_mask_pyramid", L.ptr(mkd), 0, N, 10, H0, W0, h, w, L.ptr(lvl), L.stream())
    dt = torch.bfloat16 if io else torch.float32
    feat = nhwc(t(synth.normal(31, "f", (N, C, h, w)))).to(DEV).to(dt).contiguous()
    go = nhwc(t(synth.normal(31, "go", (N, C, h, w)))).to(DEV).to(dt).contiguous()
    out = torch.empty(N, h, w, C, device=DEV, dtype=dt)
    arg = torch.empty(N, h, w, C, dtype=torch.uint8, device=DEV)
    L.call("pg_warp_mask_max_fwd_io", L.ptr(feat), None, L.ptr(wrd), L.ptr(lvl), N, 10, C, h, w, H0, W0, 0, L.ptr(out), L.ptr(arg), io, L.stream())
    print(io, "feat", float(feat.float().abs().max()), "out", float(out.float().abs().max()), "arg", arg.unique().tolist(), "go", float(go.float().abs().max()))
    d = torch.zeros(N, h, w, C, device=DEV, dtype=dt)
    L.call("pg_warp_mask_max_bwd_bbox", L.ptr(go), L.ptr(arg), L.ptr(wrd), L.ptr(lvl), None, N, 10, C, h, w, H0, W0, 0, L.ptr(d), io, L.stream())
    print("d", float(d.float().abs().max()))
    d2 = torch.zeros(N, h, w, C, device=DEV, dtype=dt)
    L.call("pg_warp_mask_max_bwd_io", L.ptr(go), L.ptr(arg), L.ptr(wrd), L.ptr(lvl), N, 10, C, h, w, H0, W0, 0, L.ptr(d2), io, L.stream())
    print("d2", float(d2.float().abs().max()))
