"""First layers at the bench shapes: forward patch kernel, and the all-taps LDS-patch weight gradient vs the generic
per-tap kernel.   usage: python tools/small_cin_bench.py  (env PG_SCW_BLOCKS overrides the persistent grid)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from gpu_util import ConvCase, E, L, DEV


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, srcs, K, s, p in (("enc k3s1 c21", [(21, False, False)], 3, 1, 1), ("enc k3s1 c18", [(18, False, False)], 3, 1, 1),
                            ("stem k4s2 c42", [(3, False, False), (18, False, False), (3, False, False), (18, False, False)], 4, 2, 0)):
    case = ConvCase(name, "conv", srcs, 64, 4, 256, 256, K, s, p, L.ACT_NONE, bias=True, scalar=True)
    acts = case.device_sources()
    gy = torch.randn(4, case.Ho, case.Wo, 64, device=DEV)
    dW = torch.zeros(K, K, 64, case.cin, device=DEV)
    fl = 2.0 * 4 * case.Ho * case.Wo * K * K * case.cin * 64

    def run():
        E._wgrad([a.src() for a in acts], 4, L.ACT_NONE, gy, 64, case.cin, True, case.Ho, case.Wo, 256, 256, K, s, p, dW,
                 scalar_x=True)
    E.SMALL_CIN_WGRAD = True
    t1 = timeit(run)
    E.SMALL_CIN_WGRAD = False
    t0 = timeit(run)
    wp = case.packed_weight()
    wt = torch.empty(case.cin * K * K * 64, device=DEV)
    out = torch.empty(4, case.Ho, case.Wo, 64, device=DEV)
    bd = case.b.to(DEV)
    tf = timeit(lambda: E._small_cin_conv(acts, 4, 256, 256, K, s, p, wp, bd, wt, out))
    print("%-14s forward (repack + patch kernel) %7.1f us (%5.1f TF)" % (name, tf, fl / tf * 1e-6))
    print("%-14s patch kernel %7.1f us (%5.1f TF)   generic %7.1f us (%5.1f TF)" % (name, t1, fl / t1 * 1e-6, t0, fl / t0 * 1e-6))
