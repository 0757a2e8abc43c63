"""Weight-gradient kernels of the bf16 data path on the north-star layer shapes (batch 32, 256x256), each launch alone:
    gpurun -- python tools/wgrad_bf16_bench.py            (PG_WGTR4=0 for the one-tap kernel only)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pta_bootstrap
pta_bootstrap.load()
from pose_transfer_amd.runtime import lib as L

N = int(os.environ.get("PG_BENCH_N", "32"))
# (name, x_is_large, Cx, Cout, Hs, Ws)
LAYERS = [("enc1 64->128", 1, 64, 128, 128, 128), ("enc2 128->256", 1, 128, 256, 64, 64), ("enc3 256->512", 1, 256, 512, 32, 32),
          ("dec5 src256->128", 0, 256, 128, 128, 128), ("dec5 src128->128", 0, 128, 128, 128, 128),
          ("dec4 src512->256", 0, 512, 256, 64, 64), ("dec4 src256->256", 0, 256, 256, 64, 64),
          ("dec3 src512->512", 0, 512, 512, 32, 32)]
for name, xl, cx, co, hs, ws in LAYERS:
    hx, wx = (2 * hs, 2 * ws) if xl else (hs, ws)
    hy, wy = (hs, ws) if xl else (2 * hs, 2 * ws)
    x = torch.randn(N, hx, wx, cx, device="cuda").to(torch.bfloat16)
    dy = torch.randn(N, hy, wy, co, device="cuda").to(torch.bfloat16)
    dW = torch.zeros(16, co, cx, device="cuda")
    run = lambda: L.call("pg_wgrad_bf16", L.ptr(x), cx, L.ptr(dy), co, xl, N, hs, ws, L.ptr(dW), cx, 0, 0, L.stream())
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * N * hs * ws * 16 * cx * co
    info = L.load().pg_last_launch_info()
    print("%-20s %8.1f us  %7.1f TFLOP/s  ksplit %d" % (name, ms * 1e3, fl / ms * 1e-9, (info >> 16) & 0x3FFF), flush=True)
