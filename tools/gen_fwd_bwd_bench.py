"""North-star sub-metric (BASELINE.json): generator forward + backward at 256x256, batch 32 — 3 * F_G * 32 = 13.25 TFLOP.
    gpurun -- python tools/gen_fwd_bwd_bench.py [batch]
Prints time, TFLOP/s and the fraction of the fp32 / bf16 MFMA peaks for the fp32 path and the bf16 data path."""
import os, sys
from types import SimpleNamespace
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pose_transfer_amd.models.pose_gan import DeformablePose_GAN
from pose_transfer_amd.runtime import engine as E
from pose_transfer_amd.utils import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
args = SimpleNamespace(size=256, batch=N, content_loss_layer="none", nn_loss_area_size=1, l1_penalty_weight=100.0)
opt = bench.make_opt(args)
_, fg, _ = bench.step_flops(256, 18)
flops = 3.0 * fg * N
inp, tgt, wr, mk = [torch.from_numpy(a).cuda() for a in synth.batch(1234, "nsub", N, 18, 256, 256)]
gout = torch.randn(N, 3, 256, 256, device="cuda")
MODES = ((0, 157.3, "fp32"), (3, 2500.0, "bf16 data path"))
if os.environ.get("PG_ONLY_BF16"):
    MODES = MODES[1:]
for prec, peak, tag in MODES:
    E.PRECISION = prec
    model = DeformablePose_GAN(opt, device="cuda:0", init_seed=0)
    eng = model.gen.engine(N)
    eng.set_dropout(None, train=True, seed=0)

    def step():
        model.gen.arena.grads.zero_()
        eng.forward(inp, wr, mk)
        eng.backward(gout)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ITERS = int(os.environ.get('PG_NS_ITERS', '10'))
    for _ in range(ITERS):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / ITERS
    tf = flops / ms / 1e9
    print("generator fwd+bwd, 256x256, batch %d, %s: %.2f ms = %.1f TFLOP/s = %.3f of the %s MFMA peak (%.0f TFLOP/s)"
          % (N, tag, ms, tf, tf / peak, "fp32" if prec == 0 else "bf16", peak), flush=True)
    del model, eng
    torch.cuda.empty_cache()
