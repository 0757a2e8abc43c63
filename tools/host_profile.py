"""cProfile of the host side of the training iteration (Python + ctypes enqueue), top functions by own time.
    gpurun -- python tools/host_profile.py"""
import os, sys, cProfile, pstats, io
from types import SimpleNamespace
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pose_transfer_amd.models.pose_gan import DeformablePose_GAN
from pose_transfer_amd.utils import synth

from pose_transfer_amd.runtime import engine as E
E.PRECISION = {"f32": 0, "bf16": 1, "bf16x3": 2, "bf16_data": 3}[sys.argv[1] if len(sys.argv) > 1 else "f32"]
args = SimpleNamespace(size=256, batch=4, content_loss_layer="none", nn_loss_area_size=1, l1_penalty_weight=100.0)
opt = bench.make_opt(args)
model = DeformablePose_GAN(opt, device="cuda:0", init_seed=0)
od = dict(vars(opt), lazy_losses=True)
batches = [[torch.from_numpy(a).cuda() for a in synth.batch(1234, "bench/%s" % s, 4, 18, 256, 256)] for s in "ABC"]
for _ in range(3):
    bench.iteration(model, batches, od)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    bench.iteration(model, batches, od)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[:45]))
