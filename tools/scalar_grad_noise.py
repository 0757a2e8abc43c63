"""How well-determined are the scalar norm gamma / beta gradients at 256 x 256?  
Runs the CPU oracle (oracle/ref_cpu.py) on the inputs of tests/test_gpu_round5.py::_step_256 in float32 and in float64 and prints,
per scalar gradient, |g32 - g64| / max(|g64|, 0.1 median |scalar gradient|) — the metric of the GPU tests.  CPU only.
    python tools/scalar_grad_noise.py [seed] [l1|nn]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pta_bootstrap; pta_bootstrap.load()
from pose_transfer_amd.utils import synth
import ref_cpu as R

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 93
name = sys.argv[2] if len(sys.argv) > 2 else "l1"
P, H, W, N = 18, 256, 256, 2
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
enc, dec = synth.nfilters((H, W))
gpar = {k: t(v) for k, v in synth.init_params(seed, "g256/%s/gen" % name, synth.generator_spec(P, enc, dec), 0.1).items()}
dpar = {k: t(v) for k, v in synth.init_params(seed, "g256/%s/disc" % name, synth.discriminator_spec(3 + 2 * P + 3), 0.1).items()}
hb = [[t(a) for a in synth.batch(seed, "g256/%s/%s" % (name, s), N, P, H, W)] for s in "ABC"]
hd = [[t(m) for m in synth.dropout_masks(seed, "g256/%s/d%s" % (name, s), N)] for s in "AC"]
cfg = dict(pose_dim=P, image_size=(H, W), batch_size=N, gan_penalty_weight=1.0, l1_penalty_weight=100.0, learning_rate=2e-4,
           content_loss_layer="none", nn_loss_area_size=1, nfilters_enc=enc, nfilters_dec=dec)
res = {}
for dt in (torch.float32, torch.float64):
    c = lambda x: x.to(dt) if x.is_floating_point() else x
    tr = R.Trainer(cfg, {k: c(v) for k, v in gpar.items()}, {k: c(v) for k, v in dpar.items()}, None)
    tr.dis_update(c(hb[0][0]), c(hb[0][1]), c(hb[0][2]), c(hb[0][3]), c(hb[1][0]), c(hb[1][1]), [c(m) for m in hd[0]])
    dg = {("d/" + k): v.double() for k, v in tr.last_disc_grads.items()}
    tr.gen_update(c(hb[2][0]), c(hb[2][1]), c(hb[2][2]), c(hb[2][3]), [c(m) for m in hd[1]])
    dg.update({("g/" + k): v.double() for k, v in tr.last_gen_grads.items()})
    res[dt] = dg
g32, g64 = res[torch.float32], res[torch.float64]
for pre in ("d/", "g/"):
    scal = [abs(float(v)) for k, v in g64.items() if k.startswith(pre) and v.numel() == 1]
    floor = 0.1 * float(np.median(scal))
    worst = []
    for k, v in g64.items():
        if not k.startswith(pre):
            continue
        if v.numel() == 1:
            worst.append((abs(float(g32[k]) - float(v)) / max(abs(float(v)), floor, 1e-12), k, float(v), float(g32[k])))
    worst.sort(reverse=True)
    print("seed %d %s %s: %d scalar gradients, floor %.3e; float32 oracle vs float64 oracle, worst five:" % (seed, name, pre, len(worst), floor))
    for w in worst[:5]:
        print("   %.4f  %-40s f64 %+.5e  f32 %+.5e" % w)
    tw = max(float((g32[k] - v).abs().max() / v.abs().max()) for k, v in g64.items() if k.startswith(pre) and v.numel() > 1)
    print("   tensors: worst max-abs difference / tensor max = %.2e" % tw)
