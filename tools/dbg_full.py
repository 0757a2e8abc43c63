import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pta_bootstrap; pta_bootstrap.load()
import ref_cpu as R
from pose_transfer_amd.models.networks import Deformable_Generator
from pose_transfer_amd.utils import synth
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
P, size = 18, (64, 64)
enc, dec = synth.nfilters(size)
par = {k: t(v) for k, v in synth.init_params(71, "full/gen", synth.generator_spec(P, enc, dec), norm_jitter=0.2).items()}
inp, tgt, wr, mk = [t(a) for a in synth.batch(71, "full", 2, P, *size)]
drops = [t(m) for m in synth.dropout_masks(71, "full", 2)]
go = t(synth.normal(71, "full/go", (2, 3, 64, 64)))
W = wr[:, :1] if mode == "full" else wr
M = None if mode == "full" else mk
gen = Deformable_Generator(3 + 2 * P, P, size, enc, dec, mode)
gen.load_state_dict(par); gen.zero_grad()
out = gen(inp.cuda(), W.cuda(), None if M is None else M.cuda(), drop_masks=[d.cuda() for d in drops])
(out * go.cuda()).sum().backward()
def oracle(dt):
    pr = {k: v.to(dt).requires_grad_(True) for k, v in par.items()}
    o = R.generator_forward(inp.to(dt), W.to(dt), None if M is None else M.to(dt), pr, P, enc, dec, size, [d.to(dt) for d in drops])
    return o, dict(zip(pr.keys(), torch.autograd.grad((o * go.to(dt)).sum(), list(pr.values()))))
o64, g64 = oracle(torch.float64); o32, g32 = oracle(torch.float32)
print("out err dev-f64 %.2e  f32-f64 %.2e" % (float((out.cpu().double() - o64).abs().max()), float((o32.double() - o64).abs().max())))
for k, g in gen.arena.grad_dict().items():
    sc = max(float(g64[k].abs().max()), 1e-8)
    d = float((g.cpu().double() - g64[k]).abs().max()) / sc
    nz = float((g32[k].double() - g64[k]).abs().max()) / sc
    dd = float((g.cpu().double() - g32[k].double()).abs().max()) / sc
    print("%-34s dev-f64 %.2e  f32-f64 %.2e  dev-f32 %.2e %s" % (k, d, nz, dd, "<<<" if d > 2e-3 + 4 * nz else ""))
