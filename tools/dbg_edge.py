import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from gpu_util import DEV, E, L, R, maxdiff, synth, t
from pose_transfer_amd.models.networks import Deformable_Generator
def run(tag, n, size, pdim, seed=29):
    enc, dec = synth.nfilters(size)
    spec = synth.generator_spec(pdim, enc, dec)
    par = {k: t(v) for k, v in synth.init_params(seed, tag, spec, norm_jitter=0.2).items()}
    inp, tgt, wr, mk = [t(a) for a in synth.batch(seed, tag, n, pdim, *size)]
    drops = [t(m) for m in synth.dropout_masks(seed, tag, n)]
    go = t(synth.normal(seed, tag + "/go", (n, 3) + tuple(size)))
    pr = {k: v.clone().requires_grad_(True) for k, v in par.items()}
    out_ref = R.generator_forward(inp, wr, mk, pr, pdim, enc, dec, size, drops)
    gref = dict(zip(pr.keys(), torch.autograd.grad((out_ref * go).sum(), list(pr.values()))))
    gen = Deformable_Generator(3 + 2 * pdim, pdim, size, enc, dec, "mask")
    gen.load_state_dict(par); gen.zero_grad()
    out = gen(inp.to(DEV), wr.to(DEV), mk.to(DEV), drop_masks=[d.to(DEV) for d in drops])
    (out * go.to(DEV)).sum().backward()
    got = gen.arena.grad_dict()
    print(tag, n, size, pdim, "out", float(maxdiff(out, out_ref)))
    for k in gref:
        scale = max(float(gref[k].abs().max()), 1e-8)
        d = (got[k].cpu() - gref[k]).abs()
        r = float(d.max()) / scale
        if r > 1e-3: print("   ", k, "rel", round(r, 5), "frac>2e-3", float((d > 2e-3 * scale).float().mean()), "scale", scale)
run("p16_128", 2, (128, 128), 16)
run("p18_128", 2, (128, 128), 18)
run("p16_128", 2, (128, 128), 16, seed=31)
