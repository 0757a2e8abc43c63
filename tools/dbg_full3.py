import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pta_bootstrap; pta_bootstrap.load()
from pose_transfer_amd.models.networks import Deformable_Generator
from pose_transfer_amd.runtime import engine as E
from pose_transfer_amd.utils import synth
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
mode = sys.argv[1]
P, size = 18, (64, 64)
enc, dec = synth.nfilters(size)
par = {k: t(v) for k, v in synth.init_params(71, "full/gen", synth.generator_spec(P, enc, dec), norm_jitter=0.2).items()}
inp, tgt, wr, mk = [t(a) for a in synth.batch(71, "full", 2, P, *size)]
drops = [t(m) for m in synth.dropout_masks(71, "full", 2)]
W = wr[:, :1] if mode == "full" else wr
gen = Deformable_Generator(3 + 2 * P, P, size, enc, dec, mode)
gen.load_state_dict(par)
eng = gen.engine(2)
eng.set_dropout([d.cuda() for d in drops])
snap = {}
for ws in (256 << 20, 0):
    E.SPLITK_WS_BYTES = ws
    eng.forward(inp.cuda(), W.cuda().float(), None if mode == "full" else mk.cuda())
    torch.cuda.synchronize()
    cur = {}
    for e in eng.encs:
        for l, x in enumerate(eng.e_raw[e]): cur["%s.%d" % (e, l)] = x.clone()
    for i, x in enumerate(eng.d_raw): cur["dec.%d" % i] = x.clone()
    for l, x in enumerate(eng.w_out): cur["warp.%d" % l] = x.clone()
    for i, st in enumerate(eng.d_norm): cur["dec.%d.aff" % i] = st.aff.clone(); cur["dec.%d.mr" % i] = st.mr.clone()
    snap[ws] = cur
for k in snap[0]:
    a, b = snap[256 << 20][k], snap[0][k]
    d = (a - b).abs()
    print("%-22s max|x| %.3e  max diff %.3e  n(diff>1e-4*max) %d  nan %d" % (k, float(b.abs().max()), float(d.max()), int((d > 1e-4 * b.abs().max()).sum()), int(torch.isnan(a).sum())))
