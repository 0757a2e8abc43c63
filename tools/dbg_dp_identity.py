"""debug: DP identity per parameter tensor at several sizes (gpurun -- python tools/dbg_dp_identity.py)"""
import os, sys
from types import SimpleNamespace
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pta_bootstrap; pta_bootstrap.load()
from pose_transfer_amd.models.pose_gan import DeformablePose_GAN
from pose_transfer_amd.utils import synth
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
def opt_(size, P, N):
    return SimpleNamespace(image_size=size, use_input_pose=True, pose_dim=P, batch_size=N, num_stacks=4, gen_type="baseline",
                           dataset="fasion", warp_skip="mask", learning_rate=2e-4, content_loss_layer="none",
                           nn_loss_area_size=1, gan_penalty_weight=1.0, l1_penalty_weight=100.0)
for (H, N) in [(64, 2), (128, 2), (256, 2), (512, 2), (256, 4)]:
    P = 18
    b = [t(a) for a in synth.batch(77, "prop%d" % H, N, P, H, H)]
    d = [t(m) for m in synth.dropout_masks(77, "prop%d" % H, N)]
    opt = opt_((H, H), P, N)
    model = DeformablePose_GAN(opt, device="cuda", init_seed=5)
    gsd = {k: v.clone() for k, v in model.gen.state_dict().items()}
    _, _, gl = model.gen_update(b[0], b[1], {"warps": b[2], "masks": b[3], "drop_masks": d}, vars(opt))
    gfull = {k: v.clone() for k, v in model.gen.arena.grad_dict().items()}
    n2 = N // 2
    opt2 = opt_((H, H), P, n2)
    m2 = DeformablePose_GAN(opt2, device="cuda", init_seed=5)
    acc = None
    gls = []
    for r in range(2):
        m2.gen.load_state_dict(gsd)
        sl = slice(n2 * r, n2 * (r + 1))
        _, _, g2 = m2.gen_update(b[0][sl].contiguous(), b[1][sl].contiguous(), {"warps": b[2][sl].contiguous(), "masks": b[3][sl].contiguous(), "drop_masks": [x[sl].contiguous() for x in d]}, vars(opt2))
        gls.append(g2)
        gd = m2.gen.arena.grad_dict()
        acc = gd if acc is None else {k: acc[k] + gd[k] for k in gd}
    print("H", H, "N", N, "loss full", gl, "shards", gls)
    worst = []
    for k in gfull:
        a = acc[k] / 2
        sc = float(gfull[k].abs().max())
        worst.append((float((a - gfull[k]).abs().max()) / max(sc, 1e-12), k, sc))
    worst.sort(reverse=True)
    for w in worst[:6]:
        print("   %.3e  %-40s scale %.3e" % w)
    del model, m2
    torch.cuda.empty_cache()
