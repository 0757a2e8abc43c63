"""In-process A/B of per-launch switches on the north-star pass (generator forward + backward, 256 x 256, batch 32, bf16 data path):
the switches are read by the library at every launch (getenv), so ONE process alternates the arms round by round — box-to-box and
process-to-process drift (+- 0.1 ms on a 17 ms pass) cancels.
    gpurun -- python tools/quad_inproc_ab.py [rounds] [passes]"""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pta_bootstrap  # noqa: E402

pta_bootstrap.load()
from pose_transfer_amd.models.pose_gan import DeformablePose_GAN  # noqa: E402
from pose_transfer_amd.runtime import engine as E  # noqa: E402
from pose_transfer_amd.utils import synth  # noqa: E402

ARMS = [("quad off", {"PG_BIG_QUAD": "0"}), ("quad 4-wave", {"PG_BIG_QUAD": "1", "PG_QUAD_WAVES": "4"}),
        ("quad 8-wave", {"PG_BIG_QUAD": "1", "PG_QUAD_WAVES": "8"})]
# PG_AB_ARMS="name:K=V,K=V;name2:K=V" replaces the arms (the first one is the reference); only PER-LAUNCH switches make sense here
if os.environ.get("PG_AB_ARMS"):
    ARMS = []
    for a in os.environ["PG_AB_ARMS"].split(";"):
        nm, _, kv = a.partition(":")
        ARMS.append((nm, dict(x.split("=", 1) for x in kv.split(",") if x)))
KEYS = sorted({k for _, e in ARMS for k in e})


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    N, size, kp = int(os.environ.get("PG_AB_BATCH", "32")), 256, 18
    E.PRECISION = 3
    o = SimpleNamespace(image_size=(size, size), use_input_pose=True, pose_dim=kp, batch_size=N, num_stacks=4, gen_type="baseline",
                        dataset="fasion", warp_skip="mask", learning_rate=2e-4, content_loss_layer="none", nn_loss_area_size=1,
                        gan_penalty_weight=1.0, l1_penalty_weight=100.0)
    model = DeformablePose_GAN(o, device="cuda:0", init_seed=0)
    dev = lambda arrs: [torch.from_numpy(a).cuda() for a in arrs]
    inp, _, wr, mk = dev(synth.batch(1234, "ns/A", N, kp, size, size))
    gout = torch.from_numpy(synth.normal(1234, "ns/gout", (N, 3, size, size))).cuda()
    eng = model.gen.engine(N)
    eng.set_dropout(None, train=True, seed=0)

    def one_pass():
        model.gen.zero_grad()
        eng.forward(inp, wr, mk)
        eng.backward(gout)

    if os.environ.get("PG_AB_MODE") == "step":        # the full training iteration (dis_update + gen_update, 3 batches) instead of the pass
        import bench
        batches = [dev(synth.batch(1234, "ns/%s" % s_, N, kp, size, size)) for s_ in "ABC"]
        od = dict(vars(o), lazy_losses=True)

        def one_pass():  # noqa: F811
            bench.iteration(model, batches, od)

    def setenv(env):
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)

    for _, env in ARMS:
        setenv(env)
        for _ in range(2):
            one_pass()
    torch.cuda.synchronize()
    res = {nm: [] for nm, _ in ARMS}
    for r in range(rounds):
        order = ARMS if r % 2 == 0 else ARMS[::-1]
        for nm, env in order:
            setenv(env)
            one_pass()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(passes):
                one_pass()
            e1.record()
            torch.cuda.synchronize()
            res[nm].append(e0.elapsed_time(e1) / passes)
    for nm, _ in ARMS:
        a = np.array(res[nm])
        print("%-16s mean %.3f ms  median %.3f  min %.3f  max %.3f  (%d rounds x %d passes)" % (nm, a.mean(), np.median(a), a.min(), a.max(), rounds, passes))
    base = np.array(res[ARMS[0][0]])
    for nm, _ in ARMS[1:]:
        d = np.array(res[nm]) - base
        print("%-16s - ref: mean %+.3f ms, per-round %s" % (nm, d.mean(), " ".join("%+.2f" % x for x in d)))


if __name__ == "__main__":
    main()
