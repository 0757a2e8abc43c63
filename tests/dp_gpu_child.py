"""Child process of the device-side data-parallel tests (tests/test_gpu_round2.py), launched through
torch.distributed.run: every rank trains on its shard of a seeded global batch of 4 at 64x64 and rank 0 stores the
(all-reduced, averaged) gradient arenas of iteration 0, the losses and the parameters after two iterations.
PG_GLOBAL_BATCH=1: one rank takes the whole global batch (the reference run of the DP identity)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pta_bootstrap  # noqa: E402

pta_bootstrap.load()
from pose_transfer_amd.models.pose_gan import DeformablePose_GAN  # noqa: E402
from pose_transfer_amd.runtime import dp  # noqa: E402
from pose_transfer_amd.utils import synth  # noqa: E402


def main(out_path):
    world = dp.init_from_env("nccl")
    rank = dp.rank()
    dev = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev)
    P, H, W, G = 18, 64, 64, 4
    n = G if os.environ.get("PG_GLOBAL_BATCH") == "1" else G // max(world, 2) if world > 1 else 2
    if world == 1 and os.environ.get("PG_GLOBAL_BATCH") != "1":
        n = 2                      # single-rank reducer test: plain batch of 2
    opt = SimpleNamespace(image_size=(H, W), use_input_pose=True, pose_dim=P, batch_size=n, num_stacks=4, gen_type="baseline",
                          dataset="fasion", warp_skip="mask", learning_rate=2e-4, content_loss_layer="none",
                          nn_loss_area_size=1, gan_penalty_weight=1.0, l1_penalty_weight=100.0)
    model = DeformablePose_GAN(opt, device=dev, init_seed=5)
    od = vars(opt)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    sl = slice(rank * n, rank * n + n)
    res = {"losses": []}
    div = lambda red: 1 if red is None else red.divisor

    def reduced(red, arena):
        """what the optimiser read: the all-reduced sums (bf16 buckets: the packed copy) divided by the reducer's divisor"""
        if red is None:
            return arena.grads.cpu()
        g32, g16 = red.grad_source()
        return ((g16.float() if g16 is not None else g32) / red.divisor).cpu()

    for it in range(2):
        b = [[t(a)[sl].contiguous().to(dev) for a in synth.batch(55, "dp/it%d/%s" % (it, s), G, P, H, W)] for s in "ABC"]
        d = [[t(m)[sl].contiguous().to(dev) for m in synth.dropout_masks(55, "dp/it%d/d%s" % (it, s), G)] for s in "AC"]
        dl = model.dis_update(b[0][0], b[0][1], {"warps": b[0][2], "masks": b[0][3], "drop_masks": d[0]}, b[1][0], b[1][1], od)
        if it == 0:
            res["disc_grads"] = (model.disc.arena.grads / div(model.d_reducer)).cpu()
            res["disc_grads_reduced"] = reduced(model.d_reducer, model.disc.arena)
        _, _, gl = model.gen_update(b[2][0], b[2][1], {"warps": b[2][2], "masks": b[2][3], "drop_masks": d[1]}, od)
        if it == 0:
            res["gen_grads"] = (model.gen.arena.grads / div(model.g_reducer)).cpu()
            res["gen_grads_reduced"] = reduced(model.g_reducer, model.gen.arena)
        res["losses"].append((dl, gl))
    torch.cuda.synchronize()
    res["gen"] = model.gen.arena.params.cpu()
    res["disc"] = model.disc.arena.params.cpu()
    res["buckets"] = 0 if model.g_reducer is None else model.g_reducer.launch_count
    res["world"] = world
    res["divisor"] = div(model.g_reducer)
    res["grad_dtype"] = "none" if model.g_reducer is None else model.g_reducer.grad_dtype
    if rank == 0:
        torch.save(res, out_path)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
