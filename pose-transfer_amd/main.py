"""Training driver with the reference's loop shape (reference src_deformable/main.py:44-159): per iteration
`training_ratio` x [two batches -> dis_update] then one batch -> gen_update.  The reference's datasets are
private, so batches come from the synthetic generator (utils/synth.py).

Single GPU:   python pose-transfer_amd/main.py --dataset fasion --pose_dim 18 --batch_size 4 --steps 20
Multi GPU:    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 pose-transfer_amd/main.py ...
"""
import os
import sys
import time

import numpy as np
import torch

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import pta_bootstrap
    pta_bootstrap.load()
    from pose_transfer_amd.models.pose_gan import DeformablePose_GAN
    from pose_transfer_amd.opts import opts
    from pose_transfer_amd.runtime import dp
    from pose_transfer_amd.utils import synth
else:
    from .models.pose_gan import DeformablePose_GAN
    from .opts import opts
    from .runtime import dp
    from .utils import synth


def synthetic_batch(opt, it, tag, device):
    inp, tgt, wr, mk = synth.batch(opt.seed + dp.rank(), "it%d/%s" % (it, tag), opt.batch_size, opt.pose_dim,
                                   *opt.image_size)
    f = lambda a: torch.from_numpy(a).to(device)
    return f(inp), f(tgt), f(wr), f(mk)


def main(argv=None):
    opt = opts().parse(argv)
    from pose_transfer_amd.runtime import engine as _E
    _E.PRECISION = {"f32": 0, "bf16": 1, "bf16x3": 2, "bf16_data": 3}[opt.precision]
    dp.init_from_env()
    device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(device)
    model = DeformablePose_GAN(opt, device=device)
    start_epoch = model.resume(opt.checkpoints_dir) if opt.resume == 1 else 1
    od = vars(opt)
    done = 0
    for epoch in range(start_epoch, opt.number_of_epochs + 1):
        gen_losses, disc_losses = [], []
        t0 = time.time()
        for it in range(opt.iters_per_epoch):
            for _ in range(opt.training_ratio):
                a = synthetic_batch(opt, done, "A", device)
                b = synthetic_batch(opt, done, "B", device)
                disc_losses.append(model.dis_update(a[0], a[1], {"warps": a[2], "masks": a[3]}, b[0], b[1], od))
            c = synthetic_batch(opt, done, "C", device)
            out, _, gl = model.gen_update(c[0], c[1], {"warps": c[2], "masks": c[3]}, od)
            gen_losses.append(gl)
            done += 1
            if it % opt.display_ratio == 0 and dp.rank() == 0:
                g = np.mean(np.array(gen_losses), axis=0)
                d = np.mean(np.array(disc_losses), axis=0)
                print("Epoch : {0:d} | Progress : {1:.2f} | Gen Total {2:.3f} LL {3:.3f} Ad {4:.3f} | "
                      "Disc Total {5:.3f} True {6:.3f} Fake {7:.3f} | {8:.2f} img/s".format(
                          epoch, it / opt.iters_per_epoch, g[0], g[1], g[2], d[0], d[1], d[2],
                          (it + 1) * opt.batch_size * dp.world_size() / max(time.time() - t0, 1e-9)), flush=True)
            if opt.steps and done >= opt.steps:
                return
        if epoch % opt.checkpoint_ratio == 0 and dp.rank() == 0:
            model.save(opt.checkpoints_dir, epoch)


if __name__ == "__main__":
    main()
