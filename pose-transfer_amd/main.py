"""Training driver with the reference's loop shape (reference src_deformable/main.py:44-159): per iteration
`training_ratio` x [two batches -> dis_update] then one batch -> gen_update; every `display_ratio` iterations the loss
means are printed and a test batch is run through the generator; every `checkpoint_ratio` epochs `gen_%03d.pkl` /
`disc_%03d.pkl` are written.

Batches come from `PoseTransfer_Dataset` (`--synthetic 0`, the reference's CSV / image layout under `--data_Dir`,
datasets/PoseTransfer_Dataset.py) or from the synthetic generator (`--synthetic 1`, default: no data set ships with
the repo; utils/synth.py).

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
from pose_transfer_amd.models.pose_gan import DeformablePose_GAN  # noqa: E402
from pose_transfer_amd.opts import opts  # noqa: E402
from pose_transfer_amd.runtime import dp  # noqa: E402
from pose_transfer_amd.utils import synth  # noqa: E402


class SyntheticSource:
    """Dataset-shaped batches from the counter-based generator: (input, target, warps, masks) or, for
    gen_type=stacked, (input, target, interpol_pose, interpol_warps, interpol_masks) (reference Dataset.py:185-188).
    --synthetic_ring K: K batches are generated once and handed out round-robin (the host-side generator costs tens of
    milliseconds per 256 x 256 batch — more than a training iteration on the bf16 data path; bench.py's loop cycles 3)."""

    def __init__(self, opt, device, split="train"):
        self.opt, self.device, self.split, self.count = opt, device, split, 0
        self.ring = [self._make(i) for i in range(int(getattr(opt, "synthetic_ring", 0) or 0))]

    def _make(self, count):
        o = self.opt
        tag = "%s/it%d" % (self.split, count)
        f = lambda a: torch.from_numpy(a).to(self.device)
        H, W = o.image_size
        if o.gen_type == "stacked":
            S = o.num_stacks
            inp, tgt, _, _ = synth.batch(o.seed + dp.rank(), tag, o.batch_size, o.pose_dim, H, W)
            poses = np.concatenate([synth.heatmaps(o.seed + dp.rank(), "%s/ip%d" % (tag, s), o.batch_size, o.pose_dim, H, W)
                                    for s in range(S)], axis=1)
            wm = [synth.warps_and_masks(o.seed + dp.rank(), "%s/iw%d" % (tag, s), o.batch_size, H, W) for s in range(S)]
            return (f(inp), f(tgt), f(poses), f(np.stack([w for w, _ in wm], 1)), f(np.stack([m for _, m in wm], 1)))
        inp, tgt, wr, mk = synth.batch(o.seed + dp.rank(), tag, o.batch_size, o.pose_dim, H, W)
        if o.warp_skip != "mask":
            wr, mk = wr[:, :1], mk[:, :1]
        return f(inp), f(tgt), f(wr).float(), f(mk)

    def next(self):
        k = self.count
        self.count += 1
        return self.ring[k % len(self.ring)] if self.ring else self._make(k)


def make_sources(opt, device):
    if opt.synthetic:
        return SyntheticSource(opt, device, "train"), SyntheticSource(opt, device, "test")
    from pose_transfer_amd.datasets.PoseTransfer_Dataset import PoseTransfer_Dataset, BatchPipeline
    # every rank builds the SAME epoch permutation (seed + epoch) and takes its own slice of each global batch: the rank
    # slices of an epoch are disjoint and their union is the permutation (tests/test_host_cpu.py)
    train = BatchPipeline(PoseTransfer_Dataset(vars(opt), "train"), opt.batch_size, device, shuffle=True,
                          seed=opt.seed, workers=opt.num_workers, rank=dp.rank(), world=dp.world_size())
    test = BatchPipeline(PoseTransfer_Dataset(vars(opt), "test"), opt.batch_size, device, shuffle=True,
                         seed=opt.seed + 7919 + dp.rank(), workers=max(1, opt.num_workers // 2))
    return train, test


def other_inputs(opt, batch):
    """reference main.py:81-86,104-108: what gen_update / dis_update receive besides input and target."""
    if opt.gen_type == "stacked":
        return {"interpol_pose": batch[2], "interpol_warps": batch[3].float(), "interpol_masks": batch[4]}
    return {"warps": batch[2].float(), "masks": batch[3]}


def build(opt, device):
    from pose_transfer_amd.runtime import engine as _E
    _E.PRECISION = {"f32": 0, "bf16": 1, "bf16x3": 2, "bf16_data": 3}[opt.precision]
    return DeformablePose_GAN(opt, device=device)


class LossLog:
    """The loss triples of an epoch.  With lazy_losses the updates return 3-float DEVICE tensors (models/pose_gan.py) and nothing
    synchronises per update; `means()` — called where the reference prints (main.py:117-127), once per display_ratio iterations —
    moves what has accumulated since the last call to the host in one copy."""

    def __init__(self):
        self.host, self.pending = [], []

    def append(self, triple):
        (self.pending if torch.is_tensor(triple) else self.host).append(triple)

    def flush(self):
        if self.pending:
            self.host.extend(torch.stack(self.pending).cpu().tolist())
            self.pending = []

    def means(self):
        self.flush()
        return np.mean(np.array(self.host, dtype=np.float64), axis=0)


def main(argv=None):
    opt = opts().parse(argv)
    dp.init_from_env()
    device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(device)
    model = build(opt, device)
    start_epoch = model.resume(opt.checkpoints_dir) if opt.resume == 1 else 1
    train, test = make_sources(opt, device)
    od = dict(vars(opt), lazy_losses=bool(opt.lazy_losses))
    done = 0
    t_steady, n_steady = None, 0
    model.iteration = (start_epoch - 1) * opt.iters_per_epoch      # dropout stream continues across a resume

    def finish(epoch):
        # steady-state throughput of THIS loop (what bench.py reports as main_py_img_s): iterations after --timing_skip
        torch.cuda.synchronize()
        stats = {"iterations": done, "timed_iterations": n_steady, "img_s": None, "lazy_losses": bool(opt.lazy_losses)}
        if t_steady is not None and n_steady > 0:
            stats["img_s"] = n_steady * opt.batch_size * dp.world_size() / max(time.perf_counter() - t_steady, 1e-9)
            if dp.rank() == 0:
                print("main.py: %d iterations after the first %d: %.2f img/s (%s losses)" % (
                    n_steady, opt.timing_skip, stats["img_s"], "lazy" if opt.lazy_losses else "eager"), flush=True)
        model.last_run_stats = stats
        if getattr(opt, "save_at_end", 0) and dp.rank() == 0:
            model.save(opt.checkpoints_dir, epoch)
        return model

    for epoch in range(start_epoch, opt.number_of_epochs + 1):
        gen_losses, disc_losses = LossLog(), LossLog()
        t0 = time.time()
        for it in range(opt.iters_per_epoch):
            c = None
            for k in range(opt.training_ratio):
                a, b = train.next(), train.next()
                if k == opt.training_ratio - 1:
                    # the generator update's batch, drawn in the reference's order (after the last discriminator pair) but BEFORE
                    # that discriminator update runs: its generator forward is enqueued ahead (models/pose_gan.py)
                    c = train.next()
                    oc = other_inputs(opt, c)
                    model.prefetch_gen_forward(c[0], oc)
                disc_losses.append(model.dis_update(a[0], a[1], other_inputs(opt, a), b[0], b[1], od))
            if c is None:                  # --training_ratio 0: a generator-only run (reference main.py:100-108 draws the batch after the loop)
                c = train.next()
                oc = other_inputs(opt, c)
            out, outputs, gl = model.gen_update(c[0], c[1], oc, od)
            gen_losses.append(gl)
            done += 1
            model.iteration += 1
            if done == opt.timing_skip:
                torch.cuda.synchronize()
                t_steady = time.perf_counter()
            elif t_steady is not None:
                n_steady += 1
            if it % opt.display_ratio == 0 and dp.rank() == 0:
                g = gen_losses.means()
                d = disc_losses.means() if disc_losses.host or disc_losses.pending else np.zeros(3)
                print("Epoch : {0:d} | Progress : {1:.2f} | Gen Total {2:.3f} LL {3:.3f} Ad {4:.3f} | "
                      "Disc Total {5:.3f} True {6:.3f} Fake {7:.3f} | {8:.2f} img/s".format(
                          epoch, it / opt.iters_per_epoch, g[0], g[1], g[2], d[0], d[1], d[2],
                          (it + 1) * opt.batch_size * dp.world_size() / max(time.time() - t0, 1e-9)), flush=True)
                model.last_display = {"gen": [float(v) for v in g], "disc": [float(v) for v in d], "iterations": done}
                if getattr(opt, "save_samples", 0):
                    # reference main.py:118-147: the current train batch and one test batch as image grids
                    from pose_transfer_amd.test import save_grid
                    save_grid(model, opt, c, out, outputs, os.path.join(opt.output_dir, "train", "%05d.png" % done))
                    tb = test.next()
                    save_grid(model, opt, tb, None, None, os.path.join(opt.output_dir, "test", "%05d.png" % done))
            if opt.steps and done >= opt.steps:
                return finish(epoch)
        if epoch % opt.checkpoint_ratio == 0 and dp.rank() == 0:
            model.save(opt.checkpoints_dir, epoch)
    return finish(opt.number_of_epochs)


if __name__ == "__main__":
    main()
