"""The deformable skips' kernels alone at batch N on the bf16 data path, per level, with the generator's own buffers (masks, transforms,
arg-max planes from a real forward + backward pass):   gpurun -- python tools/warp_bench.py [N]
Prints microseconds per launch and the rate over the algorithmic bytes (forward: read C x 2 B + write C x (2 + 1) B per pixel;
backward: read C x (2 + 1) B + read-modify-write C x 2 B x 2 per pixel)."""
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pta_bootstrap  # noqa: E402

pta_bootstrap.load()
from pose_transfer_amd.models.pose_gan import DeformablePose_GAN  # noqa: E402
from pose_transfer_amd.runtime import engine as E  # noqa: E402
from pose_transfer_amd.runtime import lib as L  # noqa: E402
from pose_transfer_amd.utils import synth  # noqa: E402


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    E.PRECISION = 3
    os.environ["PG_NO_AUX_STREAM"] = "1"
    opt = SimpleNamespace(image_size=(256, 256), use_input_pose=True, pose_dim=18, batch_size=N, num_stacks=4, gen_type="baseline",
                          dataset="fasion", warp_skip="mask", learning_rate=2e-4, content_loss_layer="none", nn_loss_area_size=1,
                          gan_penalty_weight=1.0, l1_penalty_weight=100.0)
    model = DeformablePose_GAN(opt, device="cuda", init_seed=0)
    eng = model.gen.engine(N)
    inp, tgt, wr, mk = [torch.from_numpy(a).to("cuda") for a in synth.batch(5, "warpbench", N, 18, 256, 256)]
    eng.set_dropout(train=True, seed=1)
    out = eng.forward(inp, wr, mk)
    eng.backward(torch.randn_like(out).contiguous())
    torch.cuda.synchronize()
    H = W = 256
    tf = tb = 0.0
    for l in range(eng.nwarp):
        a = eng._enc_act("encoder_app", l)
        h, w, C = eng.hw[l][0], eng.hw[l][1], eng.enc[l]

        def fwd():
            L.call("pg_warp_mask_max_fwd_io", L.ptr(a.t), L.ptr(a.aff), L.ptr(eng.warps), L.ptr(eng.lvl_masks[l]), N, eng.T, C, h, w, H, W,
                   eng.align, L.ptr(eng.w_out[l]), L.ptr(eng.w_arg[l]), 7, L.stream())

        def bwd():
            L.call("pg_warp_mask_max_bwd_bbox", L.ptr(eng.w_g[l]), L.ptr(eng.w_arg[l]), L.ptr(eng.warps), L.ptr(eng.lvl_masks[l]),
                   L.ptr(eng.mask_bbox), N, eng.T, C, h, w, H, W, eng.align, L.ptr(eng.e_dz["encoder_app"][l]), 3, L.stream())

        uf, ub = timed(fwd), timed(bwd)
        px = N * h * w * C
        print("level %d (%3d x %3d x %3d): forward %7.1f us = %.2f TB/s | backward %7.1f us = %.2f TB/s" % (
            l, h, w, C, uf, px * 5 / uf / 1e6, ub, px * 7 / ub / 1e6))
        tf += uf
        tb += ub
    print("sum: forward %.1f us, backward %.1f us" % (tf, tb))


if __name__ == "__main__":
    main()
