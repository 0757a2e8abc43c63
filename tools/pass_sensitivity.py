"""Which kernels' time is EXPOSED in the north-star pass (generator forward + backward, 256 x 256, batch 32, bf16 data path)?
For each kernel class the pass is run with every launch of the class issued TWICE (results are wrong, times are not) and compared
with the unmodified pass, arms alternating inside one process.  exposure = (pass_doubled - pass) / (the class's single-stream time):
1 = every microsecond of the class is on the pass's critical resource, 0 = hidden behind other streams' work.
    gpurun -- python tools/pass_sensitivity.py [rounds] [passes]"""
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
from pose_transfer_amd.runtime import engine as E  # noqa: E402
from pose_transfer_amd.runtime import lib as L  # noqa: E402
from pose_transfer_amd.utils import synth  # noqa: E402

CLASSES = {
    "conv forward (pg_conv, plain epilogue)": lambda n, d: n == "pg_conv" and d.epilogue == 0,
    "conv data gradient (pg_conv, scatter)": lambda n, d: n == "pg_conv" and d.epilogue == 1,
    "weight gradients (pg_wgrad_bf16_ex)": lambda n, d: n == "pg_wgrad_bf16_ex",
    "first-layer weight gradients": lambda n, d: n.startswith("pg_stem_wgrad"),
    "norm backward apply": lambda n, d: n.startswith("pg_norm_bwd_apply"),
    "materialise (+ norm fold)": lambda n, d: n.startswith("pg_materialise_bf16"),
    "first-layer forward": lambda n, d: n.startswith("pg_stem_conv"),
    "warp forward + mask pyramid": lambda n, d: n.startswith("pg_warp_mask_max_fwd") or n in ("pg_mask_pyramid", "pg_mask_bbox"),
    "warp backward": lambda n, d: n.startswith("pg_warp_mask_max_bwd"),
    "output convolution forward": lambda n, d: n == "pg_out_conv_fwd_fused",
    "output convolution backward": lambda n, d: n.startswith("pg_out_conv_bwd") or n.startswith("pg_out_conv_dgrad"),
}


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    N, size, kp = int(os.environ.get("PG_AB_BATCH", "32")), 256, 18
    E.PRECISION = {"f32": 0, "bf16_data": 3}[os.environ.get("PG_AB_PRECISION", "bf16_data")]
    o = SimpleNamespace(image_size=(size, size), use_input_pose=True, pose_dim=kp, batch_size=N, num_stacks=4, gen_type="baseline",
                        dataset="fasion", warp_skip="mask", learning_rate=2e-4, content_loss_layer="none", nn_loss_area_size=1,
                        gan_penalty_weight=1.0, l1_penalty_weight=100.0)
    model = DeformablePose_GAN(o, device="cuda:0", init_seed=0)
    dev = lambda arrs: [torch.from_numpy(a).cuda() for a in arrs]
    inp, _, wr, mk = dev(synth.batch(1234, "ns/A", N, kp, size, size))
    gout = torch.from_numpy(synth.normal(1234, "ns/gout", (N, 3, size, size))).cuda()
    eng = model.gen.engine(N)
    eng.set_dropout(None, train=True, seed=0)
    lib = L.load()
    raw_conv = lib.pg_conv
    state = {"pred": None, "count": {}, "time": None}

    def conv_wrapper(d, st):
        rc = raw_conv(d, st)
        desc = d._obj if hasattr(d, "_obj") else d
        if state["pred"] is not None and state["pred"]("pg_conv", desc):
            raw_conv(d, st)
        return rc

    def hook(name, args, launch):
        r = launch()
        if state["pred"] is not None and state["pred"](name, None):
            launch()
        return r

    lib.pg_conv = conv_wrapper
    L.CALL_HOOK = hook

    def one_pass():
        model.gen.zero_grad()
        eng.forward(inp, wr, mk)
        eng.backward(gout)

    if os.environ.get("PG_AB_MODE") == "step":        # the full training iteration (dis_update + gen_update, 3 batches) instead of the pass
        import bench
        batches = [dev(synth.batch(1234, "ns/%s" % s_, N, kp, size, size)) for s_ in "ABC"]
        od = dict(vars(o), lazy_losses=True)
        raw_wg = lib.pg_conv_wgrad

        def wg_wrapper(d, st):
            rc = raw_wg(d, st)
            if state["pred"] is not None and state["pred"]("pg_conv_wgrad", None):
                raw_wg(d, st)
            return rc

        lib.pg_conv_wgrad = wg_wrapper
        CLASSES.pop("weight gradients (pg_wgrad_bf16_ex)")
        CLASSES["weight gradients (pg_wgrad_bf16_ex / pg_conv_wgrad)"] = lambda n, d: n in ("pg_wgrad_bf16_ex", "pg_conv_wgrad")
        CLASSES["first-layer weight gradients"] = lambda n, d: n.startswith("pg_stem_wgrad") or n == "pg_small_cin_wgrad"
        CLASSES["first-layer forward"] = lambda n, d: n.startswith("pg_stem_conv") or n in ("pg_small_cin_conv", "pg_repack_small_cin")
        CLASSES["norm finalize / stats"] = lambda n, d: n in ("pg_norm_finalize", "pg_norm_stats")
        CLASSES["norm backward (reduce + apply)"] = lambda n, d: n.startswith("pg_norm_bwd")
        CLASSES.pop("norm backward apply")
        CLASSES["Adam"] = lambda n, d: n.startswith("pg_adam")
        CLASSES["losses, tanh, edge layers of D"] = lambda n, d: n in ("pg_gan_logloss", "pg_l1_loss", "pg_tanh_bwd", "pg_add2", "pg_small_cout_dgrad",
                                                                        "pg_small_cin_dgrad", "pg_small_cin_dgrad_io", "pg_bias_grad", "pg_bias_grad_bf16")
        CLASSES["dropout masks, zero fills, stream waits"] = lambda n, d: n in ("pg_dropout_mask", "pg_zero", "pg_stream_wait", "pg_copy")
        if os.environ.get("PG_AB_GLUE"):               # the glue calls one by one
            for k in list(CLASSES):
                CLASSES.pop(k)
            for nm in ("pg_stream_wait", "pg_zero", "pg_dropout_mask", "pg_copy", "pg_norm_finalize", "pg_add2", "pg_tanh_bwd", "pg_gan_logloss",
                       "pg_l1_loss", "pg_mask_pyramid", "pg_mask_bbox", "pg_stem_pack_bf16", "pg_repack_small_cin", "pg_bias_grad"):
                CLASSES[nm] = (lambda want: (lambda n, d: n == want))(nm)

        def one_pass():  # noqa: F811
            bench.iteration(model, batches, od)

    def timed(pred):
        state["pred"] = pred
        one_pass()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(passes):
            one_pass()
        e1.record()
        torch.cuda.synchronize()
        state["pred"] = None
        return e0.elapsed_time(e1) / passes

    for _ in range(3):
        one_pass()
    torch.cuda.synchronize()
    # single-stream time of each class: one un-doubled and one doubled single-stream pass, the difference is the class alone
    side, E.SIDE_STREAM = E.SIDE_STREAM, False
    one_pass(); torch.cuda.synchronize()
    ss_base = np.median([timed(None) for _ in range(3)])
    ss = {nm: max(np.median([timed(pred) for _ in range(2)]) - ss_base, 1e-6) for nm, pred in CLASSES.items()}
    E.SIDE_STREAM = side
    one_pass(); torch.cuda.synchronize()
    base, dbl = [], {nm: [] for nm in CLASSES}
    for r in range(rounds):
        base.append(timed(None))
        for nm, pred in CLASSES.items():
            dbl[nm].append(timed(pred))
        base.append(timed(None))
    b = float(np.mean(base))
    print("%s, batch %d, %s: %.3f ms (multi-stream), %.3f ms single-stream; %d rounds x %d passes" % (os.environ.get("PG_AB_MODE", "pass"), N, os.environ.get("PG_AB_PRECISION", "bf16_data"), b, ss_base, rounds, passes))
    print("%-44s %10s %12s %9s" % ("class (every launch issued twice)", "alone ms", "pass + ms", "exposure"))
    tot_a = tot_d = 0.0
    for nm in CLASSES:
        d = float(np.mean(dbl[nm])) - b
        tot_a += ss[nm]; tot_d += d
        print("%-44s %10.3f %12.3f %9.2f" % (nm, ss[nm], d, d / ss[nm]))
    print("%-44s %10.3f %12.3f %9.2f" % ("sum", tot_a, tot_d, tot_d / tot_a))


if __name__ == "__main__":
    main()
