"""Pin the oracle (oracle/ref_cpu.py) against tensors captured from the real reference
(tests/golden/*.npz, made by oracle/make_golden.py).  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_cpu as R  # noqa: E402
from pose_transfer_amd.utils import synth  # noqa: E402

P = 18


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def tp(d):
    return {k: t(v) for k, v in d.items()}


def summarize(x):
    f = x.detach().reshape(-1).double()
    idx = torch.linspace(0, f.numel() - 1, 32).long()
    return np.concatenate([[f.sum().item(), f.abs().sum().item(), f.abs().max().item()], f[idx].numpy()])


@pytest.fixture(scope="module")
def ops():
    return np.load(os.path.join(GOLDEN, "ops.npz"))


def test_block_down_up(ops):
    x = t(synth.normal(11, "blk/x", (2, 8, 12, 10)) * 1.7 + 0.3)
    y = R.block_down(x, t(synth.xavier_uniform(11, "blk/w", (16, 8, 4, 4))), torch.tensor(1.3), torch.tensor(-0.2))
    assert np.abs(y.numpy() - ops["block_down"]).max() < 2e-6
    xu = t(synth.normal(11, "blku/x", (2, 8, 5, 6)))
    dm = t(synth.dropout_masks(11, "blku", 2, (4,))[0])
    y = R.block_up(xu, t(synth.xavier_uniform(11, "blku/w", (8, 4, 4, 4))), torch.tensor(0.7), torch.tensor(0.1), dm)
    assert np.abs(y.numpy() - ops["block_up"]).max() < 2e-6


WARP_CASES = [("w256s4", (256, 256), 4, 8), ("w128x64s2", (128, 64), 2, 8), ("w224s8", (224, 224), 8, 8),
              ("w64s1", (64, 64), 1, 4), ("w96x80s2", (96, 80), 2, 4)]


@pytest.mark.parametrize("name,size,s,c", WARP_CASES)
@pytest.mark.parametrize("ac", [False, True])
def test_warp_mask_max(ops, name, size, s, c, ac):
    h, w = size[0] // s, size[1] // s
    feat = t(synth.normal(12, name + "/f", (2, c, h, w))).requires_grad_(True)
    wr, mk = synth.warps_and_masks(12, name, 2, size[0], size[1])
    out = R.warp_mask_max(feat, t(wr), t(mk), size, align_corners=ac)
    go = t(synth.normal(12, name + "/go", tuple(out.shape)))
    (gin,) = torch.autograd.grad((out * go).sum(), feat)
    tag = name + ("_ac1" if ac else "_ac0")
    # coordinates follow the reference's fp32 order; residual = grid_sample internals
    assert np.abs(out.detach().numpy() - ops[tag + "_out"]).max() < 3e-4
    # arg-max over T can flip where two candidates tie to within fp32 rounding (typically sample ~ 0 vs a
    # masked-out 0): the output is continuous there, the routed gradient is not -> allow <0.05% outliers
    d = np.abs(gin.numpy() - ops[tag + "_gin"])
    assert (d > 3e-4).mean() < 5e-4 and np.median(d) < 1e-6


@pytest.mark.parametrize("a", [3, 5])
def test_nn_loss(ops, a):
    pred = t(synth.normal(13, "nn%d/p" % a, (2, 6, 12, 9))).requires_grad_(True)
    gt = t(synth.normal(13, "nn%d/g" % a, (2, 6, 12, 9)))
    l = R.nn_loss(pred, gt, a, a)
    (g,) = torch.autograd.grad(l, pred)
    assert abs(l.item() - float(ops["nn%d_loss" % a])) < 1e-5
    assert np.abs(g.numpy() - ops["nn%d_grad" % a]).max() < 1e-7


def test_vgg_features(ops):
    vw = t(synth.xavier_uniform(14, "vgg/w", (64, 3, 3, 3)))
    vb = t(synth.uniform(14, "vgg/b", (64,), -0.1, 0.1))
    vx = t(synth.uniform(14, "vgg/x", (2, 3, 10, 14), -1, 1))
    f = R.vgg_features(vx, vw, vb)
    assert np.abs(f.numpy() - ops["vgg_feat"]).max() < 2e-6
    assert list(ops["layer_inds"]) == [1, 19]


def test_discriminator(ops):
    dpar = tp(synth.init_params(15, "disc", synth.discriminator_spec(42), norm_jitter=0.2))
    dx = t(synth.uniform(15, "disc/x", (3, 42, 64, 64), -1, 1))
    assert np.abs(R.discriminator_forward(dx, dpar).numpy() - ops["disc_out"]).max() < 2e-6
    dx2 = t(synth.uniform(15, "disc/x2", (2, 42, 96, 80), -1, 1))
    o = R.discriminator_forward(dx2, dpar).numpy()
    assert o.shape == ops["disc_out_96x80"].shape
    assert np.abs(o - ops["disc_out_96x80"]).max() < 2e-6


@pytest.mark.parametrize("name,size", [("g64", (64, 64)), ("g64x32", (64, 32))])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_generator(name, size, mode):
    g = np.load(os.path.join(GOLDEN, "generator.npz"))
    enc, dec = synth.nfilters(size)
    gpar = tp(synth.init_params(21, name, synth.generator_spec(P, enc, dec), norm_jitter=0.2))
    inp, tgt, wr, mk = synth.batch(21, name, 2, P, *size)
    drops = [t(m) for m in synth.dropout_masks(21, name, 2)] if mode == "train" else None
    with torch.no_grad():
        out = R.generator_forward(t(inp), t(wr), t(mk), gpar, P, enc, dec, size, drops)
    assert np.abs(out.numpy() - g["%s_%s_out" % (name, mode)]).max() < 1e-4


def test_generator_7level():
    g = np.load(os.path.join(GOLDEN, "generator.npz"))
    enc, dec = synth.nfilters((256, 256))
    gpar = tp(synth.init_params(22, "g128", synth.generator_spec(P, enc, dec), norm_jitter=0.2))
    inp, tgt, wr, mk = synth.batch(22, "g128", 2, P, 128, 128)
    drops = [t(m) for m in synth.dropout_masks(22, "g128", 2)]
    with torch.no_grad():
        out = R.generator_forward(t(inp), t(wr), t(mk), gpar, P, enc, dec, (128, 128), drops)
    assert np.abs(out[0].numpy() - g["g128_train_out_n0"]).max() < 1e-4
    assert np.abs(summarize(out) - g["g128_train_summary"])[2:].max() < 1e-4


def _check_summary(got, ref, what, rtol=2e-3, scalar_rtol=2e-2):
    """samples + max-abs relative to the tensor's max-abs; sums relative to its abs-sum.
    Scalar gamma/beta grads are cancelling sums over a whole activation: fp32 summation order (ATen
    instance_norm backward vs the closed form) moves them by ~1e-3 relative -> looser `scalar_rtol`
    (None = skip)."""
    if np.all(ref[3:] == ref[3]):          # 1-element tensor (all 32 strided samples identical)
        if scalar_rtol is None:
            return
        rtol = max(rtol, scalar_rtol)
    scale = max(ref[2], 1e-12)
    assert np.abs(got[2:] - ref[2:]).max() <= rtol * scale + 1e-9, what
    assert abs(got[0] - ref[0]) <= rtol * max(ref[1], 1e-12) + 1e-9, what
    assert abs(got[1] - ref[1]) <= rtol * max(ref[1], 1e-12) + 1e-9, what


def _check_par(got, ref, what, it, lr=2e-4):
    """Parameters after Adam: iteration 0 tight; afterwards sign flips of near-zero grads move single
    elements by up to 2*lr per step, so compare element samples with an absolute 3*lr budget."""
    if it == 0:
        _check_summary(got, ref, what, 1e-4, 1e-4)
    else:
        assert np.abs(got[3:] - ref[3:]).max() <= 3 * lr, what


@pytest.mark.parametrize("name,content,area,l1w", [("step_l1", "none", 1, 100.0), ("step_nn", "block1_conv2", 5, 0.01)])
def test_two_training_iterations(name, content, area, l1w):
    fix = np.load(os.path.join(GOLDEN, name + ".npz"))
    H = W = 64
    N = 2
    enc, dec = synth.nfilters((H, W))
    cfg = dict(pose_dim=P, image_size=(H, W), batch_size=N, gan_penalty_weight=1.0, l1_penalty_weight=l1w,
               learning_rate=2e-4, content_loss_layer=content, nn_loss_area_size=area,
               nfilters_enc=enc, nfilters_dec=dec)
    vgg = (t(synth.xavier_uniform(14, "vgg/w", (64, 3, 3, 3))), t(synth.uniform(14, "vgg/b", (64,), -0.1, 0.1)))
    tr = R.Trainer(cfg, tp(synth.init_params(31, name + "/gen", synth.generator_spec(P, enc, dec), 0.1)),
                   tp(synth.init_params(31, name + "/disc", synth.discriminator_spec(42), 0.1)), vgg)
    for it in range(2):
        bA = [t(a) for a in synth.batch(31, "%s/it%d/A" % (name, it), N, P, H, W)]
        bB = [t(a) for a in synth.batch(31, "%s/it%d/B" % (name, it), N, P, H, W)]
        bC = [t(a) for a in synth.batch(31, "%s/it%d/C" % (name, it), N, P, H, W)]
        dA = [t(m) for m in synth.dropout_masks(31, "%s/it%d/dA" % (name, it), N)]
        dC = [t(m) for m in synth.dropout_masks(31, "%s/it%d/dC" % (name, it), N)]
        dl = tr.dis_update(bA[0], bA[1], bA[2], bA[3], bB[0], bB[1], dA)
        np.testing.assert_allclose(dl, fix["it%d_dis_losses" % it], rtol=1e-4 if it == 0 else 2e-3, atol=1e-6)
        # Iteration 0 is compared tightly.  Adam's first step is ~ lr*sign(g), so elements with |g| ~ 1e-8
        # flip between two fp32-equivalent implementations; iteration 1 therefore starts from parameters
        # that differ by O(lr) in a few places and is compared loosely (losses and out_gen stay tight).
        g_rtol, g_srtol, p_rtol = (2e-3, 2e-2, 1e-4) if it == 0 else (5e-2, None, 5e-3)
        for k in tr.dp:
            _check_summary(summarize(tr.last_disc_grads[k]), fix["it%d_dgrad_%s" % (it, k)], "dgrad " + k, g_rtol, g_srtol)
            _check_par(summarize(tr.dp[k]), fix["it%d_dpar_%s" % (it, k)], "dpar " + k, it)
        og, gl = tr.gen_update(bC[0], bC[1], bC[2], bC[3], dC)
        np.testing.assert_allclose(gl, fix["it%d_gen_losses" % it], rtol=1e-4 if it == 0 else 2e-3, atol=1e-6)
        assert np.abs(og.numpy() - fix["it%d_out_gen" % it]).max() < (1e-4 if it == 0 else 1e-3)
        for k in tr.gp:
            _check_summary(summarize(tr.last_gen_grads[k]), fix["it%d_ggrad_%s" % (it, k)], "ggrad " + k, g_rtol, g_srtol)
            _check_par(summarize(tr.gp[k]), fix["it%d_gpar_%s" % (it, k)], "gpar " + k, it)


def test_baseline_step():
    fix = np.load(os.path.join(GOLDEN, "baseline_step.npz"))
    H, W, N = 128, 64, 2
    enc, dec = synth.nfilters((H, W))
    cfg = dict(pose_dim=P, image_size=(H, W), batch_size=N, gan_penalty_weight=1.0, l1_penalty_weight=100.0,
               learning_rate=2e-4, content_loss_layer="none", nn_loss_area_size=1, deformable=False,
               nfilters_enc=enc, nfilters_dec=dec)
    gspec = synth.generator_spec(P, enc, dec, num_skips=1, deformable=False)
    tr = R.Trainer(cfg, tp(synth.init_params(41, "base/gen", gspec, 0.1)),
                   tp(synth.init_params(41, "base/disc", synth.discriminator_spec(42), 0.1)))
    bA = [t(a) for a in synth.batch(41, "base/A", N, P, H, W)]
    bB = [t(a) for a in synth.batch(41, "base/B", N, P, H, W)]
    bC = [t(a) for a in synth.batch(41, "base/C", N, P, H, W)]
    dA = [t(m) for m in synth.dropout_masks(41, "base/dA", N)]
    dC = [t(m) for m in synth.dropout_masks(41, "base/dC", N)]
    dl = tr.dis_update(bA[0], bA[1], None, None, bB[0], bB[1], dA)
    np.testing.assert_allclose(dl, fix["dis_losses"], rtol=1e-4)
    og, gl = tr.gen_update(bC[0], bC[1], None, None, dC)
    np.testing.assert_allclose(gl, fix["gen_losses"], rtol=1e-4)
    assert np.abs(og.numpy() - fix["out_gen"]).max() < 1e-4
    for k in tr.gp:
        _check_summary(summarize(tr.last_gen_grads[k]), fix["ggrad_" + k], "ggrad " + k)


def test_mask_pyramid_power_of_two_is_centre_mean():
    """for exact 2^k factors INTER_LINEAR is the mean of the 2x2 centre pixels."""
    _, mk = synth.warps_and_masks(5, "mp", 2, 32, 16)
    m = t(mk)
    for f in (2, 4, 8):
        got = R.mask_pyramid(m, 32 // f, 16 // f)
        c = f // 2 - 1
        blocks = m.view(2, 10, 32 // f, f, 16 // f, f)
        want = blocks[:, :, :, c:c + 2, :, c:c + 2].mean(dim=(3, 5))
        assert torch.equal(got, want.float())


def test_aten_warp_matches_closed_form():
    """The timed CPU baseline uses the ATen affine_grid/grid_sample path; it must agree with the closed form."""
    feat = t(synth.normal(12, "aw/f", (2, 8, 32, 24)))
    wr, mk = synth.warps_and_masks(12, "aw", 2, 128, 96)
    a = R.warp_mask_max(feat, t(wr), t(mk), (128, 96))
    b = R.warp_mask_max(feat, t(wr), t(mk), (128, 96), aten=True)
    assert np.abs(a.numpy() - b.numpy()).max() < 3e-4


def test_host_cords_to_map_vs_reference_golden():
    """Host-side mirror of the reference's cords_to_map (kept for numpy callers) against the captured reference maps."""
    import pta_bootstrap
    pta_bootstrap.load()
    from pose_transfer_amd.utils import pose_utils as PU
    g = np.load(os.path.join(GOLDEN, "heatmaps.npz"))
    for tag in ("a", "b"):
        cords, ref = g[tag + "_cords"], g[tag + "_maps"]
        for n in range(cords.shape[0]):
            assert np.array_equal(PU.cords_to_map(cords[n], ref.shape[1:3]), ref[n])
