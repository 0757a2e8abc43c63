"""Golden-vector generator: runs the REAL reference (imported read-only from /root/reference) on
seeded synthetic inputs and stores its outputs as small fixtures under tests/golden/.

Runs ONLY in the build container (needs /root/reference); the fixtures it writes are data
(inputs are regenerated from ``pose_transfer_amd.utils.synth``; only outputs are stored).
Nothing from the reference is copied: it is imported under the shims SURVEY.md §8c lists
(missing third-party modules stubbed, ``.cuda()`` made a no-op, ``torch.load`` of the
authors' private checkpoint neutralised).

    python oracle/make_golden.py            # regenerates tests/golden/*.npz
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pta_bootstrap  # noqa: E402

pta_bootstrap.load()
from pose_transfer_amd.utils import synth  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
_SENTINEL = object()
_DROP_QUEUE = []          # explicit (N,C) multipliers consumed by the patched Dropout2d
_ALIGN = {"v": None}      # None = framework default; True = torch-0.4 behaviour


def cv2_resize_shim(img, dsize, *a, **k):
    """OpenCV INTER_LINEAR as documented: half-pixel centres, edge clamp; identity at same size."""
    w, h = dsize
    h0, w0 = img.shape[:2]
    if (h0, w0) == (h, w):
        return img.copy()

    def taps(dst, src):
        s = (np.arange(dst, dtype=np.float64) + 0.5) * (src / dst) - 0.5
        i0 = np.floor(s)
        f = s - i0
        i0 = i0.astype(np.int64)
        f = np.where(i0 < 0, 0.0, f)
        return np.clip(i0, 0, src - 1), np.clip(i0 + 1, 0, src - 1), f

    y0, y1, fy = taps(h, h0)
    x0, x1, fx = taps(w, w0)
    fx = fx.reshape(1, w, *([1] * (img.ndim - 2)))
    fy = fy.reshape(h, 1, *([1] * (img.ndim - 2)))
    top = img[y0][:, x0] * (1 - fx) + img[y0][:, x1] * fx
    bot = img[y1][:, x0] * (1 - fx) + img[y1][:, x1] * fx
    return top * (1 - fy) + bot * fy


def install_shims(src_dir):
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("keras")
    mod("keras.optimizers", Adam=object)
    dummy = lambda *a, **k: None
    for name in ("skimage", "skimage.io", "skimage.transform", "skimage.measure", "skimage.draw"):
        mod(name, imread=dummy, warp_coords=dummy, circle=dummy, line_aa=dummy, polygon=dummy,
            estimate_transform=dummy, grid_points_in_poly=dummy)
    sys.modules["skimage"].measure = sys.modules["skimage.measure"]
    sys.modules["skimage"].transform = sys.modules["skimage.transform"]
    mod("cv2", resize=cv2_resize_shim)
    try:
        import scipy.ndimage.filters  # noqa: F401
    except Exception:
        import scipy.ndimage as ndi
        mod("scipy.ndimage.filters", gaussian_filter=ndi.gaussian_filter)

    def vgg19(pretrained=False):
        cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
        layers, cin = [], 3
        for v in cfg:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        m = nn.Module()
        m.features = nn.Sequential(*layers)
        return m

    tv = mod("torchvision")
    tv.models = mod("torchvision.models", vgg19=vgg19)

    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    torch.load = lambda *a, **k: _SENTINEL
    _orig_lsd = nn.Module.load_state_dict

    def lsd(self, sd, *a, **k):
        if sd is _SENTINEL:
            return None
        return _orig_lsd(self, sd, *a, **k)

    nn.Module.load_state_dict = lsd

    def drop_fwd(self, x):
        if not _DROP_QUEUE:
            return x
        m = _DROP_QUEUE.pop(0)
        return x * m.view(m.shape[0], m.shape[1], 1, 1)

    nn.Dropout2d.forward = drop_fwd

    import torch.nn.functional as F
    _ag, _gs = F.affine_grid, F.grid_sample

    def ag(theta, size, align_corners=None):
        return _ag(theta, size, align_corners=_ALIGN["v"] if _ALIGN["v"] is not None else False)

    def gs(inp, grid, mode="bilinear", padding_mode="zeros", align_corners=None):
        return _gs(inp, grid, mode=mode, padding_mode=padding_mode,
                   align_corners=_ALIGN["v"] if _ALIGN["v"] is not None else False)

    F.affine_grid, F.grid_sample = ag, gs
    for k in [k for k in sys.modules if k in ("models", "utils", "opts") or k.startswith(("models.", "utils."))]:
        del sys.modules[k]
    sys.path.insert(0, src_dir)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def summarize(x):
    """sum, abs-sum, max-abs + 32 strided samples: enough to pin a big tensor in ~150 bytes."""
    f = x.detach().reshape(-1).double()
    idx = torch.linspace(0, f.numel() - 1, 32).long()
    return np.concatenate([[f.sum().item(), f.abs().sum().item(), f.abs().max().item()],
                           f[idx].numpy()]).astype(np.float64)


def load_sd(module, params):
    module.load_state_dict({k: t(v) for k, v in params.items()})


def heatmap_fixture(rpu):
    """tests/golden/heatmaps.npz: the reference's cords_to_map (utils/pose_utils.py:79-86) on seeded key-points with
    missing entries (-1), integer coordinates as load_pose_cords_from_strings produces them."""
    fix = {}
    for tag, (h, w), p in (("a", (32, 24), 18), ("b", (20, 28), 16)):
        ky = np.floor(synth.uniform(31, "hm/%s/y" % tag, (2, p)) * h).astype(np.int64)
        kx = np.floor(synth.uniform(31, "hm/%s/x" % tag, (2, p)) * w).astype(np.int64)
        miss = synth.uniform(31, "hm/%s/m" % tag, (2, p)) < 0.2
        ky[miss] = -1
        kx[miss & (synth.uniform(31, "hm/%s/m2" % tag, (2, p)) < 0.5)] = -1
        cords = np.stack([ky, kx], -1)                       # (2, p, 2) as (y, x)
        fix[tag + "_cords"] = cords
        fix[tag + "_maps"] = np.stack([rpu.cords_to_map(cords[n], (h, w)) for n in range(2)])    # (2, h, w, p)
    np.savez_compressed(os.path.join(OUT, "heatmaps.npz"), **fix)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    install_shims(os.path.join(REF, "src_deformable"))
    if "--only-heatmaps" in sys.argv:
        from utils import pose_utils as rpu
        heatmap_fixture(rpu)
        return
    from models import networks as rnet
    from models import pose_gan as rgan
    from utils import pose_transform as rpt
    from utils import pose_utils as rpu

    P = 18
    # ------------------------------------------------------------------ per-op fixtures
    ops = {}
    x = t(synth.normal(11, "blk/x", (2, 8, 12, 10)) * 1.7 + 0.3)
    blk = rnet.Block(8, 16)
    load_sd(blk, {"net.1.weight": synth.xavier_uniform(11, "blk/w", (16, 8, 4, 4)),
                  "net.2.weight": np.array([1.3], np.float32), "net.2.bias": np.array([-0.2], np.float32)})
    ops["block_down"] = blk(x).detach().numpy()
    xu = t(synth.normal(11, "blku/x", (2, 8, 5, 6)))
    blku = rnet.Block(8, 4, down=False, leaky=False, dropout=True)
    load_sd(blku, {"net.1.weight": synth.xavier_uniform(11, "blku/w", (8, 4, 4, 4)),
                   "net.3.weight": np.array([0.7], np.float32), "net.3.bias": np.array([0.1], np.float32)})
    dm = t(synth.dropout_masks(11, "blku", 2, (4,))[0])
    _DROP_QUEUE.append(dm)
    ops["block_up"] = blku(xu).detach().numpy()

    warp_cases = [("w256s4", (256, 256), 4, 8), ("w128x64s2", (128, 64), 2, 8), ("w224s8", (224, 224), 8, 8),
                  ("w64s1", (64, 64), 1, 4), ("w96x80s2", (96, 80), 2, 4)]
    for name, (H0, W0), s, c in warp_cases:
        h, w = H0 // s, W0 // s
        feat = t(synth.normal(12, name + "/f", (2, c, h, w))).requires_grad_(True)
        wr, mk = synth.warps_and_masks(12, name, 2, H0, W0)
        for ac in (None, True):
            _ALIGN["v"] = ac
            lay = rpt.AffineTransformLayer(10, (H0, W0), "mask")
            out = lay(feat, t(wr), t(mk).double())
            go = t(synth.normal(12, name + "/go", tuple(out.shape)))
            (gin,) = torch.autograd.grad((out * go).sum(), feat)
            tag = name + ("_ac1" if ac else "_ac0")
            ops[tag + "_out"] = out.detach().numpy()
            ops[tag + "_gin"] = gin.numpy()
        _ALIGN["v"] = None

    gm = rgan.DeformablePose_GAN.__new__(rgan.DeformablePose_GAN)   # nn_loss only needs `self`
    for a in (3, 5):
        pred = t(synth.normal(13, "nn%d/p" % a, (2, 6, 12, 9))).requires_grad_(True)
        gt = t(synth.normal(13, "nn%d/g" % a, (2, 6, 12, 9)))
        l = rgan.DeformablePose_GAN.nn_loss(gm, pred, gt, a, a)
        (g,) = torch.autograd.grad(l, pred)
        ops["nn%d_loss" % a] = np.array(l.item())
        ops["nn%d_grad" % a] = g.numpy()

    import torchvision.models as tvm
    vgg = tvm.vgg19()
    vw = synth.xavier_uniform(14, "vgg/w", (64, 3, 3, 3))
    vb = synth.uniform(14, "vgg/b", (64,), -0.1, 0.1)
    vgg.features[0].weight.data = t(vw)
    vgg.features[0].bias.data = t(vb)
    vx = t(synth.uniform(14, "vgg/x", (2, 3, 10, 14), -1, 1))
    ops["vgg_feat"] = rpu.Feature_Extractor(vgg, input=vx, layer_name="block1_conv2").detach().numpy()
    ops["layer_inds"] = np.array([rpu.get_layer_ind("block1_conv2"), rpu.get_layer_ind("block4_conv1")])

    dspec = synth.discriminator_spec(3 + 2 * P + 3)
    dpar = synth.init_params(15, "disc", dspec, norm_jitter=0.2)
    disc = rnet.Discriminator(3 + 2 * P + 3)
    load_sd(disc, dpar)
    dx = t(synth.uniform(15, "disc/x", (3, 42, 64, 64), -1, 1))
    ops["disc_out"] = disc(dx).detach().numpy()
    dx2 = t(synth.uniform(15, "disc/x2", (2, 42, 96, 80), -1, 1))
    ops["disc_out_96x80"] = disc(dx2).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **ops)
    heatmap_fixture(rpu)

    # ------------------------------------------------------------------ whole generator, 6 levels, 64x64
    gen_fix = {}
    for name, (H, W) in (("g64", (64, 64)), ("g64x32", (64, 32))):
        enc, dec = synth.nfilters((H, W))
        gpar = synth.init_params(21, name, synth.generator_spec(P, enc, dec), norm_jitter=0.2)
        gen = rnet.Deformable_Generator(3 + 2 * P, P, (H, W), enc, dec, "mask")
        load_sd(gen, gpar)
        inp, tgt, wr, mk = synth.batch(21, name, 2, P, H, W)
        for mode in ("eval", "train"):
            del _DROP_QUEUE[:]
            if mode == "train":
                _DROP_QUEUE.extend(t(m) for m in synth.dropout_masks(21, name, 2))
            out = gen(t(inp), t(wr), t(mk).double())
            gen_fix["%s_%s_out" % (name, mode)] = out.detach().numpy()
    # 7-level net at 128x128 (the >=256 architecture, pose_gan.py:17-18) — summary only
    enc7, dec7 = synth.nfilters((256, 256))
    gpar = synth.init_params(22, "g128", synth.generator_spec(P, enc7, dec7), norm_jitter=0.2)
    gen = rnet.Deformable_Generator(3 + 2 * P, P, (128, 128), enc7, dec7, "mask")
    load_sd(gen, gpar)
    inp, tgt, wr, mk = synth.batch(22, "g128", 2, P, 128, 128)
    del _DROP_QUEUE[:]
    _DROP_QUEUE.extend(t(m) for m in synth.dropout_masks(22, "g128", 2))
    out = gen(t(inp), t(wr), t(mk).double())
    gen_fix["g128_train_out_n0"] = out[0].detach().numpy()
    gen_fix["g128_train_summary"] = summarize(out)
    np.savez_compressed(os.path.join(OUT, "generator.npz"), **gen_fix)

    # ------------------------------------------------------------------ two full training iterations
    for name, content, area, l1w in (("step_l1", "none", 1, 100.0), ("step_nn", "block1_conv2", 5, 0.01)):
        H = W = 64
        N = 2
        enc, dec = synth.nfilters((H, W))
        opt = types.SimpleNamespace(image_size=(H, W), use_input_pose=True, pose_dim=P, batch_size=N,
                                    num_stacks=4, gen_type="baseline", dataset="fasion", warp_skip="mask",
                                    learning_rate=2e-4, content_loss_layer=content, nn_loss_area_size=area,
                                    gan_penalty_weight=1.0, l1_penalty_weight=l1w)
        model = rgan.DeformablePose_GAN(opt)
        load_sd(model.gen, synth.init_params(31, name + "/gen", synth.generator_spec(P, enc, dec), 0.1))
        load_sd(model.disc, synth.init_params(31, name + "/disc", dspec, 0.1))
        if content != "none":
            model.content_model.features[0].weight.data = t(vw)
            model.content_model.features[0].bias.data = t(vb)
        fix = {}
        od = vars(opt)
        for it in range(2):
            bA = synth.batch(31, "%s/it%d/A" % (name, it), N, P, H, W)
            bB = synth.batch(31, "%s/it%d/B" % (name, it), N, P, H, W)
            bC = synth.batch(31, "%s/it%d/C" % (name, it), N, P, H, W)
            del _DROP_QUEUE[:]
            _DROP_QUEUE.extend(t(m) for m in synth.dropout_masks(31, "%s/it%d/dA" % (name, it), N))
            dl = model.dis_update(t(bA[0]), t(bA[1]), {"warps": t(bA[2]), "masks": t(bA[3]).double()},
                                  t(bB[0]), t(bB[1]), od)
            fix["it%d_dis_losses" % it] = np.array(dl)
            for k, p in model.disc.named_parameters():
                fix["it%d_dgrad_%s" % (it, k)] = summarize(p.grad)
                fix["it%d_dpar_%s" % (it, k)] = summarize(p)
            del _DROP_QUEUE[:]
            _DROP_QUEUE.extend(t(m) for m in synth.dropout_masks(31, "%s/it%d/dC" % (name, it), N))
            og, _, gl = model.gen_update(t(bC[0]), t(bC[1]), {"warps": t(bC[2]), "masks": t(bC[3]).double()}, od)
            fix["it%d_gen_losses" % it] = np.array(gl)
            fix["it%d_out_gen" % it] = og.detach().numpy()
            for k, p in model.gen.named_parameters():
                fix["it%d_ggrad_%s" % (it, k)] = summarize(p.grad)
                fix["it%d_gpar_%s" % (it, k)] = summarize(p)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **fix)
        print(name, fix["it0_dis_losses"], fix["it0_gen_losses"], fix["it1_gen_losses"])

    # ------------------------------------------------------------------ src_baseline generator + one step (config 1 plumbing)
    install_shims(os.path.join(REF, "src_baseline"))
    from models import networks as bnet
    from models import pose_gan as bgan
    H, W, N = 128, 64, 2
    enc, dec = synth.nfilters((H, W))
    opt = types.SimpleNamespace(image_size=(H, W), use_input_pose=True, pose_dim=P, batch_size=N, num_stacks=4,
                                gen_type="baseline", dataset="market", warp_skip="none", learning_rate=2e-4,
                                content_loss_layer="none", nn_loss_area_size=1, gan_penalty_weight=1.0,
                                l1_penalty_weight=100.0, checkMode=0)
    model = bgan.Pose_GAN(opt)
    gspec = synth.generator_spec(P, enc, dec, num_skips=1, deformable=False)
    load_sd(model.gen, synth.init_params(41, "base/gen", gspec, 0.1))
    load_sd(model.disc, synth.init_params(41, "base/disc", dspec, 0.1))
    fix = {}
    bA = synth.batch(41, "base/A", N, P, H, W)
    bB = synth.batch(41, "base/B", N, P, H, W)
    bC = synth.batch(41, "base/C", N, P, H, W)
    od = vars(opt)
    del _DROP_QUEUE[:]
    _DROP_QUEUE.extend(t(m) for m in synth.dropout_masks(41, "base/dA", N))
    dl = model.dis_update(t(bA[0]), t(bA[1]), None, t(bB[0]), t(bB[1]), od)
    fix["dis_losses"] = np.array(dl)
    del _DROP_QUEUE[:]
    _DROP_QUEUE.extend(t(m) for m in synth.dropout_masks(41, "base/dC", N))
    og, _, gl = model.gen_update(t(bC[0]), t(bC[1]), None, od)
    fix["gen_losses"] = np.array(gl)
    fix["out_gen"] = og.detach().numpy()
    for k, p in model.gen.named_parameters():
        fix["ggrad_" + k] = summarize(p.grad)
    np.savez_compressed(os.path.join(OUT, "baseline_step.npz"), **fix)
    print("baseline", dl, gl)


if __name__ == "__main__":
    main()
