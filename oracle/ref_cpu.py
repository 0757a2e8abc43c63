"""ORACLE — CPU restatement of the reference's Deformable-GAN training path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it; the product package
(``pose-transfer_amd/``) never does and fails loudly when its HIP library is missing.

What it is: the algorithm of ``/root/reference/src_deformable`` (``dis_update`` + ``gen_update``
at ``warp_skip=mask``, ``gen_type=baseline``) written again from the closed forms in SURVEY.md
App. A, in plain PyTorch **CPU** ops, functional style (explicit parameter dicts keyed by the
reference's ``state_dict`` names, explicit dropout masks).  Each function cites the reference
lines it restates.  Arithmetic that the reference itself delegates to PyTorch ATen
(conv2d / conv_transpose2d / Adam) is delegated to the same ATen CPU kernels here; everything
the reference composes itself (per-sample norm, affine warp + mask + max, mask pyramid,
VGG pre-process quirk, nearest-neighbour loss, GAN log losses, update order) is spelled out.

Pinning status (see tests/golden/README.md and DESIGN.md §oracle):
  * pinned against tensors captured from the imported reference (tests/golden/*.npz,
    generator: oracle/make_golden.py) — generator, discriminator, warp layer, nn-loss, VGG
    extractor, two full dis_update+gen_update iterations;
  * ``mask_pyramid`` restates OpenCV INTER_LINEAR, but cv2 is absent here (the golden run uses the
    same documented formula as a shim): **parity unpinned** for non-integer scale factors;
  * VGG-19 ImageNet weights are not available offline: arithmetic pinned, trained values unpinned.
"""
import math

import torch
import torch.nn.functional as F

T_WARPS = 10
NORM_EPS = 1e-3          # reference models/networks.py:159  InstanceNorm3d(1, eps=1e-3, affine=True)
VGG_MEAN = (0.485, 0.456, 0.406)   # reference utils/pose_utils.py:326
VGG_STD = (0.229, 0.224, 0.225)    # reference utils/pose_utils.py:327


# --------------------------------------------------------------------------------------------- a1
def sample_norm(x, gamma, beta, eps=NORM_EPS):
    """Per-sample normalisation over (C,H,W) with ONE scalar gamma/beta.
    reference models/networks.py:159,166-169 — InstanceNorm3d(1) applied on x.unsqueeze(1)."""
    n = x.shape[0]
    flat = x.reshape(n, -1)
    mu = flat.mean(dim=1)
    var = flat.var(dim=1, unbiased=False)
    shp = (n,) + (1,) * (x.dim() - 1)
    return (x - mu.view(shp)) / torch.sqrt(var.view(shp) + eps) * gamma + beta


def _drop(x, mask):
    """Channel dropout with an explicit multiplier mask (N,C) in {0, 1/(1-p)}.
    reference models/networks.py:161 nn.Dropout2d() (train mode everywhere: no .eval() in main.py)."""
    if mask is None:
        return x
    return x * mask.view(mask.shape[0], mask.shape[1], 1, 1)


def block_down(x, w, gamma=None, beta=None, leaky=True):
    """reference models/networks.py:142-172 Block(down=True): act -> Conv2d(k4,s2,p1,no bias) -> [norm]."""
    x = F.leaky_relu(x, 0.2) if leaky else F.relu(x)
    y = F.conv2d(x, w, None, stride=2, padding=1)
    if gamma is not None:
        y = sample_norm(y, gamma, beta)
    return y


def block_up(x, w, gamma=None, beta=None, drop=None):
    """reference models/networks.py:156-161 Block(down=False, leaky=False):
    ReLU -> ConvTranspose2d(k4,s2,no bias) -> Cropping2D(1) -> norm -> [Dropout2d]."""
    y = F.conv_transpose2d(F.relu(x), w, None, stride=2)
    y = y[:, :, 1:-1, 1:-1]                      # networks.py:134-139 Cropping2D(1)
    if gamma is not None:
        y = sample_norm(y, gamma, beta)
    return _drop(y, drop)


# --------------------------------------------------------------------------------------------- a2
def encoder_forward(x, p, prefix, nlev):
    """reference models/networks.py:175-202: Conv2d(k3,p1,bias) then Blocks; returns ALL level outputs."""
    outs = [F.conv2d(x, p[prefix + ".net.0.weight"], p[prefix + ".net.0.bias"], padding=1)]
    for i in range(1, nlev):
        g = p.get("%s.net.%d.net.2.weight" % (prefix, i))
        b = p.get("%s.net.%d.net.2.bias" % (prefix, i))
        outs.append(block_down(outs[-1], p["%s.net.%d.net.1.weight" % (prefix, i)], g, b))
    return outs


# --------------------------------------------------------------------------------------------- a3
def decoder_forward(skips, p, ndec, drops=None):
    """reference models/networks.py:204-250: cat([out, skip]) BEFORE each block; dropout on blocks 0-2;
    tail ReLU -> Conv2d(k3,p1,bias) -> Tanh."""
    out = None
    for i in range(ndec):
        x = skips[-(i + 1)] if i == 0 else torch.cat([out, skips[-(i + 1)]], 1)
        if i < ndec - 1:
            d = drops[i] if (drops is not None and i < 3) else None
            out = block_up(x, p["decoder.net.%d.net.1.weight" % i],
                           p["decoder.net.%d.net.3.weight" % i], p["decoder.net.%d.net.3.bias" % i], d)
        else:
            out = F.conv2d(F.relu(x), p["decoder.net.%d.weight" % (i + 1)],
                           p["decoder.net.%d.bias" % (i + 1)], padding=1)
    return torch.tanh(out)


# --------------------------------------------------------------------------------------------- a6 (mask pyramid)
def mask_pyramid(masks, h, w):
    """cv2.resize(mask_HWT, (w, h)) with the default INTER_LINEAR, restated:
    source coordinate sx = (j+0.5)*(W0/w) - 0.5, the two taps clamped to the edge; identity when the
    size is unchanged.  reference utils/pose_transform.py:84-87.  Computed in float64 like the
    reference (masks stay float64 until `.float()` at :87), result cast to float32.
    PARITY UNPINNED against real OpenCV (absent offline); exact for the 2^k factors of every config."""
    n, t, h0, w0 = masks.shape
    m = masks.to(torch.float64)
    if (h0, w0) == (h, w):
        return m.to(torch.float32)

    def taps(dst, src):
        s = (torch.arange(dst, dtype=torch.float64) + 0.5) * (src / dst) - 0.5
        i0 = torch.floor(s)
        f = s - i0
        i0 = i0.to(torch.int64)
        i1 = i0 + 1
        f = torch.where(i0 < 0, torch.zeros_like(f), f)
        return i0.clamp(0, src - 1), i1.clamp(0, src - 1), f

    y0, y1, fy = taps(h, h0)
    x0, x1, fx = taps(w, w0)
    top = m[:, :, y0][:, :, :, x0] * (1 - fx) + m[:, :, y0][:, :, :, x1] * fx
    bot = m[:, :, y1][:, :, :, x0] * (1 - fx) + m[:, :, y1][:, :, :, x1] * fx
    out = top * (1 - fy).view(1, 1, h, 1) + bot * fy.view(1, 1, h, 1)
    return out.to(torch.float32)


# --------------------------------------------------------------------------------------------- a5
def normalize_transforms(warps6, h, w):
    """reference utils/pose_transform.py:48-58 — IN-PLACE, SEQUENTIAL: theta02 uses the updated theta01.
    warps6: (..., 6) = [a0 a1 a2 b0 b1 b2] in pixel units (inverse map, output -> input)."""
    a0, a1, a2, b0, b1, b2 = [warps6[..., i] for i in range(6)]
    t00 = a0
    t01 = a1 * w / h
    t02 = a2 * 2 / h + t00 + t01 - 1
    t10 = b0 * h / w
    t11 = b1
    t12 = b2 * 2 / w + t10 + t11 - 1
    return t00, t01, t02, t10, t11, t12


def affine_sample(feat, warps, init_size, align_corners=False):
    """AffineLayer.forward: replicate, normalise theta, affine_grid, bilinear grid_sample (zeros padding).
    reference utils/pose_transform.py:20-46 (+ :72-76 for the per-level translation rescale).
    Closed form of SURVEY.md App. A.2 steps 1-4, evaluated in the reference's fp32 operation order.
    Returns (N,T,C,h,w)."""
    n, c, h, w = feat.shape
    t = warps.shape[1]
    mul = torch.tensor([1, 1, init_size[0] / h, 1, 1, init_size[1] / w, 1, 1], dtype=feat.dtype)
    wr = (warps.to(feat.dtype) / mul)[..., :6]                                    # pose_transform.py:72-76
    t00, t01, t02, t10, t11, t12 = normalize_transforms(wr, h, w)                 # each (N,T)
    jj = torch.arange(w, dtype=feat.dtype)
    ii = torch.arange(h, dtype=feat.dtype)
    if align_corners:
        xs = jj * 2 / (w - 1) - 1 if w > 1 else torch.zeros_like(jj)
        ys = ii * 2 / (h - 1) - 1 if h > 1 else torch.zeros_like(ii)
    else:
        xs = (jj * 2 + 1) / w - 1                                                  # torch affine_grid base grid
        ys = (ii * 2 + 1) / h - 1
    xs = xs.view(1, 1, 1, w)
    ys = ys.view(1, 1, h, 1)
    e = lambda v: v.view(n, t, 1, 1)
    gx = e(t00) * xs + e(t01) * ys + e(t02)
    gy = e(t10) * xs + e(t11) * ys + e(t12)
    if align_corners:
        ix = (gx + 1) / 2 * (w - 1)
        iy = (gy + 1) / 2 * (h - 1)
    else:
        ix = ((gx + 1) * w - 1) / 2
        iy = ((gy + 1) * h - 1) / 2
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    fx = ix - x0
    fy = iy - y0
    x0 = x0.to(torch.int64)
    y0 = y0.to(torch.int64)
    flat = feat.reshape(n, 1, c, h * w).expand(n, t, c, h * w)
    out = torch.zeros(n, t, c, h, w, dtype=feat.dtype)
    for dy, dx, wgt in ((0, 0, (1 - fx) * (1 - fy)), (0, 1, fx * (1 - fy)),
                        (1, 0, (1 - fx) * fy), (1, 1, fx * fy)):
        xx = x0 + dx
        yy = y0 + dy
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        idx = (yy.clamp(0, h - 1) * w + xx.clamp(0, w - 1)).view(n, t, 1, h * w).expand(n, t, c, h * w)
        v = torch.gather(flat, 3, idx).view(n, t, c, h, w)
        out = out + v * (wgt * ok.to(feat.dtype)).view(n, t, 1, h, w)
    return out


def affine_sample_aten(feat, warps, init_size, align_corners=False):
    """Same as affine_sample but through the very ATen calls the reference makes (F.affine_grid +
    F.grid_sample, utils/pose_transform.py:37-39).  Used for the timed CPU baseline (bench.py) so that the
    baseline pays the reference's cost profile; agrees with the closed form to ~1e-4 (tests/test_oracle_golden.py)."""
    n, c, h, w = feat.shape
    t = warps.shape[1]
    mul = torch.tensor([1, 1, init_size[0] / h, 1, 1, init_size[1] / w, 1, 1], dtype=feat.dtype)
    wr = (warps.to(feat.dtype) / mul)[..., :6]
    t00, t01, t02, t10, t11, t12 = normalize_transforms(wr, h, w)
    theta = torch.stack([t00, t01, t02, t10, t11, t12], -1).view(n * t, 2, 3)
    x = feat.unsqueeze(1).expand(n, t, c, h, w).reshape(n * t, c, h, w)
    grid = F.affine_grid(theta, (n * t, c, h, w), align_corners=align_corners)
    return F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=align_corners).view(n, t, c, h, w)


# --------------------------------------------------------------------------------------------- a6
def warp_mask_max(feat, warps, masks, init_size, align_corners=False, aten=False):
    """AffineTransformLayer.forward at warp_skip='mask': warp T copies, multiply by the resized masks,
    max over T.  reference utils/pose_transform.py:69-92.  masks: (N,T,H0,W0) at full resolution."""
    n, c, h, w = feat.shape
    warped = (affine_sample_aten if aten else affine_sample)(feat, warps, init_size, align_corners)
    if masks is None:          # warp_skip 'full' / 'none': one transform, no mask (pose_transform.py:80,88)
        return warped.max(dim=1)[0]
    m = mask_pyramid(masks, h, w).view(n, -1, 1, h, w).to(feat.dtype)
    return (warped * m).max(dim=1)[0]


# --------------------------------------------------------------------------------------------- a4
def split_input(x, pose_dim):
    """get_imgpose: reference utils/pose_utils.py:227-233 (use_input_pose=True)."""
    return x[:, :3], x[:, 3:3 + pose_dim], x[:, 3 + pose_dim:]


def generator_forward(inp, warps, masks, p, pose_dim, nfilters_enc, nfilters_dec, init_size,
                      drops=None, align_corners=False, return_skips=False, aten_warp=False):
    """Deformable_Generator.forward + concatenate_skips: reference models/networks.py:269-288.
    masks=None is warp_skip 'full' / 'none': warps (N,1,>=6), a single unmasked transform (networks.py:283)."""
    img, src_pose, tgt_pose = split_input(inp, pose_dim)
    nlev = len(nfilters_enc)
    sk_app = encoder_forward(torch.cat([img, src_pose], 1), p, "encoder_app", nlev)
    sk_pose = encoder_forward(tgt_pose, p, "encoder_pose", nlev)
    skips = []
    for i, (a, q) in enumerate(zip(sk_app, sk_pose)):
        if i < 4:                                                                  # networks.py:282
            a = warp_mask_max(a, warps, masks, init_size, align_corners, aten_warp)
        skips.append(torch.cat([a, q], 1))
    out = decoder_forward(skips, p, len(nfilters_dec), drops)
    return (out, sk_app, sk_pose, skips) if return_skips else out


def stacked_generator_forward(inp, target_pose, warps, masks, p, pose_dim, num_stacks, nfilters_enc, nfilters_dec,
                              init_size, drops=None, align_corners=False):
    """Stacked_Generator.forward (reference models/networks.py:306-327): the SAME generator applied num_stacks
    times; stage i sees [previous output, pose_{i-1}, pose_i] (stage 0: [image, input pose, pose_0]).
    target_pose (N, num_stacks*P, H, W); warps (N, num_stacks, T, 8); masks (N, num_stacks, T, H, W) | None;
    drops: per-stage list of dropout-mask lists | None.  Returns the list of stage outputs."""
    img, init_pose, _ = split_input(inp, pose_dim)
    P = pose_dim
    outs = []
    out = None
    for i in range(num_stacks):
        tp = target_pose[:, i * P:(i + 1) * P]
        if i == 0:
            x = torch.cat([img, init_pose, tp], 1)
        else:
            x = torch.cat([out, target_pose[:, (i - 1) * P:i * P], tp], 1)
        out = generator_forward(x, warps[:, i], None if masks is None else masks[:, i], p, P, nfilters_enc,
                                nfilters_dec, init_size, None if drops is None else drops[i], align_corners)
        outs.append(out)
    return outs


def baseline_generator_forward(inp, p, nfilters_enc, nfilters_dec, drops=None):
    """src_baseline Generator: one encoder over the whole input, decoder(num_skips=1), no warps.
    reference src_baseline/models/networks.py:238-253."""
    skips = encoder_forward(inp, p, "encoder", len(nfilters_enc))
    return decoder_forward(skips, p, len(nfilters_dec), drops)


# --------------------------------------------------------------------------------------------- a7
def discriminator_forward(x, p):
    """reference models/networks.py:337-357: Conv2d(k4,s2,p0,bias) -> Blocks -> Block(.,1,bn=False)
    -> Sigmoid -> Flatten."""
    y = F.conv2d(x, p["net.0.weight"], p["net.0.bias"], stride=2)
    i = 1
    while ("net.%d.net.1.weight" % i) in p:
        y = block_down(y, p["net.%d.net.1.weight" % i], p.get("net.%d.net.2.weight" % i),
                       p.get("net.%d.net.2.bias" % i))
        i += 1
    return torch.sigmoid(y).reshape(y.shape[0], -1)


# --------------------------------------------------------------------------------------------- a12
def vgg_preprocess(x):
    """The reference's `x.view(N,H,W,C)` is a REINTERPRET, not a permute: per sample, the element at
    flat offset k gets mean[k%3], std[k%3].  reference utils/pose_utils.py:322-331."""
    n, c, h, w = x.shape
    mean = torch.tensor(VGG_MEAN, dtype=x.dtype)
    std = torch.tensor(VGG_STD, dtype=x.dtype)
    v = x.reshape(n, -1, 3)
    return ((v - mean) / std).reshape(n, c, h, w)


def vgg_features(x, w, b):
    """Feature_Extractor(layer 'block1_conv2') = features[0..1] = ReLU(conv1_1(prep(x))) only.
    reference utils/pose_utils.py:312-338 (get_layer_ind -> 0+2-1 = 1)."""
    return F.relu(F.conv2d(vgg_preprocess(x), w, b, padding=1))


# --------------------------------------------------------------------------------------------- a11
def nn_loss(pred, gt, nh=3, nw=3):
    """Nearest-neighbour L1: pad GT by -10000, min over nh*nw shifts of sum_c |ref - pred|, mean over (N,H,W).
    reference models/pose_gan.py:173-199.  Evaluated shift-by-shift (no 25x materialisation)."""
    vp, hp = nh // 2, nw // 2
    # ConstantPad2d((v_pad, v_pad, h_pad, h_pad)): (left, right, top, bottom) — pose_gan.py:176
    padded = F.pad(gt, (vp, vp, hp, hp), value=-10000.0)
    n, c, h, w = pred.shape
    best = None
    for i in range(nh):
        for j in range(nw):
            d = (padded[:, :, i:i + h, j:j + w] - pred).abs().sum(dim=1)
            best = d if best is None else torch.minimum(best, d)
    return best.mean()


# --------------------------------------------------------------------------------------------- A.6
def gan_logloss(o, real):
    """sum_n -mean_k log(o_nk + 1e-7)  (real / generator)  or  sum_n -mean_k log(1 - o_nk + 1e-7) (fake).
    reference models/pose_gan.py:90-98,140-160 (per-sample python loop, summed)."""
    v = torch.log(o + 1e-7) if real else torch.log(1 - o + 1e-7)
    return -(v.mean(dim=1)).sum()


def disc_input(inp, judged, pose_dim):
    """[img(3), src_pose(P), image_to_judge(3), tgt_pose(P)] — reference models/pose_gan.py:84-86,131-135."""
    img, sp, tp = split_input(inp, pose_dim)
    return torch.cat([img, sp, judged, tp], 1)


# --------------------------------------------------------------------------------------------- a13
class Adam:
    """torch.optim.Adam(lr, betas=(0.5,0.999), eps=1e-8) restated (no weight decay / amsgrad).
    reference models/pose_gan.py:50-51."""

    def __init__(self, params, lr=2e-4, b1=0.5, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.t = 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    def step(self, params, grads):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for k in params:
            g = grads[k]
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            params[k] = params[k] - (self.lr / bc1) * (self.m[k] / denom)
        return params


# --------------------------------------------------------------------------------------------- a9 / a10 / a14
class Trainer:
    """Functional restatement of DeformablePose_GAN (reference models/pose_gan.py:11-171).

    cfg keys: pose_dim, image_size, batch_size, gan_penalty_weight, l1_penalty_weight, learning_rate,
    content_loss_layer ('none' | 'block1_conv2'), nn_loss_area_size, align_corners, deformable."""

    def __init__(self, cfg, gen_params, disc_params, vgg=None):
        self.cfg = dict(cfg)
        self.gp = {k: v.clone() for k, v in gen_params.items()}
        self.dp = {k: v.clone() for k, v in disc_params.items()}
        self.vgg = vgg
        self.gopt = Adam(self.gp, cfg.get("learning_rate", 2e-4))
        self.dopt = Adam(self.dp, cfg.get("learning_rate", 2e-4))
        self.enc, self.dec = cfg["nfilters_enc"], cfg["nfilters_dec"]
        self.last_gen_grads = None
        self.last_disc_grads = None

    def gen(self, gp, inp, warps, masks, drops):
        c = self.cfg
        if c.get("gen_type", "baseline") == "stacked":
            # warps = {'interpol_pose', 'interpol_warps', 'interpol_masks'} (reference pose_gan.py:72-77)
            return stacked_generator_forward(inp, warps["interpol_pose"], warps["interpol_warps"],
                                             warps.get("interpol_masks"), gp, c["pose_dim"], c["num_stacks"], self.enc,
                                             self.dec, c["image_size"], drops, c.get("align_corners", False))[-1]
        if c.get("deformable", True):
            return generator_forward(inp, warps, masks, gp, c["pose_dim"], self.enc, self.dec,
                                     c["image_size"], drops, c.get("align_corners", False),
                                     aten_warp=c.get("aten_warp", False))
        return baseline_generator_forward(inp, gp, self.enc, self.dec, drops)

    def dis_update(self, inp, target, warps, masks, real_inp, real_target, drops=None, average_fn=None):
        """reference models/pose_gan.py:117-171.  out_gen is detached: bit-identical."""
        c = self.cfg
        with torch.no_grad():
            out_gen = self.gen(self.gp, inp, warps, masks, drops)
        dp = {k: v.clone().requires_grad_(True) for k, v in self.dp.items()}
        data = torch.cat([disc_input(real_inp, real_target, c["pose_dim"]),
                          disc_input(inp, out_gen, c["pose_dim"])], 0)
        res = discriminator_forward(data, dp)
        nb = c["batch_size"]
        l_true = gan_logloss(res[:nb], True) * c["gan_penalty_weight"] / nb
        l_fake = gan_logloss(res[nb:], False) * c["gan_penalty_weight"] / nb
        loss = l_true + l_fake
        grads = dict(zip(dp.keys(), torch.autograd.grad(loss, list(dp.values()))))
        if average_fn is not None:
            grads = average_fn(grads)
        self.last_disc_grads = grads
        self.dp = self.dopt.step(self.dp, grads)
        return [loss.item(), l_true.item(), l_fake.item()]

    def gen_update(self, inp, target, warps, masks, drops=None, average_fn=None):
        """reference models/pose_gan.py:69-115."""
        c = self.cfg
        gp = {k: v.clone().requires_grad_(True) for k, v in self.gp.items()}
        out_gen = self.gen(gp, inp, warps, masks, drops)
        out_dis = discriminator_forward(disc_input(inp, out_gen, c["pose_dim"]), self.dp)
        ad = gan_logloss(out_dis, True)
        if c.get("content_loss_layer", "none") != "none":
            fg = vgg_features(out_gen, self.vgg[0], self.vgg[1])
            ft = vgg_features(target, self.vgg[0], self.vgg[1])
            ll = nn_loss(fg, ft, c["nn_loss_area_size"], c["nn_loss_area_size"])
        else:
            ll = (out_gen - target).abs().mean()                                   # nn.L1Loss, pose_gan.py:66,105
        ad = ad * c["gan_penalty_weight"] / c["batch_size"]
        ll = ll * c["l1_penalty_weight"]
        total = ad + ll
        grads = dict(zip(gp.keys(), torch.autograd.grad(total, list(gp.values()))))
        if average_fn is not None:
            grads = average_fn(grads)
        self.last_gen_grads = grads
        self.gp = self.gopt.step(self.gp, grads)
        return out_gen.detach(), [total.item(), ll.item(), ad.item()]
