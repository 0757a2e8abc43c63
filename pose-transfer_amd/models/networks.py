"""Network modules with the reference's surface (reference src_deformable/models/networks.py:130-357)
and ``state_dict`` key names/shapes, executed by the HIP engines (runtime/engine.py).

``Deformable_Generator(input_nc, pose_dim, image_size, nfilters_enc, nfilters_dec, warp_skip,
use_input_pose).forward(input, warps, masks)`` and ``Discriminator(input_nc).forward(x)`` take and
return NCHW fp32 tensors like the reference; parameters live in a flat packed arena
(``.arena``) and are exposed as ``nn.Parameter`` views whose ``.grad`` aliases the gradient arena,
so ``loss.backward(); optimizer.step()`` works, while the trainer (models/pose_gan.py) drives the
engines directly without an autograd tape.
"""
import os

import torch
import torch.nn as nn

from ..runtime import engine as E
from ..runtime import lib as L
from ..utils import synth


def xavier_weights_init(module, seed=0):
    """Glorot-uniform conv weights, zero biases (reference models/networks.py:26-31), norm gamma=1/beta=0."""
    spec = module.arena.spec
    module.load_state_dict({k: torch.from_numpy(v) for k, v in synth.init_params(seed, module._init_tag, spec).items()})


class _ArenaModule(nn.Module):
    """Base: owns a ParamArena; registers packed-layout parameter views; converts state_dicts."""

    def _setup(self, spec, order, device):
        self.arena = E.ParamArena(spec, order, device)
        self._pnames = {}
        for k, _ in spec:
            name = k.replace(".", "__")
            p = nn.Parameter(self.arena.p(k), requires_grad=True)
            p.grad = self.arena.g(k)
            self.register_parameter(name, p)
            self._pnames[k] = name
        self._engines = {}

    def state_dict(self, *a, **k):
        """Reference-format state_dict: OIHW conv / IOHW conv-transpose weights under the reference's keys."""
        return self.arena.state_dict()

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self.arena.keys if k not in sd]
        if missing and strict:
            raise KeyError("missing keys in state_dict: %s" % missing[:4])
        self.arena.load_state_dict(sd)

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def cuda(self, device=None):
        return self

    def _apply(self, fn, recurse=True):
        return self          # storage is the arena; .to()/.cuda() are no-ops


class _GenFn(torch.autograd.Function):
    """autograd glue of the module-level API: the engine keeps the activations of ONE forward per (batch, stage), so
    the output is saved with the node (a later forward at the same batch size overwrites `eng.out`)."""

    @staticmethod
    def forward(ctx, inp, warps, masks, mod, anchor, stage):
        eng = mod.engine(inp.shape[0], stage)
        out = eng.forward(inp.contiguous(), warps, masks).clone()
        ctx.mod, ctx.n, ctx.stage = mod, inp.shape[0], stage
        ctx.need_in = inp.requires_grad
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, gout):
        (out,) = ctx.saved_tensors
        eng = ctx.mod.engine(ctx.n, ctx.stage)
        g = gout.contiguous().clone()
        L.call("pg_tanh_bwd", L.ptr(g), L.ptr(out), g.numel(), L.stream())
        gin = None
        if ctx.need_in:       # only the 3 image channels of `input` carry a gradient (the poses are data)
            gin = torch.zeros(eng.N, 3 + 2 * eng.P, eng.H, eng.W, dtype=torch.float32, device=g.device)
            gimg = torch.empty(eng.N, 3, eng.H, eng.W, dtype=torch.float32, device=g.device)
            eng.backward(g, image_grad=gimg)
            gin[:, :3] = gimg
        else:
            eng.backward(g)
        return gin, None, None, None, None, None


class Deformable_Generator(_ArenaModule):
    """reference models/networks.py:252-288.  warp_skip='mask' (10 masked limb warps) is the hot path; 'full' and
    'none' build the SAME two-encoder network with ONE unmasked transform on levels 0-3, exactly as the reference does
    (networks.py:257 compares against 'None' with a capital N, so num_skips is always 2; :283 picks 10 or 1 transforms)."""

    _init_tag = "gen"

    def __init__(self, input_nc, pose_dim, image_size, nfilters_enc, nfilters_dec, warp_skip="mask",
                 use_input_pose=True, align_corners=False, device="cuda"):
        super().__init__()
        if warp_skip not in ("mask", "full", "none"):
            raise Exception("Invalid warp_skip")
        if not use_input_pose:
            raise Exception("use_input_pose=False is not supported")
        assert input_nc == 3 + 2 * pose_dim
        self.input_nc, self.pose_dim, self.image_size = input_nc, pose_dim, tuple(image_size)
        self.nfilters_enc, self.nfilters_dec = tuple(nfilters_enc), tuple(nfilters_dec)
        self.warp_skip, self.use_input_pose, self.num_skips = warp_skip, use_input_pose, 2
        self.align_corners = align_corners
        self.device = device
        spec = synth.generator_spec(pose_dim, self.nfilters_enc, self.nfilters_dec)
        self._setup(spec, E.generator_param_order(spec, len(self.nfilters_enc), len(self.nfilters_dec)), device)
        self._anchor = nn.Parameter(torch.zeros(1, device=device))   # keeps autograd connected to the module
        self.number_of_transforms = 10 if warp_skip == "mask" else 1
        self.drop_seed = 0

    def engine(self, n, stage=0):
        """Engine (activation buffers + schedules) of one forward at batch n; `stage` separates the chained forwards
        of the stacked generator, which must all stay alive until the backward pass."""
        # engines are built for ONE storage mode (bf16 STORAGE on the bf16 data path, runtime/engine.py)
        key = (n, stage, E.bf16_store() and getattr(self, "bf16_store_ok", True))
        if key not in self._engines:
            self._engines[key] = E.GeneratorEngine(
                self.arena, n, self.image_size[0], self.image_size[1], self.pose_dim, self.nfilters_enc,
                self.nfilters_dec, True, self.align_corners, self.device, n_warps=self.number_of_transforms,
                masked=self.warp_skip == "mask", bf16_ok=getattr(self, "bf16_store_ok", True))
            self._engines[key].drop_stream = "drop/s%d" % stage
        return self._engines[key]

    def forward(self, input, warps, masks=None, drop_masks=None, stage=0):
        """Dropout2d follows nn.Module.train()/.eval() like the reference's blocks (networks.py:161); explicit
        `drop_masks` (parity tests) override the device RNG."""
        eng = self.engine(input.shape[0], stage)
        eng.set_dropout(drop_masks, train=self.training, seed=self.drop_seed)
        return _GenFn.apply(input, warps.float(), masks, self, self._anchor, stage)


class Stacked_Generator(nn.Module):
    """reference models/networks.py:290-327: ONE Deformable_Generator applied num_stacks times with shared weights;
    stage i sees [previous output, pose_{i-1}, pose_i] (stage 0: [image, input pose, pose_0]).  state_dict keys carry
    the reference's `generator.` prefix."""

    def __init__(self, input_nc, num_stacks, image_size, pose_dim, nfilters_enc, nfilters_dec, warp_skip="mask",
                 use_input_pose=True, align_corners=False, device="cuda"):
        super().__init__()
        self.input_nc, self.num_stacks, self.pose_dim, self.image_size = input_nc, num_stacks, pose_dim, tuple(image_size)
        self.nfilters_enc, self.nfilters_dec, self.use_input_pose = tuple(nfilters_enc), tuple(nfilters_dec), use_input_pose
        self.generator = Deformable_Generator(input_nc, pose_dim, image_size, nfilters_enc, nfilters_dec, warp_skip,
                                              use_input_pose, align_corners, device)
        # (round 6) the chained backward's image gradient of the first layer reads the bf16 gradient tensor (pg_small_cin_dgrad_io):
        # the stages run in bf16 STORAGE on the bf16 data path like the single-stage generator; PG_STACKED_F32_STORE=1 = the round-5 mode
        self.generator.bf16_store_ok = os.environ.get("PG_STACKED_F32_STORE") is None

    @property
    def arena(self):
        return self.generator.arena

    def zero_grad(self, set_to_none=False):
        self.generator.zero_grad()

    def state_dict(self, *a, **k):
        return {"generator." + key: v for key, v in self.generator.state_dict().items()}

    def load_state_dict(self, sd, strict=True):
        """accepts the stacked model's own `generator.*` keys or a plain Deformable_Generator checkpoint (the reference
        initialises the stack from `gen_090.pkl` of the single-stage model, pose_gan.py:30-32)."""
        if any(k.startswith("generator.") for k in sd):
            sd = {k[len("generator."):]: v for k, v in sd.items() if k.startswith("generator.")}
        self.generator.load_state_dict(sd, strict)

    def stage_input(self, i, input, target_pose, prev_out):
        """networks.py:313-323 — the torch.cat of each stage (a device copy into one NCHW tensor)."""
        P = self.pose_dim
        tp = target_pose[:, i * P:(i + 1) * P]
        if i == 0:
            return torch.cat([input[:, :3 + P], tp], dim=1)
        return torch.cat([prev_out, target_pose[:, (i - 1) * P:i * P], tp], dim=1)

    def forward(self, input, target_pose, target_warps, target_masks=None, drop_masks=None):
        outputs, out = [], None
        for i in range(self.num_stacks):
            inp = self.stage_input(i, input, target_pose, out)
            out = self.generator(inp, target_warps[:, i], None if target_masks is None else target_masks[:, i].contiguous(),
                                 None if drop_masks is None else drop_masks[i], stage=i)
            outputs.append(out)
        return outputs


class Generator(_ArenaModule):
    """src_baseline Generator (reference src_baseline/models/networks.py:238-253): one encoder, no warps."""

    _init_tag = "gen"

    def __init__(self, input_nc, nfilters_enc, nfilters_dec, use_input_pose=True, pose_dim=None, image_size=(128, 64),
                 device="cuda"):
        super().__init__()
        self.pose_dim = pose_dim if pose_dim is not None else (input_nc - 3) // 2
        self.input_nc, self.image_size = input_nc, tuple(image_size)
        self.nfilters_enc, self.nfilters_dec = tuple(nfilters_enc), tuple(nfilters_dec)
        self.device = device
        spec = synth.generator_spec(self.pose_dim, self.nfilters_enc, self.nfilters_dec, num_skips=1, deformable=False)
        self._setup(spec, E.generator_param_order(spec, len(self.nfilters_enc), len(self.nfilters_dec), False), device)
        self._anchor = nn.Parameter(torch.zeros(1, device=device))
        self.align_corners = False
        self.drop_seed = 0

    def engine(self, n, stage=0):
        key = (n, stage, E.bf16_store())
        if key not in self._engines:
            self._engines[key] = E.GeneratorEngine(self.arena, n, self.image_size[0], self.image_size[1],
                                                   self.pose_dim, self.nfilters_enc, self.nfilters_dec, False,
                                                   False, self.device)
        return self._engines[key]

    def forward(self, input, drop_masks=None):
        eng = self.engine(input.shape[0])
        eng.set_dropout(drop_masks, train=self.training, seed=self.drop_seed)
        return _GenFn.apply(input, None, None, self, self._anchor, 0)


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, anchor):
        eng = mod.engine(x.shape[0])
        logits = eng.forward([(x.contiguous(),)])
        sig = torch.empty_like(logits)
        L.call("pg_gan_logloss", L.ptr(logits), logits.numel(), 0, 0.0, None, None, L.ptr(sig), L.stream())
        ctx.mod, ctx.n, ctx.xshape = mod, x.shape[0], x.shape
        ctx.save_for_backward(sig)
        return sig

    @staticmethod
    def backward(ctx, gout):
        (sig,) = ctx.saved_tensors
        eng = ctx.mod.engine(ctx.n)
        dlog = (gout * sig * (1 - sig)).contiguous()          # autograd glue for nn.Sigmoid (tiny: M x K)
        n, c, h, w = ctx.xshape
        gx = torch.zeros(ctx.xshape, dtype=torch.float32, device=sig.device)
        gimg = torch.empty(n, 3, h, w, dtype=torch.float32, device=sig.device)
        eng.backward(dlog, need_wgrad=True, image_grad=[gimg])
        P = eng.P
        gx[:, 3 + P:6 + P] = gimg                              # only the judged image needs a gradient
        return gx, None, None


class Discriminator(_ArenaModule):
    """reference models/networks.py:329-357: Conv(k4,s2,p0,bias) -> 3 Blocks -> Block(512,1,bn=False) -> Sigmoid
    -> Flatten.  forward(x) takes the concatenated (M, 3+2P+3, H, W) input like the reference."""

    _init_tag = "disc"

    def __init__(self, input_nc, warp_skip=False, use_input_pose=True, checkMode=0, image_size=(256, 256),
                 device="cuda"):
        super().__init__()
        self.input_nc, self.image_size, self.device = input_nc, tuple(image_size), device
        self.pose_dim = (input_nc - 6) // 2
        spec = synth.discriminator_spec(input_nc, checkMode)
        self._setup(spec, E.discriminator_param_order(spec), device)
        self._anchor = nn.Parameter(torch.zeros(1, device=device))

    def engine(self, m):
        key = (m, E.bf16_store() and E.DISC_BF16_STORE)          # engines are built for ONE storage mode (runtime/engine.py)
        if key not in self._engines:
            self._engines[key] = E.DiscriminatorEngine(self.arena, m, self.image_size[0], self.image_size[1],
                                                       self.pose_dim, self.device)
        return self._engines[key]

    def forward(self, x):
        return _DiscFn.apply(x, self, self._anchor)
