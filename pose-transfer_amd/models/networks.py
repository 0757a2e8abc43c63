"""Network modules with the reference's surface (reference src_deformable/models/networks.py:130-357)
and ``state_dict`` key names/shapes, executed by the HIP engines (runtime/engine.py).

``Deformable_Generator(input_nc, pose_dim, image_size, nfilters_enc, nfilters_dec, warp_skip,
use_input_pose).forward(input, warps, masks)`` and ``Discriminator(input_nc).forward(x)`` take and
return NCHW fp32 tensors like the reference; parameters live in a flat packed arena
(``.arena``) and are exposed as ``nn.Parameter`` views whose ``.grad`` aliases the gradient arena,
so ``loss.backward(); optimizer.step()`` works, while the trainer (models/pose_gan.py) drives the
engines directly without an autograd tape.
"""
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
    @staticmethod
    def forward(ctx, inp, warps, masks, mod, anchor):
        eng = mod.engine(inp.shape[0])
        out = eng.forward(inp.contiguous(), warps, masks)
        ctx.mod, ctx.n = mod, inp.shape[0]
        return out.clone()

    @staticmethod
    def backward(ctx, gout):
        eng = ctx.mod.engine(ctx.n)
        g = gout.contiguous().clone()
        L.call("pg_tanh_bwd", L.ptr(g), L.ptr(eng.out), g.numel(), L.stream())
        eng.backward(g)
        return None, None, None, None, None


class Deformable_Generator(_ArenaModule):
    """reference models/networks.py:252-288.  warp_skip='mask' (10 masked affine warps) is the hot path;
    any other value than 'mask' is rejected ('full'/stacked are out of scope)."""

    _init_tag = "gen"

    def __init__(self, input_nc, pose_dim, image_size, nfilters_enc, nfilters_dec, warp_skip="mask",
                 use_input_pose=True, align_corners=False, device="cuda"):
        super().__init__()
        if warp_skip != "mask":
            raise Exception("Invalid warp_skip for the MI355X build: only 'mask' is supported")
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
        self.training_dropout = True

    def engine(self, n):
        if n not in self._engines:
            self._engines[n] = E.GeneratorEngine(self.arena, n, self.image_size[0], self.image_size[1], self.pose_dim,
                                                 self.nfilters_enc, self.nfilters_dec, True, self.align_corners,
                                                 self.device)
        return self._engines[n]

    def forward(self, input, warps, masks, drop_masks=None):
        eng = self.engine(input.shape[0])
        eng.set_dropout(drop_masks, train=self.training_dropout)
        return _GenFn.apply(input, warps.float(), masks, self, self._anchor)


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
        self.training_dropout = True
        self.align_corners = False

    def engine(self, n):
        if n not in self._engines:
            self._engines[n] = E.GeneratorEngine(self.arena, n, self.image_size[0], self.image_size[1], self.pose_dim,
                                                 self.nfilters_enc, self.nfilters_dec, False, False, self.device)
        return self._engines[n]

    def forward(self, input, drop_masks=None):
        eng = self.engine(input.shape[0])
        eng.set_dropout(drop_masks, train=self.training_dropout)
        return _GenFn.apply(input, None, None, self, self._anchor)


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
        if m not in self._engines:
            self._engines[m] = E.DiscriminatorEngine(self.arena, m, self.image_size[0], self.image_size[1],
                                                     self.pose_dim, self.device)
        return self._engines[m]

    def forward(self, x):
        return _DiscFn.apply(x, self, self._anchor)
