"""DeformablePose_GAN trainer with the reference's surface (reference src_deformable/models/pose_gan.py:11-220):
``gen_update(input, target, other_inputs, opt) -> (out_gen, [], [total, ll, ad])`` and
``dis_update(input, target, other_inputs, real_inp, real_target, opt) -> [total, true, fake]``.

Differences from the reference, all result-preserving:
  * ``dis_update`` does not back-propagate through the generator (bit-identical: those grads are zeroed
    before use, pose_gan.py:70) and ``gen_update`` skips the discriminator's weight gradients;
  * losses are reduced on the device; ``.item()`` happens once per call only when ``lazy_losses`` is off;
  * data parallel: one process per GPU, gradients all-reduced (RCCL) in backward-ordered buckets while the
    rest of the backward is still running (runtime/dp.py), then a single fused Adam launch per network.
"""
import os

import torch
import torch.nn as nn

from ..runtime import dp as DP
from ..runtime import engine as E
from ..runtime import lib as L
from ..utils import pose_utils, synth
from .networks import Deformable_Generator, Discriminator, Generator, Stacked_Generator, xavier_weights_init


_PF_STREAMS = {}            # device index -> the prefetch stream (prefetch_gen_forward)
GEN_PREFETCH = os.environ.get("PG_NO_GEN_PREFETCH") is None
# (round 6) the prefetched forward keeps a second GeneratorEngine (a full set of activation / operand buffers)
# resident; it pays where launches leave CUs idle — small per-GPU batches (256^2 batch 4: 632 -> 707 img/s) — and buys nothing at
# batch 32 or 512^2.  On only up to this many input pixels per batch (default: 12 images of 256 x 256).
GEN_PREFETCH_MAX_PIX = int(os.environ.get("PG_GEN_PREFETCH_MAX_PIX", str(12 * 256 * 256)))
ZERO_ON_PREFETCH = os.environ.get("PG_NO_ZERO_ON_PREFETCH") is None      # (round 6) gen.zero_grad() on the prefetch stream


class FusedAdam:
    """Stand-in for torch.optim.Adam(params, lr, betas=(0.5,0.999)) over a ParamArena (pose_gan.py:50-51)."""

    def __init__(self, module, lr, betas=(0.5, 0.999), eps=1e-8):
        self.module, self.lr, self.betas, self.eps = module, lr, betas, eps

    def zero_grad(self):
        self.module.arena.zero_grad()

    def step(self, grad_scale=1.0, grads_bf16=None):
        self.module.arena.adam_step(self.lr, self.betas[0], self.betas[1], self.eps, grad_scale, grads_bf16)

    def state_dict(self):
        a = self.module.arena
        return {"step": a.step, "exp_avg": a.m.clone(), "exp_avg_sq": a.v.clone()}


class DeformablePose_GAN(nn.Module):
    def __init__(self, opt, device="cuda", init_seed=0):
        super().__init__()
        # adding extra layers for larger image size — reference pose_gan.py:17-18
        nfilters_encoder, nfilters_decoder = synth.nfilters(opt.image_size)
        if not opt.use_input_pose:
            # the reference's Deformable_Generator.forward concatenates `inp_pose = None` in this mode (networks.py:270-271:
            # torch.cat([inp_app, None]) raises TypeError) — the path does not run there either
            raise Exception("use_input_pose=0 is not supported (the reference's deformable generator raises in this mode too)")
        input_nc = 3 + 2 * opt.pose_dim
        self.batch_size = opt.batch_size
        self.pose_dim = opt.pose_dim
        self.image_size = tuple(opt.image_size)
        self.device = device
        self.num_stacks = getattr(opt, "num_stacks", 4)
        self.gen_type = opt.gen_type
        # src_baseline's single-encoder Generator (BASELINE.json configs[0]) is selected EXPLICITLY (`src_baseline`
        # attribute / the Pose_GAN subclass), never through warp_skip: in src_deformable every warp_skip value builds
        # the two-encoder Deformable_Generator (reference networks.py:253-288)
        self.deformable = not getattr(opt, "src_baseline", False)
        warp_skip = getattr(opt, "warp_skip", "mask")
        self.warp_skip = warp_skip
        align = bool(getattr(opt, "align_corners", 0))
        if not self.deformable:
            if opt.gen_type != "baseline":
                raise Exception("Invalid gen_type")
            self.gen = Generator(input_nc, nfilters_encoder, nfilters_decoder, pose_dim=self.pose_dim,
                                 image_size=opt.image_size, device=device)
        elif opt.gen_type == "baseline":
            self.gen = Deformable_Generator(input_nc, self.pose_dim, opt.image_size, nfilters_encoder, nfilters_decoder,
                                            warp_skip, use_input_pose=True, align_corners=align, device=device)
        elif opt.gen_type == "stacked":     # reference pose_gan.py:28-33
            self.gen = Stacked_Generator(input_nc, self.num_stacks, opt.image_size, self.pose_dim, nfilters_encoder,
                                         nfilters_decoder, warp_skip, use_input_pose=True, align_corners=align,
                                         device=device)
        else:
            raise Exception("Invalid gen_type")
        self.disc = Discriminator(input_nc + 3, use_input_pose=True, image_size=opt.image_size, device=device)
        # the reference loads a private pretrained discriminator unconditionally (pose_gan.py:40-42); here it is
        # optional and the default is the reference's own init recipe (networks.py:26-31)
        xavier_weights_init(self._core, init_seed)
        xavier_weights_init(self.disc, init_seed + 1)
        pre = getattr(opt, "discriminator_checkpoint", None)
        if pre:
            self.disc.load_state_dict(torch.load(pre, map_location="cpu"))
        pre = getattr(opt, "generator_checkpoint", None)
        if pre:
            self.gen.load_state_dict(torch.load(pre, map_location="cpu"))
        lr = opt.learning_rate
        self.disc_opt = FusedAdam(self.disc, lr)
        self.gen_opt = FusedAdam(self._core, lr)
        self.content_loss_layer = opt.content_loss_layer
        self.nn_loss_area_size = opt.nn_loss_area_size
        self.vgg_w = self.vgg_b = None
        if self.content_loss_layer != "none":
            if pose_utils.get_layer_ind(self.content_loss_layer) != 1:
                raise Exception("only content_loss_layer=block1_conv2 (vgg19.features[:2]) is implemented")
            self.set_vgg_weights(*_default_vgg_conv1(getattr(opt, "vgg_weights", None)))
        self.world = DP.world_size()
        # PG_FORCE_REDUCER=1 exercises the bucketed all-reduce path even at world size 1 (single-GPU test of the DP code)
        use_red = self.world > 1 or (os.environ.get("PG_FORCE_REDUCER") == "1" and DP.dist.is_initialized())
        self.g_reducer = DP.GradReducer(self._core.arena, max(self.world, 2) if use_red else 1) if use_red else None
        self.d_reducer = DP.GradReducer(self.disc.arena, max(self.world, 2) if use_red else 1) if use_red else None
        self._loss = torch.zeros(8, dtype=torch.float32, device=device)
        self._bufs = {}
        # dropout RNG stream = f(opt.seed, rank, global iteration): a resumed run continues the sequence instead of
        # replaying it (main.py sets `iteration`); engines of different batch sizes / stages use different streams
        self.seed = int(getattr(opt, "seed", 0))
        self.iteration = 0

    @property
    def _core(self):
        """the module that owns the generator's parameter arena"""
        return self.gen.generator if self.gen_type == "stacked" else self.gen

    # ------------------------------------------------------------------------------------------
    def set_vgg_weights(self, w, b):
        self.vgg_w = w.to(self.device, torch.float32).contiguous()
        self.vgg_b = b.to(self.device, torch.float32).contiguous()

    def _buf(self, name, shape, dtype=torch.float32):
        key = (name, tuple(shape), dtype)
        if key not in self._bufs:
            self._bufs[key] = torch.empty(shape, dtype=dtype, device=self.device)
        return self._bufs[key]

    def _drop_setup(self, eng, drop_masks, stage, call):
        # (in a HIP-graph replay session the iteration number comes from the device counter, not from this string)
        replay = E.REPLAY_CTR is not None
        it = 0 if replay else self.iteration
        # repeated calls within ONE iteration (--training_ratio > 1: several dis_update per gen_update, main.py:78-88)
        # draw fresh masks, as the reference's Dropout2d does: the k-th repeat of a (call, stage) gets its own stream
        if getattr(self, "_drop_it", None) != it or replay:
            # (a replay session pins `it` to 0 and records exactly one dis_update + gen_update: no repeats, so the stream
            # names of its warm-up iterations and of the recorded one are the same)
            self._drop_it, self._drop_n = it, {}
        k = self._drop_n.get((call, stage), 0)
        self._drop_n[(call, stage)] = k + 1
        eng.drop_stream = "drop/r%d/i%d/%s%s/s%d" % (DP.rank(), it, call, "" if k == 0 else ".%d" % k, stage)
        eng._drop_counter = 0
        eng.set_dropout(drop_masks, train=True, seed=self.seed)

    # ---- generator forward of the NEXT gen_update, issued ahead (round 5) --------------------------------------------
    # dis_update steps only the discriminator (reference pose_gan.py:117-171), so the generator forward gen_update starts
    # with (pose_gan.py:72-79) sees the same weights whether it is enqueued after dis_update or BEFORE it.  At small per-GPU
    # batches (BASELINE.json configs[3]: 4 per GPU) most launches of an iteration leave most CUs idle; enqueued ahead on its
    # own stream and into its own engine (activation buffers), that forward runs NEXT TO dis_update's work instead of
    # after it.  Same kernels, same dropout stream (`call="g"` of this iteration), same results — gen_update recognises the
    # prefetched pass by its input tensors and the arena's weight version and otherwise computes the forward as before.
    # Single-stage generators only, not in replay sessions, small per-GPU batches only (GEN_PREFETCH_MAX_PIX).  PG_NO_GEN_PREFETCH=1
    # switches it off.
    def prefetch_gen_forward(self, input, other_inputs):
        self._pf = None
        if not (GEN_PREFETCH and self.gen_type != "stacked" and self.deformable and E.REPLAY_CTR is None and E.SIDE_STREAM
                and torch.is_tensor(input) and input.is_cuda
                and input.shape[0] * input.shape[2] * input.shape[3] <= GEN_PREFETCH_MAX_PIX):
            return False
        input = input.contiguous()
        if getattr(self, "_pf_stream", None) is None:
            # ONE prefetch stream per device, shared by every model of the process: torch hands out pool streams round-robin and HIP
            # maps them onto a few hardware queues, so a fresh stream per model moved the (main, side, auxiliary, prefetch) set onto
            # another queue combination with every model created — the third configuration leg of bench.py ran 5 % slower than the
            # same leg in a fresh process (1013 -> 961 img/s) because two of its streams shared a hardware queue.
            dev_i = input.device.index if input.device.index is not None else torch.cuda.current_device()
            st = _PF_STREAMS.get(dev_i)
            if st is None:
                st = _PF_STREAMS[dev_i] = torch.cuda.Stream(device=input.device)
            self._pf_stream = st
        if E.PRECISION == 3:
            self._core.arena.bf16_params()      # (first use after load_state_dict converts: on THIS stream, before the fork)
        L.call("pg_stream_wait", E._raw(self._pf_stream), L.stream())        # (the inputs were produced on this stream)
        with torch.cuda.stream(self._pf_stream):
            # (round 6) the generator's gradient arena (328 MB at 256^2) is zeroed here, next to dis_update's work, instead of at the
            # start of gen_update: dis_update never touches it, the last reader (the previous gen_update's Adam) is behind the fork, and
            # gen_update joins this stream before its first gradient write.  tools/pass_sensitivity.py: the zero fills cost 0.09 - 0.14 ms
            # of a batch-4 iteration on the main stream.
            zeroed = ZERO_ON_PREFETCH
            if zeroed:
                self.gen.zero_grad()
            engs, out = self._gen_forward(input, other_inputs, (other_inputs or {}).get("drop_masks"), engine_stage=1)
        self._pf = {"key": self._pf_key(input, other_inputs), "engs": engs, "out": out, "input": input, "zeroed": zeroed}
        return True

    def _pf_key(self, input, other_inputs):
        oi = other_inputs or {}
        ptr = lambda v: (v.data_ptr(), tuple(v.shape), v.dtype) if torch.is_tensor(v) else None
        return (ptr(input), ptr(oi.get("warps")), ptr(oi.get("masks")),
                tuple(ptr(m) for m in (oi.get("drop_masks") or ())), self._core.arena.version(), self.iteration)

    def _take_prefetched(self, input, other_inputs):
        pf, self._pf = getattr(self, "_pf", None), None
        if pf is None:
            return None
        if pf["key"] != self._pf_key(input, other_inputs):
            # not the pass gen_update asks for (other tensors, weights changed since): forget it, and give the dropout stream of
            # this iteration's generator update back so that the forward computed now draws the masks it would have drawn
            L.call("pg_stream_wait", L.stream(), E._raw(self._pf_stream))
            if getattr(self, "_drop_n", None):
                self._drop_n.pop(("g", 0), None)
            return None
        L.call("pg_stream_wait", L.stream(), E._raw(self._pf_stream))        # this stream continues behind the prefetched pass
        self._pf_zeroed = bool(pf.get("zeroed"))
        return pf["engs"], pf["out"]

    def _gen_forward(self, input, other_inputs, drop_masks, call="g", engine_stage=0):
        """One generator forward (the chained stages of the stacked generator).  Returns ([engines], out_gen)."""
        other_inputs = other_inputs or {}
        if self.gen_type == "stacked":       # reference pose_gan.py:72-77
            tp, wr = other_inputs["interpol_pose"], other_inputs["interpol_warps"].float()
            mk = other_inputs.get("interpol_masks") if self.warp_skip == "mask" else None
            engs, out = [], None
            for i in range(self.num_stacks):
                eng = self._core.engine(input.shape[0], i)
                self._drop_setup(eng, None if drop_masks is None else drop_masks[i], i, call)
                x = self.gen.stage_input(i, input, tp, out)
                out = eng.forward(x, wr[:, i], None if mk is None else mk[:, i].contiguous())
                engs.append(eng)
            return engs, out
        eng = self._core.engine(input.shape[0], engine_stage) if self.deformable else self._core.engine(input.shape[0])
        self._drop_setup(eng, drop_masks, 0, call)
        if self.deformable:
            return [eng], eng.forward(input, other_inputs["warps"].float(),
                                      other_inputs.get("masks") if self.warp_skip == "mask" else None)
        return [eng], eng.forward(input)

    def _losses(self, lo, lazy):
        if lazy:
            return self._loss[lo:lo + 3].clone()
        return [float(v) for v in self._loss[lo:lo + 3].tolist()]

    # ------------------------------------------------------------------------------------------
    def gen_update(self, input, target, other_inputs, opt):
        """reference pose_gan.py:69-115."""
        n, (H, W) = input.shape[0], self.image_size
        input, target = input.contiguous(), target.contiguous()
        E.dev_zero(self._loss[0:3])
        self._pf_zeroed = False
        pre = self._take_prefetched(input, other_inputs)
        if not self._pf_zeroed:          # (the prefetched pass zeroed the gradient arena on its stream, behind which this one continues)
            self.gen.zero_grad()
        if pre is not None:
            engs, out_gen = pre
        else:
            engs, out_gen = self._gen_forward(input, other_inputs, (other_inputs or {}).get("drop_masks"))
        # discriminator on [img, src_pose, out_gen, tgt_pose] — forward + data-gradient only
        deng = self.disc.engine(n)
        logits = deng.forward([(input, out_gen)])
        dlog = self._buf("dlog_g", logits.shape)
        K = logits.shape[1]
        w_gan = float(opt["gan_penalty_weight"]) / self.batch_size
        L.call("pg_gan_logloss", L.ptr(logits), logits.numel(), 0, w_gan / K, L.ptr(self._loss[2:]), L.ptr(dlog), None,
               L.stream())
        gout = self._buf("gout", out_gen.shape)
        deng.backward(dlog, need_wgrad=False, image_grad=[gout])
        l1w = float(opt["l1_penalty_weight"])
        if self.content_loss_layer != "none":
            fg = self._buf("feat_g", (n, H, W, 64))
            ft = self._buf("feat_t", (n, H, W, 64))
            L.call("pg_vgg_conv1_relu_fwd", L.ptr(out_gen), L.ptr(self.vgg_w), L.ptr(self.vgg_b), n, H, W, L.ptr(fg), L.stream())
            L.call("pg_vgg_conv1_relu_fwd", L.ptr(target), L.ptr(self.vgg_w), L.ptr(self.vgg_b), n, H, W, L.ptr(ft), L.stream())
            dfg = self._buf("dfeat_g", (n, H, W, 64))
            L.call("pg_nn_loss", L.ptr(fg), L.ptr(ft), n, H, W, 64, int(self.nn_loss_area_size), l1w / (n * H * W), 1,
                   L.ptr(self._loss[1:]), L.ptr(dfg), L.stream())
            L.call("pg_vgg_conv1_dgrad", L.ptr(dfg), L.ptr(self.vgg_w), n, H, W, L.ptr(gout), L.stream())
        else:
            L.call("pg_l1_loss", L.ptr(out_gen), L.ptr(target), out_gen.numel(), l1w / out_gen.numel(),
                   L.ptr(self._loss[1:]), L.ptr(gout), 1, L.stream())
        L.call("pg_tanh_bwd", L.ptr(gout), L.ptr(out_gen), gout.numel(), L.stream())
        if self.g_reducer is not None:
            self.g_reducer.begin()
        # single process, one stage: the optimiser step runs in ranges UNDER the backward pass (engine.EagerAdam)
        eager = None
        if (self.g_reducer is None and len(engs) == 1 and E.EAGER_ADAM and E.REPLAY_CTR is None and E.SIDE_STREAM
                and hasattr(engs[0], "param_release_cb") and input.is_cuda):
            if getattr(self, "_eager_g", None) is None:
                self._eager_g = E.EagerAdam(self._core.arena, self.gen_opt.lr, self.gen_opt.betas[0], self.gen_opt.betas[1],
                                            self.gen_opt.eps)
            eager = self._eager_g
            eager.begin()
        # stacked: back through the chained stages; stage i's image input is stage i-1's output (networks.py:320).  The
        # shared weights' gradients accumulate in the arena; only the LAST backward (stage 0) reports them ready.
        for i in range(len(engs) - 1, -1, -1):
            eng = engs[i]
            eng.grad_ready_cb = self.g_reducer.mark_ready if (self.g_reducer is not None and i == 0) else None
            if hasattr(eng, "param_release_cb"):
                eng.param_release_cb = eager.release if eager is not None else None
            if i > 0:
                gprev = self._buf("gprev%d" % (i & 1), gout.shape)
                eng.backward(gout, image_grad=gprev)
                L.call("pg_tanh_bwd", L.ptr(gprev), L.ptr(engs[i - 1].out), gprev.numel(), L.stream())
                gout = gprev
            else:
                eng.backward(gout)
        scale, gb = 1.0, None
        if self.g_reducer is not None:
            self.g_reducer.finish()
            scale, gb = 1.0 / self.g_reducer.divisor, self.g_reducer.grad_source()[1]
        if eager is not None:
            engs[0].param_release_cb = None
            eager.finish()
        else:
            self.gen_opt.step(grad_scale=scale, grads_bf16=gb)
        lp = L.ptr(self._loss)
        L.call("pg_add2", lp, lp + 4, lp + 8, 1, L.stream())        # total = ll + ad  (pose_gan.py:109)
        losses = self._losses(0, opt.get("lazy_losses", False))
        # a fresh tensor like the reference's (the engine's output buffer is overwritten by the next forward)
        outputs = [e.out.clone() for e in engs] if self.gen_type == "stacked" else []
        return (outputs[-1] if outputs else out_gen.clone()), outputs, losses

    def dis_update(self, input, target, other_inputs, real_inp, real_target, opt):
        """reference pose_gan.py:117-171 (out_gen detached: SURVEY App. A.7 (i))."""
        n = input.shape[0]
        input, real_inp, real_target = input.contiguous(), real_inp.contiguous(), real_target.contiguous()
        self.disc.zero_grad()
        E.dev_zero(self._loss[4:7])
        engs, out_gen = self._gen_forward(input, other_inputs, (other_inputs or {}).get("drop_masks"), call="d")
        deng = self.disc.engine(2 * n)
        logits = deng.forward([(real_inp, real_target), (input, out_gen)])     # cat((real, fake), 0) — pose_gan.py:136
        K = logits.shape[1]
        dlog = self._buf("dlog_d", logits.shape)
        w_gan = float(opt["gan_penalty_weight"]) / self.batch_size
        L.call("pg_gan_logloss", L.ptr(logits), n * K, 0, w_gan / K, L.ptr(self._loss[5:]), L.ptr(dlog), None, L.stream())
        L.call("pg_gan_logloss", logits.data_ptr() + 4 * n * K, n * K, 1, w_gan / K, L.ptr(self._loss[6:]),
               dlog.data_ptr() + 4 * n * K, None, L.stream())
        if self.d_reducer is not None:
            self.d_reducer.begin()
            deng.grad_ready_cb = self.d_reducer.mark_ready
        deng.backward(dlog, need_wgrad=True)
        scale, gb = 1.0, None
        if self.d_reducer is not None:
            self.d_reducer.finish()
            scale, gb = 1.0 / self.d_reducer.divisor, self.d_reducer.grad_source()[1]
        self.disc_opt.step(grad_scale=scale, grads_bf16=gb)
        lp = L.ptr(self._loss)
        L.call("pg_add2", lp + 16, lp + 20, lp + 24, 1, L.stream())
        return self._losses(4, opt.get("lazy_losses", False))

    # ------------------------------------------------------------------------------------------
    def nn_loss(self, predicted, ground_truth, nh=3, nw=3):
        """reference pose_gan.py:173-199 on NCHW tensors (module-level API; value only)."""
        assert nh == nw
        n, c, h, w = predicted.shape
        cp = 4
        while cp < c:
            cp *= 2
        P = torch.zeros(n, h, w, cp, dtype=torch.float32, device=self.device)
        G = torch.zeros(n, h, w, cp, dtype=torch.float32, device=self.device)
        P[..., :c] = predicted.permute(0, 2, 3, 1)
        G[..., :c] = ground_truth.permute(0, 2, 3, 1)
        loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        L.call("pg_nn_loss", L.ptr(P), L.ptr(G), n, h, w, cp, nh, 1.0 / (n * h * w), 0, L.ptr(loss), None, L.stream())
        return loss[0]

    def resume(self, save_dir):
        """reference pose_gan.py:201-214 (same file names, same return value = epoch of the last checkpoint)."""
        last = pose_utils.get_model_list(save_dir, "gen")
        if last is None:
            return 1
        self.gen.load_state_dict(torch.load(last, map_location="cpu"))
        epoch = int(last[-7:-4])
        last = pose_utils.get_model_list(save_dir, "dis")
        if last is None:
            return 1
        epoch = int(last[-7:-4])
        self.disc.load_state_dict(torch.load(last, map_location="cpu"))
        return epoch

    def save(self, save_dir, epoch):
        """reference pose_gan.py:216-220: gen_%03d.pkl / disc_%03d.pkl = torch.save(state_dict) in reference layout."""
        os.makedirs(save_dir, exist_ok=True)
        torch.save({k: v.cpu() for k, v in self.gen.state_dict().items()}, os.path.join(save_dir, "gen_{0:03d}.pkl".format(epoch)))
        torch.save({k: v.cpu() for k, v in self.disc.state_dict().items()}, os.path.join(save_dir, "disc_{0:03d}.pkl".format(epoch)))


class Pose_GAN(DeformablePose_GAN):
    """src_baseline trainer (reference src_baseline/models/pose_gan.py:10-142; BASELINE.json configs[0]): single-encoder
    Generator, no warps, L1 loss; same update methods."""

    def __init__(self, opt, device="cuda", init_seed=0):
        import copy
        opt = copy.copy(opt)
        opt.src_baseline = True
        super().__init__(opt, device=device, init_seed=init_seed)


def _default_vgg_conv1(path=None):
    """conv1_1 of VGG-19.  The ImageNet weights (`vgg19-dcbb9e9d.pth`) are not available offline: load them from
    `path` (a torchvision state_dict) when given, else use seeded synthetic weights (trained values unpinned)."""
    if path and os.path.exists(path):
        sd = torch.load(path, map_location="cpu")
        return sd["features.0.weight"], sd["features.0.bias"]
    return (torch.from_numpy(synth.xavier_uniform(14, "vgg/w", (64, 3, 3, 3))),
            torch.from_numpy(synth.uniform(14, "vgg/b", (64,), -0.1, 0.1)))
