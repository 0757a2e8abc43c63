"""Execution engines: the explicit forward/backward kernel schedules of the generator and the
discriminator on top of libposegan_hip (no autograd tape, no torch compute ops).

Data layout: every activation is fp32 NHWC and is stored RAW (the output of its
convolution).  The reference's per-sample norm (models/networks.py:159,166-169), dropout (:161),
pre-activation (:150,152) and torch.cat (:241,245,271,284,286) are never materialised: each tensor
carries a per-sample affine ``aff`` [N,2] and an optional channel mask [N,C] that consumers apply on
load.  Parameters live in flat arenas (params / grads / Adam m / Adam v) in packed kernel layout.
"""
import contextlib
import ctypes
import os
import math
import weakref

import numpy as np
import torch

from . import lib as L
from ..utils import synth

T_WARPS = 10
NORM_EPS = 1e-3   # reference models/networks.py:159


class KernelProfiler:
    """Per-launch timing of the contraction kernels with HIP events recorded on the launch stream
    (bench.py's `roofline` leg).  Not active inside the timed region."""

    CONV_TILES = {0: "128x128", 1: "128x64", 2: "64x64", 3: "128x32", 4: "256x256", 5: "256x128", 6: "512x64", 7: "512x128", 8: "256x256p", 9: "256x128p",
                  10: "512x64p", 11: "256x256m", 12: "256x128m", 13: "quad128", 14: "quad128m"}      # 13 / 14: igemm_bf16_quad.hip
    WGRAD_TILES = {0: "128x64", 1: "64x64", 2: "32x64", 3: "128x128", 4: "64x32", 5: "patch64xTapsCin", 6: "bf16-tr-256", 7: "bf16-stem"}

    def __init__(self):
        self.records = []

    def launch(self, kind, flops, fn):
        import ctypes
        lib = L.load()
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        L.check(lib.pg_event_create(ctypes.byref(e0)), "pg_event_create")
        L.check(lib.pg_event_create(ctypes.byref(e1)), "pg_event_create")
        L.check(lib.pg_event_record(e0, L.stream()), "pg_event_record")
        fn()
        L.check(lib.pg_event_record(e1, L.stream()), "pg_event_record")
        info = lib.pg_last_launch_info()
        if info & (1 << 30):
            name = "wgrad_igemm<%s,xs%d,ys%d>" % (self.WGRAD_TILES[info & 15], (info >> 4) & 15, (info >> 8) & 15)
        else:
            name = "conv_igemm<%s,A%d,B%d>" % (self.CONV_TILES[info & 15], (info >> 4) & 15, (info >> 8) & 15)
        self.records.append((name, kind, flops, (info >> 16) & 0x3FFF, e0, e1))

    def summary(self, per_launch=None, repeats=1):
        """Per-family totals; `per_launch` (a list) additionally receives one (name, kind, flops, ksplit, ms) per launch.
        repeats > 1: the records hold that many identical iterations back to back; each launch is credited with the MINIMUM
        of its repeats (a one-off stall — an event-pool or buffer allocation inside one repeat — showed up as a 30-45 ms
        "launch" otherwise)."""
        import ctypes
        lib = L.load()
        times = []
        for name, kind, flops, ks, e0, e1 in self.records:
            ms = ctypes.c_float()
            L.check(lib.pg_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "pg_event_elapsed_ms")
            lib.pg_event_destroy(e0)
            lib.pg_event_destroy(e1)
            times.append(ms.value)
        n = len(self.records) // max(1, repeats)
        aligned = repeats > 1 and n * repeats == len(self.records) and all(
            self.records[i][:3] == self.records[i + r * n][:3] for r in range(1, repeats) for i in range(n))
        if not aligned:
            n, repeats = len(self.records), 1
        out = {}
        for i in range(n):
            name, kind, flops, ks = self.records[i][:4]
            ms = min(times[i + r * n] for r in range(repeats))
            if per_launch is not None:
                per_launch.append((name, kind, flops, ks, ms))
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0})
            d["launches"] += 1
            d["ms"] += ms
            d["flops"] += flops
        self.records = []
        return out


PROFILER = None       # set to a KernelProfiler to time every contraction launch


# ------------------------------------------------------------------------------------------ arenas
def _pack(key, w):
    """reference state_dict tensor -> packed kernel layout [KH][KW][Cout][Cin]."""
    if w.dim() != 4:
        return w.contiguous()
    if _is_convT(key):
        return w.permute(2, 3, 1, 0).contiguous()      # (Cin,Cout,KH,KW) ConvTranspose2d
    return w.permute(2, 3, 0, 1).contiguous()          # (Cout,Cin,KH,KW) Conv2d


def _unpack(key, w):
    if w.dim() != 4:
        return w.clone()
    if _is_convT(key):
        return w.permute(3, 2, 0, 1).contiguous()
    return w.permute(2, 3, 0, 1).contiguous()


def _is_convT(key):
    # decoder up-blocks hold nn.ConvTranspose2d at `.net.1` (reference models/networks.py:156)
    return key.startswith("decoder.net.") and key.endswith(".net.1.weight")


def _packed_shape(key, shape):
    if len(shape) != 4:
        return tuple(shape)
    if _is_convT(key):
        return (shape[2], shape[3], shape[1], shape[0])
    return (shape[2], shape[3], shape[0], shape[1])


class ParamArena:
    """Flat fp32 device buffers holding every parameter of one network in packed layout, plus the
    matching gradient and Adam-moment buffers.  `order` lists keys in backward-completion order so
    that DP gradient buckets are contiguous ranges."""

    ALIGN = 64

    def __init__(self, spec, order, device):
        shapes = dict(spec)
        self.spec = list(spec)
        self.keys = list(order)
        assert sorted(self.keys) == sorted(shapes)
        self.ref_shape = shapes
        self.off, self.numel, self.pshape = {}, {}, {}
        o = 0
        for k in self.keys:
            n = int(np.prod(shapes[k]))
            self.off[k], self.numel[k] = o, n
            self.pshape[k] = _packed_shape(k, shapes[k])
            o += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.total = o
        self.params = torch.zeros(o, dtype=torch.float32, device=device)
        self.grads = torch.zeros(o, dtype=torch.float32, device=device)
        self.m = torch.zeros(o, dtype=torch.float32, device=device)
        self.v = torch.zeros(o, dtype=torch.float32, device=device)
        self.step = 0
        self.params_bf16 = None       # bf16 data path: bf16 copy of the whole arena, kept current by Adam / load_state_dict
        self.bf16_version = -1
        self.params_bf16_t, self.bf16_t_version = None, -1      # ... and the per-tap transposed copy (bf16_params_t)
        ARENAS[self.params.untyped_storage().data_ptr()] = self

    def p(self, key):
        return self.params[self.off[key]:self.off[key] + self.numel[key]].view(self.pshape[key])

    def g(self, key):
        return self.grads[self.off[key]:self.off[key] + self.numel[key]].view(self.pshape[key])

    def _bump_version(self):
        key = self.params.untyped_storage().data_ptr()
        WEIGHT_VERSION[key] = WEIGHT_VERSION.get(key, 0) + 1

    def version(self):
        return WEIGHT_VERSION.get(self.params.untyped_storage().data_ptr(), 0)

    def bf16_params(self):
        """bf16 copy of the arena in the same packed layout ([tap][Cout][Cin] = the K-contiguous forward operand).
        Written by the Adam launch itself (pg_adam_ex); converted here only after load_state_dict / the first use."""
        if self.params_bf16 is None:
            self.params_bf16 = torch.empty(self.total, dtype=torch.bfloat16, device=self.params.device)
        if self.bf16_version != self.version():
            L.call("pg_pack_bf16", L.ptr(self.params), L.ptr(self.params_bf16), self.total, L.stream())
            self.bf16_version = self.version()
        return self.params_bf16

    def bf16_params_t(self):
        """per-tap TRANSPOSED bf16 copy of every convolution weight ([tap][Cin][Cout]: the K-contiguous operand of the data
        gradients), same offsets as the arena, rebuilt by ONE launch per optimiser step (pg_weights_to_bf16_batch)"""
        if self.params_bf16_t is None:
            self.params_bf16_t = torch.empty(self.total, dtype=torch.bfloat16, device=self.params.device)
            rec, tile0 = [], 0
            for k in self.keys:
                ps = self.pshape[k]
                if len(ps) != 4:
                    continue
                taps, co, ci = ps[0] * ps[1], ps[2], ps[3]
                rec.append((self.off[k], taps, co, ci, tile0))
                tile0 += taps * ((co + 63) // 64) * ((ci + 31) // 32)       # 64 x 32 tiles (csrc/optim.hip)
            tab = np.zeros(len(rec), dtype=np.dtype([("off", "<i8"), ("taps", "<i4"), ("co", "<i4"), ("ci", "<i4"), ("t0", "<i4")]))
            for i, r in enumerate(rec):
                tab[i] = r
            self._wt_tab = torch.from_numpy(tab.view(np.uint8).copy()).to(self.params.device)
            self._wt_n, self._wt_tiles = len(rec), tile0
        if self.bf16_t_version != self.version():
            L.call("pg_weights_to_bf16_batch", L.ptr(self.params), L.ptr(self._wt_tab), self._wt_n, self._wt_tiles,
                   L.ptr(self.params_bf16_t), L.stream())
            self.bf16_t_version = self.version()
        return self.params_bf16_t

    def load_state_dict(self, sd):
        self._bump_version()
        for k in self.keys:
            w = sd[k]
            if not torch.is_tensor(w):
                w = torch.from_numpy(np.ascontiguousarray(w))
            assert tuple(w.shape) == tuple(self.ref_shape[k]), (k, tuple(w.shape), self.ref_shape[k])
            self.p(k).copy_(_pack(k, w.to(torch.float32)).to(self.params.device))

    def state_dict(self):
        return {k: _unpack(k, self.p(k)) for k, _ in self.spec}

    def grad_dict(self):
        return {k: _unpack(k, self.g(k)) for k, _ in self.spec}

    def moment_dicts(self):
        """Adam first/second moments in reference layout (tests / checkpointing of optimizer state)."""
        mv = lambda buf, k: buf[self.off[k]:self.off[k] + self.numel[k]].view(self.pshape[k])
        return ({k: _unpack(k, mv(self.m, k)) for k, _ in self.spec},
                {k: _unpack(k, mv(self.v, k)) for k, _ in self.spec})

    def zero_grad(self):
        dev_zero(self.grads)

    # ---- the same optimiser step issued in RANGES while the backward pass is still running (round 5, EagerAdam below)
    def adam_begin(self, lr, b1=0.5, b2=0.999, eps=1e-8):
        self.step += 1
        bc1 = 1.0 - b1 ** self.step
        bc2 = 1.0 - b2 ** self.step
        want_bf16 = PRECISION == 3 and self.params.is_cuda
        if want_bf16 and self.params_bf16 is None:
            self.params_bf16 = torch.empty(self.total, dtype=torch.bfloat16, device=self.params.device)
        self._eager = (float(b1), float(b2), eps, lr / bc1, math.sqrt(bc2), want_bf16)

    def adam_range(self, lo, hi):
        """Adam on the arena elements [lo, hi) (multiples of ALIGN) on the current stream"""
        b1, b2, eps, step_size, bc2s, want_bf16 = self._eager
        n = hi - lo
        if n <= 0:
            return
        o4 = 4 * lo
        if not want_bf16:
            L.call("pg_adam", self.params.data_ptr() + o4, self.grads.data_ptr() + o4, self.m.data_ptr() + o4,
                   self.v.data_ptr() + o4, n, b1, b2, eps, step_size, bc2s, 1.0, L.stream())
            return
        L.call("pg_adam_ex", self.params.data_ptr() + o4, self.grads.data_ptr() + o4, None, self.m.data_ptr() + o4,
               self.v.data_ptr() + o4, n, b1, b2, eps, step_size, bc2s, 1.0, self.params_bf16.data_ptr() + 2 * lo, L.stream())

    def adam_end(self):
        """every range has been issued: the parameters changed (the per-tap transposed bf16 copy is stale; the forward-layout
        copy was written by the range launches)"""
        want_bf16 = self._eager[5]
        self._eager = None
        self._bump_version()
        if want_bf16:
            self.bf16_version = self.version()

    def adam_step(self, lr, b1=0.5, b2=0.999, eps=1e-8, grad_scale=1.0, grads_bf16=None):
        """torch.optim.Adam semantics (reference models/pose_gan.py:50-51); bias corrections in double.
        grads_bf16: read the (all-reduced) bf16 gradient sums instead of the fp32 arena (runtime/dp.py bf16 buckets)."""
        self.step += 1
        self._bump_version()
        if REPLAY_CTR is not None:
            # replay-safe form: the graph freezes arguments, so the step number is step_base + the device counter (the first
            # optimiser step recorded in replay mode has counter 0); host bookkeeping (self.step) is kept by runtime/graph.py
            if getattr(self, "replay_base", None) is None:
                self.replay_base = self.step
            want_bf16 = PRECISION == 3 and self.params.is_cuda
            if want_bf16 and self.params_bf16 is None:
                self.params_bf16 = torch.empty(self.total, dtype=torch.bfloat16, device=self.params.device)
            L.call("pg_adam_ctr", L.ptr(self.params), L.ptr(self.grads), L.ptr(grads_bf16), L.ptr(self.m), L.ptr(self.v),
                   self.total, float(b1), float(b2), eps, lr, self.replay_base, L.ptr(REPLAY_CTR), grad_scale,
                   L.ptr(self.params_bf16) if want_bf16 else None, L.stream())
            if want_bf16:
                self.bf16_version = self.version()
            return
        bc1 = 1.0 - b1 ** self.step
        bc2 = 1.0 - b2 ** self.step
        want_bf16 = PRECISION == 3 and self.params.is_cuda
        if grads_bf16 is None and not want_bf16:
            L.call("pg_adam", L.ptr(self.params), L.ptr(self.grads), L.ptr(self.m), L.ptr(self.v), self.total,
                   b1, b2, eps, lr / bc1, math.sqrt(bc2), grad_scale, L.stream())
            return
        if want_bf16 and self.params_bf16 is None:
            self.params_bf16 = torch.empty(self.total, dtype=torch.bfloat16, device=self.params.device)
        L.call("pg_adam_ex", L.ptr(self.params), L.ptr(self.grads), L.ptr(grads_bf16), L.ptr(self.m), L.ptr(self.v),
               self.total, b1, b2, eps, lr / bc1, math.sqrt(bc2), grad_scale, L.ptr(self.params_bf16) if want_bf16 else None,
               L.stream())
        if want_bf16:
            self.bf16_version = self.version()


# ------------------------------------------------------------------------------------------ activation handle
class Act:
    """A raw NHWC activation with its deferred per-sample affine / channel mask."""

    def __init__(self, t, C, aff=None, mask=None, mr=None, strides=None, base_ptr=None):
        self.t, self.C, self.aff, self.mask, self.mr = t, C, aff, mask, mr
        self.strides = strides        # (sN,sC,sH,sW) for small-C strided sources
        self.base_ptr = base_ptr      # explicit device pointer (channel slice of a bigger tensor)
        self.dz = None                # gradient buffer wrt the post-norm value

    def src(self, with_prologue=True):
        s = L.Src()
        s.ptr = self.base_ptr if self.base_ptr is not None else L.ptr(self.t)
        s.C = self.C
        if with_prologue:
            s.aff = L.ptr(self.aff)
            s.mask = L.ptr(self.mask)
        if self.strides is not None:
            s.sN, s.sC, s.sH, s.sW = self.strides
        return s


# MFMA operand precision of the forward / data-gradient contractions: 0 = fp32 (reference parity, default),
# 1 = bf16 operands, 2 = bf16x3 split (include/posegan_hip.h PG_PREC_*).  Weight gradients always run in fp32.
# 3 = bf16 DATA path: sources are materialised once as bf16 tensors (normalised, activated, masked), weights are
# converted per optimiser step, and the contraction DMAs bf16 tiles straight into LDS (fp32 accumulate / outputs).
PRECISION = {"f32": 0, "bf16": 1, "bf16x3": 2, "bf16_data": 3}[os.environ.get("PG_PRECISION", "f32")]

ARENAS = weakref.WeakValueDictionary()    # parameter-arena storage pointer -> ParamArena (weak: arenas die with their model)
WEIGHT_VERSION = {}         # parameter-arena storage -> version, bumped whenever its parameters change (optimiser step,
                            # load_state_dict): the bf16 weight copies of THAT arena go stale
_BF_SRC = {}                # device -> list of bf16 scratch buffers, one per source slot (stream-ordered reuse)
_BF_W = {}                  # (data_ptr, numel) -> [version, nt, t]


def _bf16_weight(W, taps, Cout, Cin, transposed):
    arena = ARENAS.get(W.untyped_storage().data_ptr())
    if arena is not None and not transposed:
        # K-contiguous forward operand = the packed arena layout: a view of the arena's bf16 copy (written by pg_adam_ex)
        off = (W.data_ptr() - arena.params.data_ptr()) // 4
        return arena.bf16_params()[off:off + W.numel()]
    if arena is None:
        # a weight outside any arena (kernel tests, module-level API, the padded output-conv weight): converted on every call
        # into a buffer that lives as long as the weight tensor does (stable pointers: launch tape / HIP graph; a cache of
        # CONTENTS keyed by address would serve stale copies when the allocator reuses the address)
        keep = _BF_W_EXT.get((W.data_ptr(), W.numel(), bool(transposed)))
        if keep is None or keep[0]() is not W:
            keep = (weakref.ref(W), torch.empty(W.numel(), dtype=torch.bfloat16, device=W.device))
            _BF_W_EXT[(W.data_ptr(), W.numel(), bool(transposed))] = keep
        buf = keep[1]
        L.call("pg_weights_to_bf16", L.ptr(W), taps, Cout, Cin, None if transposed else L.ptr(buf),
               L.ptr(buf) if transposed else None, L.stream())
        return buf
    if BATCH_WT and W.dim() == 4:
        off = (W.data_ptr() - arena.params.data_ptr()) // 4
        return arena.bf16_params_t()[off:off + W.numel()]
    key = (W.data_ptr(), W.numel())
    ver = arena.version()
    ent = _BF_W.get(key)
    if ent is None:
        ent = _BF_W[key] = [-1, -1, None, None]          # versions of the [nt, t] copies, their (persistent) buffers
    idx = 1 if transposed else 0
    if ent[2 + idx] is None:
        ent[2 + idx] = torch.empty(W.numel(), dtype=torch.bfloat16, device=W.device)
    if ent[idx] != ver:
        buf = ent[2 + idx]
        L.call("pg_weights_to_bf16", L.ptr(W), taps, Cout, Cin, None if transposed else L.ptr(buf),
               L.ptr(buf) if transposed else None, L.stream())
        ent[idx] = ver
    return ent[2 + idx]


OUT_FWD_FUSED = os.environ.get("PG_NO_OUT_FWD_FUSED") is None     # ablation switch: the output convolution's forward as materialise + contraction + tap gather
WARP_BBOX = os.environ.get("PG_NO_WARP_BBOX") is None    # ablation switch: warp backward without the mask bounding boxes
BATCH_WT = os.environ.get("PG_NO_BATCH_WT") is None      # ablation switch: one pg_weights_to_bf16 launch per layer and step
_BF_W_EXT = {}              # (data_ptr, numel, transposed) -> (weakref to the weight tensor, its bf16 buffer)
# bf16 STORAGE (round 3): on the bf16 data path the GENERATOR keeps its raw activations and the gradients flowing through
# them as bf16 tensors (PG_NO_BF16_STORE=1: fp32 storage as in round 2).  Kernels that take raw device pointers learn the
# dtype from this registry (data_ptr -> tensor) or from per-descriptor flags (lib.make_dst, pg_conv_t.out_bf16).
BF16_STORE = os.environ.get("PG_NO_BF16_STORE") is None
_BF16_TENSORS = weakref.WeakValueDictionary()


def bf16_store():
    return PRECISION == 3 and BF16_STORE


def _reg_bf16(t):
    if t.dtype == torch.bfloat16:
        _BF16_TENSORS[t.data_ptr()] = t
    return t


def _is_bf16_ptr(ptr):
    return int(ptr or 0) in _BF16_TENSORS


_BF_CTX = None              # bf16 operand cache of the pass that is running (set by the engines, see BfCache)
_BF_CTX_X = None            # during a backward pass: the forward pass's cache (activated inputs = weight-gradient operands)


# (round 4) pg_norm_finalize folded into the materialisation pass that reads the normalised tensor first: NormState.forward
# registers the layer here instead of launching the 5 us finalize kernel; the first BfCache.get / get2 that asks for an operand
# with this affine passes the statistics to pg_materialise_bf16_norm, which computes the affine itself and publishes it.  Every
# other reader of an affine (warp kernels, fp32 prologues, backward) must call _flush_norm first — or find it already published.
FOLD_FINALIZE = os.environ.get("PG_NO_FOLD_FINALIZE") is None
_PENDING_NORM = {}          # data_ptr of the `aff` tensor -> (NormState, gamma, beta, N, Lr)


def _flush_norm(aff_ptr):
    """launch the stand-alone finalize of a norm layer whose affine is still pending (a consumer needs it NOW)"""
    pend = _PENDING_NORM.pop(int(aff_ptr or 0), None) if _PENDING_NORM else None
    if pend is not None:
        st, gamma, beta, N, Lr = pend
        L.call("pg_norm_finalize", L.ptr(st.sums), L.ptr(gamma), L.ptr(beta), N, Lr, NORM_EPS, L.ptr(st.mr), L.ptr(st.aff), L.stream())


def _flush_all_norms():
    for k in list(_PENDING_NORM):
        _flush_norm(k)


def _materialise(ptr, x_bf16, aff, mask, act, N, HW, C, out, out2=None, act2=0):
    """pg_materialise_bf16_ex, or its form with the pending finalize of `aff`'s norm layer folded in"""
    pend = _PENDING_NORM.pop(int(aff or 0), None) if (_PENDING_NORM and aff) else None
    if pend is not None:
        st, gamma, beta, n_, Lr = pend
        L.call("pg_materialise_bf16_norm", ptr, x_bf16, L.ptr(st.sums), L.ptr(gamma), L.ptr(beta), Lr, NORM_EPS, L.ptr(st.mr),
               L.ptr(st.aff), mask, act, N, HW, C, out, out2, act2, L.stream())
    else:
        L.call("pg_materialise_bf16_ex", ptr, x_bf16, aff, mask, act, N, HW, C, out, out2, act2, L.stream())


class BfCache:
    """bf16 operand tensors of ONE engine pass, keyed by (tensor, channels, activation, deferred affine, mask): a tensor
    that feeds several contractions (forward + weight gradient, data gradient + weight gradient) is converted ONCE per
    pass; the buffers persist across iterations, `begin()` only invalidates their contents."""

    def __init__(self):
        self.buf, self.valid = {}, set()

    def begin(self):
        self.valid.clear()

    @staticmethod
    def key(ptr, C, act, aff, mask):
        return (int(ptr or 0), int(C), int(act), int(aff or 0), int(mask or 0))

    def lookup(self, ptr, C, act, aff, mask):
        k = self.key(ptr, C, act, aff, mask)
        return self.buf[k] if k in self.valid else None

    def reserve(self, ptr, C, act, aff, mask, need, dev):
        """the buffer of this key, marked valid: the caller's kernel writes the operand itself"""
        k = self.key(ptr, C, act, aff, mask)
        b = self._buffer(k, need, dev)
        self.valid.add(k)
        return b

    def adopt(self, ptr, C, act, aff, mask, tensor):
        """`tensor` (bf16, written by its producer) IS the operand of this key: no materialisation pass"""
        k = self.key(ptr, C, act, aff, mask)
        self.buf[k] = tensor
        self.valid.add(k)
        return tensor

    def _buffer(self, k, need, dev):
        b = self.buf.get(k)
        if b is None or b.numel() < need or b.data_ptr() in _BF16_TENSORS:     # never write into an adopted storage tensor
            b = self.buf[k] = torch.empty(need, dtype=torch.bfloat16, device=dev)
            self.valid.discard(k)
        return b

    def get(self, ptr, C, act, aff, mask, N, HW, dev):
        """the bf16 NHWC tensor bf16(act((a*x+b)*mask)), materialised on the current stream if this pass has not yet.
        A raw tensor in bf16 STORAGE that needs no prologue (gradients) is its own operand."""
        k = self.key(ptr, C, act, aff, mask)
        if k in self.valid:
            return self.buf[k]
        src = _BF16_TENSORS.get(int(ptr or 0))
        if src is not None and act == L.ACT_NONE and not aff and not mask:
            return self.adopt(ptr, C, act, aff, mask, src)
        b = self._buffer(k, N * HW * C, dev)
        _materialise(ptr, 1 if src is not None else 0, aff, mask, act, N, HW, C, L.ptr(b))
        self.valid.add(k)
        return b

    def get2(self, ptr, C, act, act2, aff, mask, N, HW, dev):
        """both activated copies of one normalised tensor in ONE pass over the raw tensor (an encoder skip is read through
        LeakyReLU by the next level and through ReLU by the decoder)"""
        k1, k2 = self.key(ptr, C, act, aff, mask), self.key(ptr, C, act2, aff, mask)
        if k1 in self.valid and k2 in self.valid:
            return
        b1, b2 = self._buffer(k1, N * HW * C, dev), self._buffer(k2, N * HW * C, dev)
        _materialise(ptr, 1 if _is_bf16_ptr(ptr) else 0, aff, mask, act, N, HW, C, L.ptr(b1), L.ptr(b2), act2)
        self.valid.add(k1)
        self.valid.add(k2)


def _bf16_sources(srcs, N, Hi, Wi, act, dev):
    """Materialise every source as a bf16 NHWC tensor: bf16(act((a*x+b)*mask)); returns pure Src descriptors."""
    pool = _BF_SRC.setdefault(dev, [None] * L.PG_MAX_SRC)
    out = []
    for j, s in enumerate(srcs):
        if _BF_CTX is not None:
            t = _BF_CTX.get(s.ptr, s.C, act, s.aff, s.mask, N, Hi * Wi, dev)
        else:
            need = N * Hi * Wi * s.C
            if pool[j] is None or pool[j].numel() < need:
                pool[j] = torch.empty(need, dtype=torch.bfloat16, device=dev)
            _materialise(s.ptr, 1 if _is_bf16_ptr(s.ptr) else 0, s.aff, s.mask, act, N, Hi * Wi, s.C, L.ptr(pool[j]))
            t = pool[j]
        q = L.Src()
        q.ptr, q.C = L.ptr(t), s.C
        out.append(q)
    return out



def _conv(srcs, N, Hi, Wi, act, mode, K, stride, pad, Ho, Wo, W, wCout, wCin, transposed=False, scalar_in=False,
          out=None, out_strides=None, bias=None, out_act=L.OUT_NONE, dsts=None, n_off=0, n_cnt=0, ksplit=0, stats=None,
          allow_n32=False):
    prec = PRECISION
    if prec == 3:
        ncols = n_cnt if n_cnt > 0 else (wCin if transposed else wCout)
        # 32 output columns have ONE bf16 kernel: the 512 x 64 tile with half its columns masked (plain epilogue, a launch large
        # enough to be taken by it) — only the output convolution's tap launch asks for it (allow_n32); pg_conv refuses loudly
        # if such a launch falls through to the 128 x 32 tile, which has no bf16 instantiation
        ok = (not scalar_in and n_off == 0 and n_cnt == 0 and (ncols > 32 or (ncols == 32 and allow_n32))
              and all(s.C % 64 == 0 for s in srcs) and isinstance(W, torch.Tensor))
        if ok:      # bf16 tensors in, K-contiguous bf16 weights (per-tap transposed copy for the data-gradient)
            srcs = _bf16_sources(srcs, N, Hi, Wi, act, W.device)
            W = _bf16_weight(W, K * K, wCout, wCin, transposed)
            if transposed:
                wCout, wCin, transposed = wCin, wCout, False
            act = L.ACT_NONE
        else:
            if any(_is_bf16_ptr(s.ptr) for s in srcs) or (isinstance(out, torch.Tensor) and out.dtype == torch.bfloat16):
                raise RuntimeError("bf16 storage: this convolution is not eligible for the bf16 data path (channels %s -> %d)"
                                   % ([s.C for s in srcs], ncols))
            for s_ in srcs:          # the fp32 prologue reads the deferred affine itself
                _flush_norm(s_.aff)
            prec = 0
    d = L.ConvDesc()
    for i, s in enumerate(srcs):
        d.src[i] = s
    d.nsrc, d.N, d.Hi, d.Wi = len(srcs), N, Hi, Wi
    d.act, d.scalar_in = act, 1 if scalar_in else 0
    d.mode, d.KH, d.KW, d.stride, d.pad, d.Ho, d.Wo = mode, K, K, stride, pad, Ho, Wo
    d.w_transposed = 1 if transposed else 0
    d.W, d.wCout, d.wCin = L.ptr(W), wCout, wCin
    d.n_off, d.n_cnt = n_off, n_cnt
    if dsts is None:
        ncnt = n_cnt if n_cnt > 0 else (wCin if transposed else wCout)
        d.epilogue = 0
        d.out = out if isinstance(out, int) else L.ptr(out)
        d.out_bf16 = 1 if (isinstance(out, torch.Tensor) and out.dtype == torch.bfloat16) else 0
        d.bias = L.ptr(bias)
        d.out_act = out_act
        if out_strides is None:
            out_strides = (Ho * Wo * ncnt, 1, Wo * ncnt, ncnt)
        d.oN, d.oC, d.oH, d.oW = out_strides
    else:
        d.epilogue = 1
        for i, t in enumerate(dsts):
            d.dst[i] = t
        d.ndst = len(dsts)
    d.ksplit = ksplit
    d.precision = prec
    d.stats = L.ptr(stats)
    if SPLITK_WS_BYTES > 0 and os.environ.get("PG_WS_SKIP") != str(d.epilogue):     # PG_WS_SKIP: debugging switch
        dev_ = W.device if isinstance(W, torch.Tensor) else torch.device("cuda", torch.cuda.current_device())
        # one scratch per (device, stream): launches on different streams run concurrently (the two encoder chains, round 5)
        wkey = (dev_, L.stream().value)
        ws = _SPLITK_WS.get(wkey)
        if ws is None:
            ws = _SPLITK_WS[wkey] = torch.empty(SPLITK_WS_BYTES // 4, dtype=torch.float32, device=dev_)
        d.workspace, d.workspace_bytes = ws.data_ptr(), SPLITK_WS_BYTES
    if PROFILER is not None:
        sp = (Ho * Wo) if mode == 0 else (Hi * Wi)
        ncnt_ = n_cnt if n_cnt > 0 else (wCin if transposed else wCout)
        kdim = wCout if transposed else wCin
        PROFILER.launch("conv", 2.0 * N * sp * K * K * kdim * ncnt_,
                        lambda: L.check(L.load().pg_conv(d, L.stream()), "pg_conv"))
        return L.load().pg_last_launch_info()
    L.check(L.load().pg_conv(d, L.stream()), "pg_conv")
    return L.load().pg_last_launch_info()      # tile code, split-K, PG_INFO_BSUMS (thread-local, set by the call above)


def _conv_dgrad(gy_src, N, Hi, Wi, mode, K, stride, pad, Ho, Wo, W, Cout, Cin, dsts, ksplit=0):
    """Data-gradient of a layer whose packed weight is W [K][K][Cout][Cin]: contraction over (taps, Cout) of the
    upstream gradient (N,Hi,Wi,Cout) into the layer input's (N,Ho,Wo,Cin), scattered to `dsts` with act'/mask applied.
    (A per-tap pre-transposed weight copy was measured: no gain over reading the [k][n] operand directly.)"""
    return _conv([gy_src], N, Hi, Wi, L.ACT_NONE, mode, K, stride, pad, Ho, Wo, W, Cout, Cin, transposed=True, dsts=dsts,
                 ksplit=ksplit)


REPLAY_CTR = None         # device uint64 counter of a HIP-graph replay session (runtime/graph.py); None = host-side scalars


NORM_BWD_BF16 = os.environ.get("PG_NO_NORM_BWD_BF16") is None    # ablation switch: separate materialisation of dy
FUSE_NORM_SUMS = os.environ.get("PG_NO_FUSED_NORM_SUMS") is None   # ablation switch: norm backward's reduce pass always runs
STEM_EMIT_BF16 = os.environ.get("PG_NO_STEM_EMIT_BF16") is None  # ablation switch: separate materialisation of the level-0 output
STEM_BF16 = os.environ.get("PG_NO_STEM_BF16") is None     # ablation switch: fp32 first-layer kernels on the bf16 data path
DISC_BF16_STORE = os.environ.get("PG_DISC_F32_STORE") is None   # (round 6) bf16 STORAGE for the discriminator; the switch = round 5's fp32 storage


def stem_pack_floats(K, cin):
    """Size (in floats) of the weight-repack scratch of a first layer: fp32 [Cin][K*K][64] or the packed bf16 filter."""
    return max(cin * K * K * 64, (int(L.load().pg_stem_pack_elems(K, cin)) + 1) // 2)


def _stem_conv_stored(acts, N, Hi, Wi, K, stride, pad, W, bias, wt_buf, raw, outs):
    """First-layer convolution in bf16 STORAGE: `raw` (bf16 NHWC tensor) receives bf16(conv + bias); `outs` = up to two
    (activation, bf16 tensor) pairs written in the same pass (the operands of the layers that read this output)."""
    cin = sum(a.C for a in acts)
    L.call("pg_stem_pack_bf16", L.ptr(W), K, cin, L.ptr(wt_buf), L.stream())
    arr = (L.Src * len(acts))(*[a.src() for a in acts])
    o = list(outs) + [(L.ACT_NONE, None)] * (2 - len(outs))
    L.call("pg_stem_conv_bf16_v3", arr, len(acts), N, Hi, Wi, K, stride, pad, L.ptr(wt_buf), L.ptr(bias), None,
           L.ptr(raw), L.ACT_NONE, L.ptr(o[0][1]), o[0][0], L.ptr(o[1][1]), o[1][0], L.stream())


def _small_cin_conv(acts, N, Hi, Wi, K, stride, pad, W, bias, wt_buf, out, next_act=None, bf_ptr=None):
    """First-layer convolution (few NCHW input channels -> 64 NHWC): repack the weights, then the patch kernel.
    `next_act`: activation of the layer that reads the output next — on the bf16 data path the kernel then also writes that
    layer's bf16 operand into the pass's operand cache."""
    cin = sum(a.C for a in acts)
    if PRECISION == 3 and STEM_BF16 and cin <= 80:
        L.call("pg_stem_pack_bf16", L.ptr(W), K, cin, L.ptr(wt_buf), L.stream())
        arr = (L.Src * len(acts))(*[a.src() for a in acts])
        optr = out if isinstance(out, int) else L.ptr(out)
        bf = bf_ptr                      # a slice of an operand the caller reserved itself (discriminator: one call per pair)
        if bf is None and next_act is not None and _BF_CTX is not None and STEM_EMIT_BF16:
            Ho, Wo = (Hi + 2 * pad - K) // stride + 1, (Wi + 2 * pad - K) // stride + 1
            bf = L.ptr(_BF_CTX.reserve(optr, 64, next_act, None, None, N * Ho * Wo * 64, W.device))
        L.call("pg_stem_conv_bf16_ex", arr, len(acts), N, Hi, Wi, K, stride, pad, L.ptr(wt_buf), L.ptr(bias), optr, bf,
               next_act if bf is not None else L.ACT_NONE, L.stream())
        return
    L.call("pg_repack_small_cin", L.ptr(W), K, K, 64, cin, L.ptr(wt_buf), L.stream())
    arr = (L.Src * len(acts))(*[a.src() for a in acts])
    L.call("pg_small_cin_conv", arr, len(acts), N, Hi, Wi, K, stride, pad, L.ptr(wt_buf), L.ptr(bias),
           out if isinstance(out, int) else L.ptr(out), L.stream())


# scratch of split-K pg_conv launches (ksplit partial tiles + one fix-up kernel instead of float atomics); 0 disables
SPLITK_WS_BYTES = int(os.environ.get("PG_SPLITK_WS_MB", "256")) << 20
_SPLITK_WS = {}
# Weight gradients are off the critical path of a backward pass (only the optimiser needs them), so they are enqueued
# on a SIDE stream: the small deep layers leave most CUs idle (one short round of workgroups), and the weight-gradient
# of layer l then overlaps the data-gradient of layer l and the norm backward of layer l-1 on the main stream
# (+1.4 % fp32, +3 % on the bf16 data path; running the two encoders of the forward pass on two streams: +-0).
# Order: the side stream waits for the main stream at the call (dz complete, and with it every parameter gradient the
# main stream wrote for the same layer before the call: norm gamma / beta, biases); the main stream waits for the side
# stream at the end of the pass only.  A gradient range handed to the data-parallel reducer is ordered by the reducer's
# own events (runtime/dp.py: _wait_producers): the invariant it relies on — every main-stream write into the gradient
# arena of a layer is enqueued BEFORE that layer's _wgrad call, and _ready() for the layer comes after it — is stress-tested
# by tests/test_gpu_round3.py::test_reducer_stream_order_under_main_stream_delay.
SIDE_STREAM = os.environ.get("PG_NO_SIDE_STREAM") is None
_SIDE = {}


def _raw(stream):
    return ctypes.c_void_p(stream.cuda_stream)


def dev_zero(t):
    """t.zero_() as a library call (stream-ordered on torch's current stream; recorded on a launch tape)"""
    if not t.is_cuda:
        t.zero_()
        return
    L.call("pg_zero", L.ptr(t), t.numel() * t.element_size(), L.stream())


def dev_copy(dst, src):
    """dst.copy_(src) for equal dtype / contiguous tensors as a library call; anything else goes through torch"""
    if (dst.is_cuda and dst.dtype == src.dtype and dst.is_contiguous() and src.is_contiguous() and dst.numel() == src.numel()
            and dst.device == src.device):
        L.call("pg_copy", L.ptr(dst), L.ptr(src), dst.numel() * dst.element_size(), L.stream())
    else:
        if REPLAY_CTR is not None and dst.is_cuda:
            # a replay session (launch tape / HIP graph) records library enqueues only: a torch copy would silently drop out
            raise RuntimeError("dev_copy: dtype / layout mismatch (%s %s -> %s %s) needs a torch copy, which a replay session cannot "
                               "record" % (src.dtype, tuple(src.shape), dst.dtype, tuple(dst.shape)))
        dst.copy_(src.reshape(dst.shape) if src.numel() == dst.numel() else src)


def _side_stream():
    dev = torch.cuda.current_device()
    st = _SIDE.get(dev)
    if st is None:
        st = _SIDE[dev] = torch.cuda.Stream(device=dev)
    return st


def _join_side():
    """main stream waits for everything enqueued on the side stream so far."""
    if SIDE_STREAM and _SIDE.get(torch.cuda.current_device()) is not None:
        L.call("pg_stream_wait", L.stream(), _raw(_SIDE[torch.cuda.current_device()]))


# Third stream (round 4): the deformable skips' kernels — mask pyramid, limb-mask boxes, warp forward, warp backward — are
# instruction- / latency-bound passes over the four high-resolution levels that depend only on the SHALLOW encoder levels
# (forward) / the shallow decoder blocks (backward).  The DEEP layers (16^2 ... 4^2 maps: ~25 short launches per pass that
# leave most CUs idle) do not touch them, so the warp kernels run on an auxiliary stream next to that chain:
#   forward :  app 0..3, pose 0..3 | fork: [aux: warp l = 0..3]  [main: app 4.., pose 4.., decoder blocks on unwarped levels] | join
#   backward:  ... decoder block of level 3 | fork: [aux: warp backward l = 3..0]  [main: deep decoder blocks, encoder levels
#              whose input gradient no warp writes] | join before the first data gradient that accumulates into a warped level
AUX_STREAM = os.environ.get("PG_NO_AUX_STREAM") is None
_AUX = {}


def _aux_on():
    return AUX_STREAM and SIDE_STREAM and torch.cuda.is_available()


def _aux_stream():
    dev = torch.cuda.current_device()
    st = _AUX.get(dev)
    if st is None:
        st = _AUX[dev] = torch.cuda.Stream(device=dev)
    return st


@contextlib.contextmanager
def _on_aux(fork=True):
    """run the body's launches on the auxiliary stream (fork: it first waits for everything on the main stream so far)"""
    if not _aux_on():
        yield
        return
    aux = _aux_stream()
    if fork:
        L.call("pg_stream_wait", _raw(aux), L.stream())
    with torch.cuda.stream(aux):
        yield


def _join_aux():
    """main stream waits for everything enqueued on the auxiliary stream so far"""
    if _aux_on() and _AUX.get(torch.cuda.current_device()) is not None:
        L.call("pg_stream_wait", L.stream(), _raw(_AUX[torch.cuda.current_device()]))


# Fourth stream (round 5): the two encoders of the deformable generator are independent chains between the network input and
# the decoder (forward) / between the decoder's data gradients and the first layers (backward).  Their 16^2 ... 4^2 levels are
# short launches that leave most CUs idle (and, at batch 4 per GPU — the per-GPU shape of BASELINE.json configs[3] — so does
# every level): the pose encoder's levels >= ENC_PAR_LEVEL run on a second stream next to the appearance encoder's.
#   PG_ENC_PAR=0 switches it off; PG_ENC_PAR_LEVEL=l pins the first concurrent level (default: chosen per engine from the
#   workgroup count of the level's forward launch, GeneratorEngine._enc_par_level)
ENC_PAR = os.environ.get("PG_ENC_PAR", "1") != "0"
_ENC2 = {}


def _enc2_on():
    return ENC_PAR and SIDE_STREAM and torch.cuda.is_available()


def _enc2_stream():
    dev = torch.cuda.current_device()
    st = _ENC2.get(dev)
    if st is None:
        st = _ENC2[dev] = torch.cuda.Stream(device=dev)
    return st


@contextlib.contextmanager
def _on_enc2(fork=False):
    """run the body's launches on the second encoder stream (fork: it first waits for everything on the main stream so far)"""
    st = _enc2_stream()
    if fork:
        L.call("pg_stream_wait", _raw(st), L.stream())
    with torch.cuda.stream(st):
        yield


def _join_enc2():
    """main stream waits for everything enqueued on the second encoder stream so far"""
    if _ENC2.get(torch.cuda.current_device()) is not None:
        L.call("pg_stream_wait", L.stream(), _raw(_ENC2[torch.cuda.current_device()]))


OUT_CONV_STREAM = os.environ.get("PG_NO_OUT_CONV_STREAM") is None   # ablation switch: K=32 pg_conv launch instead
_BF_WG = {}        # device -> [small operand, large operand planes, fp32 product] scratch of the bf16 weight gradient
SMALL_CIN_DGRAD = os.environ.get("PG_NO_SMALL_CIN_DGRAD") is None   # ablation switch: GEMM-N = 3 pg_conv launch instead
SMALL_CIN_WGRAD = os.environ.get("PG_NO_SMALL_CIN_WGRAD") is None   # ablation switch: generic per-tap kernel
SMALL_CIN_WGRAD_WS = 512 * 64 * 704     # floats: per-workgroup partials of the first-layer weight gradient (<= PG_SMALL_CIN_WGRAD_WS)
_SCW_WS = {}
WGRAD_BF16_MIN_FLOPS = float(os.environ.get("PG_WG_THR", "4e9"))   # below this the tap products are launch-bound: fp32 kernel


def _wgrad_bf16(srcs, N, act, dY, Cout, Cin, x_is_large, Hs, Ws, Hl, Wl, dW):
    """Weight gradient of a k4/s2/p1 Block convolution (conv: x large, dY small; conv-transpose + crop: x small, dY
    large) on the bf16 data path: both tensors are written once as channel-major, zero-bordered bf16 images on the
    SMALL pixel grid (the large one as its four stride-2 phase planes), after which tap (r, s) of the gradient is an NT
    GEMM over the pixel axis whose large-side operand is only shifted by dyq*Wp + dxq elements (pg_gemm_taps_bf16)."""
    dev = dW.device
    Wp = (Ws + 2 + 7) // 8 * 8
    Kp = (N * (Hs + 2) * Wp + 63) // 64 * 64
    slack = Wp + 64
    Cs, Cl = (Cout, Cin) if x_is_large else (Cin, Cout)
    need = (Cs * Kp + 2 * slack, 4 * Cl * Kp + 2 * slack, 16 * Cout * Cin)
    pool = _BF_WG.setdefault(dev, [None, None, None])
    for i, (n_el, dt) in enumerate(zip(need, (torch.bfloat16, torch.bfloat16, torch.float32))):
        if pool[i] is None or pool[i].numel() < n_el:
            pool[i] = torch.zeros(n_el, dtype=dt, device=dev)
    small = pool[0].data_ptr() + 2 * slack
    large = pool[1].data_ptr() + 2 * slack
    dYp = dY if isinstance(dY, int) else L.ptr(dY)

    def put(ptr_x, aff, mask, a, C, H, W, sub, py, px, out_ptr):
        L.call("pg_channel_major_bf16", ptr_x, aff, mask, a, N, H, W, C, sub, py, px, Hs, Ws, Wp, Kp, out_ptr, L.stream())

    c0 = 0
    for s_ in srcs:                      # the (virtually concatenated) input, activated like the forward prologue
        if x_is_large:
            for ph in range(4):
                put(s_.ptr, s_.aff, s_.mask, act, s_.C, Hl, Wl, 2, ph >> 1, ph & 1, large + 2 * ((ph * Cin + c0) * Kp))
        else:
            put(s_.ptr, s_.aff, s_.mask, act, s_.C, Hs, Ws, 1, 0, 0, small + 2 * (c0 * Kp))
        c0 += s_.C
    if x_is_large:
        put(dYp, None, None, L.ACT_NONE, Cout, Hs, Ws, 1, 0, 0, small)
    else:
        for ph in range(4):
            put(dYp, None, None, L.ACT_NONE, Cout, Hl, Wl, 2, ph >> 1, ph & 1, large + 2 * (ph * Cout * Kp))
    offs = []
    for r in range(4):
        for q in range(4):
            py, px = (r - 1) % 2, (q - 1) % 2
            dyq, dxq = (r - 1 - py) // 2, (q - 1 - px) // 2
            offs.append((py * 2 + px) * Cl * Kp + dyq * Wp + dxq)
    zero = torch.zeros(16, dtype=torch.int64)
    shift = torch.tensor(offs, dtype=torch.int64)
    if x_is_large:      # A = gradient [Cout][Kp] (small grid), B = input planes
        a_ptr, b_ptr, a_off, b_off = small, large, zero, shift
    else:               # A = gradient planes, B = input [Cin][Kp]
        a_ptr, b_ptr, a_off, b_off = large, small, shift, zero
    prod = pool[2][:16 * Cout * Cin]
    L.call("pg_gemm_taps_bf16", a_ptr, b_ptr, Cout, Cin, Kp, 16, a_off.data_ptr(), b_off.data_ptr(), L.ptr(prod), L.stream())
    L.call("pg_add2", L.ptr(dW), L.ptr(dW), L.ptr(prod), 16 * Cout * Cin, L.stream())


_BF_WGT = {}       # device -> bf16 scratch of the transposing-read weight gradient: [x per source slot ..., dY]
# pixel-count floor of the transposing-read weight gradient (0 = none): with the small-tile rule of pg_wgrad_bf16 for few-pixel
# layers, routing them to the fp32 kernel instead measured the same (batch 4: 505 vs 507 img/s)
WGRAD_TR_MIN_PIXELS = int(os.environ.get("PG_WGTR_MIN_PIXELS", "0"))
WGRAD_TR64 = os.environ.get("PG_NO_WGRAD_TR64") is None    # ablation switch: 64-channel layers through channel-major copies + NT GEMMs
WGRAD_TR = os.environ.get("PG_NO_WGRAD_TR") is None        # ablation switch: channel-major copies + NT GEMMs instead


def _tr_channels_ok(Cout, cs):
    """pg_wgrad_bf16 tiles: multiples of 128 channels on both operands, or exactly 64 on ONE of them (per source launch)."""
    big = lambda c: c % 128 == 0
    if big(Cout):
        return all(big(c) or (c == 64 and WGRAD_TR64) for c in cs)
    return WGRAD_TR64 and Cout == 64 and all(big(c) for c in cs)


def _k4s2_geometry(Hs, Ws, Hl, Wl, x_is_large):
    """k4 s2 p1: large = 2 small (generator blocks) or, for a Conv2d on an odd map, 2 small + 1 (discriminator)."""
    if Hl == 2 * Hs and Wl == 2 * Ws:
        return True
    return bool(x_is_large) and Hs == (Hl - 2) // 2 + 1 and Ws == (Wl - 2) // 2 + 1


def _wgrad_tr_ok(srcs, Cout, K, stride, pad, scalar_x, y_strides, cout_store, Cin, Hs, Ws, Hl, Wl, dW, N, x_is_large=True,
                 dY=None):
    # operands in bf16 STORAGE have no fp32 kernel to fall back to: every size goes to the transposing-read kernels
    stored = _is_bf16_ptr(dY if isinstance(dY, int) else L.ptr(dY)) or any(_is_bf16_ptr(s_.ptr) for s_ in srcs)
    return (PRECISION == 3 and WGRAD_TR and K == 4 and stride == 2 and pad == 1 and not scalar_x and y_strides is None
            and cout_store == 0 and _k4s2_geometry(Hs, Ws, Hl, Wl, x_is_large) and isinstance(dW, torch.Tensor)
            and _tr_channels_ok(Cout, [s_.C for s_ in srcs])
            and (stored or (2.0 * N * Hs * Ws * 16 * Cin * Cout >= WGRAD_BF16_MIN_FLOPS and N * Hs * Ws >= WGRAD_TR_MIN_PIXELS)))


def _wgrad_bf16_tr(srcs, N, act, dY, Cout, Cin, x_is_large, Hs, Ws, Hl, Wl, dW, dy_bf16=None):
    """Weight gradient on the bf16 data path straight from pixel-major bf16 tensors (csrc/wgrad_bf16.hip): pg_wgrad_bf16
    adds each source's column block into dW — no channel-major copies, no product buffer.  The operands are the bf16
    tensors the forward pass (activated inputs) and the data-gradient (dY) already materialised (BfCache); anything
    missing is converted into this function's OWN scratch (it runs on the weight-gradient side stream)."""
    dev = dW.device
    pool = _BF_WGT.setdefault(dev, [None] * (L.PG_MAX_SRC + 1))
    Hx, Wx = (Hl, Wl) if x_is_large else (Hs, Ws)
    Hy, Wy = (Hs, Ws) if x_is_large else (Hl, Wl)

    def buf(slot, need):
        if pool[slot] is None or pool[slot].numel() < need:
            pool[slot] = torch.empty(need, dtype=torch.bfloat16, device=dev)
        return pool[slot]

    dyb = dy_bf16
    if dyb is None:
        dyp = dY if isinstance(dY, int) else L.ptr(dY)
        dyb = _BF16_TENSORS.get(int(dyp))                 # bf16 STORAGE: the gradient tensor is the operand
        if dyb is None:
            dyb = buf(L.PG_MAX_SRC, N * Hy * Wy * Cout)
            L.call("pg_materialise_bf16", dyp, None, None, L.ACT_NONE, N, Hy * Wy, Cout, L.ptr(dyb), L.stream())
    c0 = 0
    for j, s_ in enumerate(srcs):
        xb = _BF_CTX_X.lookup(s_.ptr, s_.C, act, s_.aff, s_.mask) if _BF_CTX_X is not None else None
        if xb is None:
            xb = buf(j, N * Hx * Wx * s_.C)
            L.call("pg_materialise_bf16_ex", s_.ptr, 1 if _is_bf16_ptr(s_.ptr) else 0, s_.aff, s_.mask, act, N, Hx * Wx, s_.C,
                   L.ptr(xb), None, 0, L.stream())
        L.call("pg_wgrad_bf16_ex", L.ptr(xb), s_.C, L.ptr(dyb), Cout, 1 if x_is_large else 0, N, Hs, Ws, Hl, Wl, L.ptr(dW), Cin,
               c0, 0, L.stream())
        c0 += s_.C


def _wgrad(srcs, N, act, dY, Cout, Cin, x_is_large, Hs, Ws, Hl, Wl, K, stride, pad, dW, scalar_x=False,
           y_strides=None, ksplit=0, cout_store=0, dbias=None):
    """dbias (first layers only): the layer's bias-gradient tensor; returns True when the weight-gradient pass produced it as
    well (pg_stem_wgrad_bf16_v2: a constant-one input channel) — otherwise the caller runs the bias-gradient kernel."""
    dyb = None
    if _BF_CTX is not None and _wgrad_tr_ok(srcs, Cout, K, stride, pad, scalar_x, y_strides, cout_store, Cin, Hs, Ws, Hl, Wl, dW, N,
                                            x_is_large, dY):
        # the bf16 gradient is shared with the data-gradient contraction of the same layer: convert it once, on the MAIN stream
        Hy, Wy = (Hs, Ws) if x_is_large else (Hl, Wl)
        dyb = _BF_CTX.get(dY if isinstance(dY, int) else L.ptr(dY), Cout, L.ACT_NONE, None, None, N, Hy * Wy, dW.device)
    if not SIDE_STREAM or not torch.cuda.is_available():
        return _wgrad_main(srcs, N, act, dY, Cout, Cin, x_is_large, Hs, Ws, Hl, Wl, K, stride, pad, dW, scalar_x,
                           y_strides, ksplit, cout_store, dyb, dbias)
    side = _side_stream()
    L.call("pg_stream_wait", _raw(side), L.stream())
    with torch.cuda.stream(side):
        return _wgrad_main(srcs, N, act, dY, Cout, Cin, x_is_large, Hs, Ws, Hl, Wl, K, stride, pad, dW, scalar_x, y_strides,
                           ksplit, cout_store, dyb, dbias)


STEM_BIAS_FUSED = os.environ.get("PG_NO_STEM_BIAS_FUSED") is None    # ablation switch: separate bias-gradient launches


def _stem_bias_fusable(K, stride, pad, cin):
    """Will the first layer's weight-gradient pass (pg_stem_wgrad_bf16_v2) deliver the bias gradient too?  Mirrors the launch code
    (csrc/stem_bf16.hip launch_stem_wgrad: the bf16 stem kernel, a spare channel slot for the constant-one channel, a tap whose
    input pixel exists for every output pixel).  The caller decides BEFORE the call: a bias gradient that needs its own kernel is
    enqueued in front of the weight-gradient call, so that the data-parallel reducer's ordering invariant holds (every main-stream
    write into a layer's gradients precedes the layer's _wgrad call)."""
    if not (PRECISION == 3 and STEM_BF16 and STEM_BIAS_FUSED and SMALL_CIN_WGRAD):
        return False
    if K == 3 and stride == 1:
        if cin > 36:
            return False
        cp = 24 if cin <= 24 else 36
        return cin < cp and pad == 1
    if K == 4 and stride == 2:
        if cin > 72:
            return False
        cp = 44 if cin <= 44 else 72
        return cin < cp and pad == 0
    return False


def _wgrad_main(srcs, N, act, dY, Cout, Cin, x_is_large, Hs, Ws, Hl, Wl, K, stride, pad, dW, scalar_x=False,
                y_strides=None, ksplit=0, cout_store=0, dy_bf16=None, dbias=None):
    if _wgrad_tr_ok(srcs, Cout, K, stride, pad, scalar_x, y_strides, cout_store, Cin, Hs, Ws, Hl, Wl, dW, N, x_is_large, dY):
        run = lambda: _wgrad_bf16_tr(srcs, N, act, dY, Cout, Cin, x_is_large, Hs, Ws, Hl, Wl, dW, dy_bf16)
        if PROFILER is not None:
            PROFILER.launch("wgrad", 2.0 * N * Hs * Ws * K * K * Cin * Cout, run)
            return
        return run()
    if _is_bf16_ptr(dY if isinstance(dY, int) else L.ptr(dY)) and not scalar_x:
        raise RuntimeError("bf16 storage: this weight gradient is not eligible for the transposing-read kernels "
                           "(Cout %d, sources %s, k%d s%d)" % (Cout, [s_.C for s_ in srcs], K, stride))
    if (PRECISION == 3 and K == 4 and stride == 2 and pad == 1 and not scalar_x and y_strides is None and cout_store == 0
            and Cin > 32 and Hl == 2 * Hs and Wl == 2 * Ws and isinstance(dW, torch.Tensor)
            and 2.0 * N * Hs * Ws * 16 * Cin * Cout >= WGRAD_BF16_MIN_FLOPS):
        if PROFILER is not None:
            PROFILER.launch("wgrad", 2.0 * N * Hs * Ws * K * K * Cin * Cout,
                            lambda: _wgrad_bf16(srcs, N, act, dY, Cout, Cin, x_is_large, Hs, Ws, Hl, Wl, dW))
            return
        return _wgrad_bf16(srcs, N, act, dY, Cout, Cin, x_is_large, Hs, Ws, Hl, Wl, dW)
    if (scalar_x and x_is_large and Cout == 64 and y_strides is None and cout_store == 0 and ksplit == 0
            and act == L.ACT_NONE and SMALL_CIN_WGRAD
            and ((K == 3 and stride == 1 and Cin <= (36 if PRECISION == 3 and STEM_BF16 else 35))
                 or (K == 4 and stride == 2 and Cin <= (72 if PRECISION == 3 and STEM_BF16 else 88)))):
        # first layers (raw NCHW inputs): all taps in one pass over dY, patch gathered from LDS
        arr = (L.Src * len(srcs))(*srcs)
        dYp = dY if isinstance(dY, int) else L.ptr(dY)
        ws = _SCW_WS.get(dW.device)
        if ws is None:
            ws = _SCW_WS[dW.device] = torch.empty(SMALL_CIN_WGRAD_WS, dtype=torch.float32, device=dW.device)
        fn = "pg_stem_wgrad_bf16" if PRECISION == 3 and STEM_BF16 else "pg_small_cin_wgrad"
        db = L.ptr(dbias) if (dbias is not None and STEM_BIAS_FUSED) else None
        if _is_bf16_ptr(dYp):
            assert fn == "pg_stem_wgrad_bf16", "bf16 storage needs the bf16 first-layer kernels"
            run = lambda: L.call("pg_stem_wgrad_bf16_v2", arr, len(srcs), N, Hl, Wl, K, stride, pad, dYp, 1, L.ptr(dW), db, L.ptr(ws),
                                 ws.numel(), L.stream())
        elif fn == "pg_stem_wgrad_bf16":
            run = lambda: L.call("pg_stem_wgrad_bf16_v2", arr, len(srcs), N, Hl, Wl, K, stride, pad, dYp, 0, L.ptr(dW), db, L.ptr(ws),
                                 ws.numel(), L.stream())
        else:
            run = lambda: L.call(fn, arr, len(srcs), N, Hl, Wl, K, stride, pad, dYp, L.ptr(dW), L.ptr(ws), ws.numel(), L.stream())
        if PROFILER is not None:
            PROFILER.launch("wgrad", 2.0 * N * Hs * Ws * K * K * Cin * Cout, run)
        else:
            run()
        return bool(fn == "pg_stem_wgrad_bf16" and db is not None and (L.load().pg_last_launch_info() & L.INFO_STEM_BIAS))
    d = L.WgradDesc()
    for i, s in enumerate(srcs):
        d.src[i] = s
    d.nsrc, d.N, d.act, d.scalar_x = len(srcs), N, act, 1 if scalar_x else 0
    d.dY = dY if isinstance(dY, int) else L.ptr(dY)
    if y_strides is not None:
        d.scalar_y = 1
        d.yN, d.yC, d.yH, d.yW = y_strides
    d.x_is_large = 1 if x_is_large else 0
    d.Hs, d.Ws, d.Hl, d.Wl = Hs, Ws, Hl, Wl
    d.KH, d.KW, d.stride, d.pad = K, K, stride, pad
    d.dW, d.Cout, d.Cin = L.ptr(dW), Cout, Cin
    d.ksplit = ksplit
    d.cout_store = cout_store
    if PROFILER is not None:
        PROFILER.launch("wgrad", 2.0 * N * Hs * Ws * K * K * Cin * Cout,
                        lambda: L.check(L.load().pg_conv_wgrad(d, L.stream()), "pg_conv_wgrad"))
        return
    L.check(L.load().pg_conv_wgrad(d, L.stream()), "pg_conv_wgrad")


# Test aid (tests/test_gpu_round3.py): PG_DEBUG_MAIN_DELAY_US=n puts a busy-wait of n microseconds on the MAIN stream in front
# of every gradient the main stream writes into the parameter-gradient arena (norm gamma / beta, conv biases), i.e. it makes
# those writes LATE relative to everything the side and communication streams do — the stress case of dp._wait_producers.
_MAIN_DELAY_US = int(os.environ.get("PG_DEBUG_MAIN_DELAY_US", "0"))


def _debug_delay():
    if _MAIN_DELAY_US > 0:
        L.call("pg_debug_spin", _MAIN_DELAY_US, L.stream())


class NormScratch:
    """All per-sample statistics buffers of one engine in ONE allocation, so that a forward (backward) pass zeroes
    them with a single memset instead of one tiny fill launch per norm layer."""

    def __init__(self, count, N, device):
        self.sums = torch.zeros(count, N, L.STAT_SLOTS, 2, dtype=torch.float64, device=device)
        # backward: the reduce pass's sums [N][2] and (round 4) the sums written by the PRODUCER of a gradient (pg_dst_t.bsums,
        # [N][STAT_SLOTS][2]) per norm layer — one allocation (`bwd`), one memset per backward pass
        nb, nf = count * N * 2, count * N * L.STAT_SLOTS * 2
        self.bwd = torch.zeros(nb + nf, dtype=torch.float64, device=device)
        self.bsums = self.bwd[:nb].view(count, N, 2)
        self.fsums = self.bwd[nb:].view(count, N, L.STAT_SLOTS, 2)
        self.used = 0

    def take(self):
        i = self.used
        self.used += 1
        return self.sums[i], self.bsums[i], self.fsums[i]


class NormState:
    """Scratch for one per-sample norm: stats (double), mean/rstd, the emitted affine, backward sums."""

    def __init__(self, N, device, scratch=None):
        if scratch is not None:
            self.sums, self.bsums, self.fsums = scratch.take()
            self.shared = True
        else:
            self.sums = torch.zeros(N, L.STAT_SLOTS, 2, dtype=torch.float64, device=device)
            self.bsums = torch.zeros(N, 2, dtype=torch.float64, device=device)
            self.fsums = None
            self.shared = False
        self.mr = torch.zeros(N, 2, dtype=torch.float32, device=device)
        self.aff = torch.zeros(N, 2, dtype=torch.float32, device=device)

    def stats_target(self):
        """The (zeroed) statistics buffer a producing pg_conv accumulates into (pg_conv_t.stats)."""
        if not self.shared:
            dev_zero(self.sums)
        return self.sums

    def forward(self, y, N, Lr, gamma, beta, have_stats=False):
        if not have_stats:
            if not self.shared:
                dev_zero(self.sums)
            L.call("pg_norm_stats", L.ptr(y), N, Lr, L.ptr(self.sums), L.stream())
        if FOLD_FINALIZE and PRECISION == 3 and _BF_CTX is not None and y.is_cuda:
            # bf16 data path: the first reader of the normalised tensor is a materialisation pass — it finalizes (see _PENDING_NORM)
            _PENDING_NORM[self.aff.data_ptr()] = (self, gamma, beta, N, Lr)
            return
        L.call("pg_norm_finalize", L.ptr(self.sums), L.ptr(gamma), L.ptr(beta), N, Lr, NORM_EPS, L.ptr(self.mr),
               L.ptr(self.aff), L.stream())

    def backward(self, dz, y, N, Lr, gamma, dgamma, dbeta, C=0, fused=0, beta=None):
        """dz <- dy in place.  On the bf16 data path (C = channels of the tensor, a multiple of 64) the apply kernel also
        writes dy as the bf16 operand the data- / weight-gradient contractions of this layer will ask the pass's operand
        cache for, so their materialisation pass (4 B read + 2 B write per element) does not run.
        fused (round 4): 1 / 2 = the kernel that wrote dz already accumulated the two per-sample sums into `fsums`
        (pg_dst_t.bsums; sums_mode of pg_norm_bwd_apply_v2) — the reduce pass over (dz, y) does not run."""
        if not self.shared:
            dev_zero(self.bsums)
        _debug_delay()
        io = (1 if dz.dtype == torch.bfloat16 else 0) | (2 if y.dtype == torch.bfloat16 else 0)
        if io:
            # bf16 STORAGE: dz <- dy in place in bf16; the result is the operand of the layer's gradient contractions
            if not fused:
                L.call("pg_norm_bwd_reduce_ex", L.ptr(dz), L.ptr(y), L.ptr(self.mr), N, Lr, L.ptr(self.bsums), io, L.stream())
            bf = None
            if not (io & 1) and PRECISION == 3 and NORM_BWD_BF16 and _BF_CTX is not None and C > 0 and C % 64 == 0:
                bf = _BF_CTX.reserve(L.ptr(dz), C, L.ACT_NONE, None, None, N * Lr, dz.device)
            if fused == 2:
                # sums in activated-operand form divide by gamma: a layer whose gamma is too small takes the plain reduce pass
                # (decided on the device; a well-conditioned layer's launch exits at once — csrc/norm.hip norm_gamma_small)
                L.call("pg_norm_bwd_reduce_guard", L.ptr(dz), L.ptr(y), L.ptr(self.mr), L.ptr(gamma), L.ptr(beta), N, Lr,
                       L.ptr(self.bsums), io, L.stream())
            L.call("pg_norm_bwd_apply_v3", L.ptr(dz), L.ptr(y), L.ptr(self.mr), L.ptr(self.fsums if fused else self.bsums), L.ptr(gamma),
                   L.ptr(beta), N, Lr, L.ptr(dgamma), L.ptr(dbeta), L.ptr(bf), io, int(fused),
                   L.ptr(self.bsums) if fused == 2 else None, L.stream())
            if (io & 1) and _BF_CTX is not None and C > 0:
                _BF_CTX.adopt(L.ptr(dz), C, L.ACT_NONE, None, None, dz)
            return
        bf = None
        if PRECISION == 3 and NORM_BWD_BF16 and _BF_CTX is not None and C > 0 and C % 64 == 0:
            bf = _BF_CTX.reserve(L.ptr(dz), C, L.ACT_NONE, None, None, N * Lr, dz.device)
        if fused:
            L.call("pg_norm_bwd_apply_v2", L.ptr(dz), L.ptr(y), L.ptr(self.mr), L.ptr(self.fsums), L.ptr(gamma), L.ptr(beta), N, Lr,
                   L.ptr(dgamma), L.ptr(dbeta), L.ptr(bf), 0, int(fused), L.stream())
            return
        L.call("pg_norm_bwd_reduce", L.ptr(dz), L.ptr(y), L.ptr(self.mr), N, Lr, L.ptr(self.bsums), L.stream())
        if bf is not None:
            L.call("pg_norm_bwd_apply_ex", L.ptr(dz), L.ptr(y), L.ptr(self.mr), L.ptr(self.bsums), L.ptr(gamma), N, Lr,
                   L.ptr(dgamma), L.ptr(dbeta), L.ptr(bf), L.stream())
        else:
            L.call("pg_norm_bwd_apply", L.ptr(dz), L.ptr(y), L.ptr(self.mr), L.ptr(self.bsums), L.ptr(gamma), N, Lr,
                   L.ptr(dgamma), L.ptr(dbeta), L.stream())


# Adam under the backward pass (round 5).  The optimiser step is a pure streaming pass (28 - 30 bytes per parameter: 0.39 ms for
# the generator's 82 M parameters — 6 % of the batch-4 bf16 iteration, after which the next forward has to wait for it) and the
# deep layers, which hold almost all parameters, finish their gradients EARLY in the backward pass.  A layer's parameters are
# released as soon as (a) its gradients are complete — weight gradient enqueued on the side stream, norm / bias gradients on the
# stream that runs the layer — and (b) the pass has read its weights for the last time (the layer's data gradient is enqueued).
# Released ranges (the arena is laid out in backward-completion order: an advancing offset) are updated on the weight-gradient
# side stream, which first waits for the releasing stream.  Same arithmetic per element as the single launch: bit-identical
# parameters.  Single process only (a data-parallel run must all-reduce first), not in replay sessions, not for the stacked
# generator (its stages accumulate into the shared arena).
# OFF by default since the end of round 5 (PG_EAGER_ADAM=1 switches it on, PG_NO_EAGER_ADAM=1 still forces it off): first measured
# as a gain (bf16 batch 4 577 -> 592 img/s next to the second encoder stream), it became a loss once the small-pixel-count weight
# gradients on the same stream got their smaller tiles and fewer splits — 17 range launches of 35 us queue up between them:
# bf16 batch 4 635 / 643 -> 647 / 662 img/s without it (two boxes), configs[2] 947 -> 955, batch 32 1117 -> 1126, fp32 batch 4
# 174.1 -> 174.8.  The single launch after the pass runs at the copy rate (0.39 ms at 6.3 TB/s).
EAGER_ADAM = os.environ.get("PG_EAGER_ADAM") == "1" and os.environ.get("PG_NO_EAGER_ADAM") is None
EAGER_ADAM_MIN = int(os.environ.get("PG_EAGER_ADAM_MIN", str(1 << 20)))      # elements per range launch (4 MB of parameters)


class EagerAdam:
    def __init__(self, arena, lr, b1=0.5, b2=0.999, eps=1e-8):
        self.A, self.hyper = arena, (lr, b1, b2, eps)
        self.index = {k: i for i, k in enumerate(arena.keys)}

    def begin(self):
        self.free = [False] * len(self.A.keys)
        self.next_key, self.done = 0, 0
        self.streams = {}            # streams that released keys since the last range launch (the two encoder chains)
        self.A.adam_begin(*self.hyper)

    def _end_offset(self, i):
        return self.A.off[self.A.keys[i]] if i < len(self.A.keys) else self.A.total

    def release(self, keys):
        """called on the stream that ran the layers' last reader: their gradients are complete, their weights free"""
        for k in keys:
            self.free[self.index[k]] = True
        cur = L.stream()
        self.streams[cur.value] = cur
        while self.next_key < len(self.free) and self.free[self.next_key]:
            self.next_key += 1
        upto = self._end_offset(self.next_key)
        if upto - self.done >= EAGER_ADAM_MIN and SIDE_STREAM and torch.cuda.is_available():
            side = _side_stream()
            for st in self.streams.values():      # every stream that ran a last reader of the range (events recorded NOW cover them)
                L.call("pg_stream_wait", _raw(side), st)
            self.streams = {}
            with torch.cuda.stream(side):
                self.A.adam_range(self.done, upto)
            self.done = upto

    def finish(self):
        """after the pass (the caller's stream has joined the side stream): whatever is left"""
        self.A.adam_range(self.done, self.A.total)
        self.done = self.A.total
        self.A.adam_end()


def generator_param_order(spec, nlev, ndec, deformable=True):
    """Backward-completion order of the generator's parameters (decoder tail first, encoder level 0 last)."""
    keys = [k for k, _ in spec]
    order = []

    def take(prefix):
        for k in keys:
            if k.startswith(prefix) and k not in order:
                order.append(k)

    take("decoder.net.%d." % ndec)           # final conv (module index ndec = len(nfilters_dec))
    for i in range(ndec - 2, -1, -1):
        take("decoder.net.%d." % i)
    encs = ("encoder_app", "encoder_pose") if deformable else ("encoder",)
    for l in range(nlev - 1, -1, -1):
        for e in encs:
            take("%s.net.%d." % (e, l))
    assert len(order) == len(keys), (len(order), len(keys))
    return order


# ------------------------------------------------------------------------------------------ generator
class GeneratorEngine:
    """Deformable_Generator (reference models/networks.py:252-288) / src_baseline Generator
    (src_baseline/models/networks.py:238-253) forward + backward for a fixed (N,H,W)."""

    def __init__(self, arena, N, H, W, pose_dim, nfilters_enc, nfilters_dec, deformable=True, align_corners=False,
                 device="cuda", n_warps=T_WARPS, masked=True, bf16_ok=True):
        # n_warps / masked: warp_skip='mask' -> 10 masked limb transforms; 'full' / 'none' -> ONE unmasked transform
        # (reference networks.py:283: AffineTransformLayer(10 if warp_skip == 'mask' else 1, ...))
        self.A, self.N, self.H, self.W, self.P = arena, N, H, W, pose_dim
        self.T, self.masked = (n_warps, masked) if deformable else (0, False)
        self.enc, self.dec = tuple(nfilters_enc), tuple(nfilters_dec)
        self.nlev, self.ndec = len(self.enc), len(self.dec)
        self.deformable, self.align = deformable, 1 if align_corners else 0
        self.dev = device
        self.encs = ("encoder_app", "encoder_pose") if deformable else ("encoder",)
        assert self.ndec == self.nlev
        f32 = dict(dtype=torch.float32, device=device)
        # bf16 STORAGE (round 3): raw activations and their gradients as bf16 on the bf16 data path (first layers need the
        # bf16 stem kernels: <= 80 input channels, 64 outputs)
        cin_max = (3 + 2 * pose_dim) if not deformable else (3 + pose_dim)
        cin_fin_ = (2 if deformable else 1) * self.enc[0] + self.dec[-2]
        chans_ok = (all(c % 128 == 0 for c in self.enc[1:]) and all(c % 128 == 0 for c in self.dec[:-2])
                    and (self.dec[-2] % 128 == 0 or self.dec[-2] == 64) and cin_fin_ <= 256 and cin_fin_ % 32 == 0)
        # every layer must be eligible for the bf16 kernels (there is no fp32 fallback inside the storage mode): 64 first-layer
        # outputs, multiples of 128 channels elsewhere (one 64-channel operand per weight gradient), an output convolution of
        # <= 256 input channels (pg_out_conv_bwd_direct).  Models that are not fall back to fp32 STORAGE here, at build time.
        self.bfs = bool(bf16_store() and bf16_ok and self.enc[0] == 64 and cin_max <= 80 and STEM_BF16 and chans_ok
                        and torch.cuda.is_available())
        act = dict(dtype=torch.bfloat16 if self.bfs else torch.float32, device=device)
        A_ = lambda *shape: _reg_bf16(torch.empty(*shape, **act))
        hw = [(H >> l, W >> l) for l in range(self.nlev)]
        assert all(h << l == H and w << l == W for l, (h, w) in enumerate(hw)), "H,W must be divisible by 2^(levels-1)"
        self.hw = hw
        # encoder activations (raw), grads, norm state
        self.e_raw = {e: [A_(N, hw[l][0], hw[l][1], self.enc[l]) for l in range(self.nlev)] for e in self.encs}
        self.e_dz = {e: [A_(N, hw[l][0], hw[l][1], self.enc[l]) for l in range(self.nlev)] for e in self.encs}
        self.nscr = NormScratch(len(self.encs) * self.nlev + self.ndec, N, device)
        self.e_norm = {e: [NormState(N, device, self.nscr) if 0 < l < self.nlev - 1 else None for l in range(self.nlev)] for e in self.encs}
        # warped appearance skips (levels 0..3)
        self.nwarp = min(4, self.nlev) if deformable else 0
        self.w_out = [A_(N, hw[l][0], hw[l][1], self.enc[l]) for l in range(self.nwarp)]     # bf16 STORAGE: relu(out)
        self.w_g = [A_(N, hw[l][0], hw[l][1], self.enc[l]) for l in range(self.nwarp)]
        self.w_arg = [torch.empty(N, hw[l][0], hw[l][1], self.enc[l], dtype=torch.uint8, device=device) for l in range(self.nwarp)]
        self.lvl_masks = [(torch.empty if self.masked else torch.ones)(N, hw[l][0], hw[l][1], self.T, **f32)
                          for l in range(self.nwarp)]
        # decoder up-block outputs: block i lives at level nlev-2-i
        self.d_raw, self.d_dz, self.d_norm = [], [], []
        for i in range(self.ndec - 1):
            h, w = hw[self.nlev - 2 - i]
            self.d_raw.append(A_(N, h, w, self.dec[i]))
            self.d_dz.append(A_(N, h, w, self.dec[i]))
            self.d_norm.append(NormState(N, device, self.nscr))
        self.drop = [torch.ones(N, self.dec[i], **f32) for i in range(min(3, self.ndec - 1))]
        self.out = torch.empty(N, 3, H, W, **f32)
        cin0 = {"encoder_app": 3 + pose_dim, "encoder_pose": pose_dim, "encoder": 3 + 2 * pose_dim}
        self.wt0 = {e: torch.empty(stem_pack_floats(3, cin0[e]), **f32) for e in self.encs}    # [Cin][9][64] repack of conv 0
        # bf16 STORAGE: the 27 tap columns are padded to 32 when the launch is large enough for the 512 x 64 bf16 kernel (which
        # masks the upper half of its tile: half the fp32 bytes), else to the 64 columns every bf16 kernel takes
        self.fin_cols = 27 if not self.bfs else (32 if (N * H * W) // 512 >= int(os.environ.get("PG_BF16_BIG_MIN", "192")) and
                                                 os.environ.get("PG_NO_FIN32") is None else 64)
        self.y_taps = torch.empty(N, H, W, self.fin_cols, **f32)       # output conv as a 1x1 with N = 9 taps x 3 channels
        cin_fin = (2 if deformable else 1) * self.enc[0] + self.dec[-2]
        self.wt_fin = torch.zeros(self.fin_cols, cin_fin, **f32) if self.bfs else None    # bf16 STORAGE: weight padded to the tile
        self.fin_ws = torch.empty(1024 * cin_fin * 28, **f32) if (self.bfs and cin_fin <= 256) else None
        self.g_taps = torch.empty(N, H, W, 32, **f32)       # im2col of d(pre-tanh): weight- and data-gradient operand
        self.wt_out = torch.zeros((2 if deformable else 1) * self.enc[0] + self.dec[-2], 32, **f32)   # [cin][(tap, co)]
        self.warps = torch.empty(N, max(self.T, 1), 8, **f32)
        self.mask_bbox = torch.empty(N, max(self.T, 1), 4, dtype=torch.int32, device=device) if self.masked else None
        self.input = None
        self._drop_counter = 0
        self.drop_stream = "drop"      # mixed into the dropout key (the trainer sets seed / rank / global iteration)
        self.grad_ready_cb = None      # DP hook: called with the parameter keys whose gradients are complete
        self.param_release_cb = None   # EagerAdam hook: keys whose gradients are complete AND whose weights this pass no longer reads
        self._bf_fwd, self._bf_bwd = BfCache(), BfCache()      # bf16 data path: operand tensors of the last forward / backward
        self._fsum = {}                # (kind, index) -> sums_mode: norm layers whose backward sums a producer's epilogue wrote

    # -------------------------------------------------------------------------------- helpers
    def _ready(self, *prefixes):
        # the reducer orders its collective against BOTH producer streams with events (runtime/dp.py: _wait_producers);
        # the main stream keeps running the data-gradient chain, the side stream the weight gradients
        if self.grad_ready_cb is not None:
            self.grad_ready_cb([k for k in self.A.keys if k.startswith(prefixes)])

    def _release(self, *prefixes):
        if self.param_release_cb is not None:
            self.param_release_cb([k for k in self.A.keys if k.startswith(prefixes)])

    def _enc_par_level(self):
        """First encoder level whose pose-encoder launches go to the second encoder stream (ENC_PAR), nlev = none."""
        if not (self.deformable and len(self.encs) == 2 and _enc2_on()):
            return self.nlev
        pin = os.environ.get("PG_ENC_PAR_LEVEL")
        if pin is not None:
            return max(0, min(self.nlev, int(pin)))
        # measured (round 5, one box, generator forward + backward at batch 32 / bf16 batch-4 step / fp32 batch-4 step):
        # off 17.8 ms / 576 img/s / 173.6 img/s; from level 4 (the 16^2 ... 4^2 levels only) 17.8 / - / -; from level 3 17.5;
        # from level 1 17.3 / 592 / 175.5; from level 0 - / 590 / 175.2.  The first layers stay on the main stream.
        # End of the round, after the weight-gradient retune and with Adam's range launches gone, the picture at SMALL batch turned:
        # bf16 batch 4, level 1 / 2 / 3 / 4 / off: 657 / 675 / 682 / 687 / 699 img/s; configs[2] (batch 8) level 2 / 3 / 4: 973 / 987 /
        # 993; fp32 batch 4, level 1 / 2 / 3 / 4 / off: 175.1 / 174.4 / 177.5 / 178.1 / 178.0; batch 32 (pass / step): level 1
        # 17.01 ms, level 3 17.14 - 17.21, off 17.40; step 1143 - 1145 img/s at every level.  More streams only add contention between
        # launches that are latency-bound anyway: the second encoder stream is for large batches.  Batch 8 / 12 / 16 steps, off against
        # level 1: 890 / 851, 1008 / 962, 1060 / 1031 img/s; 512^2 batch 8: 291 / 280; batch 32: the step is the same at every level
        # (1143 - 1145), the generator forward + backward alone gains 2 % from level 1 (17.01 against 17.40 ms).
        if self.N <= 24:
            return self.nlev
        return min(1, self.nlev)

    def _enc_in_src(self, e, inp):
        """NCHW channel slice of `input` feeding encoder level 0 (get_imgpose, utils/pose_utils.py:227-233)."""
        C = inp.shape[1]
        HW = self.H * self.W
        strides = (C * HW, HW, self.W, 1)
        if e == "encoder_app":
            return Act(inp, 3 + self.P, strides=strides)
        if e == "encoder_pose":
            return Act(inp, self.P, strides=strides, base_ptr=inp.data_ptr() + 4 * (3 + self.P) * HW)
        return Act(inp, C, strides=strides)

    def _enc_act(self, e, l):
        st = self.e_norm[e][l]
        return Act(self.e_raw[e][l], self.enc[l], aff=st.aff if st is not None else None)

    def _dec_sources(self, i):
        """K-sources of decoder block i (the reference's cat([out, skip]), networks.py:241,245), level nlev-1-i."""
        l = self.nlev - 1 - i
        srcs = []
        if i > 0:
            srcs.append(("dec", i - 1, Act(self.d_raw[i - 1], self.dec[i - 1], aff=self.d_norm[i - 1].aff,
                                           mask=self.drop[i - 1] if (i - 1) < len(self.drop) and self.use_drop else None)))
        if self.deformable:
            if l < self.nwarp:
                srcs.append(("warp", l, Act(self.w_out[l], self.enc[l])))
            else:
                srcs.append(("app", l, self._enc_act("encoder_app", l)))
            srcs.append(("pose", l, self._enc_act("encoder_pose", l)))
        else:
            srcs.append(("enc", l, self._enc_act("encoder", l)))
        return srcs

    # -------------------------------------------------------------------------------- forward
    def set_dropout(self, masks=None, train=True, seed=0):
        """masks: list of (N,C) multiplier tensors (parity tests) | None -> device RNG (train) / identity (eval)."""
        self.use_drop = bool(train) or masks is not None
        if masks is not None:
            for d, m in zip(self.drop, masks):
                dev_copy(d, m)
        elif train:
            for i, d in enumerate(self.drop):
                self._drop_counter += 1
                key = int(synth._stream_key(seed, "%s/N%d/%d/%d" % (self.drop_stream, self.N, self._drop_counter, i)))
                if REPLAY_CTR is not None:     # the key is frozen into the graph: the device counter makes every replay differ
                    L.call("pg_dropout_mask_ctr", L.ptr(d), d.numel(), key, 0.5, L.ptr(REPLAY_CTR), L.stream())
                else:
                    L.call("pg_dropout_mask", L.ptr(d), d.numel(), key, 0.5, L.stream())

    def forward(self, inp, warps=None, masks=None):
        global _BF_CTX
        self._bf_fwd.begin()
        _BF_CTX = self._bf_fwd if PRECISION == 3 else None
        try:
            return self._forward(inp, warps, masks)
        finally:
            _BF_CTX = None

    def _forward(self, inp, warps=None, masks=None):
        """inp (N,3+2P,H,W) NCHW fp32; warps (N,10,8); masks (N,10,H,W) fp32|fp64.  Returns out_gen NCHW."""
        A, N, H, W = self.A, self.N, self.H, self.W
        assert tuple(inp.shape) == (N, 3 + 2 * self.P, H, W) and inp.is_contiguous() and inp.dtype == torch.float32
        self.input = inp
        dev_zero(self.nscr.sums)
        if not hasattr(self, "use_drop"):
            self.set_dropout(None, train=True)
        if self.deformable:
            T = self.T
            # (N,T,8) as the reference Dataset emits it; estimate_uniform_transform may hand over 9 values per row
            # (pose_transform.py:322) — only the first six are ever read (pose_transform.py:28)
            w8 = warps.reshape(N, T, -1)
            dev_copy(self.warps, w8 if w8.shape[-1] == 8 else w8[:, :, :8])
            if self.masked:
                assert masks.is_contiguous() and tuple(masks.shape) == (N, T, H, W)
                with _on_aux():
                    for l in range(self.nwarp):
                        L.call("pg_mask_pyramid", L.ptr(masks), 1 if masks.dtype == torch.float64 else 0, N, T, H, W,
                               self.hw[l][0], self.hw[l][1], L.ptr(self.lvl_masks[l]), L.stream())
                    if WARP_BBOX:       # bounding boxes of the non-zero mask regions: the warp backward skips what cannot contribute
                        L.call("pg_mask_bbox", L.ptr(masks), 1 if masks.dtype == torch.float64 else 0, N, T, H, W, L.ptr(self.mask_bbox),
                               L.stream())
        # ---- encoders (reference networks.py:193-202)
        bfs = self.bfs
        assert bfs == (bf16_store() and bfs), "the engine was built for another storage mode (PRECISION changed?)"
        npx = lambda l: self.hw[l][0] * self.hw[l][1]
        # (round 3, tried and dropped: the pose encoder's chain on the side stream next to the appearance encoder's — 22.28 ms
        # against 22.02 ms for the north-star pass, and the 224^2 data-parallel identity test failed under it)
        # shallow levels of both encoders first: everything the warps need exists, then the warps go to the auxiliary stream
        # while the deep levels and the deep decoder blocks run here
        cut = self.nwarp if (self.nwarp > 0 and _aux_on()) else self.nlev
        pl = self._enc_par_level()       # first level of the pose encoder that runs on the second encoder stream (nlev: none)
        if pl < self.nlev:
            if PRECISION == 3:
                A.bf16_params()          # (first use after load_state_dict converts: on THIS stream, before the chains fork)
            e_app, e_pose = self.encs
            self._forward_encoder(e_pose, inp, bfs, npx, 0, pl)
            with _on_enc2(fork=True):
                self._forward_encoder(e_pose, inp, bfs, npx, pl, self.nlev)
            encs_main = (e_app,)
        else:
            encs_main = self.encs
        for e in encs_main:
            self._forward_encoder(e, inp, bfs, npx, 0, cut)
        if cut < self.nlev:
            for l in range(self.nwarp):      # the deepest warped level's affine is still pending: publish it on THIS stream, before the fork
                _flush_norm(L.ptr(self._enc_act("encoder_app", l).aff))
            with _on_aux():
                self._forward_warps(bfs)
            for e in encs_main:
                self._forward_encoder(e, inp, bfs, npx, cut, self.nlev)
        else:
            _join_aux()                  # the mask pyramid / boxes were enqueued on the auxiliary stream (nwarp == nlev: no fork below)
            self._forward_warps(bfs)
        if pl < self.nlev:
            _join_enc2()                 # the decoder's first block reads both encoders
        self._forward_rest(inp, bfs, joined=cut >= self.nlev)
        _flush_all_norms()
        return self.out

    def _forward_encoder(self, e, inp, bfs, npx, l0, l1):
        """levels l0 .. l1-1 of encoder `e` (level 0 = the first convolution)"""
        A, N, H, W = self.A, self.N, self.H, self.W
        if l0 == 0:
            s0 = self._enc_in_src(e, inp)
            if bfs:
                # level 0 has no norm: the stem writes the raw tensor and the operand(s) of its readers in one pass — the next
                # encoder level (LeakyReLU) and, unless the skip is warped first, the decoder (ReLU)
                r0 = self.e_raw[e][0]
                outs = [(L.ACT_LEAKY, _BF_CTX.reserve(L.ptr(r0), 64, L.ACT_LEAKY, None, None, r0.numel(), r0.device))]
                if not (self.deformable and e == "encoder_app" and self.nwarp > 0):
                    outs.append((L.ACT_RELU, _BF_CTX.reserve(L.ptr(r0), 64, L.ACT_RELU, None, None, r0.numel(), r0.device)))
                _stem_conv_stored([s0], N, H, W, 3, 1, 1, A.p(e + ".net.0.weight"), A.p(e + ".net.0.bias"), self.wt0[e], r0, outs)
            elif self.enc[0] == 64:
                _small_cin_conv([s0], N, H, W, 3, 1, 1, A.p(e + ".net.0.weight"), A.p(e + ".net.0.bias"), self.wt0[e],
                                self.e_raw[e][0], next_act=L.ACT_LEAKY)
            else:
                _conv([s0.src()], N, H, W, L.ACT_NONE, 0, 3, 1, 1, H, W, A.p(e + ".net.0.weight"), self.enc[0], s0.C,
                      scalar_in=True, out=self.e_raw[e][0], bias=A.p(e + ".net.0.bias"))
        for l in range(max(1, l0), l1):
            hi, wi = self.hw[l - 1]
            ho, wo = self.hw[l]
            has_norm = l < self.nlev - 1
            xin = self._enc_act(e, l - 1)
            if bfs and l - 1 >= 1 and not (self.deformable and e == "encoder_app" and l - 1 < self.nwarp):
                # skip l-1 is read twice with different activations: both operands in one pass over the raw tensor
                _BF_CTX.get2(L.ptr(xin.t), xin.C, L.ACT_LEAKY, L.ACT_RELU, L.ptr(xin.aff), None, N, npx(l - 1), xin.t.device)
            _conv([xin.src()], N, hi, wi, L.ACT_LEAKY, 0, 4, 2, 1, ho, wo,
                  A.p("%s.net.%d.net.1.weight" % (e, l)), self.enc[l], self.enc[l - 1], out=self.e_raw[e][l],
                  stats=self.e_norm[e][l].stats_target() if has_norm else None)
            if has_norm:
                self.e_norm[e][l].forward(self.e_raw[e][l], N, ho * wo * self.enc[l],
                                          A.p("%s.net.%d.net.2.weight" % (e, l)), A.p("%s.net.%d.net.2.bias" % (e, l)),
                                          have_stats=True)

    def _forward_warps(self, bfs):
        """deformable skips (reference networks.py:279-288, utils/pose_transform.py:69-92)"""
        N, H, W = self.N, self.H, self.W
        for l in range(self.nwarp):
            a = self._enc_act("encoder_app", l)
            _flush_norm(L.ptr(a.aff))        # (the warp may run before the level's materialisation pass: auxiliary stream)
            if bfs:     # bf16 in, relu(out) bf16 out: the stored tensor IS the decoder's operand (and its ReLU-derivative input)
                L.call("pg_warp_mask_max_fwd_io", L.ptr(a.t), L.ptr(a.aff), L.ptr(self.warps), L.ptr(self.lvl_masks[l]), N,
                       self.T, self.enc[l], self.hw[l][0], self.hw[l][1], H, W, self.align, L.ptr(self.w_out[l]),
                       L.ptr(self.w_arg[l]), 7, L.stream())
                _BF_CTX.adopt(L.ptr(self.w_out[l]), self.enc[l], L.ACT_RELU, None, None, self.w_out[l])
                continue
            L.call("pg_warp_mask_max_fwd", L.ptr(a.t), L.ptr(a.aff), L.ptr(self.warps), L.ptr(self.lvl_masks[l]), N,
                   self.T, self.enc[l], self.hw[l][0], self.hw[l][1], H, W, self.align, L.ptr(self.w_out[l]),
                   L.ptr(self.w_arg[l]), L.stream())

    def _forward_rest(self, inp, bfs, joined=True):
        A, N, H, W = self.A, self.N, self.H, self.W
        # ---- decoder (reference networks.py:236-250)
        for i in range(self.ndec - 1):
            srcs = self._dec_sources(i)
            if not joined and any(kind == "warp" for kind, _, _ in srcs):
                _join_aux()              # the first block that reads a warped skip
                joined = True
            hi, wi = self.hw[self.nlev - 1 - i]
            ho, wo = 2 * hi, 2 * wi
            cin = sum(a.C for _, _, a in srcs)
            _conv([a.src() for _, _, a in srcs], N, hi, wi, L.ACT_RELU, 1, 4, 2, 1, ho, wo,
                  A.p("decoder.net.%d.net.1.weight" % i), self.dec[i], cin, out=self.d_raw[i],
                  stats=self.d_norm[i].stats_target())
            self.d_norm[i].forward(self.d_raw[i], N, ho * wo * self.dec[i], A.p("decoder.net.%d.net.3.weight" % i),
                                   A.p("decoder.net.%d.net.3.bias" % i), have_stats=True)
        if not joined:
            _join_aux()
        i = self.ndec - 1
        srcs = self._dec_sources(i)
        cin = sum(a.C for _, _, a in srcs)
        # 256->3 output conv (networks.py:228) re-associated: 1x1 conv to 27 = 9 taps x 3 channels, then tap gather
        # + bias + tanh (csrc/edge.hip) — a 3-wide GEMM-N would leave 29/32 of every MFMA tile empty
        if bfs and self._out_conv_fused(srcs, cin, i):
            return self.out
        if bfs:
            # bf16 STORAGE: the three sources are bf16 operands already (normalised block output, relu'd warp, the stem's
            # ReLU copy); the 27 tap columns are padded to the 64-column tile of the bf16 kernels
            dev_copy(self.wt_fin[:27], A.p("decoder.net.%d.weight" % (i + 1)).view(27, cin))
            _conv([a.src() for _, _, a in srcs], N, H, W, L.ACT_RELU, 0, 1, 1, 0, H, W, self.wt_fin, self.fin_cols, cin, out=self.y_taps,
                  allow_n32=True)
            L.call("pg_tap_gather_pitch", L.ptr(self.y_taps), self.fin_cols, N, H, W, L.ptr(A.p("decoder.net.%d.bias" % (i + 1))),
                   L.OUT_TANH, L.ptr(self.out), 3 * H * W, H * W, W, 1, L.stream())
            return self.out
        _conv([a.src() for _, _, a in srcs], N, H, W, L.ACT_RELU, 0, 1, 1, 0, H, W,
              A.p("decoder.net.%d.weight" % (i + 1)), 27, cin, out=self.y_taps)
        L.call("pg_tap_gather", L.ptr(self.y_taps), N, H, W, 3, 3, 1, 3, L.ptr(A.p("decoder.net.%d.bias" % (i + 1))),
               L.OUT_TANH, L.ptr(self.out), 3 * H * W, H * W, W, 1, L.stream())
        return self.out

    def _out_conv_fused(self, srcs, cin, i):
        """(round 5) the output convolution's forward as ONE streaming pass (csrc/out_conv_fwd.hip): the last block's raw bf16
        output is normalised + ReLU'd in registers (its activated operand is still written: the backward pass reads it), the other
        sources are the bf16 operands their producers wrote, the 27 tap columns never touch HBM.  False: not eligible, the caller
        runs the materialise / contraction / tap-gather chain."""
        A, N, H, W = self.A, self.N, self.H, self.W
        (k0, _, a0), rest = srcs[0], srcs[1:]
        if (not OUT_FWD_FUSED or k0 != "dec" or a0.mask is not None or not _is_bf16_ptr(L.ptr(a0.t)) or len(rest) > 2
                or tuple(a.C for _, _, a in srcs) not in ((128, 64, 64), (128, 64), (64, 64))
                or _BF_CTX.lookup(L.ptr(a0.t), a0.C, L.ACT_RELU, L.ptr(a0.aff), None) is not None):
            return False
        dev = a0.t.device
        ops = [_BF_CTX.get(L.ptr(a.t) if a.base_ptr is None else a.base_ptr, a.C, L.ACT_RELU, L.ptr(a.aff), L.ptr(a.mask), N, H * W, dev)
               for _, _, a in rest]
        op0 = _BF_CTX.reserve(L.ptr(a0.t), a0.C, L.ACT_RELU, L.ptr(a0.aff), None, N * H * W * a0.C, dev)
        pend = _PENDING_NORM.pop(int(L.ptr(a0.aff) or 0), None) if (_PENDING_NORM and a0.aff is not None) else None
        if pend is not None:
            st, gamma, beta, _n, Lr = pend
            fold = (None, L.ptr(st.sums), L.ptr(gamma), L.ptr(beta), Lr, NORM_EPS, L.ptr(st.mr), L.ptr(st.aff))
        else:
            fold = (L.ptr(a0.aff), None, None, None, 0, NORM_EPS, None, None)
        x1, c1 = (L.ptr(ops[0]), rest[0][2].C) if len(rest) > 0 else (None, 0)
        x2, c2 = (L.ptr(ops[1]), rest[1][2].C) if len(rest) > 1 else (None, 0)
        L.call("pg_out_conv_fwd_fused", L.ptr(a0.t), a0.C, *fold, L.ptr(op0), x1, c1, x2, c2,
               L.ptr(A.p("decoder.net.%d.weight" % (i + 1))), L.ptr(A.p("decoder.net.%d.bias" % (i + 1))), N, H, W, L.OUT_TANH,
               L.ptr(self.out), L.stream())
        return True

    # -------------------------------------------------------------------------------- backward
    def _dsts_for(self, srcs, first_write):
        """Epilogue destinations of a data-gradient wrt the concat `srcs` (consumer activation = ReLU)."""
        dsts = []
        for kind, idx, a in srcs:
            if kind == "dec":
                # the decoder block's output gradient has ONE writer (this launch): it may carry the norm backward's sums
                bs = self.d_norm[idx].fsums if FUSE_NORM_SUMS else None
                dsts.append(L.make_dst(self.d_dz[idx], a.C, fwd=a.t, aff=a.aff, mask=a.mask, act=L.ACT_RELU, bsums=bs))
            elif kind == "warp":
                dsts.append(L.make_dst(self.w_g[idx], a.C, fwd=a.t, act=L.ACT_RELU))
            else:
                e = {"app": "encoder_app", "pose": "encoder_pose", "enc": "encoder"}[kind]
                dsts.append(L.make_dst(self.e_dz[e][idx], a.C, fwd=a.t, aff=a.aff, act=L.ACT_RELU,
                                       accumulate=not first_write))
        return dsts

    def backward(self, dpre, image_grad=None):
        global _BF_CTX, _BF_CTX_X
        self._bf_bwd.begin()
        _BF_CTX, _BF_CTX_X = (self._bf_bwd, self._bf_fwd) if PRECISION == 3 else (None, None)
        try:
            return self._backward(dpre, image_grad)
        finally:
            _BF_CTX = _BF_CTX_X = None
            _join_side()

    def _backward(self, dpre, image_grad=None):
        """dpre: gradient wrt the pre-tanh output, NCHW (N,3,H,W), contiguous.  Accumulates into arena.grads.
        image_grad: optional NCHW (N,3,H,W) buffer receiving d/d(input[:, :3]) — the stacked generator chains stage i's
        image input to stage i-1's output (reference networks.py:320)."""
        A, N, H, W = self.A, self.N, self.H, self.W
        assert dpre.is_contiguous() and tuple(dpre.shape) == (N, 3, H, W)
        ystr = (3 * H * W, H * W, W, 1)
        dev_zero(self.nscr.bwd if FUSE_NORM_SUMS else self.nscr.bsums)
        self._fsum = {}
        # ---- final conv k3s1p1 (+bias, tanh handled by the caller)
        i = self.ndec - 1
        srcs = self._dec_sources(i)
        cin = sum(a.C for _, _, a in srcs)
        wkey = "decoder.net.%d.weight" % (i + 1)
        _debug_delay()
        L.call("pg_bias_grad", L.ptr(dpre), N, H * W, 3, 3 * H * W, 1, H * W, L.ptr(A.g("decoder.net.%d.bias" % (i + 1))),
               L.stream())
        if not self.bfs:
            L.call("pg_im2col_taps", L.ptr(dpre), ystr[0], ystr[1], ystr[2], ystr[3], N, H, W, 3, 3, 1, 3, 32,
                   L.ptr(self.g_taps), L.stream())
        if self.bfs:
            # bf16 STORAGE: the im2col'd gradient stays the fp32 [pixel][32] tensor of the streaming kernels; every destination's
            # forward tensor is the ACTIVATED bf16 operand of the forward pass (its sign gives relu', its value is the
            # weight-gradient operand).  Two lean streaming passes (csrc/out_conv_dgrad.hip): weight gradient on the side
            # stream, data gradient on the main stream.  (Tried: the data gradient as a K = 64 bf16 contraction with the padded
            # weight — 8192 one-K-tile workgroups: 1.05 ms at batch 32 against 0.9 ms streaming.)
            L.call("pg_transpose_f32", L.ptr(A.p(wkey)), 27, cin, L.ptr(self.wt_out), 32, L.stream())
            dsts = []
            for (kind, idx, a), d0 in zip(srcs, self._dsts_for(srcs, True)):
                xop = self._bf_fwd.lookup(L.ptr(a.t), a.C, L.ACT_RELU, L.ptr(a.aff), L.ptr(a.mask))
                assert xop is not None, "forward operand of the output convolution is not in the pass cache"
                gt = {"dec": lambda: self.d_dz[idx], "warp": lambda: self.w_g[idx]}.get(
                    kind, lambda: self.e_dz[{"app": "encoder_app", "pose": "encoder_pose", "enc": "encoder"}[kind]][idx])()
                # the last block's output gradient is written here and nowhere else: the fused pass may add the norm
                # backward's sums in their activated-operand form (sums_mode 2; no dropout mask on this source)
                bs = self.d_norm[idx].fsums if (kind == "dec" and FUSE_NORM_SUMS and a.mask is None) else None
                dsts.append(L.make_dst(gt, a.C, fwd=xop.reshape(-1)[:gt.numel()].view(gt.shape), act=L.ACT_RELU,
                                       accumulate=bool(d0.accumulate), bsums=bs))
            arr = (L.Dst * len(dsts))(*dsts)
            # no im2col'd copy of the gradient: the kernels gather each pixel's 27 values from dpre; the weight-gradient pass
            # goes to the side stream like every other weight gradient
            side = _side_stream() if SIDE_STREAM else None
            if side is not None:
                L.call("pg_stream_wait", _raw(side), L.stream())
            L.call("pg_out_conv_bwd_direct", L.ptr(dpre), 1, L.ptr(self.wt_out), N, H, W, arr, len(dsts), L.ptr(A.g(wkey)),
                   L.ptr(self.fin_ws), self.fin_ws.numel(), ctypes.c_void_p(side.cuda_stream) if side is not None else None, L.stream())
            if L.load().pg_last_launch_info() & L.INFO_BSUMS:
                for (kind, idx, a), dd in zip(srcs, dsts):
                    if dd.bsums:
                        self._fsum[("d", idx)] = 2
            self._ready("decoder.net.%d." % (i + 1))
        else:
            self._backward_final_fp32(srcs, cin, wkey, i)
        self._release("decoder.net.%d." % (i + 1))       # (its data gradient read `wt_out`, a copy of the weight made above)
        self._backward_rest(image_grad)

    def _backward_final_fp32(self, srcs, cin, wkey, i):
        A, N, H, W = self.A, self.N, self.H, self.W
        _wgrad([a.src() for _, _, a in srcs], N, L.ACT_RELU, self.g_taps, 32, cin, True, H, W, H, W, 1, 1, 0, A.g(wkey),
               cout_store=27)
        self._ready("decoder.net.%d." % (i + 1))
        # data-gradient of the 3-channel output conv = one K=32 contraction of the same im2col'd gradient with the
        # weight viewed as [27 (tap, co)][cin] (transposed, zero-padded to 32), scattered with relu' / dropout mask
        L.call("pg_transpose_f32", L.ptr(A.p(wkey)), 27, cin, L.ptr(self.wt_out), 32, L.stream())
        dsts = self._dsts_for(srcs, True)
        if cin <= 256 and all(t.C % 4 == 0 for t in dsts) and OUT_CONV_STREAM:
            arr = (L.Dst * len(dsts))(*dsts)          # K = 27: no GEMM — the streaming kernel (one wave per pixel)
            L.call("pg_out_conv_dgrad", L.ptr(self.g_taps), L.ptr(self.wt_out), N, H, W, arr, len(dsts), L.stream())
            info = L.load().pg_last_launch_info()
        else:
            info = _conv([Act(self.g_taps, 32).src()], N, H, W, L.ACT_NONE, 0, 1, 1, 0, H, W, self.wt_out, cin, 32, dsts=dsts)
        if (info or 0) & L.INFO_BSUMS:          # the last block's output gradient has this one writer: sums_mode 1 (raw forward values)
            for (kind, idx, a), dd in zip(srcs, dsts):
                if kind == "dec" and dd.bsums:
                    self._fsum[("d", idx)] = 1

    def _backward_rest(self, image_grad=None):
        A, N, H, W = self.A, self.N, self.H, self.W

        def warp_bwd(levels):
            for l in levels:
                L.call("pg_warp_mask_max_bwd_bbox", L.ptr(self.w_g[l]), L.ptr(self.w_arg[l]), L.ptr(self.warps),
                       L.ptr(self.lvl_masks[l]), L.ptr(self.mask_bbox) if (self.masked and WARP_BBOX) else None, N, self.T, self.enc[l],
                       self.hw[l][0], self.hw[l][1], H, W, self.align, L.ptr(self.e_dz["encoder_app"][l]), 3 if self.bfs else 0,
                       L.stream())

        # decoder block that writes the deepest warped skip's gradient (level nwarp-1); -1: no fork (few levels / no aux stream)
        i_fork = (self.nlev - self.nwarp) if (self.nwarp > 0 and _aux_on() and 0 < self.nlev - self.nwarp <= self.ndec - 2) else -1
        forked = False
        # ---- up blocks
        for i in range(self.ndec - 2, -1, -1):
            srcs = self._dec_sources(i)
            cin = sum(a.C for _, _, a in srcs)
            hi, wi = self.hw[self.nlev - 1 - i]
            ho, wo = 2 * hi, 2 * wi
            wkey = "decoder.net.%d.net.1.weight" % i
            self.d_norm[i].backward(self.d_dz[i], self.d_raw[i], N, ho * wo * self.dec[i],
                                    A.p("decoder.net.%d.net.3.weight" % i), A.g("decoder.net.%d.net.3.weight" % i),
                                    A.g("decoder.net.%d.net.3.bias" % i), C=self.dec[i], fused=self._fsum.pop(("d", i), 0),
                                    beta=A.p("decoder.net.%d.net.3.bias" % i))
            dy = self.d_dz[i]
            _wgrad([a.src() for _, _, a in srcs], N, L.ACT_RELU, dy, self.dec[i], cin, False, hi, wi, ho, wo, 4, 2, 1,
                   A.g(wkey))
            self._ready("decoder.net.%d." % i)
            info = _conv_dgrad(Act(dy, self.dec[i]).src(), N, ho, wo, 0, 4, 2, 1, hi, wi, A.p(wkey), self.dec[i], cin,
                               self._dsts_for(srcs, True))
            if i > 0 and (info or 0) & L.INFO_BSUMS:
                self._fsum[("d", i - 1)] = 1
            self._release("decoder.net.%d." % i)
            if i == i_fork:
                # every warped skip's gradient w_g[0 .. nwarp-1] is complete (block i wrote the deepest one): the warp
                # backward goes to the auxiliary stream, deepest level first (the first one the encoder chain needs)
                with _on_aux():
                    warp_bwd(range(self.nwarp - 1, -1, -1))
                forked = True
        # ---- deformable skips
        if not forked:
            warp_bwd(range(self.nwarp))
        # ---- encoders (round 5: the pose encoder's levels >= pl on the second encoder stream, next to the appearance encoder's)
        pl = self._enc_par_level()
        par_open = pl < self.nlev
        if par_open:
            with _on_enc2(fork=True):    # every decoder data gradient (the first writers of both encoders' gradients) is enqueued
                pass
        for l in range(self.nlev - 1, 0, -1):
            if forked and l - 1 < self.nwarp:
                _join_aux()              # this level's data gradient accumulates into a gradient the warp backward wrote
                forked = False
            if par_open and l < pl:
                _join_enc2()             # the remaining pose levels run on this stream again
                par_open = False
            for e in self.encs:
                par = par_open and e == self.encs[-1]
                with (_on_enc2() if par else contextlib.nullcontext()):
                    self._backward_encoder_level(e, l)
        if par_open:
            _join_enc2()
        if forked:
            _join_aux()
        self._backward_stems(image_grad)

    def _backward_encoder_level(self, e, l):
        A, N = self.A, self.N
        hi, wi = self.hw[l - 1]
        ho, wo = self.hw[l]
        wkey = "%s.net.%d.net.1.weight" % (e, l)
        dz = self.e_dz[e][l]
        if l < self.nlev - 1:
            self.e_norm[e][l].backward(dz, self.e_raw[e][l], N, ho * wo * self.enc[l],
                                       A.p("%s.net.%d.net.2.weight" % (e, l)), A.g("%s.net.%d.net.2.weight" % (e, l)),
                                       A.g("%s.net.%d.net.2.bias" % (e, l)), C=self.enc[l], fused=self._fsum.pop((e, l), 0))
        xin = self._enc_act(e, l - 1)
        _wgrad([xin.src()], N, L.ACT_LEAKY, dz, self.enc[l], self.enc[l - 1], True, ho, wo, hi, wi, 4, 2, 1,
               A.g(wkey))
        self._ready("%s.net.%d." % (e, l))
        # this launch is the LAST writer of level l-1's gradient (skip / warp contributions were written before): it may
        # carry the sums of that level's norm backward
        nst = self.e_norm[e][l - 1]
        bs = nst.fsums if (nst is not None and FUSE_NORM_SUMS) else None
        info = _conv_dgrad(Act(dz, self.enc[l]).src(), N, ho, wo, 1, 4, 2, 1, hi, wi, A.p(wkey), self.enc[l],
                           self.enc[l - 1],
                           [L.make_dst(self.e_dz[e][l - 1], self.enc[l - 1], fwd=xin.t, aff=xin.aff, act=L.ACT_LEAKY,
                                       accumulate=True, bsums=bs)])
        if bs is not None and (info or 0) & L.INFO_BSUMS:
            self._fsum[(e, l - 1)] = 1
        self._release("%s.net.%d." % (e, l))

    def _backward_stems(self, image_grad=None):
        A, N, H, W = self.A, self.N, self.H, self.W
        for e in self.encs:
            dz = self.e_dz[e][0]
            s0 = self._enc_in_src(e, self.input)
            # (round 4) the first layer's weight-gradient pass delivers the bias gradient as well where it can; where it cannot,
            # the bias-gradient kernel goes FIRST (main stream, before the weight-gradient call: the reducer's ordering invariant)
            fuse = self.enc[0] == 64 and _stem_bias_fusable(3, 1, 1, s0.C)
            if not fuse:
                _debug_delay()
                if self.bfs:
                    L.call("pg_bias_grad_bf16", L.ptr(dz), N * H * W, self.enc[0], L.ptr(A.g(e + ".net.0.bias")), L.stream())
                else:
                    L.call("pg_bias_grad", L.ptr(dz), N * H * W, 1, self.enc[0], self.enc[0], 0, 1,
                           L.ptr(A.g(e + ".net.0.bias")), L.stream())
            done = _wgrad([s0.src()], N, L.ACT_NONE, dz, self.enc[0], s0.C, True, H, W, H, W, 3, 1, 1,
                          A.g(e + ".net.0.weight"), scalar_x=True, dbias=A.g(e + ".net.0.bias") if fuse else None)
            assert bool(done) == bool(fuse), "first-layer bias gradient: the launch code and _stem_bias_fusable disagree"
            self._ready(e + ".net.0.")
            if image_grad is not None and e in ("encoder_app", "encoder"):
                # data-gradient of the k3/s1/p1 first convolution restricted to its 3 image channels, written NCHW
                assert image_grad.is_contiguous() and tuple(image_grad.shape) == (N, 3, H, W)
                if self.enc[0] == 64 and (SMALL_CIN_DGRAD or self.bfs):
                    # (round 6) bf16 STORAGE: dz is a bf16 tensor — the streaming kernel reads it as such (io_flags bit 0), so the
                    # chained stages of the stacked generator run in the storage mode of the single-stage generator
                    L.call("pg_small_cin_dgrad_io", L.ptr(dz), L.ptr(A.p(e + ".net.0.weight")), N, H, W, 3, 1, 1, H, W, s0.C, 0, 3,
                           L.ptr(image_grad), 3 * H * W, H * W, W, 1, 1 if self.bfs else 0, L.stream())
                else:
                    _conv([Act(dz, self.enc[0]).src()], N, H, W, L.ACT_NONE, 1, 3, 1, 1, H, W, A.p(e + ".net.0.weight"),
                          self.enc[0], s0.C, transposed=True, out=image_grad, out_strides=(3 * H * W, H * W, W, 1),
                          n_off=0, n_cnt=3)
            self._release(e + ".net.0.")


# ------------------------------------------------------------------------------------------ discriminator
def discriminator_param_order(spec):
    keys = [k for k, _ in spec]
    idx = sorted({int(k.split(".")[1]) for k in keys}, reverse=True)
    order = []
    for i in idx:
        order += [k for k in keys if k.startswith("net.%d." % i)]
    return order


class DiscriminatorEngine:
    """Discriminator (reference models/networks.py:329-357) forward + backward for a fixed batch M."""

    def __init__(self, arena, M, H, W, pose_dim, device="cuda"):
        self.A, self.M, self.H, self.W, self.P = arena, M, H, W, pose_dim
        self.chans = [64]
        j = 1
        while ("net.%d.net.1.weight" % j) in arena.off:
            self.chans.append(arena.ref_shape["net.%d.net.1.weight" % j][0])
            j += 1
        self.nblk = len(self.chans)            # stem + blocks; last block has Cout=1 and no norm
        f32 = dict(dtype=torch.float32, device=device)
        hs, ws = [(H - 4) // 2 + 1], [(W - 4) // 2 + 1]
        for j in range(1, self.nblk):
            hs.append((hs[-1] + 2 - 4) // 2 + 1)
            ws.append((ws[-1] + 2 - 4) // 2 + 1)
        assert hs[-1] >= 1 and ws[-1] >= 1
        self.hs, self.ws = hs, ws
        # bf16 STORAGE (round 6; the generator's since round 3): the stem's and the first blocks' raw outputs and the gradients that
        # flow through them are bf16 tensors — everything up to the LAST normalised block, whose output (15 x 15 x 512 at 256^2: 6 %
        # of the network's activation elements) and gradient stay fp32 because the three kernels of the one-output-channel last
        # block (forward, weight gradient, data gradient: csrc/edge.hip, pg_conv_wgrad) read fp32 tensors.  No new kernel is needed
        # below that: the stem writes bf16 (pg_stem_conv_bf16_v3), the 256-row / generic bf16 kernels store and scatter bf16, norm
        # forward / backward and the weight gradients take either dtype, the stem's weight / image gradients read bf16
        # (pg_stem_wgrad_bf16_v2, pg_small_cin_dgrad_io).  PG_DISC_F32_STORE=1 = the round-5 mode.
        self.bfs = bool(bf16_store() and STEM_BF16 and STEM_EMIT_BF16 and DISC_BF16_STORE and 3 + 2 * pose_dim + 3 <= 80 and self.nblk >= 4
                        and all(c % 128 == 0 for c in self.chans[1:-1]) and torch.cuda.is_available())
        nbf = self.nblk - 2 if self.bfs else 0          # blocks 0 .. nbf - 1 in bf16
        dt = lambda j: dict(dtype=torch.bfloat16 if j < nbf else torch.float32, device=device)
        self.raw = [_reg_bf16(torch.empty(M, hs[j], ws[j], self.chans[j], **dt(j))) for j in range(self.nblk)]
        self.dz = [_reg_bf16(torch.empty(M, hs[j], ws[j], self.chans[j], **dt(j))) for j in range(self.nblk)]
        self._dz0_views = {}
        self.nscr = NormScratch(self.nblk, M, device)
        self.norm = [NormState(M, device, self.nscr) if 0 < j < self.nblk - 1 else None for j in range(self.nblk)]
        self.wt0 = torch.empty(stem_pack_floats(4, 3 + 2 * pose_dim + 3), **f32)          # [Cin][16][64] repack of the stem
        self.K = hs[-1] * ws[-1]               # outputs per image (49 at 256^2)
        self.inputs = None
        self.grad_ready_cb = None
        self._bf_fwd, self._bf_bwd = BfCache(), BfCache()

    def _ready(self, *prefixes):
        # the reducer orders its collective against BOTH producer streams with events (runtime/dp.py: _wait_producers);
        # the main stream keeps running the data-gradient chain, the side stream the weight gradients
        if self.grad_ready_cb is not None:
            self.grad_ready_cb([k for k in self.A.keys if k.startswith(prefixes)])

    def _stem_srcs(self, pair):
        """[img(3), src_pose(P), image_to_judge(3), tgt_pose(P)] without the cat (pose_gan.py:86,133,135).
        pair = (input NCHW (n,3+2P,H,W), judged NCHW (n,3,H,W))  or  (x,) with x the already-concatenated
        (n,3+2P+3,H,W) tensor (module-level Discriminator.forward API)."""
        P, HW = self.P, self.H * self.W
        if len(pair) == 1:
            x = pair[0]
            sX = (x.shape[1] * HW, HW, self.W, 1)
            return [Act(x, 3 + P, strides=sX), Act(x, 3, strides=sX, base_ptr=x.data_ptr() + 4 * (3 + P) * HW),
                    Act(x, P, strides=sX, base_ptr=x.data_ptr() + 4 * (6 + P) * HW)]
        inp, judged = pair
        C = inp.shape[1]
        sI = (C * HW, HW, self.W, 1)
        sJ = (3 * HW, HW, self.W, 1)
        return [Act(inp, 3 + P, strides=sI), Act(judged, 3, strides=sJ),
                Act(inp, P, strides=sI, base_ptr=inp.data_ptr() + 4 * (3 + P) * HW)]

    def _act(self, j):
        st = self.norm[j]
        return Act(self.raw[j], self.chans[j], aff=st.aff if st is not None else None)

    def forward(self, pairs):
        global _BF_CTX
        self._bf_fwd.begin()
        _BF_CTX = self._bf_fwd if PRECISION == 3 else None
        try:
            return self._forward(pairs)
        finally:
            _BF_CTX = None

    def _forward(self, pairs):
        """pairs: list of (input NCHW (n,3+2P,H,W), judged NCHW (n,3,H,W)); sum n == M.  Returns logits (M,K)."""
        A, H, W = self.A, self.H, self.W
        self.inputs = pairs
        dev_zero(self.nscr.sums)
        off = 0
        bf0 = None
        if PRECISION == 3 and STEM_BF16 and STEM_EMIT_BF16 and _BF_CTX is not None and 3 + 2 * self.P + 3 <= 80:
            bf0 = _BF_CTX.reserve(L.ptr(self.raw[0]), 64, L.ACT_LEAKY, None, None, self.M * self.hs[0] * self.ws[0] * 64,
                                  self.raw[0].device)
        assert self.bfs == (bf16_store() and self.bfs), "the engine was built for another storage mode (PRECISION changed?)"
        for pair in pairs:
            n = pair[0].shape[0]
            assert all(t.is_contiguous() and t.dtype == torch.float32 for t in pair)
            srcs = self._stem_srcs(pair)
            esz = 2 if self.bfs else 4
            out_ptr = self.raw[0].data_ptr() + esz * off * self.hs[0] * self.ws[0] * 64
            bfp = None if bf0 is None else bf0.data_ptr() + 2 * off * self.hs[0] * self.ws[0] * 64
            if self.bfs:
                # bf16 STORAGE: the raw output as bf16 and the next block's LeakyReLU operand in the same pass
                assert bfp is not None
                L.call("pg_stem_pack_bf16", L.ptr(A.p("net.0.weight")), 4, sum(a.C for a in srcs), L.ptr(self.wt0), L.stream())
                arr = (L.Src * len(srcs))(*[a.src() for a in srcs])
                L.call("pg_stem_conv_bf16_v3", arr, len(srcs), n, H, W, 4, 2, 0, L.ptr(self.wt0), L.ptr(A.p("net.0.bias")), None,
                       out_ptr, L.ACT_NONE, bfp, L.ACT_LEAKY, None, L.ACT_NONE, L.stream())
            else:
                _small_cin_conv(srcs, n, H, W, 4, 2, 0, A.p("net.0.weight"), A.p("net.0.bias"), self.wt0, out_ptr,
                                next_act=L.ACT_LEAKY, bf_ptr=bfp)
            off += n
        assert off == self.M
        for j in range(1, self.nblk):
            _conv([self._act(j - 1).src()], self.M, self.hs[j - 1], self.ws[j - 1], L.ACT_LEAKY, 0, 4, 2, 1, self.hs[j],
                  self.ws[j], A.p("net.%d.net.1.weight" % j), self.chans[j], self.chans[j - 1], out=self.raw[j],
                  stats=self.norm[j].stats_target() if j < self.nblk - 1 else None)
            if j < self.nblk - 1:
                self.norm[j].forward(self.raw[j], self.M, self.hs[j] * self.ws[j] * self.chans[j],
                                     A.p("net.%d.net.2.weight" % j), A.p("net.%d.net.2.bias" % j), have_stats=True)
        _flush_all_norms()
        return self.raw[-1].view(self.M, self.K)

    def backward(self, dlogits, need_wgrad=True, image_grad=None):
        global _BF_CTX, _BF_CTX_X
        self._bf_bwd.begin()
        _BF_CTX, _BF_CTX_X = (self._bf_bwd, self._bf_fwd) if PRECISION == 3 else (None, None)
        try:
            return self._backward(dlogits, need_wgrad, image_grad)
        finally:
            _BF_CTX = _BF_CTX_X = None
            _join_side()

    def _backward(self, dlogits, need_wgrad=True, image_grad=None):
        """dlogits (M,K).  need_wgrad: accumulate weight grads (dis_update).  image_grad: list of NCHW (n,3,H,W)
        buffers (one per forward pair, or None) receiving d/d(judged image) (gen_update)."""
        A, M, H, W = self.A, self.M, self.H, self.W
        dev_zero(self.nscr.bwd if FUSE_NORM_SUMS else self.nscr.bsums)
        fsum = {}                   # block -> sums_mode: norm layers whose backward sums the producer of their gradient wrote
        j = self.nblk - 1
        ystr = (self.K, 1, self.ws[j], 1)
        wkey = "net.%d.net.1.weight" % j
        xin = self._act(j - 1)
        if need_wgrad:
            _wgrad([xin.src()], M, L.ACT_LEAKY, dlogits, 1, self.chans[j - 1], True, self.hs[j], self.ws[j],
                   self.hs[j - 1], self.ws[j - 1], 4, 2, 1, A.g(wkey), y_strides=ystr)
            self._ready("net.%d." % j)
        dst = L.make_dst(self.dz[j - 1], self.chans[j - 1], fwd=xin.t, aff=xin.aff, act=L.ACT_LEAKY)      # (1 output channel: no fused sums)
        if self.chans[j - 1] % 4 == 0 and 16 * self.chans[j - 1] * 4 <= 160 * 1024:
            # 1 output channel: K = 16 taps, no GEMM — the streaming small-Cout kernel (6 us; 73 us as a pg_conv launch)
            L.call("pg_small_cout_dgrad", L.ptr(dlogits), ystr[0], ystr[1], ystr[2], ystr[3], M, self.hs[j - 1],
                   self.ws[j - 1], 4, 4, 2, 1, 1, L.ptr(A.p(wkey)), (L.Dst * 1)(dst), 1, L.stream())
        else:
            _conv([Act(dlogits, 1, strides=ystr).src()], M, self.hs[j], self.ws[j], L.ACT_NONE, 1, 4, 2, 1, self.hs[j - 1],
                  self.ws[j - 1], A.p(wkey), 1, self.chans[j - 1], transposed=True, scalar_in=True, dsts=[dst])
        for j in range(self.nblk - 2, 0, -1):
            wkey = "net.%d.net.1.weight" % j
            dz = self.dz[j]
            self.norm[j].backward(dz, self.raw[j], M, self.hs[j] * self.ws[j] * self.chans[j],
                                  A.p("net.%d.net.2.weight" % j),
                                  A.g("net.%d.net.2.weight" % j) if need_wgrad else None,
                                  A.g("net.%d.net.2.bias" % j) if need_wgrad else None, C=self.chans[j], fused=fsum.pop(j, 0))
            xin = self._act(j - 1)
            if need_wgrad:
                _wgrad([xin.src()], M, L.ACT_LEAKY, dz, self.chans[j], self.chans[j - 1], True, self.hs[j], self.ws[j],
                       self.hs[j - 1], self.ws[j - 1], 4, 2, 1, A.g(wkey))
                self._ready("net.%d." % j)
            nst = self.norm[j - 1]          # this launch is the only writer of block j-1's gradient
            bs = nst.fsums if (nst is not None and FUSE_NORM_SUMS) else None
            info = _conv_dgrad(Act(dz, self.chans[j]).src(), M, self.hs[j], self.ws[j], 1, 4, 2, 1, self.hs[j - 1],
                               self.ws[j - 1], A.p(wkey), self.chans[j], self.chans[j - 1],
                               [L.make_dst(self.dz[j - 1], self.chans[j - 1], fwd=xin.t, aff=xin.aff, act=L.ACT_LEAKY, bsums=bs)])
            if bs is not None and (info or 0) & L.INFO_BSUMS:
                fsum[j - 1] = 1
        # stem
        dz0 = self.dz[0]
        cin = 3 + 2 * self.P + 3
        off = 0
        for pi, pair in enumerate(self.inputs):
            n = pair[0].shape[0]
            dptr = dz0.data_ptr() + (2 if self.bfs else 4) * off * self.hs[0] * self.ws[0] * 64
            if self.bfs:
                # this pair's slice of the bf16 gradient as a registered tensor: the weight-gradient dispatch recognises bf16 STORAGE
                # by the tensor behind the pointer
                v = self._dz0_views.get((off, n))
                if v is None:
                    v = self._dz0_views[(off, n)] = _reg_bf16(dz0[off:off + n])
                assert v.data_ptr() == dptr
            if need_wgrad:
                # (round 4) the stem's weight-gradient pass delivers this pair's share of the bias gradient where it can; where it
                # cannot, the bias-gradient kernel goes first (see GeneratorEngine._backward_stems)
                fuse = _stem_bias_fusable(4, 2, 0, cin)
                if not fuse:
                    _debug_delay()
                    if self.bfs:
                        L.call("pg_bias_grad_bf16", dptr, n * self.hs[0] * self.ws[0], 64, L.ptr(A.g("net.0.bias")), L.stream())
                    else:
                        L.call("pg_bias_grad", dptr, n * self.hs[0] * self.ws[0], 1, 64, 64, 0, 1, L.ptr(A.g("net.0.bias")), L.stream())
                done = _wgrad([a.src() for a in self._stem_srcs(pair)], n, L.ACT_NONE, dptr, 64, cin, True, self.hs[0],
                              self.ws[0], H, W, 4, 2, 0, A.g("net.0.weight"), scalar_x=True, dbias=A.g("net.0.bias") if fuse else None)
                assert bool(done) == bool(fuse), "stem bias gradient: the launch code and _stem_bias_fusable disagree"
            if image_grad is not None and image_grad[pi] is not None:
                g = image_grad[pi]
                s = L.Src()
                s.ptr, s.C = dptr, 64
                if SMALL_CIN_DGRAD or self.bfs:       # streaming kernel: 16 lanes per image pixel (GEMM-N = 3 wastes a 32-wide MFMA tile)
                    L.call("pg_small_cin_dgrad_io", dptr, L.ptr(A.p("net.0.weight")), n, self.hs[0], self.ws[0], 4, 2, 0, H, W, cin,
                           3 + self.P, 3, L.ptr(g), 3 * H * W, H * W, W, 1, 1 if self.bfs else 0, L.stream())
                else:
                    _conv([s], n, self.hs[0], self.ws[0], L.ACT_NONE, 1, 4, 2, 0, H, W, A.p("net.0.weight"), 64, cin,
                          transposed=True, out=g, out_strides=(3 * H * W, H * W, W, 1), n_off=3 + self.P, n_cnt=3)
            off += n
        if need_wgrad:
            self._ready("net.0.")
