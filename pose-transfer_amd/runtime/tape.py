"""Launch-tape replay of the training iteration (one process, one GPU) — round 3.

The Python host enqueues ~250 C-ABI calls per iteration (25 us each on the bf16 data path at batch 4: 6 ms of a 7.6 ms
iteration, tools/host_overhead.py).  A HIP graph of the same work replays SLOWER than the eager loop on ROCm 7.2 and loses the
two-stream overlap of the weight gradients (runtime/graph.py).  The tape is the library's own replay: while one iteration
runs the usual way, every enqueue the library makes (kernel launches, memsets, stream waits, device copies) is recorded
with a copy of its arguments (csrc/common.h: PG_KLAUNCH, csrc/api.hip: pg_tape_*); `replay()` re-issues them from ONE C call,
on the streams they were recorded on.  The contract is a graph's: every buffer is persistent, the dropout key and Adam's
step number come from a device counter (pg_dropout_mask_ctr / pg_adam_ctr / pg_counter_add), inputs live in static tensors.

    t = TapedIteration(model, batches, opt_dict)        # warm-up + recording; `batches` are the STATIC input tensors
    for real in loader:  copy real data into `batches`;  out_gen, dis_losses, gen_losses = t.replay()
    t.close()

Single process only (the data-parallel reducer orders its collectives with torch events).  The reference has no counterpart
(eager PyTorch loop, main.py:77-108)."""
import ctypes

import torch

from . import dp as DP
from . import engine as E
from . import lib as L


class TapedIteration:
    def __init__(self, model, batches, opt_dict, warmup=3, drop_masks=None):
        assert DP.world_size() == 1, "launch-tape replay is single-process only"
        assert E.REPLAY_CTR is None, "one replay session at a time"
        self.model, self.batches = model, batches
        # the tape holds ADDRESSES: every input must be the static tensor the library reads directly — a dtype / layout that
        # makes the iteration go through a torch temporary (warps.float(), a non-contiguous or 9-column warp tensor) would be
        # recorded as a copy from a freed address
        for bt in batches:
            for x in bt:
                assert x.is_cuda and x.is_contiguous(), "taped iteration: inputs must be contiguous device tensors"
            assert bt[0].dtype == torch.float32 and bt[1].dtype == torch.float32 and bt[2].dtype == torch.float32, \
                "taped iteration: images and warps must be fp32"
            assert bt[2].shape[-1] == 8 and bt[3].dtype in (torch.float32, torch.float64), \
                "taped iteration: warps need exactly 8 columns, masks fp32 / fp64"
        for dm in (drop_masks or ()):
            for x in (dm or ()):
                assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(), "taped iteration: dropout masks fp32 on the device"
        self.drop = drop_masks or (None, None)          # explicit (dis_update, gen_update) dropout masks: parity tests
        self.od = dict(opt_dict, lazy_losses=True)
        self.dev = batches[0][0].device
        self.ctr = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.arenas = [model.gen.arena, model.disc.arena]
        for a in self.arenas:
            a.replay_base = None
        E.REPLAY_CTR = self.ctr
        self.replays = 0
        self.tape = ctypes.c_void_p()
        try:
            for _ in range(warmup):                      # eager iterations in replay mode: every cache reaches its steady state
                self._iteration()
                self.replays += 1
            torch.cuda.synchronize(self.dev)
            lib = L.load()
            L.check(lib.pg_tape_begin(), "pg_tape_begin")
            try:
                self._iteration()                        # runs AND is recorded
            finally:
                n = ctypes.c_int64()
                L.check(lib.pg_tape_end(ctypes.byref(self.tape), ctypes.byref(n)), "pg_tape_end")
            self.n_ops = int(n.value)
            self.replays += 1
            self._sync_host_state()
        except Exception:
            self.close()
            raise

    def _iteration(self):
        a, b, c = self.batches
        m = self.model
        oa, oc = {"warps": a[2], "masks": a[3]}, {"warps": c[2], "masks": c[3]}
        if self.drop[0] is not None:
            oa["drop_masks"], oc["drop_masks"] = self.drop
        m.dis_update(a[0], a[1], oa, b[0], b[1], self.od)
        m.gen_update(c[0], c[1], oc, self.od)
        L.call("pg_counter_add", L.ptr(self.ctr), 1, L.stream())

    def _sync_host_state(self):
        for a in self.arenas:
            a.step = a.replay_base + self.replays - 1
            a._bump_version()
            a.bf16_version = a.version()                 # the replayed Adam launch keeps the bf16 weight copy current

    def outputs(self):
        """(out_gen, dis losses [total, true, fake], gen losses [total, ll, ad]) — live device tensors of the engines"""
        m = self.model
        eng = m.gen.engine(self.batches[2][0].shape[0])
        return eng.out, m._loss[4:7], m._loss[0:3]

    def replay(self):
        """One training iteration on the data currently in `batches` (ONE host call); returns `outputs()`."""
        L.check(L.load().pg_tape_replay(self.tape), "pg_tape_replay")
        self.replays += 1
        self._sync_host_state()
        return self.outputs()

    def close(self):
        if self.tape:
            L.load().pg_tape_destroy(self.tape)
            self.tape = ctypes.c_void_p()
        E.REPLAY_CTR = None
        for a in self.arenas:
            a.replay_base = None
