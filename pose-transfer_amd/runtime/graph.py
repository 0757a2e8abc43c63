"""HIP-graph replay of the training iteration (one process, one GPU).

Every buffer of the step is allocated once per (batch, H, W) and every kernel of the library is enqueued on torch's current
stream, so `dis_update` + `gen_update` can be captured into one HIP graph and replayed with a single host call per
iteration (the Python host otherwise issues ~600 launches: 7 ms of an 8.4 ms iteration on the bf16 data path at batch 4,
tools/host_overhead.py).  Two scalars change per iteration — the dropout key and Adam's step number; in a replay session
they come from a device counter (include/posegan_hip.h: pg_dropout_mask_ctr, pg_adam_ctr) that the graph increments itself.

    g = GraphedIteration(model, batches, opt_dict)      # warm-up + capture; `batches` are the STATIC input tensors
    for real in loader:  copy real data into `batches`;  g.replay()
    g.close()                                            # back to host-side scalars (optimiser step counts stay right)

Not available with data parallelism (the gradient reducer's communication stream is not captured): world size 1 only.
The reference has no counterpart (its loop is eager PyTorch, main.py:77-108)."""
import torch

from . import dp as DP
from . import engine as E
from . import lib as L


class GraphedIteration:
    def __init__(self, model, batches, opt_dict, warmup=3, drop_masks=None):
        assert DP.world_size() == 1, "HIP-graph replay is single-process only"
        assert E.REPLAY_CTR is None, "one replay session at a time"
        self.model, self.batches = model, batches
        self.drop = drop_masks or (None, None)          # explicit (dis_update, gen_update) dropout masks: parity tests
        self.od = dict(opt_dict, lazy_losses=True)
        self.dev = batches[0][0].device
        self.ctr = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.arenas = [model.gen.arena, model.disc.arena]
        for a in self.arenas:
            a.replay_base = None
        E.REPLAY_CTR = self.ctr
        self.replays = 0
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):                      # eager iterations in replay mode: caches reach their steady state
                self._iteration()
                self.replays += 1
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._iteration()
        self._sync_host_state()

    def _iteration(self):
        a, b, c = self.batches
        m = self.model
        oa, oc = {"warps": a[2], "masks": a[3]}, {"warps": c[2], "masks": c[3]}
        if self.drop[0] is not None:
            oa["drop_masks"], oc["drop_masks"] = self.drop
        self.dis_losses = m.dis_update(a[0], a[1], oa, b[0], b[1], self.od)
        self.out_gen, _, self.gen_losses = m.gen_update(c[0], c[1], oc, self.od)
        L.call("pg_counter_add", L.ptr(self.ctr), 1, L.stream())

    def _sync_host_state(self):
        for a in self.arenas:
            a.step = a.replay_base + self.replays - 1
        self.model.iteration = getattr(self.model, "iteration", 0)

    def replay(self):
        """One training iteration on the data currently in `batches`; returns (out_gen, dis losses, gen losses) as device
        tensors owned by the graph (valid until the next replay)."""
        self.graph.replay()
        self.replays += 1
        for a in self.arenas:
            a.step = a.replay_base + self.replays - 1
            a._bump_version()
        return self.out_gen, self.dis_losses, self.gen_losses

    def close(self):
        E.REPLAY_CTR = None
        for a in self.arenas:
            a.replay_base = None
