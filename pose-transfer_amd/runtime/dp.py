"""Data parallelism: one process per GPU, gradients averaged with all-reduce (RCCL over xGMI; `nccl` backend
IS RCCL on ROCm).  New in this build — the reference is single-process.

Contract: every loss term is a per-sample mean and the norm is per-sample, so the AVERAGE over ranks
of gradients computed on equal local shards equals the single-process gradient at the global batch.

Overlap: parameters are laid out in the arena in backward-completion order, so "gradient ready" is a monotonically
advancing offset.  As soon as a bucket worth of gradients is complete its all-reduce is issued with async_op=True:
the collective waits (on the device) for the kernels enqueued so far and then runs concurrently with the rest of
the backward pass.  The 1/world scaling is folded into the fused Adam kernel (grad_scale).
xGMI is point-to-point (7 links x ~153 GB/s): a few large buckets keep every link busy with few launches.
"""
import os

import torch
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* when launched by torch.distributed.run."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if dist.is_initialized() or (ws <= 1 and "RANK" not in os.environ):
        return world_size()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend)
    return dist.get_world_size()


def shard(t, r=None, w=None):
    """Contiguous shard of the leading (batch) dimension for rank r of w (equal shard sizes required)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    n = t.shape[0]
    assert n % w == 0, "global batch %d not divisible by world size %d" % (n, w)
    return t[r * (n // w):(r + 1) * (n // w)]


class GradReducer:
    """Bucketed SUM all-reduce of a flat gradient arena, issued as buckets complete during backward."""

    def __init__(self, arena, world, bucket_bytes=64 << 20, group=None):
        self.arena, self.world, self.group = arena, world, group
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.keys = list(arena.keys)
        self.index = {k: i for i, k in enumerate(self.keys)}
        self.begin()

    def begin(self):
        self.ready = [False] * len(self.keys)
        self.next_key = 0        # first key not yet known complete
        self.launched = 0        # arena offset up to which all-reduces were issued
        self.works = []

    def _end_offset(self, i):
        return self.arena.off[self.keys[i]] if i < len(self.keys) else self.arena.total

    def mark_ready(self, keys):
        for k in keys:
            self.ready[self.index[k]] = True
        while self.next_key < len(self.keys) and self.ready[self.next_key]:
            self.next_key += 1
        upto = self._end_offset(self.next_key)
        if upto - self.launched >= self.bucket_elems:
            self._launch(upto)

    def _launch(self, upto):
        if upto <= self.launched:
            return
        buf = self.arena.grads[self.launched:upto]
        if self.world > 1:
            self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.launched = upto

    def finish(self):
        """Issue whatever is left (keys never reported count as ready: e.g. unused parameters) and wait."""
        self._launch(self.arena.total)
        for w in self.works:
            w.wait()
        self.works = []
