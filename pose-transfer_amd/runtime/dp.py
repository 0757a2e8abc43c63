"""Data parallelism: one process per GPU, gradients averaged with all-reduce over RCCL / xGMI.
New in this build — the reference is single-process.

Contract: every loss term is a per-sample mean and the norm is per-sample, so the AVERAGE over ranks
of gradients computed on equal local shards equals the single-process gradient at the global batch.

Overlap: parameters are laid out in the arena in backward-completion order, so "gradient ready" is a monotonically
advancing offset.  As soon as a bucket worth of gradients is complete its all-reduce is enqueued on a COMMUNICATION
stream that waits — through HIP events recorded at that moment — for exactly the work that produced the bucket: the main
stream (data-gradient chain) and the weight-gradient side stream.  Neither of them waits for the collective; only the
optimiser step at the end of the pass does (`finish()`).  The 1/world scaling is folded into the fused Adam kernel
(grad_scale).  xGMI is point-to-point (7 links x ~153 GB/s): large buckets keep every link busy with few launches; the bucket size shrinks
towards the end of the arena (`_threshold`) so that what is left for `finish()` — the exposed tail — is a few MB.

Transport: on the device the collective is RCCL behind the C ABI (`pg_comm_*`, include/posegan_hip.h; the rendezvous
token travels over the torch.distributed store that torch.distributed.run already set up).  `PG_DP_BACKEND=torch` or a
CPU arena (the gloo tests) use `torch.distributed.all_reduce` instead.
Gradient format: fp32 buckets on the fp32 paths; on the bf16 data path (engine.PRECISION == 3) the default is bf16 buckets:
each finished range is packed to bf16 (pg_pack_bf16), reduced at half the bytes, and Adam reads the bf16 sums (pg_adam_ex).
`PG_DP_GRAD_DTYPE=f32|bf16` overrides either default.  What the bf16 sum costs in accuracy is measured by
tests/test_dp_cpu.py::test_bf16_bucket_sum_of_8_ranks_parameter_error (8 gloo ranks, gradients spread over 4 decades).

Debugging aid: `PG_DP_DEBUG_PEER=1` (single rank, PG_FORCE_REDUCER=1) makes every bucket's "all-reduce" non-trivial by
adding the bucket to itself on the communication stream — the sum a second rank with identical gradients would give — and
reports a divisor of 2, so a bucket that is reduced before its last producer has finished changes the result
(tests/test_gpu_round3.py::test_reducer_stream_order_under_main_stream_delay).
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import lib as L


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


dist_world = world_size


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


_COMM = {}      # device index -> pg_comm handle (one communicator per process)


def rccl_comm(device):
    """The process's RCCL communicator behind the C ABI (created on first use; rendezvous over torch.distributed)."""
    idx = torch.device(device).index or 0
    if idx in _COMM:
        return _COMM[idx]
    lib = L.load()
    w, r = world_size(), rank()
    token = [None]
    if r == 0:
        buf = ctypes.create_string_buffer(128)
        L.check(lib.pg_comm_unique_id(buf), "pg_comm_unique_id")
        token[0] = bytes(buf.raw)
    if w > 1:
        dist.broadcast_object_list(token, src=0)
    handle = ctypes.c_void_p()
    with torch.cuda.device(idx):
        L.check(lib.pg_comm_init(ctypes.c_char_p(token[0]), r, w, ctypes.byref(handle)), "pg_comm_init")
    _COMM[idx] = handle
    return handle


def destroy_comms():
    for h in _COMM.values():
        L.load().pg_comm_destroy(h)
    _COMM.clear()


class GradReducer:
    """Bucketed SUM all-reduce of a flat gradient arena, issued as buckets complete during backward."""

    def __init__(self, arena, world, bucket_bytes=64 << 20, group=None, backend=None, grad_dtype=None, min_bucket_bytes=2 << 20):
        self.arena, self.world, self.group = arena, world, group
        # test aid (tests/test_gpu_round5.py: one bucket per layer): PG_DP_BUCKET_BYTES / PG_DP_MIN_BUCKET_BYTES override the sizes
        bucket_bytes = int(os.environ.get("PG_DP_BUCKET_BYTES", bucket_bytes))
        min_bucket_bytes = int(os.environ.get("PG_DP_MIN_BUCKET_BYTES", min_bucket_bytes))
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.min_bucket_elems = max(1, min(min_bucket_bytes, bucket_bytes) // 4)
        self.keys = list(arena.keys)
        self.index = {k: i for i, k in enumerate(self.keys)}
        self.on_device = arena.grads.is_cuda
        if backend is None:
            backend = os.environ.get("PG_DP_BACKEND", "rccl" if self.on_device else "torch")
        self.backend = backend if self.on_device else "torch"
        if grad_dtype is None:
            from . import engine as E
            grad_dtype = os.environ.get("PG_DP_GRAD_DTYPE", "bf16" if (E.PRECISION == 3 and self.on_device) else "f32")
        self.grad_dtype = grad_dtype
        assert self.grad_dtype in ("f32", "bf16")
        self.bf16 = self.grad_dtype == "bf16"
        self.packed = torch.empty(arena.total, dtype=torch.bfloat16, device=arena.grads.device) if self.bf16 else None
        # debugging hooks (test aids, see the module docstring): read ONCE here and announced when active
        self.debug_peer = os.environ.get("PG_DP_DEBUG_PEER") == "1" and self.on_device and dist_world() == 1
        self.debug_no_wait = os.environ.get("PG_DP_DEBUG_NO_WAIT") == "1"
        self.wait_main = os.environ.get("PG_DP_WAIT_MAIN") == "1"
        if self.debug_peer or self.debug_no_wait:
            import sys
            print("[pose_transfer_amd.dp] DEBUG hooks active: %s — gradients are NOT those of a normal run"
                  % ", ".join(n for n, v in (("PG_DP_DEBUG_PEER", self.debug_peer), ("PG_DP_DEBUG_NO_WAIT", self.debug_no_wait)) if v),
                  file=sys.stderr)
        # what the optimiser divides the sums by: the size of the group this reducer spans (not the global world size)
        gsize = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.divisor = max(1, gsize) * (2 if self.debug_peer else 1)
        self.comm_stream = torch.cuda.Stream(device=arena.grads.device) if self.on_device else None
        self.comm = None
        if self.backend == "rccl" and (world > 1 or os.environ.get("PG_FORCE_REDUCER") == "1"):
            try:
                self.comm = rccl_comm(arena.grads.device)
            except (RuntimeError, OSError) as e:
                # Both transports are RCCL over xGMI; the C-ABI communicator only saves torch's per-collective host work.
                # If it cannot be created (librccl not loadable by dlopen, rendezvous refused), say so and keep going
                # through torch.distributed's communicator rather than losing the run.  All ranks take the same branch:
                # the failure modes are properties of the node image, not of a rank.
                import sys
                print("[pose_transfer_amd.dp] pg_comm_init failed (%s); using torch.distributed all_reduce" % e, file=sys.stderr)
                self.backend = "torch"
        self.launch_count = 0
        # per-bucket timing (bench.py's `dp` block; off in the timed region): `profile = True` brackets every collective with
        # events on the COMMUNICATION stream and measures, in finish(), how long the optimiser's stream really waits for it
        self.profile = False
        self._prof = []          # (bytes, start event | wall start, end event | wall end)
        self._exposed = None
        self.begin()

    def begin(self):
        self.ready = [False] * len(self.keys)
        self.next_key = 0        # first key not yet known complete
        self.launched = 0        # arena offset up to which all-reduces were issued
        self.works = []
        self.launch_count = 0
        self._prof = []
        self._exposed = None

    def _end_offset(self, i):
        return self.arena.off[self.keys[i]] if i < len(self.keys) else self.arena.total

    def mark_ready(self, keys):
        for k in keys:
            self.ready[self.index[k]] = True
        while self.next_key < len(self.keys) and self.ready[self.next_key]:
            self.next_key += 1
        upto = self._end_offset(self.next_key)
        if upto - self.launched >= self._threshold():
            self._launch(upto)

    def _threshold(self):
        """Bucket size that shrinks towards the end of the arena: a quarter of what is still unreduced, between
        min_bucket and bucket.  The deep layers hold most of the parameters and finish EARLY in the backward pass, the
        high-resolution layers that finish last hold almost none — with a fixed 64 MB bucket the last 36 MB of the
        generator's 164 MB waited for finish() and their all-reduce ran after the backward pass, exposed; with the
        shrinking bucket at most min_bucket (2 MB) is left for finish()."""
        remaining = self.arena.total - self.launched
        return min(self.bucket_elems, max(self.min_bucket_elems, remaining // 4))

    def _wait_producers(self, final=False):
        """The communication stream waits — through events recorded now — for the producers of the range it is about to
        reduce; the producers themselves do not wait for anything.

        Weight gradients are written by the side stream.  The few gradients the MAIN stream writes (norm gamma / beta, conv
        biases) are enqueued before the same layer's weight-gradient call, and that call makes the side stream wait for
        the main stream (engine._wgrad) — so "the side stream has reached this point" already implies them, and a bucket
        launched from a `_ready` hook only needs the side-stream event.  (An event on the main stream per bucket cost 4 %
        of the iteration on one GPU: 170 -> 163 img/s)  The last launch (finish) and runs without a
        side stream wait for the main stream as well; PG_DP_WAIT_MAIN=1 restores the conservative form everywhere."""
        from . import engine as E
        if self.debug_no_wait:      # negative control of the ordering stress test: no producer events
            return
        dev = self.arena.grads.device
        cs = self.comm_stream
        side = E._SIDE.get(dev.index if dev.index is not None else torch.cuda.current_device())
        side_on = E.SIDE_STREAM and side is not None
        if final or not side_on or self.wait_main:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            cs.wait_event(ev)
        if side_on:
            ev2 = torch.cuda.Event()
            ev2.record(side)
            cs.wait_event(ev2)

    def _launch(self, upto, final=False):
        if upto <= self.launched:
            return
        lo, n = self.launched, upto - self.launched
        self.launched = upto
        self.launch_count += 1
        if not self.on_device:                      # CPU arenas (gloo tests)
            buf = self.arena.grads[lo:upto]
            if self.bf16:                           # same data flow as on the device: pack, reduce the bf16 bucket
                self.packed[lo:upto] = buf.to(torch.bfloat16)
                buf = self.packed[lo:upto]
            if self.world > 1:
                self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                if self.profile:
                    import time
                    self._prof.append([buf.numel() * buf.element_size(), time.perf_counter(), None])
            return
        self._wait_producers(final)
        buf = self.arena.grads[lo:upto]
        ev0 = ev1 = None
        if self.profile:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self._prof.append([n * (2 if self.bf16 else 4), ev0, ev1])
        with torch.cuda.stream(self.comm_stream):
            if ev0 is not None:
                ev0.record(self.comm_stream)         # after the producer waits: the bracket holds pack + collective only
            if self.debug_peer:                     # the sum a second rank with identical gradients would contribute
                L.call("pg_add2", L.ptr(buf), L.ptr(buf), L.ptr(buf), n, L.stream())
            if self.bf16:
                pk = self.packed[lo:upto]
                L.call("pg_pack_bf16", L.ptr(buf), L.ptr(pk), n, L.stream())     # arena offsets are multiples of 64 elements
                buf = pk
            if self.comm is not None:
                L.check(L.load().pg_comm_allreduce_bucket(self.comm, L.ptr(buf), n, 1 if self.bf16 else 0, L.stream()),
                        "pg_comm_allreduce_bucket")
            elif self.world > 1:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)      # enqueued on the current (= comm) stream
            if ev1 is not None:
                ev1.record(self.comm_stream)

    def finish(self):
        """Issue whatever is left (keys never reported count as ready: e.g. unused parameters) and make the optimiser's
        stream wait for the collectives."""
        self._launch(self.arena.total, final=True)
        if self.profile and not self.on_device:
            import time
            t0 = time.perf_counter()
            for i, w in enumerate(self.works):
                w.wait()
                self._prof[i][2] = time.perf_counter()
            self._exposed = (time.perf_counter() - t0) * 1e3
        for w in self.works:
            w.wait()
        self.works = []
        if self.on_device:
            main = torch.cuda.current_stream(self.arena.grads.device)
            if self.profile:
                # exposed communication = how much later the communication stream finishes than the optimiser's stream gets here
                ea, ec = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record(main)
                ec.record(self.comm_stream)
                self._exposed = (ea, ec)
            main.wait_stream(self.comm_stream)

    def comm_profile(self):
        """What the last profiled pass (profile = True) measured — call after a device synchronisation: per-bucket bytes and
        milliseconds (device: events on the communication stream around pack + collective; CPU / gloo: launch -> completion
        wall time, an upper bound), their sum, and the EXPOSED time: how long the optimiser's stream waited in finish()."""
        buckets = []
        for nbytes, a, b in self._prof:
            if self.on_device:
                ms = a.elapsed_time(b)
            else:
                ms = ((b if b is not None else a) - a) * 1e3
            buckets.append({"bytes": int(nbytes), "ms": round(float(ms), 4)})
        exposed = self._exposed
        if isinstance(exposed, tuple):
            exposed = max(0.0, exposed[0].elapsed_time(exposed[1]))
        return {"buckets": buckets, "allreduce_ms": round(sum(b["ms"] for b in buckets), 4),
                "exposed_ms": None if exposed is None else round(float(exposed), 4), "grad_dtype": self.grad_dtype,
                "bytes": int(sum(b["bytes"] for b in buckets))}

    def grad_source(self):
        """(fp32 grads, bf16 grads or None): what the optimiser step reads after finish()."""
        return self.arena.grads, self.packed
