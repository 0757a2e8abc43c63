"""How much host (Python + ctypes) time does one training iteration cost?  CPU time vs wall time of 20 iterations.
    gpurun -- python tools/host_overhead.py"""
import os, sys, time
from types import SimpleNamespace
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (re-uses the bench's model / batch construction)
from pose_transfer_amd.models.pose_gan import DeformablePose_GAN
from pose_transfer_amd.utils import synth

from pose_transfer_amd.runtime import engine as E, lib as L
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
E.PRECISION = {"f32": 0, "bf16": 1, "bf16x3": 2, "bf16_data": 3}[prec]
args = SimpleNamespace(size=256, batch=4, content_loss_layer="none", nn_loss_area_size=1, l1_penalty_weight=100.0)
opt = bench.make_opt(args)
model = DeformablePose_GAN(opt, device="cuda:0", init_seed=0)
od = dict(vars(opt), lazy_losses=True)
batches = [[torch.from_numpy(a).cuda() for a in synth.batch(1234, "bench/%s" % s, 4, 18, 256, 256)] for s in "ABC"]
for _ in range(3):
    bench.iteration(model, batches, od)
torch.cuda.synchronize()
w0, c0 = time.perf_counter(), time.process_time()
for _ in range(20):
    bench.iteration(model, batches, od)
w1, c1 = time.perf_counter(), time.process_time()      # enqueue finished (the GPU may still be working)
torch.cuda.synchronize()
w2 = time.perf_counter()
counts = {}
def hook(name, a, launch):
    counts[name] = counts.get(name, 0) + 1
    return launch()
L.CALL_HOOK = hook
bench.iteration(model, batches, od)
L.CALL_HOOK = None
torch.cuda.synchronize()
print("%s batch 4: %d C-ABI calls per iteration; top: %s" % (prec, sum(counts.values()),
      ", ".join("%s x%d" % kv for kv in sorted(counts.items(), key=lambda kv: -kv[1])[:12])))
print("per iteration: host CPU time %.2f ms, enqueue wall %.2f ms, end-to-end wall %.2f ms" %
      ((c1 - c0) / 20 * 1e3, (w1 - w0) / 20 * 1e3, (w2 - w0) / 20 * 1e3))
# the same iteration replayed from the library's launch tape (runtime/tape.py): ONE host call per iteration
from pose_transfer_amd.runtime.tape import TapedIteration
tape = TapedIteration(model, batches, od, warmup=2)
for _ in range(3):
    tape.replay()
torch.cuda.synchronize()
w0, c0 = time.perf_counter(), time.process_time()
for _ in range(20):
    tape.replay()
w1, c1 = time.perf_counter(), time.process_time()
torch.cuda.synchronize()
w2 = time.perf_counter()
print("launch tape (%d recorded enqueues): host CPU time %.2f ms, enqueue wall %.2f ms, end-to-end wall %.2f ms per iteration" %
      (tape.n_ops, (c1 - c0) / 20 * 1e3, (w1 - w0) / 20 * 1e3, (w2 - w0) / 20 * 1e3))
tape.close()
