"""bench.py — GAN train images/sec (one iteration = dis_update + gen_update on 3 independent batches;
reference src_deformable/main.py:77-108) on synthetic 256x256 data, BASELINE.json configs[1]:
src_deformable warp_skip=mask, fasion 256x256, 18 key-points, batch 4 per GPU, fp32.

    python bench.py                       # 1 GPU, defaults finish in ~1-2 minutes incl. the CPU baseline
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `value` = global images / second with inputs resident in HBM.
`roofline`: the dominant contraction kernel family timed per launch with HIP events on the launch stream
(a separate, un-timed profiled iteration) against the fp32-MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md).
`cpu_baseline`: the CPU oracle (oracle/ref_cpu.py, torch-CPU restatement of the reference) timed on this
box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import pta_bootstrap  # noqa: E402

pta_bootstrap.load()
from pose_transfer_amd.models.pose_gan import DeformablePose_GAN  # noqa: E402
from pose_transfer_amd.runtime import dp  # noqa: E402
from pose_transfer_amd.runtime import engine as E  # noqa: E402
from pose_transfer_amd.utils import synth  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA peak (same guide; the 5 PFLOP/s headline figure includes 2:1 sparsity)
P = 18                            # key-points; --pose_dim overrides (config 3 uses 32)


def make_opt(args):
    return SimpleNamespace(image_size=(args.size, args.size), use_input_pose=True, pose_dim=P, batch_size=args.batch,
                           num_stacks=4, gen_type="baseline", dataset="fasion", warp_skip="mask", learning_rate=2e-4,
                           content_loss_layer=args.content_loss_layer, nn_loss_area_size=args.nn_loss_area_size,
                           gan_penalty_weight=1.0, l1_penalty_weight=args.l1_penalty_weight)


def iteration(model, batches, od):
    a, b, c = batches
    model.dis_update(a[0], a[1], {"warps": a[2], "masks": a[3]}, b[0], b[1], od)
    model.gen_update(c[0], c[1], {"warps": c[2], "masks": c[3]}, od)


def step_flops(size, pose_dim):
    """Algorithmic conv FLOPs per image-iteration = 4 F_G + 8 F_D."""
    enc, dec = synth.nfilters((size, size))
    H = size
    fg = 0.0
    hw = [(H >> l) ** 2 for l in range(len(enc))]
    for cin0 in (3 + pose_dim, pose_dim):
        fg += 2 * enc[0] * hw[0] * cin0 * 9
        for l in range(1, len(enc)):
            fg += 2 * enc[l] * hw[l] * enc[l - 1] * 16
    for i in range(len(dec) - 1):
        l = len(enc) - 1 - i
        cin = 2 * enc[l] + (dec[i - 1] if i > 0 else 0)
        fg += 2 * cin * hw[l] * dec[i] * 16
    fg += 2 * 3 * hw[0] * (2 * enc[0] + dec[-2]) * 9
    fd, h, cin = 0.0, size, 3 + 2 * pose_dim + 3
    for j, co in enumerate((64, 128, 256, 512, 1)):
        h = (h - 4) // 2 + 1 if j == 0 else (h + 2 - 4) // 2 + 1
        fd += 2 * co * h * h * cin * 16
        cin = co
    return 4 * fg + 8 * fd, fg, fd


def cpu_baseline(args):
    """Oracle (kind 'port') on the host cores: dis_update + gen_update at the bench resolution and per-GPU batch (capped at
    4), 1 warm-up + up to 3 timed iterations; timing stops early once 45 s of timed work are spent so that
    the default run stays within minutes."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_cpu as R
    n = min(args.batch, 4)
    size = args.size
    enc, dec = synth.nfilters((size, size))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    cfg = dict(pose_dim=P, image_size=(size, size), batch_size=n, gan_penalty_weight=1.0,
               l1_penalty_weight=args.l1_penalty_weight, learning_rate=2e-4,
               content_loss_layer=args.content_loss_layer, nn_loss_area_size=args.nn_loss_area_size,
               nfilters_enc=enc, nfilters_dec=dec, aten_warp=True)
    gp = {k: t(v) for k, v in synth.init_params(1, "cpu/gen", synth.generator_spec(P, enc, dec)).items()}
    dpar = {k: t(v) for k, v in synth.init_params(1, "cpu/disc", synth.discriminator_spec(3 + 2 * P + 3)).items()}
    vgg = (t(synth.xavier_uniform(14, "vgg/w", (64, 3, 3, 3))), t(synth.uniform(14, "vgg/b", (64,), -0.1, 0.1)))
    tr = R.Trainer(cfg, gp, dpar, vgg)
    b = [[t(a) for a in synth.batch(1, "cpu/%s" % s, n, P, size, size)] for s in "ABC"]
    d = [t(m) for m in synth.dropout_masks(1, "cpu/d", n)]
    cores = torch.get_num_threads()

    def iteration():
        t0 = time.time()
        tr.dis_update(b[0][0], b[0][1], b[0][2], b[0][3], b[1][0], b[1][1], d)
        tr.gen_update(b[2][0], b[2][1], b[2][2], b[2][3], d)
        return time.time() - t0

    warm = iteration()
    times = []
    while len(times) < 3 and (not times or sum(times) + times[-1] < 45.0):
        times.append(iteration())
    dt = sum(times) / len(times)
    return {"value": n / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "work": "4 F_G + 8 F_D port (the reference as written executes 6 F_G + 9 F_D: its dis_update also back-propagates "
                    "through the generator, pose_gan.py:129,166 — the reference itself would be ~1.4x slower than this baseline)",
            "sample": "%d timed iteration(s) after 1 warm-up (%.1f s) of dis_update+gen_update, %dx%d, batch %d, fp32, "
                      "oracle/ref_cpu.py on torch-CPU (%d threads), %.1f s per iteration"
                      % (len(times), warm, size, size, n, cores, dt)}


PREC_TEXT = {"f32": "fp32", "bf16x3": "fp32 storage, bf16x3 split MFMA operands", "bf16": "fp32 storage, bf16 MFMA operands",
             "bf16_data": "bf16 data path (bf16 operand tensors, fp32 accumulate / master weights)"}
DTYPE_TEXT = {"f32": "f32", "bf16x3": "f32 storage/accumulate, bf16x3 MFMA operands",
              "bf16": "f32 storage/accumulate, bf16 MFMA operands", "bf16_data": "bf16 operands, f32 accumulate"}
PEAK_HBM_TBS = 8.0                # HBM3E peak, same guide (6.3 TB/s is what a float4 copy achieves)


def _ival(a):
    return int(getattr(a, "value", a) or 0)


# Algorithmic HBM bytes of the memory-bound kernels (the minimum a fused kernel has to move), as a function
# of the C-ABI call's arguments (include/posegan_hip.h).  fp32 elements unless noted.
HBM_MODELS = {
    # feat read + out write + level masks (66.4 MB / image at 256^2)
    "pg_warp_mask_max_fwd": lambda a: _ival(a[4]) * _ival(a[7]) * _ival(a[8]) * (8 * _ival(a[6]) + 4 * _ival(a[5])),
    # read grad + arg-max / masks replay + read-modify-write of the input gradient (~130 MB / image = 16.5 B / element)
    "pg_warp_mask_max_bwd": lambda a: int(_ival(a[4]) * _ival(a[7]) * _ival(a[8]) * _ival(a[6]) * 16.5),
    "pg_norm_stats": lambda a: 4 * _ival(a[1]) * _ival(a[2]),
    "pg_norm_bwd_reduce": lambda a: 8 * _ival(a[3]) * _ival(a[4]),                 # dz, y
    "pg_norm_bwd_apply": lambda a: 12 * _ival(a[5]) * _ival(a[6]),                 # dz (read + write), y
    "pg_norm_bwd_apply_ex": lambda a: 14 * _ival(a[5]) * _ival(a[6]),              # + the bf16 copy of dy
    "pg_adam": lambda a: 28 * _ival(a[4]),                                           # p, g, m, v read; p, m, v written
    "pg_adam_ex": lambda a: 28 * _ival(a[5]),
    "pg_nn_loss": lambda a: 12 * _ival(a[2]) * _ival(a[3]) * _ival(a[4]) * _ival(a[5]),       # P, G read; dP written
    "pg_vgg_conv1_relu_fwd": lambda a: _ival(a[3]) * _ival(a[4]) * _ival(a[5]) * (12 + 256),
    "pg_vgg_conv1_dgrad": lambda a: _ival(a[2]) * _ival(a[3]) * _ival(a[4]) * (256 + 24),
    "pg_small_cin_conv": lambda a: _ival(a[2]) * _ival(a[3]) * _ival(a[4]) * 4 * sum(a[0][i].C for i in range(_ival(a[1])))
                         + 256 * _ival(a[2]) * ((_ival(a[3]) + 2 * _ival(a[7]) - _ival(a[5])) // _ival(a[6]) + 1)
                         * ((_ival(a[4]) + 2 * _ival(a[7]) - _ival(a[5])) // _ival(a[6]) + 1),
    "pg_small_cin_wgrad": lambda a: _ival(a[2]) * _ival(a[3]) * _ival(a[4]) * 4 * sum(a[0][i].C for i in range(_ival(a[1])))
                          + 256 * _ival(a[2]) * ((_ival(a[3]) + 2 * _ival(a[7]) - _ival(a[5])) // _ival(a[6]) + 1)
                          * ((_ival(a[4]) + 2 * _ival(a[7]) - _ival(a[5])) // _ival(a[6]) + 1),
    # gradient towards nc image channels: dY (256 B per stem-output pixel) read, nc x 4 B per image pixel written
    "pg_small_cin_dgrad": lambda a: _ival(a[2]) * (256 * _ival(a[3]) * _ival(a[4]) + 4 * _ival(a[12]) * _ival(a[8]) * _ival(a[9])),
    "pg_stem_conv_bf16": lambda a: HBM_MODELS["pg_small_cin_conv"](a),
    "pg_stem_conv_bf16_ex": lambda a: HBM_MODELS["pg_small_cin_conv"](a) * (1.5 if _ival(a[11]) else 1.0),    # + bf16 copy
    "pg_stem_wgrad_bf16": lambda a: HBM_MODELS["pg_small_cin_wgrad"](a),
    # last conv (256 -> 3): tap tensor (27 -> 32 columns) + NCHW image; data-gradient: im2col'd gradient + fwd read / grad write
    "pg_tap_gather": lambda a: _ival(a[1]) * _ival(a[2]) * _ival(a[3]) * (27 * 4 + 12),
    "pg_out_conv_dgrad": lambda a: _ival(a[2]) * _ival(a[3]) * _ival(a[4]) * (128 + 8 * sum(a[5][i].C for i in range(_ival(a[6])))),
    "pg_materialise_bf16": lambda a: 6 * _ival(a[4]) * _ival(a[5]) * _ival(a[6]),                 # fp32 in, bf16 out
    "pg_channel_major_bf16": lambda a: 6 * _ival(a[4]) * _ival(a[5]) * _ival(a[6]) * _ival(a[7]) // max(1, _ival(a[8]) ** 2),
    # ---- round 3: the `_io` / bf16-STORAGE forms (io_flags: element sizes follow the flags)
    "pg_warp_mask_max_fwd_io": lambda a: _ival(a[4]) * _ival(a[7]) * _ival(a[8]) * (
        _ival(a[6]) * ((2 if _ival(a[14]) & 1 else 4) + (2 if _ival(a[14]) & 2 else 4)) + 4 * _ival(a[5])),
    "pg_warp_mask_max_bwd_io": lambda a: int(_ival(a[4]) * _ival(a[7]) * _ival(a[8]) * _ival(a[6]) * (8.25 if _ival(a[13]) == 3 else 16.5)),
    # (round 3) the backward with the limb masks' bounding boxes: same bytes, `bbox` sits in front of N
    "pg_warp_mask_max_bwd_bbox": lambda a: int(_ival(a[5]) * _ival(a[8]) * _ival(a[9]) * _ival(a[7]) * (8.25 if _ival(a[14]) == 3 else 16.5)),
    "pg_norm_bwd_reduce_ex": lambda a: (4 if _ival(a[6]) == 3 else 8) * _ival(a[3]) * _ival(a[4]),
    "pg_norm_bwd_apply_io": lambda a: (6 if _ival(a[10]) == 3 else (14 if a[9] else 12)) * _ival(a[5]) * _ival(a[6]),
    "pg_materialise_bf16_ex": lambda a: ((2 if _ival(a[1]) else 4) + (4 if a[9] else 2)) * _ival(a[5]) * _ival(a[6]) * _ival(a[7]),
    "pg_stem_conv_bf16_v3": lambda a: _ival(a[2]) * _ival(a[3]) * _ival(a[4]) * 4 * sum(a[0][i].C for i in range(_ival(a[1])))
                            + (256 * (1 if a[10] else 0) + 128 * sum(1 for j in (11, 13, 15) if a[j])) * _ival(a[2])
                            * ((_ival(a[3]) + 2 * _ival(a[7]) - _ival(a[5])) // _ival(a[6]) + 1)
                            * ((_ival(a[4]) + 2 * _ival(a[7]) - _ival(a[5])) // _ival(a[6]) + 1),
    "pg_stem_wgrad_bf16_ex": lambda a: _ival(a[2]) * _ival(a[3]) * _ival(a[4]) * 4 * sum(a[0][i].C for i in range(_ival(a[1])))
                             + (128 if _ival(a[9]) else 256) * _ival(a[2])
                             * ((_ival(a[3]) + 2 * _ival(a[7]) - _ival(a[5])) // _ival(a[6]) + 1)
                             * ((_ival(a[4]) + 2 * _ival(a[7]) - _ival(a[5])) // _ival(a[6]) + 1),
    "pg_bias_grad_bf16": lambda a: 2 * _ival(a[1]) * _ival(a[2]),
    "pg_tap_gather_pitch": lambda a: _ival(a[2]) * _ival(a[3]) * _ival(a[4]) * (27 * 4 + 12),
    # output-conv backward, bf16 storage, ONE pass: dpre (12 B / pixel) + activated operand read once + gradient write (2 + 2 B / channel)
    "pg_out_conv_bwd_direct": lambda a: _ival(a[3]) * _ival(a[4]) * _ival(a[5]) * (12 + 4 * sum(a[6][i].C for i in range(_ival(a[7])))),
    "pg_l1_loss": lambda a: 12 * _ival(a[2]),
    "pg_tanh_bwd": lambda a: 12 * _ival(a[2]),
}


class HbmProfiler:
    """HIP events (on the launch stream) around every call of a memory-bound entry point during ONE un-timed iteration."""

    def __init__(self):
        import ctypes
        from pose_transfer_amd.runtime import lib as L
        self.L, self.ct, self.rec = L, ctypes, []

    def hook(self, name, args, launch):
        model = HBM_MODELS.get(name)
        if model is None:
            return launch()
        L, ct = self.L, self.ct
        lib = L.load()
        e0, e1 = ct.c_void_p(), ct.c_void_p()
        lib.pg_event_create(ct.byref(e0)); lib.pg_event_create(ct.byref(e1))
        lib.pg_event_record(e0, L.stream())
        launch()
        lib.pg_event_record(e1, L.stream())
        self.rec.append((name, model(args), e0, e1))

    def summary(self):
        lib, ct = self.L.load(), self.ct
        out = {}
        for name, nbytes, e0, e1 in self.rec:
            ms = ct.c_float()
            lib.pg_event_elapsed_ms(e0, e1, ct.byref(ms))
            lib.pg_event_destroy(e0); lib.pg_event_destroy(e1)
            d = out.setdefault(name, {"calls": 0, "ms": 0.0, "bytes": 0})
            d["calls"] += 1; d["ms"] += ms.value; d["bytes"] += nbytes
        res = []
        for name, d in sorted(out.items(), key=lambda kv: -kv[1]["ms"]):
            tbs = d["bytes"] / max(d["ms"], 1e-9) * 1e-9
            res.append({"kernel": name, "calls": d["calls"], "ms": round(d["ms"], 4), "algorithmic_MB": round(d["bytes"] / 1e6, 2),
                        "TB_per_s": round(tbs, 3), "frac_of_hbm_peak": round(tbs / PEAK_HBM_TBS, 4)})
        return res


def pmc_traffic(kernel):
    """(HBM bytes per launch of `kernel`, source file) from the COMMITTED rocprofv3 PMC passes (profiles/round*_pmc.json,
    made by tools/pmc_bench.sh on the default workload — counters cannot be collected from inside this process):
    2*FETCH_SIZE + WRITE_SIZE, in bytes (MI355X_MICROARCH.md: FETCH_SIZE reads half of a wide coalesced stream on
    gfx950).  (None, None) when no PMC summary is available."""
    for name in ("round3_pmc.json", "round2_pmc.json", "round1_pmc.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            k = d["kernels"].get(kernel)
            if k is not None:
                return int((2 * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024), "profiles/" + name
        except Exception:
            continue
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # >= 10 warm-up + >= 50 timed iterations
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU (BASELINE.json configs[1]: 4)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--pose_dim", type=int, default=18)
    ap.add_argument("--content_loss_layer", default="none")
    ap.add_argument("--nn_loss_area_size", type=int, default=1)
    ap.add_argument("--l1_penalty_weight", type=float, default=100.0)
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3", "bf16", "bf16_data"],
                    help="MFMA operand format of the fwd/dgrad contractions (default f32 = the reference's arithmetic; "
                         "the other modes are extra, non-headline measurements)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the iteration as ONE captured HIP graph (runtime/graph.py; single GPU only) — an extra, "
                         "non-default measurement of the host-bound small-batch configurations")
    ap.add_argument("--tape", action="store_true",
                    help="replay the iteration from the library's launch tape (runtime/tape.py: ONE host call per iteration, the "
                         "launches stay on their streams; single GPU only) — removes the Python enqueue cost of the small-batch "
                         "configurations; an extra measurement, the default line is the eager loop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--launch-table", default=None, help="write one line per contraction launch of the profiled iteration here")
    args = ap.parse_args()
    global P
    P = args.pose_dim
    E.PRECISION = {"f32": 0, "bf16": 1, "bf16x3": 2, "bf16_data": 3}[args.precision]

    world = dp.init_from_env()
    rank = dp.rank()
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = "cuda:%d" % local
    torch.cuda.set_device(device)

    opt = make_opt(args)
    model = DeformablePose_GAN(opt, device=device, init_seed=0)
    od = dict(vars(opt), lazy_losses=True)
    dev = lambda arrs: [torch.from_numpy(a).to(device) for a in arrs]
    batches = [dev(synth.batch(1234 + rank, "bench/%s" % s, args.batch, P, args.size, args.size)) for s in "ABC"]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    graphed = None
    if args.graph:
        assert world == 1, "--graph is single-GPU only"
        from pose_transfer_amd.runtime.graph import GraphedIteration
        graphed = GraphedIteration(model, batches, od, warmup=max(2, args.warmup))
    if args.tape:
        assert world == 1 and not args.graph, "--tape is single-GPU only (and exclusive with --graph)"
        from pose_transfer_amd.runtime.tape import TapedIteration
        graphed = TapedIteration(model, batches, od, warmup=max(2, args.warmup))
    step = graphed.replay if graphed is not None else (lambda: iteration(model, batches, od))
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if graphed is not None:
        graphed.close()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    global_batch = args.batch * world
    ips = global_batch * args.steps / elapsed

    # ---- roofline leg: one extra profiled iteration, HIP events around every contraction launch
    roof = None
    if not args.no_kernel_profile:
        side, E.SIDE_STREAM = E.SIDE_STREAM, False      # per-kernel durations: no concurrent weight-gradient stream
        iteration(model, batches, od)                   # un-profiled single-stream pass: scratch buffers that only this
        torch.cuda.synchronize()                        # mode allocates exist before the events are placed (a first-use
        E.PROFILER = E.KernelProfiler()                 # hipMalloc showed up as a 30 ms "launch" otherwise)
        iteration(model, batches, od)                   # two profiled repeats; every launch is credited with its faster one
        iteration(model, batches, od)
        torch.cuda.synchronize()
        E.SIDE_STREAM = side
        launches = [] if args.launch_table else None
        fam = E.PROFILER.summary(launches, repeats=2)
        E.PROFILER = None
        if launches is not None and rank == 0:
            with open(args.launch_table, "w") as f:
                f.write("# idx family kind GFLOP ksplit us TFLOP/s\n")
                for i, (nm, kind, fl, ks, ms) in enumerate(launches):
                    f.write("%3d %-34s %-5s %8.2f %3d %8.1f %7.1f\n" % (i, nm, kind, fl * 1e-9, ks, ms * 1e3, fl / max(ms, 1e-9) * 1e-9))
        if fam:
            name, d = max(fam.items(), key=lambda kv: kv[1]["ms"])
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            # bf16 operand modes run their forward / data-gradient (and, on the data path, weight-gradient) contractions
            # on the bf16 matrix pipe: they are priced against its dense peak
            peak = PEAK_BF16_MFMA_TFLOPS if args.precision in ("bf16", "bf16_data") else PEAK_F32_MFMA_TFLOPS
            traffic, traffic_src = pmc_traffic(name) if args.precision == "f32" else (None, None)
            roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic, "traffic_source": traffic_src,     # committed PMC summary, NOT measured in this run
                    "gpu_event_ms_total": round(sum(v["ms"] for v in fam.values()), 3),   # HIP-event time of all contraction launches of ONE iteration
                    "launches": d["launches"], "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                    "flops_per_launch_avg": d["flops"] / d["launches"],
                    "families": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                     "tflops": round(v["flops"] / max(v["ms"], 1e-9) * 1e-9, 2)}
                                 for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}}
    # ---- HBM-bound kernels: one more un-timed, single-stream iteration with events around each memory-bound launch
    hbm = None
    if not args.no_kernel_profile:
        from pose_transfer_amd.runtime import lib as _L
        prof = HbmProfiler()
        side, E.SIDE_STREAM = E.SIDE_STREAM, False
        _L.CALL_HOOK = prof.hook
        try:
            iteration(model, batches, od)
            torch.cuda.synchronize()
        finally:
            _L.CALL_HOOK = None
            E.SIDE_STREAM = side
        hbm = prof.summary()
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    if rank == 0:
        sf, fg, fd = step_flops(args.size, P)
        out = {
            "metric": "GAN train images/sec (gen+disc step) at %dx%d" % (args.size, args.size),
            "value": round(ips, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE_TEXT[args.precision], "data": "synthetic",
            "config": {"workload": "src_deformable warp_skip=mask gen_type=baseline, %dx%d, %d kpts, batch %d/GPU, %s%s"
                                   % (args.size, args.size, P, args.batch, PREC_TEXT[args.precision],
                                      " (BASELINE.json configs[1])"
                                      if (args.size, P, args.batch, args.precision) == (256, 18, 4, "f32") else ""),
                       "global_batch": global_batch, "parallelism": "dp%d" % world,
                       "content_loss_layer": args.content_loss_layer, "nn_loss_area_size": args.nn_loss_area_size,
                       "precision": args.precision, **({"hip_graph": True} if args.graph else {}),
                       **({"launch_tape": True} if args.tape else {})},
            "step_tflops": round(sf * ips / 1e12, 2),
            "step_frac_of_f32_mfma_peak": round(sf * ips / 1e12 / (PEAK_F32_MFMA_TFLOPS * world), 4),
            "roofline": roof, "hbm_kernels": hbm, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
