"""bench.py — GAN train images/sec (one iteration = dis_update + gen_update on 3 independent batches;
reference src_deformable/main.py:77-108) on synthetic 256x256 data, BASELINE.json configs[1]:
src_deformable warp_skip=mask, fasion 256x256, 18 key-points, batch 4 per GPU, fp32.

    python bench.py                       # 1 GPU, defaults finish in ~1-2 minutes incl. the CPU baseline
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

    python bench.py --gpus N              # without a launcher: re-executes itself under torch.distributed.run with N ranks

Prints ONE JSON line (rank 0).  `value` = global images / second with inputs resident in HBM.
`parity` (N=1): the SAME first iteration (same weights, batches, dropout masks) on the device and in the CPU oracle:
max-abs(out_gen) and the relative error of the loss triples (<= 1e-3 / <= 1e-4 in fp32).
`north_star` / `bf16_data_b32_img_s` (N=1): BASELINE.json's sub-metric — generator forward+backward at 256x256, batch 32
on the bf16 data path against the dense bf16 MFMA peak — and the full training iteration in that configuration.
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
    return SimpleNamespace(image_size=(args.size, args.size), use_input_pose=True, pose_dim=_kp(args), batch_size=args.batch,
                           num_stacks=4, gen_type="baseline", dataset="fasion", warp_skip="mask", learning_rate=2e-4,
                           content_loss_layer=args.content_loss_layer, nn_loss_area_size=args.nn_loss_area_size,
                           gan_penalty_weight=1.0, l1_penalty_weight=args.l1_penalty_weight)


def _kp(args):
    return int(getattr(args, "pose_dim", P))


def iteration(model, batches, od):
    """one training iteration as main.py runs it (reference main.py:77-108): the batch of the generator update is drawn first
    and its generator forward enqueued ahead (DeformablePose_GAN.prefetch_gen_forward: dis_update does not touch the generator's
    weights), then dis_update, then gen_update"""
    a, b, c = batches
    oc = {"warps": c[2], "masks": c[3]}
    if hasattr(model, "prefetch_gen_forward"):
        model.prefetch_gen_forward(c[0], oc)
    model.dis_update(a[0], a[1], {"warps": a[2], "masks": a[3]}, b[0], b[1], od)
    model.gen_update(c[0], c[1], oc, od)


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


def parity_device_iteration(args, device, rank):
    """The FIRST training iteration of the bench configuration on the device with everything pinned: the timed model's
    initial weights (init_seed=0), the bench batches, explicit dropout masks, eager losses.  Returns what the oracle needs to
    repeat exactly that iteration (cpu_baseline) and the device's results."""
    n = min(args.batch, 4)
    o = make_opt(args)
    o.batch_size = n
    kp = _kp(args)
    model = DeformablePose_GAN(o, device=device, init_seed=0)
    gsd = {k: v.detach().cpu().clone() for k, v in model.gen.state_dict().items()}
    dsd = {k: v.detach().cpu().clone() for k, v in model.disc.state_dict().items()}
    host = [synth.batch(1234 + rank, "bench/%s" % s_, n, kp, args.size, args.size) for s_ in "ABC"]
    drops = [synth.dropout_masks(1234 + rank, "bench/d%s" % s_, n) for s_ in "AC"]
    dev = lambda arrs: [torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in arrs]
    a, b, c = [dev(h) for h in host]
    dA, dC = [dev(d) for d in drops]
    od = dict(vars(o), lazy_losses=False)
    dl = model.dis_update(a[0], a[1], {"warps": a[2], "masks": a[3], "drop_masks": dA}, b[0], b[1], od)
    og, _, gl = model.gen_update(c[0], c[1], {"warps": c[2], "masks": c[3], "drop_masks": dC}, od)
    res = {"dis": [float(v) for v in dl], "gen": [float(v) for v in gl], "out_gen": og.detach().cpu()}
    del model
    torch.cuda.empty_cache()
    return {"n": n, "gen_sd": gsd, "disc_sd": dsd, "host": host, "drops": drops, "device": res}


def cpu_baseline(args, pin=None, iters=None):
    """Oracle (kind 'port') on the host cores: dis_update + gen_update at the bench resolution and per-GPU batch (capped at
    4), 1 warm-up + 3 timed iterations.  With `pin` (parity_device_iteration) the oracle starts from the SAME
    weights, batches and dropout masks as the device did, and its warm-up iteration doubles as the parity check."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_cpu as R
    n = min(args.batch, 4)
    size = args.size
    P = _kp(args)
    iters = args.cpu_iters if iters is None else iters
    enc, dec = synth.nfilters((size, size))
    t = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
    cfg = dict(pose_dim=P, image_size=(size, size), batch_size=n, gan_penalty_weight=1.0,
               l1_penalty_weight=args.l1_penalty_weight, learning_rate=2e-4,
               content_loss_layer=args.content_loss_layer, nn_loss_area_size=args.nn_loss_area_size,
               nfilters_enc=enc, nfilters_dec=dec, aten_warp=True)
    if pin is not None:
        gp, dpar = {k: t(v) for k, v in pin["gen_sd"].items()}, {k: t(v) for k, v in pin["disc_sd"].items()}
        b = [[t(a) for a in h] for h in pin["host"]]
        dA, dC = [[t(m) for m in d] for d in pin["drops"]]
    else:
        gp = {k: t(v) for k, v in synth.init_params(1, "cpu/gen", synth.generator_spec(P, enc, dec)).items()}
        dpar = {k: t(v) for k, v in synth.init_params(1, "cpu/disc", synth.discriminator_spec(3 + 2 * P + 3)).items()}
        b = [[t(a) for a in synth.batch(1, "cpu/%s" % s, n, P, size, size)] for s in "ABC"]
        dA = dC = [t(m) for m in synth.dropout_masks(1, "cpu/d", n)]
    vgg = (t(synth.xavier_uniform(14, "vgg/w", (64, 3, 3, 3))), t(synth.uniform(14, "vgg/b", (64,), -0.1, 0.1)))
    tr = R.Trainer(cfg, gp, dpar, vgg)
    cores = torch.get_num_threads()
    last = {}

    def iteration():
        t0 = time.time()
        last["dis"] = tr.dis_update(b[0][0], b[0][1], b[0][2], b[0][3], b[1][0], b[1][1], dA)
        last["out_gen"], last["gen"] = tr.gen_update(b[2][0], b[2][1], b[2][2], b[2][3], dC)
        return time.time() - t0

    warm = iteration()
    parity = None
    if pin is not None:
        dv = pin["device"]
        rel = lambda x, y: float(max(abs(p - q) / max(abs(q), 1e-12) for p, q in zip(x, y)))
        parity = {"out_gen_max_abs": float((dv["out_gen"] - last["out_gen"]).abs().max()),
                  "dis_losses_rel": rel(dv["dis"], last["dis"]), "gen_losses_rel": rel(dv["gen"], last["gen"]),
                  "device_losses": {"dis": dv["dis"], "gen": dv["gen"]},
                  "oracle_losses": {"dis": [float(v) for v in last["dis"]], "gen": [float(v) for v in last["gen"]]},
                  "what": "first dis_update+gen_update from identical weights / batches / dropout masks, %dx%d, batch %d, %s "
                          "on the device vs oracle/ref_cpu.py (fp32, torch-CPU); bars (fp32): out_gen <= 1e-3 max-abs, losses <= "
                          "1e-4 relative (SURVEY.md 8d)" % (size, size, n, args.precision)}
    if iters <= 0:             # parity only (the extra configuration legs): the warm-up iteration WAS the check
        return {"warmup_iteration_s": round(warm, 2), "cores": cores}, parity
    times = [iteration() for _ in range(iters)]
    dt = sum(times) / len(times)
    out = {"value": n / dt, "unit": "images/s", "cores": cores, "kind": "port",
           "work": "4 F_G + 8 F_D port (the reference as written executes 6 F_G + 9 F_D: its dis_update also back-propagates "
                   "through the generator, pose_gan.py:129,166 — the reference itself would be ~1.4x slower than this baseline)",
           "iteration_s": [round(x, 2) for x in times],
           "sample": "%d timed iteration(s) after 1 warm-up (%.1f s) of dis_update+gen_update, %dx%d, batch %d, fp32, "
                     "oracle/ref_cpu.py on torch-CPU (%d threads), %.1f s per iteration"
                     % (len(times), warm, size, size, n, cores, dt)}
    return out, parity


def north_star_legs(device, passes, steps):
    """BASELINE.json north_star sub-metric under the driver's clock: generator forward + backward at 256x256, batch 32, 18
    key-points on the bf16 data path (3 F_G x 32 = 13.25 TFLOP per pass) against the dense bf16 MFMA peak, timed with HIP
    events on the launch stream; then the full training iteration (dis_update + gen_update, 3 independent batches) in the same
    configuration.  Single GPU, rank 0 only; the default (fp32) model has been freed before."""
    import ctypes
    from pose_transfer_amd.runtime import lib as _L
    N, size, kp = 32, 256, 18
    prev = E.PRECISION
    E.PRECISION = 3
    try:
        o = SimpleNamespace(image_size=(size, size), use_input_pose=True, pose_dim=kp, batch_size=N, num_stacks=4,
                            gen_type="baseline", dataset="fasion", warp_skip="mask", learning_rate=2e-4,
                            content_loss_layer="none", nn_loss_area_size=1, gan_penalty_weight=1.0, l1_penalty_weight=100.0)
        model = DeformablePose_GAN(o, device=device, init_seed=0)
        dev = lambda arrs: [torch.from_numpy(a).to(device) for a in arrs]
        batches = [dev(synth.batch(1234, "ns/%s" % s_, N, kp, size, size)) for s_ in "ABC"]
        inp, _, wr, mk = batches[0]
        gout = torch.from_numpy(synth.normal(1234, "ns/gout", (N, 3, size, size))).to(device)
        eng = model.gen.engine(N)
        eng.set_dropout(None, train=True, seed=0)

        def one_pass():
            model.gen.zero_grad()
            eng.forward(inp, wr, mk)
            eng.backward(gout)

        for _ in range(3):
            one_pass()
        torch.cuda.synchronize()
        lib = _L.load()
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        lib.pg_event_create(ctypes.byref(e0)); lib.pg_event_create(ctypes.byref(e1))
        lib.pg_event_record(e0, _L.stream())
        t0 = time.perf_counter()
        for _ in range(passes):
            one_pass()
        lib.pg_event_record(e1, _L.stream())
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / passes * 1e3
        ms = ctypes.c_float()
        lib.pg_event_elapsed_ms(e0, e1, ctypes.byref(ms))
        lib.pg_event_destroy(e0); lib.pg_event_destroy(e1)
        ms = ms.value / passes
        _, fg, _ = step_flops(size, kp)
        flops = 3.0 * fg * N
        tf = flops / ms / 1e9
        ns = {"ms": round(ms, 3), "wall_ms": round(wall, 3), "tflops": round(tf, 1),
              "frac_of_bf16_peak": round(tf / PEAK_BF16_MFMA_TFLOPS, 4), "peak": PEAK_BF16_MFMA_TFLOPS, "passes": passes,
              "flop_per_pass": flops,
              "workload": "Deformable_Generator forward + backward, 256x256, 18 kpts, batch 32, bf16 data path "
                          "(bf16 operands and storage, fp32 accumulate / statistics / master weights), warp_skip=mask, "
                          "random dropout; target >= 0.5 (<= 10.6 ms)"}
        # per-family table of the SAME pass, single stream (no concurrent weight-gradient / warp / encoder streams: a launch's
        # HIP-event duration is that kernel alone), each launch credited with the faster of two repeats — so that the dominant
        # kernel's fraction of the bf16 peak can be recomputed from this line alone
        side, E.SIDE_STREAM = E.SIDE_STREAM, False
        try:
            one_pass()
            torch.cuda.synchronize()
            E.PROFILER = E.KernelProfiler()
            one_pass()
            one_pass()
            torch.cuda.synchronize()
            fam = E.PROFILER.summary(None, repeats=2)
            E.PROFILER = None
            hp = HbmProfiler()
            _L.CALL_HOOK = hp.hook
            for _ in range(2):
                one_pass()
            torch.cuda.synchronize()
            _L.CALL_HOOK = None
            ns["families"] = {k: {"launches": v["launches"], "ms": round(v["ms"], 3), "gflop": round(v["flops"] * 1e-9, 1),
                                  "tflops": round(v["flops"] / max(v["ms"], 1e-9) * 1e-9, 1),
                                  "frac_of_bf16_peak": round(v["flops"] / max(v["ms"], 1e-9) * 1e-9 / PEAK_BF16_MFMA_TFLOPS, 4)}
                              for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}
            ns["contraction_ms_single_stream"] = round(sum(v["ms"] for v in fam.values()), 3)
            ns["hbm_kernels"] = hp.summary(repeats=2)
            ns["hbm_side_ms_single_stream"] = round(sum(e["ms"] for e in ns["hbm_kernels"]), 3)
        finally:
            E.PROFILER = None
            _L.CALL_HOOK = None
            E.SIDE_STREAM = side
        od = dict(vars(o), lazy_losses=True)
        for _ in range(3):
            iteration(model, batches, od)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            iteration(model, batches, od)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        b32 = {"value": round(N / dt, 2), "unit": "images/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
               "workload": "dis_update + gen_update, 256x256, 18 kpts, batch 32, bf16 data path, L1 loss, 1 GPU"}
        del model, eng, batches
        torch.cuda.empty_cache()
        return ns, b32
    finally:
        E.PRECISION = prev


PREC_CODE = {"f32": 0, "bf16": 1, "bf16x3": 2, "bf16_data": 3}


def config_leg(device, cfg, steps, parity_n=0):
    """One more BASELINE.json configuration under the driver's clock (single GPU, rank 0): throughput of the full iteration at
    the configuration's own per-GPU batch and — with parity_n > 0 — `parity` of the FIRST iteration at batch parity_n (same
    weights / batches / dropout masks on the device and in the CPU oracle; the norm and every loss are per-sample, so a smaller
    batch checks the same arithmetic).  cfg: size, batch, pose_dim, precision, content_loss_layer, nn_loss_area_size,
    l1_penalty_weight."""
    import copy
    prev = E.PRECISION
    E.PRECISION = PREC_CODE[cfg.precision]
    try:
        pin = pargs = None
        if parity_n > 0:
            pargs = copy.copy(cfg)
            pargs.batch = parity_n
            pin = parity_device_iteration(pargs, device, 0)
        o = make_opt(cfg)
        kp = _kp(cfg)
        model = DeformablePose_GAN(o, device=device, init_seed=0)
        dev = lambda arrs: [torch.from_numpy(a).to(device) for a in arrs]
        batches = [dev(synth.batch(1234, "leg/%s" % s_, cfg.batch, kp, cfg.size, cfg.size)) for s_ in "ABC"]
        od = dict(vars(o), lazy_losses=True)
        for _ in range(5):
            iteration(model, batches, od)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            iteration(model, batches, od)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        sf, _, _ = step_flops(cfg.size, kp)
        peak = PEAK_BF16_MFMA_TFLOPS if cfg.precision in ("bf16", "bf16_data") else PEAK_F32_MFMA_TFLOPS
        leg = {"value": round(cfg.batch / dt, 2), "unit": "images/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
               "step_tflops": round(sf * cfg.batch / dt / 1e12, 2), "step_frac_of_peak": round(sf * cfg.batch / dt / 1e12 / peak, 4),
               "peak": peak,
               "workload": "dis_update + gen_update, %dx%d, %d kpts, batch %d, %s, content_loss_layer=%s nn_loss_area_size=%d "
                           "l1_penalty_weight=%g, 1 GPU" % (cfg.size, cfg.size, kp, cfg.batch, PREC_TEXT[cfg.precision],
                                                            cfg.content_loss_layer, cfg.nn_loss_area_size, cfg.l1_penalty_weight)}
        del model, batches
        torch.cuda.empty_cache()
    finally:
        E.PRECISION = prev
    if pin is not None:
        _, leg["parity"] = cpu_baseline(pargs, pin, iters=0)
        if cfg.precision != "f32":
            leg["parity"]["what"] += ("; NOT an fp32 path: the stated bf16 envelope is tests/test_gpu_round5.py BF16_TOL "
                                      "(2 x the worst value observed over 3 seeds at 256x256, profiles/round5_bf16_tolerance.txt)")
    return leg



def main_py_leg(precision, steps=60):
    """The PRODUCT's own training loop under the driver's clock: pose-transfer_amd/main.py in this process,
    synthetic data from a ring of 3 pre-generated batches (what the loop above cycles), lazy losses read back once per display_ratio,
    configs[1]'s shape.  main() reports its steady-state rate over the iterations after --timing_skip."""
    from pose_transfer_amd import main as M
    import contextlib
    import io
    prev = E.PRECISION
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            model = M.main(["--dataset", "fasion", "--pose_dim", "18", "--batch_size", "4", "--precision", precision, "--steps", str(steps),
                            "--synthetic", "1", "--synthetic_ring", "3", "--display_ratio", "50", "--timing_skip", "10",
                            "--exp_root", "/tmp/pg_bench_main", "--expID", "bench_" + precision])
        st = dict(model.last_run_stats)
        del model
        torch.cuda.empty_cache()
        return {"value": round(st["img_s"], 2), "unit": "images/s", "timed_iterations": st["timed_iterations"],
                "lazy_losses": st["lazy_losses"],
                "workload": "pose-transfer_amd/main.py --synthetic 1 --synthetic_ring 3 --steps %d --precision %s, 256x256, 18 kpts, batch 4: "
                            "dis_update + gen_update per iteration, losses read back every 50 iterations, iterations 11.. timed "
                            "(wall clock between two device synchronisations)" % (steps, precision)}
    finally:
        E.PRECISION = prev


def forced_reducer_leg(steps=10):
    """First-contact telemetry of the data-parallel path on ONE GPU: this script again in a child process
    with PG_FORCE_REDUCER=1 — a world-size-1 process group, the bucketed RCCL all-reduce of both networks' gradient arenas on the
    communication stream, exactly the ordering an N > 1 run uses — so that `dp` (per-bucket ms, exposed wait, rccl_ranks) is non-null
    in a single-GPU record and its img/s sits next to `value` (the cost of the reducer's ordering)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PG_FORCE_REDUCER="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", "5", "--no-cpu-baseline",
           "--no-north-star", "--no-config-legs", "--no-kernel-profile", "--no-extra-legs"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        return {"value": d["value"], "unit": "images/s", "steps": steps, "rccl_ranks": d.get("rccl_ranks"),
                "dp_transport": d.get("dp_transport"), "dp": d.get("dp"),
                "workload": "the default configuration (256x256, batch 4, fp32) in a child process with PG_FORCE_REDUCER=1: world-size-1 "
                            "process group, bucketed all-reduce of both gradient arenas on the communication stream"}
    except Exception as e:      # telemetry only: never fail the bench line over it
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}


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
    # (round 5) output convolution forward in one pass: every source read once (2 B / channel), the block's activated operand and out_gen written
    "pg_out_conv_fwd_fused": lambda a: _ival(a[17]) * _ival(a[18]) * _ival(a[19])
                             * (2 * (_ival(a[1]) + _ival(a[12]) + _ival(a[14])) + 2 * _ival(a[1]) + 12),
    "pg_tap_gather_pitch": lambda a: _ival(a[2]) * _ival(a[3]) * _ival(a[4]) * (27 * 4 + 12),
    # output-conv backward, bf16 storage, ONE pass: dpre (12 B / pixel) + activated operand read once + gradient write (2 + 2 B / channel)
    "pg_out_conv_bwd_direct": lambda a: _ival(a[3]) * _ival(a[4]) * _ival(a[5]) * (12 + 4 * sum(a[6][i].C for i in range(_ival(a[7])))),
    # (round 4 / 5) norm backward apply with the sums taken from the producer; materialise with the finalize folded in
    "pg_norm_bwd_apply_v2": lambda a: (6 if _ival(a[11]) == 3 else (14 if a[10] else 12)) * _ival(a[6]) * _ival(a[7]),
    "pg_norm_bwd_apply_v3": lambda a: (6 if _ival(a[11]) == 3 else (14 if a[10] else 12)) * _ival(a[6]) * _ival(a[7]),
    "pg_materialise_bf16_norm": lambda a: ((2 if _ival(a[1]) else 4) + (4 if a[15] else 2)) * _ival(a[11]) * _ival(a[12]) * _ival(a[13]),
    "pg_l1_loss": lambda a: 12 * _ival(a[2]),
    "pg_tanh_bwd": lambda a: 12 * _ival(a[2]),
}


def _small_cin_flops(a):
    ho = (_ival(a[3]) + 2 * _ival(a[7]) - _ival(a[5])) // _ival(a[6]) + 1
    wo = (_ival(a[4]) + 2 * _ival(a[7]) - _ival(a[5])) // _ival(a[6]) + 1
    return 2.0 * _ival(a[2]) * ho * wo * 64 * sum(a[0][i].C for i in range(_ival(a[1]))) * _ival(a[5]) ** 2


# entry points of this table that are NOT memory-bound in fp32: the first layers contract K = taps x Cin (189 ... 672) on the
# fp32 matrix pipe (52 - 170 FLOP per byte); their line also carries TFLOP/s against the fp32 MFMA peak
MFMA_FLOPS = {"pg_small_cin_conv": _small_cin_flops, "pg_small_cin_wgrad": _small_cin_flops}


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
        fl = MFMA_FLOPS.get(name)
        self.rec.append((name, model(args), e0, e1, fl(args) if fl else 0.0))

    def summary(self, repeats=1):
        """Per entry point: calls, summed time, algorithmic bytes.  repeats > 1: the records hold that many identical
        iterations back to back; every call is credited with the MINIMUM over its repeats (round 3's single pass once
        reported a 58 us kernel at 0.96 ms: the bracket had swallowed a first-use allocation)."""
        lib, ct = self.L.load(), self.ct
        times = []
        for name, nbytes, e0, e1, _fl in self.rec:
            ms = ct.c_float()
            lib.pg_event_elapsed_ms(e0, e1, ct.byref(ms))
            lib.pg_event_destroy(e0); lib.pg_event_destroy(e1)
            times.append(ms.value)
        n = len(self.rec) // max(1, repeats)
        aligned = repeats > 1 and n * repeats == len(self.rec) and all(
            self.rec[i][:2] == self.rec[i + r * n][:2] for r in range(1, repeats) for i in range(n))
        if not aligned:
            n, repeats = len(self.rec), 1
        out = {}
        for i in range(n):
            name, nbytes = self.rec[i][:2]
            ms = min(times[i + r * n] for r in range(repeats))
            d = out.setdefault(name, {"calls": 0, "ms": 0.0, "bytes": 0, "flops": 0.0})
            d["calls"] += 1; d["ms"] += ms; d["bytes"] += nbytes; d["flops"] += self.rec[i][4]
        res = []
        for name, d in sorted(out.items(), key=lambda kv: -kv[1]["ms"]):
            tbs = d["bytes"] / max(d["ms"], 1e-9) * 1e-9
            e = {"kernel": name, "calls": d["calls"], "ms": round(d["ms"], 4), "algorithmic_MB": round(d["bytes"] / 1e6, 2),
                 "TB_per_s": round(tbs, 3), "frac_of_hbm_peak": round(tbs / PEAK_HBM_TBS, 4)}
            if d["flops"] > 0:
                tf = d["flops"] / max(d["ms"], 1e-9) * 1e-9
                e.update(bound="mfma", tflops=round(tf, 2), frac_of_f32_mfma_peak=round(tf / PEAK_F32_MFMA_TFLOPS, 4))
            res.append(e)
        return res


def pmc_traffic(kernel):
    """(HBM bytes per launch of `kernel`, source file) from the COMMITTED rocprofv3 PMC passes (profiles/round*_pmc.json,
    made by tools/pmc_bench.sh on the default workload — counters cannot be collected from inside this process):
    2*FETCH_SIZE + WRITE_SIZE, in bytes (MI355X_MICROARCH.md: FETCH_SIZE reads half of a wide coalesced stream on
    gfx950).  (None, None) when no PMC summary is available."""
    for name in ("round6_pmc.json", "round5_pmc.json", "round4_pmc.json", "round3_pmc.json", "round2_pmc.json", "round1_pmc.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            k = d["kernels"].get(kernel)
            if k is not None:
                return int((2 * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024), "profiles/" + name
        except Exception:
            continue
    return None, None


# family name of the launch table -> kernel symbol in profiles/round*_pmc_northstar.json (bf16 data path; that pass profiled the
# generator's forward + backward at 256 x 256, batch 32 — the launches of these families in a batch-32 training step are the same)
BF16_PMC_NAMES = {
    "conv_igemm<256x256p,A0,B0>": "void pg::conv_bf16_pair_kernel<256, false>(pg::ConvK)",
    "conv_igemm<256x128p,A0,B0>": "void pg::conv_bf16_pair_kernel<128, false>(pg::ConvK)",
    "conv_igemm<512x64p,A0,B0>": "void pg::conv_bf16_pair_kernel<64, false>(pg::ConvK)",
    "conv_igemm<256x256m,A0,B0>": "void pg::conv_bf16_pair_kernel<256, true>(pg::ConvK)",
    "conv_igemm<256x128m,A0,B0>": "void pg::conv_bf16_pair_kernel<128, true>(pg::ConvK)",
    "conv_igemm<256x256,A0,B0>": "void pg::conv_bf16_big_kernel<256, 64>(pg::ConvK)",
    "conv_igemm<quad128,A0,B0>": "void pg::conv_bf16_quad_kernel<false>(pg::ConvK)",
    "conv_igemm<quad128m,A0,B0>": "void pg::conv_bf16_quad_kernel<true>(pg::ConvK)",
}
BF16_PMC_NAMES_R4 = {k: v.replace(", false>", ">") for k, v in BF16_PMC_NAMES.items()}      # (round 4: one template parameter)


def pmc_traffic_bf16(kernel, args):
    if not (args.batch == 32 and args.size == 256):
        return None, None
    for name, names in (("round6_pmc_northstar.json", BF16_PMC_NAMES), ("round5_pmc_northstar.json", BF16_PMC_NAMES),
                        ("round4_pmc_northstar.json", BF16_PMC_NAMES_R4)):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))["kernels"]
            k = d.get(names.get(kernel, ""))
            if k is not None:
                return int((2 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024), \
                    "profiles/%s (generator forward + backward, batch 32)" % name
        except Exception:
            continue
    return None, None


def fail(msg, code=2):
    sys.stderr.write(msg + "\n")
    sys.stderr.flush()
    sys.exit(code)


def maybe_spawn(args):
    """`python bench.py --gpus N` with N > 1 and no launcher in the environment: start the N ranks ourselves (one process per
    GPU, torch.distributed.run on 127.0.0.1) and exit with the launcher's status.  Fails loudly when fewer than N devices are
    visible — a silent 1-GPU number labelled N would be worse than no number."""
    if args.gpus <= 1 or "RANK" in os.environ:
        return
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not args.dry_run and ndev < args.gpus:
        fail("bench.py: --gpus %d but only %d GPU(s) are visible on this node; not starting (the path has no CPU fallback)"
             % (args.gpus, ndev))
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.exit(subprocess.call(cmd, env=env))


def dry_run(world, rank, local):
    """Launcher / rendezvous check without a model: every rank contributes rank+1 to an all-reduce (gloo without GPUs, nccl =
    RCCL with them, plus the C-ABI communicator the trainer uses) and rank 0 prints what came back."""
    import torch.distributed as dist
    on_gpu = torch.cuda.is_available() and torch.cuda.device_count() > local
    if on_gpu:
        torch.cuda.set_device(local)
    t = torch.tensor([float(rank + 1)], device="cuda:%d" % local if on_gpu else "cpu")
    if world > 1:
        dist.all_reduce(t)
    rccl = None
    if on_gpu and world > 1:
        import ctypes
        from pose_transfer_amd.runtime import lib as _Lc
        comm = dp.rccl_comm("cuda:%d" % local)
        r_, w_ = ctypes.c_int32(-1), ctypes.c_int32(-1)
        _Lc.check(_Lc.load().pg_comm_ranks(comm, ctypes.byref(r_), ctypes.byref(w_)), "pg_comm_ranks")
        rccl = int(w_.value)
        dp.destroy_comms()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": world, "sum_of_ranks_plus_1": float(t.item()),
                          "expected": world * (world + 1) / 2.0, "backend": dist.get_backend() if world > 1 else None,
                          "rccl_ranks": rccl}), flush=True)
    ok = abs(float(t.item()) - world * (world + 1) / 2.0) < 1e-6
    if dist.is_initialized():
        dist.destroy_process_group()
    if not ok:
        fail("bench.py --dry-run: the all-reduce returned %r" % float(t.item()), 3)


def main(argv=None, model_factory=None):
    """model_factory (tests only — tests/test_dp_cpu.py): callable(opt, device, rank, world) -> an object with dis_update /
    gen_update / g_reducer / d_reducer standing in for DeformablePose_GAN on a box without a GPU, so that THIS function's
    N > 1 path (rendezvous, barriers, max-over-ranks timing, the `dp` block, the JSON line) runs end to end on gloo.  The
    product never passes it: without it a missing GPU is a hard error."""
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
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU-oracle leg (and with it `parity`)")
    ap.add_argument("--cpu-iters", type=int, default=3, help="timed oracle iterations after its warm-up (SURVEY.md 8d: 3)")
    ap.add_argument("--no-north-star", action="store_true",
                    help="skip the bf16 batch-32 legs (`north_star`, `bf16_data_b32_img_s`; N=1 only)")
    ap.add_argument("--north-star-passes", type=int, default=20)
    ap.add_argument("--no-config-legs", action="store_true",
                    help="skip the extra configuration legs (`bf16_data_b4_img_s`, `cfg2_224_p32_b8_bf16`, `cfg3_nnloss_vgg_b4`; N=1 only)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip `main_py_img_s` (the product's main.py loop, fp32 and bf16 data path) and `dp_forced_1gpu` (a child "
                         "process with PG_FORCE_REDUCER=1); N=1 only")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check: the ranks rendezvous, all-reduce their rank numbers and rank 0 prints one JSON line; no "
                         "model (works without a GPU: gloo)")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--launch-table", default=None, help="write one line per contraction launch of the profiled iteration here")
    args = ap.parse_args(argv)
    global P
    P = args.pose_dim
    E.PRECISION = PREC_CODE[args.precision]
    stand_in = model_factory is not None

    if not stand_in:
        maybe_spawn(args)             # --gpus N without a launcher: re-executes under torch.distributed.run, never returns
    world = dp.init_from_env()
    rank = dp.rank()
    if world != args.gpus:
        fail("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); start it as `python bench.py --gpus %d` or "
             "`python -m torch.distributed.run --nnodes=1 --nproc-per-node %d ... bench.py --gpus %d`"
             % (args.gpus, world, args.gpus, args.gpus, args.gpus))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        return dry_run(world, rank, local)
    if stand_in:
        device = "cpu"
        args.no_kernel_profile = args.no_north_star = args.no_cpu_baseline = args.no_config_legs = args.no_extra_legs = True
    else:
        if not torch.cuda.is_available() or torch.cuda.device_count() <= local:
            fail("bench.py: rank %d needs GPU %d but %d GPU(s) are visible (no CPU fallback exists)"
                 % (rank, local, torch.cuda.device_count() if torch.cuda.is_available() else 0))
        device = "cuda:%d" % local
        torch.cuda.set_device(device)

    def sync():
        if not stand_in:
            torch.cuda.synchronize()

    opt = make_opt(args)
    # N=1: the pinned first iteration for the `parity` field (its own model instance; the oracle repeats it in cpu_baseline)
    pin = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        pin = parity_device_iteration(args, device, rank)
    model = model_factory(opt, device, rank, world) if stand_in else DeformablePose_GAN(opt, device=device, init_seed=0)
    od = dict(vars(opt), lazy_losses=True)
    dev = lambda arrs: [torch.from_numpy(a).to(device) for a in arrs]
    batches = [dev(synth.batch(1234 + rank, "bench/%s" % s, args.batch, P, args.size, args.size)) for s in "ABC"]

    def barrier():
        sync()
        if world > 1:
            torch.distributed.barrier()
        sync()

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
    per_rank = [args.batch * args.steps / elapsed]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        allt = [torch.zeros_like(tt) for _ in range(world)]
        torch.distributed.all_gather(allt, tt)
        per_rank = [args.batch * args.steps / float(x.item()) for x in allt]
        elapsed = max(float(x.item()) for x in allt)
    global_batch = args.batch * world
    ips = global_batch * args.steps / elapsed
    # what the data-parallel transport itself reports (RCCL communicator behind the C ABI: ncclCommCount)
    rccl_ranks, transport = 1, "single process (no collective)"
    red = getattr(model, "g_reducer", None)
    dp_block = None
    if red is not None:
        if red.comm is not None:
            import ctypes
            from pose_transfer_amd.runtime import lib as _Lc
            r_, w_ = ctypes.c_int32(-1), ctypes.c_int32(-1)
            _Lc.check(_Lc.load().pg_comm_ranks(red.comm, ctypes.byref(r_), ctypes.byref(w_)), "pg_comm_ranks")
            rccl_ranks, transport = int(w_.value), "RCCL ncclAllReduce behind the C ABI (pg_comm_*), %s buckets" % red.grad_dtype
        else:
            rccl_ranks = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
            transport = "torch.distributed all_reduce (%s), %s buckets" % (
                torch.distributed.get_backend() if torch.distributed.is_initialized() else "none", red.grad_dtype)
        # ---- `dp`: one more un-timed iteration with every collective bracketed by events on the communication stream —
        # per-bucket bytes / time and the EXPOSED communication time (how long the optimiser's stream waited in finish())
        reds = [("gen", model.g_reducer), ("disc", getattr(model, "d_reducer", None))]
        for _, r in reds:
            if r is not None:
                r.profile = True
        barrier()
        iteration(model, batches, od)
        barrier()
        prof = {}
        for nm, r in reds:
            if r is not None:
                prof[nm] = r.comm_profile()
                r.profile = False
        ex = [v["exposed_ms"] for v in prof.values() if v["exposed_ms"] is not None]
        dp_block = {"ranks": world, "rccl_ranks": rccl_ranks, "transport": transport,
                    "allreduce_ms_per_step": round(sum(v["allreduce_ms"] for v in prof.values()), 4),
                    "exposed_comm_ms_per_step": round(sum(ex), 4) if ex else None,
                    "bytes_per_step": int(sum(v["bytes"] for v in prof.values())),
                    "what": "rank 0, one extra iteration after the timed region: per bucket = pack + all-reduce on the communication "
                            "stream (HIP events; gloo: launch -> completion wall time); exposed = how much later than the "
                            "optimiser's stream the communication stream finished (finish())", **prof}

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
            traffic, traffic_src = pmc_traffic(name) if args.precision == "f32" else pmc_traffic_bf16(name, args)
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
            for _ in range(3):
                iteration(model, batches, od)
            torch.cuda.synchronize()
        finally:
            _L.CALL_HOOK = None
            E.SIDE_STREAM = side
        hbm = prof.summary(repeats=3)
    ns = b32 = None
    legs = {}
    single = rank == 0 and world == 1 and not stand_in
    if single and not (args.no_north_star and args.no_config_legs and args.no_extra_legs):
        del model, batches, step, graphed
        torch.cuda.empty_cache()
    if single and not args.no_north_star:
        ns, b32 = north_star_legs(device, args.north_star_passes, args.north_star_passes)
    if single and not args.no_config_legs:
        # the remaining BASELINE.json configurations under the driver's clock:
        #   configs[1]'s shape on the bf16 data path = what each of 8 GPUs runs in configs[3] (batch 32 / 8 GPUs);
        #   configs[2] as written (224 x 224, 32 key-points, batch 8, bf16);
        #   configs[3]'s per-GPU shape (nn-loss 5 x 5 + VGG block1_conv2, l1_penalty_weight 0.01, batch 4) in fp32 with parity
        ns_ = SimpleNamespace
        par = 0 if args.no_cpu_baseline else 2
        legs["bf16_data_b4_img_s"] = config_leg(device, ns_(size=256, batch=4, pose_dim=18, precision="bf16_data",
                                                            content_loss_layer="none", nn_loss_area_size=1,
                                                            l1_penalty_weight=100.0), steps=40, parity_n=par)
        if ns is not None and legs["bf16_data_b4_img_s"].get("parity"):
            # the path `north_star.frac_of_bf16_peak` is quoted on, next to ITS parity: the bf16 data path at
            # 256 x 256 against the oracle (first iteration, batch 2; norm and losses are per sample) — NOT north_star's fp32 bar of 1e-3
            ns["parity"] = dict(legs["bf16_data_b4_img_s"]["parity"],
                                note="same kernels as the batch-32 pass timed above.  This is NOT the fp32 path: north_star's 1e-3 / 1e-4 bars "
                                     "are met by the default line's `parity` (fp32) only.  For scale: the GPU suite's bf16 bars on ITS fixtures "
                                     "(tests/test_gpu_round5.py BF16_TOL, 2 x the worst of 3 seeds) are out_gen 0.045 max-abs / 6e-3 mean, losses "
                                     "5e-2; this line's weights and batch are other ones (xavier init_seed 0, bench batch)")
        legs["cfg2_224_p32_b8_bf16"] = config_leg(device, ns_(size=224, batch=8, pose_dim=32, precision="bf16_data",
                                                              content_loss_layer="none", nn_loss_area_size=1,
                                                              l1_penalty_weight=100.0), steps=30, parity_n=par)
        legs["cfg3_nnloss_vgg_b4"] = config_leg(device, ns_(size=256, batch=4, pose_dim=18, precision="f32",
                                                            content_loss_layer="block1_conv2", nn_loss_area_size=5,
                                                            l1_penalty_weight=0.01), steps=20, parity_n=par)
        # configs[4]'s per-GPU shape (512 x 512, batch 64 over 8 GPUs = 8 per GPU, bf16) with its own parity at batch 2
        legs["cfg4_512_b8_bf16"] = config_leg(device, ns_(size=512, batch=8, pose_dim=18, precision="bf16_data",
                                                          content_loss_layer="none", nn_loss_area_size=1,
                                                          l1_penalty_weight=100.0), steps=10, parity_n=par)
    if single and not args.no_extra_legs:
        legs["main_py_img_s"] = {"f32": main_py_leg("f32", 60), "bf16_data": main_py_leg("bf16_data", 60),
                                 "compare_with": "`value` (fp32) and `bf16_data_b4_img_s.value`: the same iteration from bench.py's own loop"}
        legs["dp_forced_1gpu"] = forced_reducer_leg(10)
    cpu = parity = None
    if single and not args.no_cpu_baseline:
        cpu, parity = cpu_baseline(args, pin)

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
                       **({"launch_tape": True} if args.tape else {}),
                       **({"stand_in_model": True} if stand_in else {})},
            "step_tflops": round(sf * ips / 1e12, 2),
            "step_frac_of_f32_mfma_peak": round(sf * ips / 1e12 / (PEAK_F32_MFMA_TFLOPS * world), 4),
            "roofline": roof, "parity": parity, "north_star": ns, "bf16_data_b32_img_s": b32, **legs,
            "rccl_ranks": rccl_ranks, "dp_transport": transport, "per_rank_img_s": [round(v, 3) for v in per_rank],
            "dp": dp_block, "hbm_kernels": hbm, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return None


if __name__ == "__main__":
    main()
