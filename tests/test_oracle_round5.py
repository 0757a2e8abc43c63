"""Round-5 CPU test: the oracle at BASELINE.json configs[4]'s geometry (512 x 512, 18 key-points, 7 levels, 8 x 8 bottleneck)
against tensors captured from the REAL reference (tests/golden/g512.npz, oracle/make_golden_r5.py) — the pin behind the
512 x 512 GPU tests of tests/test_gpu_round5.py.  Generator forward only here (the training step at this size is minutes of CPU;
the GPU test compares the step with the capture directly)."""
import os
import sys

import numpy as np
import torch

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_cpu as R  # noqa: E402
from pose_transfer_amd.utils import synth  # noqa: E402

P, H, W, N, STRIDE = 18, 512, 512, 2, 10


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_oracle_generator_512_vs_reference_capture():
    """reference models/networks.py:252-288 at 512^2: eval forward (strided samples 2e-5, summary)."""
    fix = np.load(os.path.join(GOLDEN, "g512.npz"))
    enc, dec = synth.nfilters((H, W))
    gp = {k: t(v) for k, v in synth.init_params(95, "g512/gen", synth.generator_spec(P, enc, dec), norm_jitter=0.2).items()}
    inp, tgt, wr, mk = [t(a) for a in synth.batch(95, "g512", N, P, H, W)]
    with torch.no_grad():
        out = R.generator_forward(inp, wr, mk, gp, P, enc, dec, (H, W), None, False, aten_warp=True)
    d = (out[:, :, ::STRIDE, ::STRIDE] - t(fix["gen_eval_strided"])).abs().max().item()
    assert d < 2e-5, d
    f = out.reshape(-1).double()
    ref = fix["gen_eval_summary"]
    assert abs(f.sum().item() - ref[0]) < 1e-5 * out.numel() and abs(f.abs().max().item() - ref[2]) < 2e-5
