"""Round-5 golden vectors for BASELINE.json configs[4]'s geometry: runs the REAL reference (imported read-only from /root/reference
under the shims of oracle/make_golden.py) at 512 x 512, 18 key-points, 7 levels (8 x 8 bottleneck), N = 2; writes
tests/golden/g512.npz.  (configs[4] had property checks only.)

    python oracle/make_golden_r5.py            # ~5 minutes of CPU in the build container

Inputs / weights are regenerated from pose_transfer_amd.utils.synth; only reference OUTPUTS are stored (strided samples and
summaries of the big tensors: SURVEY.md 8c "Large shapes: strided samples + sum, sum|x|, max").

* gen_eval / gen_train   Deformable_Generator.forward (reference models/networks.py:252-288) in eval mode and in train mode with
                         explicit Dropout2d masks.
* l1_*                   one dis_update + gen_update (models/pose_gan.py:69-171) with the L1 pixel loss: loss triples, out_gen,
                         the summary of every parameter gradient.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_golden as MG  # noqa: E402
from pose_transfer_amd.utils import synth  # noqa: E402

t, summarize, load_sd, OUT, REF = MG.t, MG.summarize, MG.load_sd, MG.OUT, MG.REF
_DROP = MG._DROP_QUEUE
P, H, W, N = 18, 512, 512, 2          # (N = 1 is not a batch to the reference: its InstanceNorm3d(1) call reads a 4-D input as unbatched)
STRIDE = 10         # 52 x 52 samples per plane


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    MG.install_shims(os.path.join(REF, "src_deformable"))
    from models import networks as rnet
    from models import pose_gan as rgan
    enc, dec = synth.nfilters((H, W))
    assert len(enc) == 7
    fix = {}
    gpar = synth.init_params(95, "g512/gen", synth.generator_spec(P, enc, dec), norm_jitter=0.2)
    gen = rnet.Deformable_Generator(3 + 2 * P, P, (H, W), enc, dec, "mask")
    load_sd(gen, gpar)
    inp, tgt, wr, mk = synth.batch(95, "g512", N, P, H, W)
    for mode in ("eval", "train"):
        del _DROP[:]
        if mode == "train":
            _DROP.extend(t(m) for m in synth.dropout_masks(95, "g512", N))
        with torch.no_grad():
            out = gen(t(inp), t(wr), t(mk).double())
        fix["gen_%s_summary" % mode] = summarize(out)
        fix["gen_%s_strided" % mode] = out[:, :, ::STRIDE, ::STRIDE].numpy()
        print("gen", mode, fix["gen_%s_summary" % mode][:3], flush=True)

    dspec = synth.discriminator_spec(3 + 2 * P + 3)
    name = "l1"
    opt = types.SimpleNamespace(image_size=(H, W), use_input_pose=True, pose_dim=P, batch_size=N, num_stacks=4,
                                gen_type="baseline", dataset="fasion", warp_skip="mask", learning_rate=2e-4,
                                content_loss_layer="none", nn_loss_area_size=1, gan_penalty_weight=1.0, l1_penalty_weight=100.0)
    model = rgan.DeformablePose_GAN(opt)
    load_sd(model.gen, synth.init_params(96, "g512/%s/gen" % name, synth.generator_spec(P, enc, dec), 0.1))
    load_sd(model.disc, synth.init_params(96, "g512/%s/disc" % name, dspec, 0.1))
    od = vars(opt)
    bA, bB, bC = [synth.batch(96, "g512/%s/%s" % (name, s), N, P, H, W) for s in "ABC"]
    del _DROP[:]
    _DROP.extend(t(m) for m in synth.dropout_masks(96, "g512/%s/dA" % name, N))
    dl = model.dis_update(t(bA[0]), t(bA[1]), {"warps": t(bA[2]), "masks": t(bA[3]).double()}, t(bB[0]), t(bB[1]), od)
    fix[name + "_dis_losses"] = np.array(dl)
    for k, p in model.disc.named_parameters():
        fix["%s_dgrad_%s" % (name, k)] = summarize(p.grad)
    del _DROP[:]
    _DROP.extend(t(m) for m in synth.dropout_masks(96, "g512/%s/dC" % name, N))
    og, _, gl = model.gen_update(t(bC[0]), t(bC[1]), {"warps": t(bC[2]), "masks": t(bC[3]).double()}, od)
    fix[name + "_gen_losses"] = np.array(gl)
    fix[name + "_out_gen_summary"] = summarize(og)
    fix[name + "_out_gen_strided"] = og.detach()[:, :, ::STRIDE, ::STRIDE].numpy()
    for k, p in model.gen.named_parameters():
        fix["%s_ggrad_%s" % (name, k)] = summarize(p.grad)
    print(name, dl, gl, flush=True)
    np.savez_compressed(os.path.join(OUT, "g512.npz"), **fix)
    print("g512.npz", os.path.getsize(os.path.join(OUT, "g512.npz")), "bytes")


if __name__ == "__main__":
    main()
