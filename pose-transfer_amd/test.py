"""Inference / evaluation driver with the reference's shape (reference src_deformable/test.py:25-54): load the last
checkpoint of `--checkpoints_dir`, run the generator forward-only over the test split and write one image grid
[input | target pose | target | generated] per batch to `generated_images_dir` (`%05d.png`).

Forward-only reuse of the training kernels: no backward buffers are touched, the discriminator is never run.
Dropout2d stays ACTIVE by default, exactly as in the reference: its test.py never calls `.eval()` and never reads
`--use_dropout_test`, so the first three decoder blocks keep dropping channels.  `--deterministic_test 1` (new) runs
`model.gen.eval()` first — inference without dropout.

    python pose-transfer_amd/test.py --dataset fasion --pose_dim 18 --batch_size 4 --expID full_fasion --synthetic 0
"""
import os
import sys

import numpy as np
import torch

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import pta_bootstrap
    pta_bootstrap.load()
from pose_transfer_amd.opts import opts  # noqa: E402
from pose_transfer_amd.utils import pose_utils  # noqa: E402


def generate(model, opt, batch):
    """One forward of the generator on a Dataset-shaped batch; returns (out, outputs): the final image and, for the
    stacked generator, every stage output (reference test.py:43-49)."""
    with torch.no_grad():
        if opt.gen_type == "baseline":
            if getattr(model, "deformable", True):
                out = model.gen(batch[0], batch[2].float(), batch[3])
            else:
                out = model.gen(batch[0])
            return out.detach(), []
        if opt.gen_type == "stacked":
            outs = model.gen(batch[0], batch[2], batch[3].float(), batch[4])
            return outs[-1].detach(), [o.detach() for o in outs]
    raise Exception("Invalid gen type !!")


def save_grid(model, opt, batch, out=None, outputs=None, path=None):
    """Image grid of a batch (reference pose_utils.display / display_stacked); generates `out` when it is not given."""
    from PIL import Image
    if out is None:
        out, outputs = generate(model, opt, batch)
    if opt.gen_type == "stacked":
        img = pose_utils.display_stacked(batch[0], batch[2], batch[1], outputs, opt.num_stacks, opt.use_input_pose, opt.pose_dim)
    else:
        img = pose_utils.display(batch[0], batch[1], out, opt.use_input_pose, opt.pose_dim)
    if path is not None:
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        Image.fromarray(np.ascontiguousarray(img)).save(path)
    return img


def main(argv=None):
    from pose_transfer_amd import main as M
    opt = opts().parse(argv)
    device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(device)
    model = M.build(opt, device)
    epoch = model.resume(opt.checkpoints_dir)
    if getattr(opt, "deterministic_test", 0):
        model.gen.eval()
    _, test = M.make_sources(opt, device)
    n_pairs = len(test.ds) if hasattr(test, "ds") else opt.images_for_test
    num_iterations = max(1, n_pairs // opt.batch_size)
    if opt.steps:
        num_iterations = min(num_iterations, opt.steps)
    os.makedirs(opt.generated_images_dir, exist_ok=True)
    for it in range(num_iterations):
        if it % 50 == 0:
            print(it / num_iterations, flush=True)
        batch = test.next()
        save_grid(model, opt, batch, path=os.path.join(opt.generated_images_dir, "{0}.png".format(str(it).zfill(5))))
    return epoch, num_iterations


if __name__ == "__main__":
    main()
