"""Rate of the real-data input pipeline: images per second that
`BatchPipeline` can hand to the trainer at 256x256 / 18 key-points from PNG files, for 4 / 8 / 16 decode workers, next to the
device-side cost of the on-device input generation (pg_preprocess_image + pg_cords_to_map + pg_affine_transforms +
pg_pose_masks per batch, HIP events).  The consumer it has to feed runs at 170 (fp32) - 920 (bf16, batch 32) images / s.
    gpurun -- python tools/pipeline_bench.py        -> profiles/round3_pipeline_rate.txt
Two fixture sets: 'noise' PNGs (uniform random pixels: incompressible, 197 KB per file — the slowest decode a 256^2 PNG can
have) and 'smooth' PNGs (low-frequency content like a photograph on a plain background, ~60 KB)."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pta_bootstrap  # noqa: E402

pta_bootstrap.load()
import dataset_fixture as DF  # noqa: E402
from pose_transfer_amd.datasets.PoseTransfer_Dataset import PoseTransfer_Dataset, BatchPipeline  # noqa: E402
from pose_transfer_amd.runtime import lib as L  # noqa: E402


def smooth_images(opt, n_images):
    """overwrite the fixture's noise images with smooth ones (same names)"""
    from PIL import Image
    H, W = opt["image_size"]
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    for split in ("train", "test"):
        d = opt["images_dir_" + split]
        for i, name in enumerate(sorted(os.listdir(d))):
            img = np.stack([127 + 100 * np.sin(xx / (17 + i % 5) + c) * np.cos(yy / (23 + i % 7) - c) for c in range(3)], -1)
            img[H // 5:4 * H // 5, W // 3:2 * W // 3] += 20 * np.sin(xx[H // 5:4 * H // 5, W // 3:2 * W // 3, None] * 0.9)
            Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(d, name))


def rate(opt, batch, workers, nbatch=60):
    ds = PoseTransfer_Dataset(dict(opt, gen_type="baseline", num_stacks=4, batch_size=batch, use_input_pose=True, warp_skip="mask"),
                              "train")
    pipe = BatchPipeline(ds, batch, "cuda:0", shuffle=True, seed=3, workers=workers)
    for _ in range(8):
        pipe.next()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nbatch):
        b = pipe.next()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # device-side construction of one batch, alone
    raws = [ds.raw(i) for i in range(batch)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    img = torch.from_numpy(np.stack([r[0] for r in raws] + [r[1] for r in raws])).cuda()
    kp = torch.from_numpy(np.stack([r[2] for r in raws] + [r[3] for r in raws]).astype(np.float32)).cuda()
    bufs = ds.alloc(batch, "cuda:0")
    for _ in range(3):
        ds.device_batch(img, kp, batch, [(r[2], r[3]) for r in raws], "cuda:0", bufs)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        ds.device_batch(img, kp, batch, [(r[2], r[3]) for r in raws], "cuda:0", bufs)
    e1.record()
    torch.cuda.synchronize()
    pipe.pool.shutdown(wait=False)
    return batch * nbatch / dt, e0.elapsed_time(e1) / 20 * 1e3


def main():
    out = []
    for kind in ("noise", "smooth"):
        with tempfile.TemporaryDirectory() as tmp:
            opt = DF.write_dataset(tmp, "fasion", pose_dim=18, image_size=(256, 256), n_images=96, n_pairs=512, seed=5)
            if kind == "smooth":
                smooth_images(opt, 96)
            sz = np.mean([os.path.getsize(os.path.join(opt["images_dir_train"], f)) for f in os.listdir(opt["images_dir_train"])])
            for batch in (4, 32):
                for workers in (4, 8, 16):
                    r, us = rate(opt, batch, workers)
                    line = ("%-6s PNG (%3.0f KB/file) 256x256 P=18  batch %2d  workers %2d : %7.1f images/s delivered;  device-side "
                            "sample construction %6.1f us per batch (%5.2f us per image)" % (kind, sz / 1e3, batch, workers, r, us, us / batch))
                    print(line, flush=True)
                    out.append(line)
    with open(os.path.join(ROOT, "gpurun_out", "round3_pipeline_rate.txt"), "w") as f:
        f.write("# tools/pipeline_bench.py on the GPU box (%d host threads): BatchPipeline delivery rate (PNG decode on worker threads,\n"
                "# pinned staging, side-stream upload, on-device key-point geometry) and the device-side cost of one batch alone\n" % os.cpu_count())
        f.write("\n".join(out) + "\n")


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    main()
