"""Real-data input pipeline (reference src_deformable/datasets/PoseTransfer_Dataset.py:13-192), redesigned for a GPU that
consumes hundreds of images per second.

Wire formats are the reference's (opts.py:99-121, Dataset.py:26-46):
  * pairs CSV (`<dataset>-pairs-{train,test}-interpol.csv`): columns `from`, `to` (image file names);
  * annotation CSVs (`<dataset>-annotation-{train,test}.csv`, ':'-separated): `name`, `keypoints_y`, `keypoints_x`, the
    last two JSON integer lists with -1 = missing (pose_utils.py:160-163) — as in the reference both files are
    concatenated into one table indexed by name (Dataset.py:40-46);
  * images under `<dataset>-dataset/{train,test}/`, looked up in that order, an all-zero image when absent
    (Dataset.py:135-144).

What moved: the reference builds every sample on the main Python thread — image decode, P Gaussian maps, ten least-squares
limb fits and ten polygon masks per sample in NumPy / scikit-image (Dataset.py:77-108) — about 60 ms per sample.  Here a
sample is only DECODED on the host (worker threads; PIL releases the GIL): uint8 pixels and 2 x P key-point integers.
`BatchPipeline` stages a batch in pinned memory, copies it with ONE async H2D per tensor on a side stream and runs the
rest on the device, written straight into the tensors the trainer reads:
  pg_preprocess_image -> input[:, :3] / target   (pose_utils.py:216-217)
  pg_cords_to_map     -> input[:, 3:3+P], input[:, 3+P:]   (pose_utils.py:79-86)
  pg_affine_transforms / pg_pose_masks -> warps (N,10,8), masks (N,10,H,W)   (pose_transform.py:143-289)
  (warp_skip 'full': pg_uniform_transform; gen_type 'stacked': the interpolated poses of Dataset.py:146-158 as well)
Batch k+1 is decoded / uploaded / rasterised while step k trains; `next()` only makes the training stream wait for an event.
"""
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pandas as pd
import torch

from ..runtime import lib as L
from ..utils import pose_transform, pose_utils


def _imread(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


class PoseTransfer_Dataset:
    """Index over image pairs; `raw(i)` is the host part of a sample, `__getitem__` the reference-shaped sample."""

    def __init__(self, opt, split):
        self.split = split
        self.gen_type = opt["gen_type"]
        self.num_stacks = opt["num_stacks"]
        self.pose_dim = opt["pose_dim"]
        self._batch_size = 1 if split in ("test", "val") else opt["batch_size"]
        self._image_size = tuple(opt["image_size"])
        self._images_dir_train = opt["images_dir_train"]
        self._images_dir_test = opt["images_dir_test"]
        self._pairs_file_train = pd.read_csv(opt["pairs_file_train_interpol"])
        self._pairs_file_test = pd.read_csv(opt["pairs_file_test_interpol"])
        ann = pd.concat([pd.read_csv(opt["annotations_file_train"], sep=":"),
                         pd.read_csv(opt["annotations_file_test"], sep=":")], axis=0, ignore_index=True)
        self._annotations_file = ann.set_index("name")
        self._pairs = self._pairs_file_train if split == "train" else self._pairs_file_test
        self.length = len(self._pairs)
        self._use_input_pose = opt["use_input_pose"]
        self._warp_skip = opt["warp_skip"]
        self.device = opt.get("device", "cuda")
        self._kp_cache = {}

    def __len__(self):
        return self.length

    # ------------------------------------------------------------------ host part
    def keypoints(self, name):
        kp = self._kp_cache.get(name)
        if kp is None:
            row = self._annotations_file.loc[name]
            kp = pose_utils.load_pose_cords_from_strings(row["keypoints_y"], row["keypoints_x"]).astype(np.float32)
            self._kp_cache[name] = kp
        return kp

    def load_image(self, name):
        for d in (self._images_dir_train, self._images_dir_test):
            p = os.path.join(d, name)
            if os.path.exists(p):
                img = _imread(p)
                assert img.shape[:2] == self._image_size, "image %s is %s, expected %s" % (name, img.shape[:2], self._image_size)
                return img
        return np.zeros(self._image_size + (3,), dtype=np.uint8)      # the reference's "blank image" fallback

    def raw(self, index):
        pair = self._pairs.iloc[index]
        return (self.load_image(pair["from"]), self.load_image(pair["to"]), self.keypoints(pair["from"]),
                self.keypoints(pair["to"]))

    def interpol_keypoints(self, kp_from, kp_to):
        """Key-points of the num_stacks interpolation stages as the reference ends up using them for warps and masks:
        rendered to heat-maps and read back as integer peaks (Dataset.py:112-133,146-158).  Returns
        (maps (S, P, 2) float32 — centres of the rendered heat-maps, chain (S+1, P, 2) float32 — peak positions
        [input, stage 1 .. S])."""
        H, W = self._image_size
        src = pose_utils.peak_cords(kp_from, (H, W))
        dst = pose_utils.peak_cords(kp_to, (H, W))
        maps = [pose_utils.compute_interpol_pose(src, dst, i, self.num_stacks, self.pose_dim) for i in range(1, self.num_stacks + 1)]
        chain = [src] + [pose_utils.peak_cords(m, (H, W)) for m in maps]
        return np.asarray(maps, dtype=np.float32), np.asarray(chain, dtype=np.float32)

    # ------------------------------------------------------------------ device part
    def collate(self, raws, device=None, out=None):
        """[raw samples] -> the reference's batch: baseline (input, target, warps, masks); stacked (input, target,
        interpol_pose, interpol_warps, interpol_masks).  `out`: pre-allocated tensors to fill."""
        device = device or self.device
        n, P, (H, W) = len(raws), self.pose_dim, self._image_size
        img = torch.from_numpy(np.stack([r[0] for r in raws] + [r[1] for r in raws])).to(device, non_blocking=True)
        kp = torch.from_numpy(np.stack([r[2] for r in raws] + [r[3] for r in raws])).to(device, non_blocking=True)
        return self.device_batch(img, kp, n, [(r[2], r[3]) for r in raws], device, out)

    def device_batch(self, img, kp, n, kps_host, device, out=None):
        """img (2n, H, W, 3) uint8 and kp (2n, P, 2) float32 ON THE DEVICE ([from..., to...]) -> batch tensors."""
        P, (H, W) = self.pose_dim, self._image_size
        f32 = dict(dtype=torch.float32, device=device)
        T = 10 if self._warp_skip == "mask" else 1
        S = self.num_stacks
        if out is None:
            out = self.alloc(n, device)
        inp, tgt = out["input"], out["target"]
        st = L.stream()
        L.call("pg_preprocess_image", L.ptr(img), n, H, W, L.ptr(inp), *inp.stride(), st)
        L.call("pg_preprocess_image", img.data_ptr() + n * H * W * 3, n, H, W, L.ptr(tgt), *tgt.stride(), st)
        src_pose, tgt_pose = inp[:, 3:3 + P], inp[:, 3 + P:]
        L.call("pg_cords_to_map", L.ptr(kp), n, P, H, W, 6.0, L.ptr(src_pose), *src_pose.stride(), st)
        L.call("pg_cords_to_map", kp.data_ptr() + n * P * 2 * 4, n, P, H, W, 6.0, L.ptr(tgt_pose), *tgt_pose.stride(), st)
        kp_from, kp_to = kp[:n], kp[n:]
        if self.gen_type == "stacked":
            maps, chain = zip(*[self.interpol_keypoints(a, b) for a, b in kps_host])
            maps = torch.from_numpy(np.stack(maps)).to(device)               # (n, S, P, 2)
            chain = torch.from_numpy(np.stack(chain)).to(device)             # (n, S+1, P, 2)
            ip = out["interpol_pose"]                                        # (n, S*P, H, W)
            L.call("pg_cords_to_map", L.ptr(maps.view(n, S * P, 2).contiguous()), n, S * P, H, W, 6.0, L.ptr(ip), *ip.stride(), st)
            # warps / masks of [input, stage 1..S] against the previous pose of the chain (Dataset.py:112-133)
            prev = torch.cat([chain[:, :1], chain[:, :-1]], dim=1).reshape(n * (S + 1), P, 2).contiguous()
            cur = chain.reshape(n * (S + 1), P, 2).contiguous()
            if self._warp_skip == "mask":
                pose_transform.affine_transforms(prev, cur, P, device, out=out["interpol_warps"].view(n * (S + 1), 10, 8), check=False)
                pose_transform.pose_masks(cur, (H, W), P, device, out=out["interpol_masks"].view(n * (S + 1), 10, H, W), check=False)
            else:
                out["interpol_warps"].view(n * (S + 1), 1, 8).copy_(pose_transform.estimate_uniform_transform(prev, cur, P, device, check=False))
            return inp, tgt, ip, out["interpol_warps"], out["interpol_masks"]
        if self._warp_skip == "mask":
            pose_transform.affine_transforms(kp_from, kp_to, P, device, out=out["warps"], check=False)
            pose_transform.pose_masks(kp_to, (H, W), P, device, out=out["masks"], check=False)
        else:
            out["warps"].copy_(pose_transform.estimate_uniform_transform(kp_from, kp_to, P, device, check=False))
        return inp, tgt, out["warps"], out["masks"]

    def alloc(self, n, device):
        P, (H, W), S = self.pose_dim, self._image_size, self.num_stacks
        f32 = dict(dtype=torch.float32, device=device)
        T = 10 if self._warp_skip == "mask" else 1
        out = {"input": torch.empty(n, 3 + 2 * P, H, W, **f32), "target": torch.empty(n, 3, H, W, **f32)}
        if self.gen_type == "stacked":
            out["interpol_pose"] = torch.empty(n, S * P, H, W, **f32)
            out["interpol_warps"] = torch.empty(n, S + 1, T, 8, **f32)
            out["interpol_masks"] = torch.ones(n, S + 1, T, H, W, **f32) if T == 10 else torch.ones(n, S + 1, 1, 1, 1, **f32)
        else:
            out["warps"] = torch.empty(n, T, 8, **f32)
            out["masks"] = torch.ones(n, T, H, W, **f32) if T == 10 else torch.ones(n, 1, 1, 1, **f32)
        return out

    def __getitem__(self, index):
        """One reference-shaped sample (device tensors, leading batch dimension removed)."""
        raw = self.raw(index)
        for k in (raw[2], raw[3]):
            pose_transform._check_torso(k, self.pose_dim)
        return tuple(t[0] for t in self.collate([raw]))


def shard_indices(n_items, batch, rank, world, seed, shuffle, order, epoch, cursor):
    """Indices of this rank's next batch -> (indices, order, epoch, cursor).  The epoch permutation depends on (seed, epoch)
    ONLY — every rank must pass the same seed — and rank r takes rows [r*batch, (r+1)*batch) of each global batch of
    batch*world items, so that over an epoch the rank slices are disjoint and cover the permutation (minus a dropped tail)."""
    if order is None or cursor + batch * world > len(order):
        g = np.random.RandomState(seed + epoch)
        order = g.permutation(n_items) if shuffle else np.arange(n_items)
        epoch += 1
        cursor = 0
        if len(order) < batch * world:      # tiny data sets: sample with replacement
            order = g.randint(0, n_items, size=batch * world)
    lo = cursor + rank * batch
    return order[lo:lo + batch], order, epoch, cursor + batch * world


class BatchPipeline:
    """Prefetching batch source: `next()` returns device tensors of the next batch; decode (worker threads), pinned
    staging, H2D copy and the device-side sample construction of batch k+1.. run on a side stream while step k trains.
    Replaces torch.utils.data.DataLoader(dataset, batch_size, shuffle=True) of the reference (main.py:50-60) — same
    shuffled-epoch semantics, iterator restarts when exhausted (main.py:24-43 `load_sample`)."""

    RING = 8          # batch buffers in flight: an iteration holds 3 batches (dis A, dis B, gen C) while more are prefetched

    def __init__(self, dataset, batch_size, device, shuffle=True, seed=0, workers=4, depth=3, rank=0, world=1,
                 drop_last=True):
        self.ds, self.n, self.device = dataset, batch_size, device
        self.shuffle, self.seed, self.rank, self.world = shuffle, seed, rank, world
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self.depth = depth
        self.epoch, self.cursor, self.order = 0, 0, None
        self.stream = torch.cuda.Stream(device=device)
        H, W = dataset._image_size
        P = dataset.pose_dim
        self.bufs = [dataset.alloc(batch_size, device) for _ in range(self.RING)]
        self.pin_img = [torch.empty(2 * batch_size, H, W, 3, dtype=torch.uint8).pin_memory() for _ in range(self.RING)]
        self.pin_kp = [torch.empty(2 * batch_size, P, 2, dtype=torch.float32).pin_memory() for _ in range(self.RING)]
        self.dev_img = [torch.empty(2 * batch_size, H, W, 3, dtype=torch.uint8, device=device) for _ in range(self.RING)]
        self.dev_kp = [torch.empty(2 * batch_size, P, 2, dtype=torch.float32, device=device) for _ in range(self.RING)]
        self.slot = 0
        self.pending = []          # decode jobs in flight: [(future list, slot)]
        self.staged = []           # batches whose upload + device-side construction is enqueued: [(tensors, event)]
        self.ahead = 2
        for _ in range(depth + self.ahead):
            self._submit()
        for _ in range(self.ahead):
            self._stage()

    def _indices(self):
        idx, self.order, self.epoch, self.cursor = shard_indices(len(self.ds), self.n, self.rank, self.world, self.seed,
                                                                 self.shuffle, self.order, self.epoch, self.cursor)
        return idx

    def _submit(self):
        idx = self._indices()
        slot = self.slot
        self.slot = (self.slot + 1) % self.RING
        self.pending.append(([self.pool.submit(self.ds.raw, int(i)) for i in idx], slot))

    def _stage(self):
        """Take the oldest decoded batch: pinned staging, async H2D and the device-side construction on the side stream."""
        futs, slot = self.pending.pop(0)
        raws = [f.result() for f in futs]
        n = self.n
        for r in raws:
            pose_transform._check_torso(r[2], self.ds.pose_dim), pose_transform._check_torso(r[3], self.ds.pose_dim)
        pi, pk = self.pin_img[slot], self.pin_kp[slot]
        for j, r in enumerate(raws):        # (np.copyto into the pinned buffers: decoded arrays may be read-only views)
            np.copyto(pi[j].numpy(), r[0], casting="unsafe"); np.copyto(pi[n + j].numpy(), r[1], casting="unsafe")
            np.copyto(pk[j].numpy(), r[2], casting="unsafe"); np.copyto(pk[n + j].numpy(), r[3], casting="unsafe")
        # this slot's tensors were handed out RING batches ago; their consumers are already enqueued on the training
        # stream, so the side stream only has to wait for the work enqueued there so far (it then runs under the NEXT step)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self.dev_img[slot].copy_(pi, non_blocking=True)
            self.dev_kp[slot].copy_(pk, non_blocking=True)
            batch = self.ds.device_batch(self.dev_img[slot], self.dev_kp[slot], n, [(r[2], r[3]) for r in raws], self.device,
                                         self.bufs[slot])
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.staged.append((batch, ev))
        self._submit()

    def next(self):
        batch, ev = self.staged.pop(0)
        torch.cuda.current_stream(self.device).wait_event(ev)
        self._stage()
        return batch

    __next__ = next

    def __iter__(self):
        return self
