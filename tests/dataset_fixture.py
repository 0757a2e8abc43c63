"""A tiny on-disk data set in the reference's wire format (reference src_deformable/opts.py:99-121,
datasets/PoseTransfer_Dataset.py:26-46): `<dataset>-dataset/{train,test}/*.png`, `<dataset>-annotation-{train,test}.csv`
(':'-separated, columns name / keypoints_y / keypoints_x with JSON integer lists, -1 = missing) and
`<dataset>-pairs-{train,test}-interpol.csv` (columns from / to).  Used by oracle/make_golden_r2.py (through the REAL
reference Dataset) and by the tests of the build's own loader — both regenerate it from the same seeds."""
import json
import os

import numpy as np

import pta_bootstrap

pta_bootstrap.load()
from pose_transfer_amd.utils import synth  # noqa: E402

LABELS_PAF = ['nose', 'neck', 'Rsho', 'Relb', 'Rwri', 'Lsho', 'Lelb', 'Lwri', 'Rhip', 'Rkne', 'Rank', 'Lhip', 'Lkne',
              'Lank', 'Leye', 'Reye', 'Lear', 'Rear']
LABELS = ['Rank', 'Rknee', 'Rhip', 'Lhip', 'Lknee', 'Lank', 'pelv', 'spine', 'neck', 'head', 'Rwri', 'Relb', 'Rsho',
          'Lsho', 'Lelb', 'Lwri']


def keypoints(seed, tag, n, pose_dim, h, w, p_missing=0.2):
    """(n, P, 2) int64 (y, x); the four torso joints are always present (the reference's geometry needs them)."""
    ky = np.floor(synth.uniform(seed, tag + "/y", (n, pose_dim)) * h).astype(np.int64)
    kx = np.floor(synth.uniform(seed, tag + "/x", (n, pose_dim)) * w).astype(np.int64)
    miss = synth.uniform(seed, tag + "/m", (n, pose_dim)) < p_missing
    names = LABELS if pose_dim == 16 else LABELS_PAF
    for nm in ("Rhip", "Lhip", "Rsho", "Lsho"):
        miss[:, names.index(nm)] = False
    ky[miss] = -1
    kx[miss] = -1
    return np.stack([ky, kx], -1)


def write_dataset(root, dataset, pose_dim=18, image_size=(128, 64), n_images=6, n_pairs=4, seed=81):
    """Writes the files and returns the option dict a Dataset needs (the reference reads exactly these keys)."""
    from PIL import Image
    H, W = image_size
    data_dir = os.path.join(root, "data") + os.sep
    d = data_dir + dataset
    os.makedirs(d + "-dataset/train", exist_ok=True)
    os.makedirs(d + "-dataset/test", exist_ok=True)
    kps = keypoints(seed, "ds/kp", n_images, pose_dim, H, W)
    names = []
    for i in range(n_images):
        img = (synth.uniform(seed, "ds/img%d" % i, (H, W, 3)) * 256).astype(np.uint8)
        name = "img_%03d.png" % i
        names.append(name)
        Image.fromarray(img).save(os.path.join(d + "-dataset", "train" if i % 2 == 0 else "test", name))
    for split, sel in (("train", range(0, n_images, 2)), ("test", range(1, n_images, 2))):
        with open(d + "-annotation-%s.csv" % split, "w") as f:
            f.write("name:keypoints_y:keypoints_x\n")
            for i in sel:
                f.write("%s:%s:%s\n" % (names[i], json.dumps([int(v) for v in kps[i, :, 0]]),
                                        json.dumps([int(v) for v in kps[i, :, 1]])))
    pick = np.floor(synth.uniform(seed, "ds/pairs", (2, n_pairs, 2)) * n_images).astype(int)
    for s, split in enumerate(("train", "test")):
        with open(d + "-pairs-%s-interpol.csv" % split, "w") as f:
            f.write("from,to\n")
            for a, b in pick[s]:
                f.write("%s,%s\n" % (names[a], names[(b if b != a else (a + 1) % n_images)]))
    return dict(dataset=dataset, pose_dim=pose_dim, image_size=(H, W), data_Dir=data_dir,
                images_dir_train=d + "-dataset/train", images_dir_test=d + "-dataset/test",
                annotations_file_train=d + "-annotation-train.csv", annotations_file_test=d + "-annotation-test.csv",
                pairs_file_train_interpol=d + "-pairs-train-interpol.csv",
                pairs_file_test_interpol=d + "-pairs-test-interpol.csv",
                pairs_file_train=d + "-pairs-train.csv", pairs_file_test=d + "-pairs-test.csv")
