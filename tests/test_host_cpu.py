"""CPU tests of the host logic around the hot path: option parsing (every reference command line parses), the training
driver's model construction + gen_%03d.pkl / disc_%03d.pkl save/resume round trip, the data-pipeline helpers and the
wire-format parsing of the Dataset (no kernel is launched here)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests"))
import dataset_fixture as DF  # noqa: E402
from pose_transfer_amd.opts import opts  # noqa: E402
from pose_transfer_amd.utils import pose_utils, synth  # noqa: E402

# the flag sets of the reference's README / src_deformable/commands, plus every flag its opts.py defines (opts.py:14-77)
REFERENCE_COMMAND_LINES = [
    "--l1_penalty_weight 100 --batch_size 4 --number_of_epochs 90 --gen_type baseline --expID full_fasion --pose_dim 18 --dataset fasion",
    "--warp_skip mask --dataset fasion --l1_penalty_weight 0.01 --nn_loss_area_size 5 --batch_size 2 --content_loss_layer block1_conv2 --number_of_epochs 90 --gen_type baseline --pose_dim 18 --expID dsc_fasion",
    "--warp_skip full --dataset market --gen_type stacked --num_stacks 4 --checkpoint_ratio 1 --display_ratio 10",
    "--output_dir o --log_file l --checkpoints_dir c --frame_diff 5 --start_epoch 3 --pose_estimator p.h5 --images_for_test 100 "
    "--disc_type warp --generated_images_dir g --load_generated_images 1 --use_dropout_test 1 --compute_h36m_paf_split 2 "
    "--tv_penalty_weight 1 --lstruct_penalty_weight 1 --warp_agg avg --use_input_pose 1 --training_ratio 2 --resume 1 "
    "--learning_rate 1e-4 --gan_penalty_weight 2 --iters_per_epoch 10 --generator_checkpoint a --discriminator_checkpoint b --data_Dir ../d/",
]


@pytest.mark.parametrize("line", REFERENCE_COMMAND_LINES)
def test_reference_command_lines_parse(line):
    o = opts().parse(line.split())
    assert o.image_size in ((256, 256), (224, 224), (128, 128), (128, 64))
    # reference opts.py:81-121 derived attributes
    assert o.checkpoints_dir == os.path.join(o.exp_root, o.expID, "models")
    assert o.generated_images_dir == os.path.join(o.exp_root, o.expID, "results", "generated")
    assert o.annotations_file_train == o.data_Dir + o.dataset + "-annotation-train.csv"
    assert o.pairs_file_test_interpol == o.data_Dir + o.dataset + "-pairs-test-interpol.csv"
    assert o.annotations_file_train_paf.endswith("-annotation-paf-train%d.csv" % o.compute_h36m_paf_split)


def test_image_size_by_dataset():
    for ds, size in (("fasion", (256, 256)), ("h36m", (224, 224)), ("fasion128128", (128, 128)), ("market", (128, 64)),
                     ("fasion128", (128, 64))):
        assert opts().parse(["--dataset", ds]).image_size == size


@pytest.mark.parametrize("gen_type,warp_skip", [("baseline", "mask"), ("stacked", "mask"), ("baseline", "full")])
def test_driver_build_and_checkpoint_roundtrip(tmp_path, gen_type, warp_skip):
    """main.build() on the CPU (arenas are plain tensors), save() -> resume() through the reference's file names."""
    from pose_transfer_amd import main as M
    o = opts().parse(["--dataset", "market", "--pose_dim", "18", "--batch_size", "2", "--gen_type", gen_type,
                      "--warp_skip", warp_skip, "--exp_root", str(tmp_path), "--expID", "rt"])
    model = M.build(o, "cpu")
    sd0 = {k: v.clone() for k, v in model.gen.state_dict().items()}
    dd0 = {k: v.clone() for k, v in model.disc.state_dict().items()}
    if gen_type == "stacked":
        assert all(k.startswith("generator.") for k in sd0)          # reference networks.py:302
    assert "encoder_app.net.0.weight" in model._core.state_dict() and tuple(model._core.state_dict()["decoder.net.0.net.1.weight"].shape) == (1024, 512, 4, 4)
    model.save(o.checkpoints_dir, 7)
    assert sorted(os.listdir(o.checkpoints_dir)) == ["disc_007.pkl", "gen_007.pkl"]
    other = M.build(opts().parse(["--dataset", "market", "--pose_dim", "18", "--batch_size", "2", "--gen_type", gen_type,
                                  "--warp_skip", warp_skip, "--exp_root", str(tmp_path), "--expID", "rt"]), "cpu")
    other._core.arena.params.add_(1.0)
    other.disc.arena.params.add_(1.0)
    assert other.resume(o.checkpoints_dir) == 7
    for k in sd0:
        assert torch.equal(other.gen.state_dict()[k], sd0[k]), k
    for k in dd0:
        assert torch.equal(other.disc.state_dict()[k], dd0[k]), k
    assert M.build(o, "cpu").resume(str(tmp_path / "nowhere")) == 1
    # a checkpoint written by torch.save of a reference-layout state_dict loads (OIHW / IOHW tensors under the reference keys)
    sd = torch.load(os.path.join(o.checkpoints_dir, "gen_007.pkl"))
    key = ("generator." if gen_type == "stacked" else "") + "encoder_pose.net.1.net.1.weight"
    assert tuple(sd[key].shape) == (128, 64, 4, 4)


def test_invalid_configurations_raise():
    from pose_transfer_amd import main as M
    o = opts().parse(["--dataset", "market", "--pose_dim", "18"])
    o.gen_type = "nonsense"
    with pytest.raises(Exception, match="Invalid gen_type"):
        M.build(o, "cpu")
    o = opts().parse(["--dataset", "market", "--pose_dim", "18", "--content_loss_layer", "block4_conv1"])
    with pytest.raises(Exception, match="block1_conv2"):
        M.build(o, "cpu")


def test_synthetic_source_shapes():
    from pose_transfer_amd import main as M
    for gen_type, warp_skip in (("baseline", "mask"), ("baseline", "full"), ("stacked", "mask")):
        o = opts().parse(["--dataset", "market", "--pose_dim", "18", "--batch_size", "2", "--gen_type", gen_type,
                          "--warp_skip", warp_skip, "--num_stacks", "3"])
        b = M.SyntheticSource(o, "cpu").next()
        assert tuple(b[0].shape) == (2, 39, 128, 64) and tuple(b[1].shape) == (2, 3, 128, 64)
        if gen_type == "stacked":
            assert tuple(b[2].shape) == (2, 54, 128, 64) and tuple(b[3].shape) == (2, 3, 10, 8) and tuple(b[4].shape) == (2, 3, 10, 128, 64)
            oi = M.other_inputs(o, b)
            assert set(oi) == {"interpol_pose", "interpol_warps", "interpol_masks"}
        else:
            T = 10 if warp_skip == "mask" else 1
            assert tuple(b[2].shape) == (2, T, 8)


def test_synthetic_ring_and_loss_log():
    """main.py (round 6): --synthetic_ring K hands out K pre-generated batches round-robin (the fresh source's first K batches); LossLog
    keeps lazily returned loss triples as tensors and moves them to the host in one copy when the means are asked for (where the
    reference prints them, main.py:117-127) — the running means equal those of eagerly read-back floats."""
    from pose_transfer_amd import main as M
    base = ["--dataset", "market", "--pose_dim", "18", "--batch_size", "2"]
    fresh = M.SyntheticSource(opts().parse(base), "cpu")
    ring = M.SyntheticSource(opts().parse(base + ["--synthetic_ring", "2"]), "cpu")
    f = [fresh.next() for _ in range(2)]
    r = [ring.next() for _ in range(5)]
    for i in range(5):
        for a, b in zip(r[i], f[i % 2]):
            assert torch.equal(a.float(), b.float())
    assert r[2][0] is r[0][0]                                   # the ring re-uses its tensors
    lazy, eager = M.LossLog(), M.LossLog()
    rng = np.random.RandomState(3)
    for i in range(7):
        v = rng.uniform(0, 3, 3).astype(np.float32)
        lazy.append(torch.from_numpy(v.copy()))
        eager.append([float(x) for x in v])
        if i in (2, 6):
            assert len(lazy.pending) > 0
            np.testing.assert_allclose(lazy.means(), eager.means(), rtol=1e-7)
            assert not lazy.pending and len(lazy.host) == i + 1


# ---------------------------------------------------------------------------------------------- pipeline helpers
def test_peak_cords_equals_render_and_read_back():
    """peak_cords == map_to_cord(cords_to_map(.)) — what the reference does for the interpolated poses."""
    H, W = 40, 30
    # rows 6.. are OUT OF FRAME: map_to_cord's 0.1 threshold (reference pose_utils.py:55-60) keeps those whose nearest
    # in-image pixel is closer than sqrt(-72 ln 0.1) = 12.88 px and drops the others
    cords = np.array([[3, 4], [10.5, 7.25], [-1, -1], [20.49, 29.0], [0.5, 0.5], [39, 0],
                      [-5.0, 10.0], [45.0, 12.0], [-12.0, 10.0], [-13.0, 10.0], [20.0, 42.5], [50.0, 40.0], [-9.0, -9.0],
                      [-9.5, -9.5]], dtype=np.float64)
    maps = pose_utils.cords_to_map(cords, (H, W))
    back = pose_utils.map_to_cord(maps, len(cords))
    assert (back == pose_utils.peak_cords(cords, (H, W))).all()
    assert (back[2] == -1).all() and tuple(back[1]) == (10, 7) and tuple(back[4]) == (0, 0)
    assert tuple(back[6]) == (0, 10) and tuple(back[8]) == (0, 10) and (back[9] == -1).all() and (back[11] == -1).all()
    assert tuple(back[12]) == (0, 0) and (back[13] == -1).all()


def test_compute_interpol_pose_rules():
    a = np.array([[10, 10], [-1, -1], [20, 20], [-1, -1]] + [[5, 5]] * 14, dtype=np.float64)
    b = np.array([[30, 50], [8, 8], [-1, -1], [-1, -1]] + [[5, 9]] * 14, dtype=np.float64)
    p1 = pose_utils.compute_interpol_pose(a, b, 1, 4, 18)
    p3 = pose_utils.compute_interpol_pose(a, b, 3, 4, 18)
    assert np.allclose(p1[0], [15, 20]) and np.allclose(p3[0], [25, 40])
    assert (p1[1] == -1).all() and np.allclose(p3[1], [8, 8])          # appears in the second half
    assert np.allclose(p1[2], [20, 20]) and (p3[2] == -1).all()        # vanishes in the second half
    assert (p1[3] == -1).all() and (p3[3] == -1).all()
    lin = pose_utils.compute_interpol_pose(a[:16], b[:16], 2, 4, 16)
    assert np.allclose(lin[0], [20, 30])


def test_deprocess_and_grid():
    x = torch.tensor([-1.0, 0.0, 1.0, 0.5])
    assert pose_utils._deprocess_image(x).tolist() == [0, 127, 255, 191]
    batch = np.arange(4 * 2 * 3 * 1).reshape(4, 2, 3, 1)
    g = pose_utils.make_grid(batch, row=2, col=2)
    assert g.shape == (4, 6, 1)
    assert (g[:2, :3] == batch[0]).all() and (g[2:, :3] == batch[1]).all() and (g[:2, 3:] == batch[2]).all()
    g1 = pose_utils.make_grid(batch, row=2, col=2, order=1)
    assert (g1[:2, 3:] == batch[1]).all()


def test_display_grid_layout():
    N, P, H, W = 2, 18, 32, 24
    inp, tgt, _, _ = [torch.from_numpy(a) for a in synth.batch(3, "disp", N, P, H, W)]
    img = pose_utils.display(inp, tgt, tgt * 0.5, True, P)
    assert img.shape == (N * H, 4 * W, 3) and img.dtype == np.uint8
    want = pose_utils._deprocess_image(tgt).permute(0, 2, 3, 1).numpy()
    assert (img[:H, 2 * W:3 * W] == want[0]).all() and (img[H:, 2 * W:3 * W] == want[1]).all()


def test_dataset_wire_format_parsing(tmp_path):
    """pairs CSV / ':'-separated annotation CSV with JSON lists / image lookup order / blank-image fallback."""
    from pose_transfer_amd.datasets.PoseTransfer_Dataset import PoseTransfer_Dataset
    opt = DF.write_dataset(str(tmp_path), "fasion128", pose_dim=18, image_size=(128, 64), n_images=6, n_pairs=4, seed=81)
    opt.update(gen_type="baseline", num_stacks=2, warp_skip="mask", use_input_pose=True, batch_size=2)
    ds = PoseTransfer_Dataset(opt, "train")
    assert len(ds) == 4 and len(PoseTransfer_Dataset(opt, "test")) == 4
    kps = DF.keypoints(81, "ds/kp", 6, 18, 128, 64)
    import pandas as pd
    pairs = pd.read_csv(opt["pairs_file_train_interpol"])
    img_from, img_to, k_from, k_to = ds.raw(1)
    i_from, i_to = int(pairs.iloc[1]["from"][4:7]), int(pairs.iloc[1]["to"][4:7])
    assert (k_from == kps[i_from]).all() and (k_to == kps[i_to]).all() and k_from.dtype == np.float32
    want = (synth.uniform(81, "ds/img%d" % i_from, (128, 64, 3)) * 256).astype(np.uint8)
    assert img_from.dtype == np.uint8 and (img_from == want).all()
    os.remove(os.path.join(opt["images_dir_train"] if i_to % 2 == 0 else opt["images_dir_test"], "img_%03d.png" % i_to))
    assert (ds.raw(1)[1] == 0).all()                       # reference Dataset.py:141-143
    maps, chain = ds.interpol_keypoints(k_from, k_to)
    assert maps.shape == (2, 18, 2) and chain.shape == (3, 18, 2) and (chain[0] == k_from).all()


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_rank_shards_of_an_epoch_are_disjoint_and_cover_the_permutation(world):
    """every rank draws the SAME epoch permutation and takes its own slice of each global batch."""
    from pose_transfer_amd.datasets.PoseTransfer_Dataset import shard_indices
    n_items, batch, seed = 203, 3, 11
    per_epoch = n_items // (batch * world)
    seen = [[[] for _ in range(2)] for _ in range(world)]
    for r in range(world):
        order, epoch, cursor = None, 0, 0
        for it in range(2 * per_epoch):
            idx, order, epoch, cursor = shard_indices(n_items, batch, r, world, seed, True, order, epoch, cursor)
            assert len(idx) == batch
            seen[r][it // per_epoch] += list(map(int, idx))
    for ep in range(2):
        allidx = sum((seen[r][ep] for r in range(world)), [])
        assert len(allidx) == per_epoch * batch * world
        assert len(set(allidx)) == len(allidx), "a sample was served to two ranks within one epoch"
        perm = np.random.RandomState(seed + ep).permutation(n_items)
        assert set(allidx) == set(map(int, perm[:len(allidx)]))
    assert seen[0][0] != seen[0][1]           # a new permutation per epoch
