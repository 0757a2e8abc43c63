"""Command-line options with the reference's flag names (reference src_deformable/opts.py:14-97).
Only flags the hot path reads are acted upon; path/dir side effects of the reference's parse() are limited to
the experiment directories.  New flags: --synthetic, --align_corners, --vgg_weights, --steps, --lazy_losses,
--synthetic_ring, --timing_skip."""
import argparse
import os


class opts():
    def __init__(self):
        self.parser = argparse.ArgumentParser(description="Pose guided image generation using deformable skip layers (MI355X build)")

    def init(self):
        p = self.parser
        p.add_argument("--expID", default="default", help="Experiment ID")
        p.add_argument("--data_Dir", default="../data/", help="Directory with annotations and data")
        p.add_argument("--batch_size", default=4, type=int, help="Size of the batch (per process)")
        p.add_argument("--training_ratio", default=1, type=int, help="discriminator updates per generator update")
        p.add_argument("--resume", default=0, type=int, help="resume from checkpoint")
        p.add_argument("--learning_rate", default=2e-4, type=float)
        p.add_argument("--l1_penalty_weight", default=100, type=float)
        p.add_argument("--gan_penalty_weight", default=1, type=float)
        p.add_argument("--tv_penalty_weight", default=0, type=float)
        p.add_argument("--lstruct_penalty_weight", default=0, type=float)
        p.add_argument("--number_of_epochs", default=500, type=int)
        p.add_argument("--content_loss_layer", default="none", help="vgg19 layer, e.g. block1_conv2, or none")
        p.add_argument("--pose_dim", default=16, type=int)
        p.add_argument("--iters_per_epoch", default=1000, type=int)
        p.add_argument("--checkpoint_ratio", default=5, type=int)
        p.add_argument("--generator_checkpoint", default=None)
        p.add_argument("--discriminator_checkpoint", default=None)
        p.add_argument("--nn_loss_area_size", default=1, type=int)
        p.add_argument("--dataset", default="h36m", choices=["market", "fasion", "fasion128", "fasion128128", "h36m"])
        p.add_argument("--num_stacks", default=4, type=int)
        p.add_argument("--display_ratio", default=50, type=int)
        p.add_argument("--use_input_pose", default=True, type=int)
        p.add_argument("--warp_skip", default="mask", choices=["none", "full", "mask"])
        p.add_argument("--warp_agg", default="max", choices=["max", "avg"])
        p.add_argument("--gen_type", default="baseline", choices=["baseline", "stacked"])
        # reference flags that the hot path does not read: accepted so that every reference command line parses
        # (reference opts.py:17-77); --use_dropout_test is honoured by test.py, the directories are overwritten by
        # parse() exactly as in the reference (opts.py:81-84)
        p.add_argument("--output_dir", default="output/displayed_samples")
        p.add_argument("--log_file", default="output/full/fasion/log")
        p.add_argument("--checkpoints_dir", default="output/checkpoints")
        p.add_argument("--frame_diff", default=10, type=int)
        p.add_argument("--compute_h36m_paf_split", default=0, type=int)
        p.add_argument("--start_epoch", default=0, type=int)
        p.add_argument("--pose_estimator", default="pose_estimator.h5")
        p.add_argument("--images_for_test", default=12000, type=int)
        p.add_argument("--disc_type", default="call", choices=["call", "sim", "warp"])
        p.add_argument("--generated_images_dir", default="output/generated_images")
        p.add_argument("--load_generated_images", default=0, type=int)
        p.add_argument("--use_dropout_test", default=0, type=int, help="keep Dropout2d active when generating images")
        # new in this build
        p.add_argument("--synthetic", default=1, type=int,
                       help="1 = synthetic tensors (no dataset ships with the repo); 0 = PoseTransfer_Dataset on --data_Dir")
        p.add_argument("--exp_root", default="../exp/", help="root of saveDir / checkpoints_dir (reference: ../exp/)")
        p.add_argument("--num_workers", default=4, type=int, help="host threads decoding / rasterising the next batches")
        p.add_argument("--device_pose", default=1, type=int,
                       help="1 = heat-maps, limb transforms and limb masks are computed on the GPU from key-points")
        p.add_argument("--align_corners", default=0, type=int, help="1 = torch-0.4 grid semantics (SURVEY App. A.2)")
        p.add_argument("--vgg_weights", default=None, help="torchvision vgg19 state_dict for the content loss")
        p.add_argument("--steps", default=0, type=int, help="stop after this many iterations (0 = full schedule)")
        p.add_argument("--seed", default=1234, type=int)
        p.add_argument("--lazy_losses", default=1, type=int,
                       help="1 = the six loss scalars stay on the device and are read back once per --display_ratio iterations (the "
                            "reference only prints them then, main.py:117-127); 0 = a device->host read-back per update")
        p.add_argument("--synthetic_ring", default=0, type=int,
                       help="--synthetic 1: cycle through this many pre-generated device batches instead of generating one on the host per "
                            "call (0 = fresh batch every call; bench.py's loop uses 3)")
        p.add_argument("--timing_skip", default=10, type=int, help="iterations left out of the steady-state img/s main() reports")
        p.add_argument("--save_samples", default=0, type=int, help="write train / test image grids every display_ratio iterations")
        p.add_argument("--save_at_end", default=0, type=int, help="write a checkpoint when --steps stops the run")
        p.add_argument("--deterministic_test", default=0, type=int, help="test.py: gen.eval() (no Dropout2d) before generating")
        p.add_argument("--precision", default="f32", choices=["f32", "bf16x3", "bf16", "bf16_data"],
                       help="contraction operand format: f32 = the reference's arithmetic (default); bf16_data = bf16 data path "
                            "(fp32 master weights / accumulation; stated bf16 tolerance, DESIGN.md)")

    def parse(self, argv=None):
        self.init()
        self.opt = self.parser.parse_args(argv)
        o = self.opt
        o.saveDir = os.path.join(o.exp_root, o.expID)
        o.output_dir = os.path.join(o.exp_root, o.expID, "results")
        o.checkpoints_dir = os.path.join(o.exp_root, o.expID, "models")
        o.generated_images_dir = os.path.join(o.exp_root, o.expID, "results", "generated")
        # image size by data-set name — reference opts.py:90-97
        if o.dataset == "fasion":
            o.image_size = (256, 256)
        elif o.dataset == "h36m":
            o.image_size = (224, 224)
        elif o.dataset == "fasion128128":
            o.image_size = (128, 128)
        else:
            o.image_size = (128, 64)
        # data-set file layout — reference opts.py:99-121
        d = o.data_Dir + o.dataset
        o.images_dir_train = d + "-dataset/train"
        o.images_dir_test = d + "-dataset/test"
        o.annotations_file_train = d + "-annotation-train.csv"
        o.annotations_file_test = d + "-annotation-test.csv"
        o.annotations_file_train_paf = d + "-annotation-paf-train" + str(o.compute_h36m_paf_split) + ".csv"
        o.annotations_file_test_paf = d + "-annotation-paf-test" + str(o.compute_h36m_paf_split) + ".csv"
        o.pairs_file_train = d + "-pairs-train.csv"
        o.pairs_file_test = d + "-pairs-test.csv"
        o.pairs_file_train_iterative = d + "-pairs-train-iterative.csv"
        o.pairs_file_test_iterative = d + "-pairs-test-iterative.csv"
        o.pairs_file_train_interpol = d + "-pairs-train-interpol.csv"
        o.pairs_file_test_interpol = d + "-pairs-test-interpol.csv"
        o.tmp_pose_dir = "tmp/" + o.dataset + "/"
        return o
