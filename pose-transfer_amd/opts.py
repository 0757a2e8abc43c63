"""Command-line options with the reference's flag names (reference src_deformable/opts.py:14-97).
Only flags the hot path reads are acted upon; path/dir side effects of the reference's parse() are limited to
the experiment directories.  New flags: --synthetic, --align_corners, --vgg_weights, --steps."""
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
        # new in this build
        p.add_argument("--synthetic", default=1, type=int, help="train on synthetic tensors (no dataset ships with the repo)")
        p.add_argument("--align_corners", default=0, type=int, help="1 = torch-0.4 grid semantics (SURVEY App. A.2)")
        p.add_argument("--vgg_weights", default=None, help="torchvision vgg19 state_dict for the content loss")
        p.add_argument("--steps", default=0, type=int, help="stop after this many iterations (0 = full schedule)")
        p.add_argument("--seed", default=1234, type=int)
        p.add_argument("--precision", default="f32", choices=["f32", "bf16x3", "bf16", "bf16_data"],
                       help="contraction operand format: f32 = the reference's arithmetic (default); bf16_data = bf16 data path "
                            "(fp32 master weights / accumulation; stated bf16 tolerance, DESIGN.md)")

    def parse(self, argv=None):
        self.init()
        self.opt = self.parser.parse_args(argv)
        o = self.opt
        o.saveDir = os.path.join("../exp/", o.expID)
        o.output_dir = os.path.join("../exp/", o.expID, "results")
        o.checkpoints_dir = os.path.join("../exp/", o.expID, "models")
        # image size by data-set name — reference opts.py:90-97
        if o.dataset == "fasion":
            o.image_size = (256, 256)
        elif o.dataset == "h36m":
            o.image_size = (224, 224)
        elif o.dataset == "fasion128128":
            o.image_size = (128, 128)
        else:
            o.image_size = (128, 64)
        return o
