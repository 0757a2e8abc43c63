"""bench.py configuration legs in a given order inside ONE process (hardware-queue mapping of the streams: models/pose_gan.py _PF_STREAMS).
    gpurun -- python tools/leg_order_test.py cfg2,b4,cfg2"""
import sys, os
sys.path.insert(0, os.getcwd())
import bench, torch
from types import SimpleNamespace as ns_
cfg2 = ns_(size=224, batch=8, pose_dim=32, precision="bf16_data", content_loss_layer="none", nn_loss_area_size=1, l1_penalty_weight=100.0)
b4 = ns_(size=256, batch=4, pose_dim=18, precision="bf16_data", content_loss_layer="none", nn_loss_area_size=1, l1_penalty_weight=100.0)
order = sys.argv[1]
for name in order.split(","):
    leg = bench.config_leg("cuda:0", cfg2 if name == "cfg2" else b4, steps=30, parity_n=0)
    print(name, leg["value"], flush=True)
