import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_train_ops as T
from amuse_amd import train_ops
orig = train_ops.register_lane
train_ops.register_lane = lambda s, lane=1: orig(s, 0)     # sabotage: the side stream shares lane 0
try:
    T.test_two_layer_chains_on_two_streams_do_not_share_scratch()
    print("NOT DETECTED")
except AssertionError:
    print("detected: sharing a lane breaks bitwise equality")
