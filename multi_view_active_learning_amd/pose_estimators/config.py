"""Model configuration with the reference's field names and defaults
(pose_estimators/config.py:10-56): TYPE, STRIDE=4, HRNET.{PRETRAINED_LAYERS,
FINAL_CONV_KERNEL, STAGE2..4.{NUM_MODULES,NUM_BRANCHES,BLOCK,NUM_BLOCKS,NUM_CHANNELS,
FUSE_METHOD}}.  The shipped default is HRNet-W32; ``hrnet_w48()`` is the same tree with
channels 48/96/192/384 (the reference leaves that to the caller's YAML)."""
from ..cfgnode import CfgNode as CN


def _stage(modules, branches, channels):
    s = CN()
    s.NUM_MODULES = modules
    s.NUM_BRANCHES = branches
    s.BLOCK = "BASIC"
    s.NUM_BLOCKS = [4] * branches
    s.NUM_CHANNELS = list(channels)
    s.FUSE_METHOD = "SUM"
    return s


def _hrnet(width):
    h = CN()
    h.PRETRAINED_LAYERS = [
        "conv1", "bn1", "conv2", "bn2", "layer1", "transition1", "stage2", "transition2", "stage3",
    ]
    h.FINAL_CONV_KERNEL = 1
    h.STAGE2 = _stage(1, 2, (width, 2 * width))
    h.STAGE3 = _stage(4, 3, (width, 2 * width, 4 * width))
    h.STAGE4 = _stage(3, 4, (width, 2 * width, 4 * width, 8 * width))
    return h


def get_default_configs():
    c = CN()
    c.TYPE = "POSE_RESNET"
    c.LOAD_CNN_WEIGHTS = True
    c.STRIDE = 4
    c.HRNET = _hrnet(32)
    return c


def hrnet_w48():
    return _hrnet(48)
