"""Experiment configuration tree with the reference's field names and defaults
(config.py:14-106, dataset/config.py:10-51).  Only the hot-path knobs are consumed by
this package: POSE_ESTIMATOR.{TYPE,STRIDE,HRNET.*}, DATA.{NUM_JOINTS,INPUT_WIDTH,
INPUT_HEIGHT,TYPE}, AL.{STRATEGY,USE_SOFTARGMAX,USE_REPROJECTION_XE,REPROJECTION_SIGMA,
MPE_CONFIG,HP_CONFIG,BSB_CONFIG,ITER_AMOUNT,INFERENCE.BATCH_SIZE}, TRAIN.{BATCH_SIZE,
LOSS_CLIP_VALUE,OPTIM.*}, NUM_GPUS; the rest is carried so that a reference YAML merges."""
from .cfgnode import CfgNode as CN
from .pose_estimators.config import get_default_configs as _pose_defaults


def _data_defaults():
    d = CN()
    d.INPUT_WIDTH = 256
    d.INPUT_HEIGHT = 256
    d.SCALE_BBOX = 1.0
    d.SIGMA = 1.0
    d.PSEUDO_LABEL_SIGMA = 1.0
    d.TYPE = "panoptic"  # or "ih26m"
    d.EPOCH_SIZE = 2000
    d.NUM_JOINTS = 19  # 42 for ih26m
    d.NUM_AUG = 0
    d.AUG_MAGNITUDE = 0
    d.USE_ROTATION = True
    d.USE_IMAGE_AUG = True
    d.USE_CONST_AUG_MAGNITUDE = True
    return d


def get_default_configs():
    c = CN()
    c.EXPR_NAME = "EXPR"
    c.EXPR_TYPE = "SUPERVISED"
    c.COMMENT = "N/A"
    c.RANDOM_SEED = 1307
    c.NUM_GPUS = 1

    c.SAL = CN()
    c.SAL.NUM_FRAMES = [0, 20, 20, 30, 30, 40, 40, 50, 50, 50]
    c.SAL.INLIER_THRESHOLD = 7
    c.SAL.CLUSTER_FILE_PATH = ""
    c.SAL.NUM_CLUSTERS = 10

    c.AL = CN()
    c.AL.STRATEGY = "RANDOM"  # HP | BSB | RANDOM | MPE | TRIANGULATION | CORESET
    c.AL.INITIAL_AMOUNT = 200
    c.AL.ITER_AMOUNT = 100
    c.AL.START_ITER = 0
    c.AL.ITERATIONS = 10
    c.AL.USE_SOFTARGMAX = False
    c.AL.USE_REPROJECTION_XE = False
    c.AL.REPROJECTION_SIGMA = 1.0
    c.AL.MPE_CONFIG = "AVG"
    c.AL.BSB_CONFIG = "AVG"
    c.AL.HP_CONFIG = "AVG"
    c.AL.INFERENCE = CN()
    c.AL.INFERENCE.BATCH_SIZE = 2
    c.AL.INFERENCE.NUM_WORKERS = 2

    c.TRAIN = CN()
    c.TRAIN.LOSS_CLIP_VALUE = 10.0
    c.TRAIN.BATCH_SIZE = 2
    c.TRAIN.NUM_WORKERS = 2
    c.TRAIN.LOG_EVERY_ITER = 500
    c.TRAIN.OPTIM = CN()
    c.TRAIN.OPTIM.TOTAL_STEPS = 5000
    c.TRAIN.OPTIM.LR = 0.001
    c.TRAIN.OPTIM.LR_DECAY_STEP_SIZE = 3000

    c.EVAL = CN()
    c.EVAL.METRIC = "3DPCK"

    c.POSE_ESTIMATOR = _pose_defaults()
    c.DATA = _data_defaults()
    return c
