from .config import get_default_configs, hrnet_w48
from .hrnet import PoseHighResolutionNet
from .loss import Pose2DMeanSquaredError
from .pose_estimator import PoseEstimator
from .pose_resnet import PoseResNet


def get_pose_net(cfg):
    """Thin factory with the reference's selection logic (workflow.py:125-139):
    ``cfg.POSE_ESTIMATOR.TYPE`` in {"POSE_RESNET", "HRNET"}, ``cfg.DATA.NUM_JOINTS``."""
    kind = cfg.POSE_ESTIMATOR.TYPE
    if kind == "POSE_RESNET":
        return PoseResNet(cfg.DATA.NUM_JOINTS)
    if kind == "HRNET":
        return PoseHighResolutionNet(cfg.DATA.NUM_JOINTS, hrnet_cfg=cfg.POSE_ESTIMATOR.HRNET)
    raise NotImplementedError(kind)
