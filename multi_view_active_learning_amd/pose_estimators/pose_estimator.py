"""Abstract base of the heat-map networks (reference pose_estimators/pose_estimator.py:13-23)."""
import abc

import torch


class PoseEstimator(abc.ABC, torch.nn.Module):
    """``nn.Module`` with ``num_joints`` and an abstract ``forward(x)``:
    x (N, 3, H, W) fp32 NCHW -> heat-maps (N, num_joints, H/4, W/4) fp32 NCHW."""

    def __init__(self, num_joints):
        super().__init__()
        self.num_joints = num_joints

    @abc.abstractmethod
    def forward(self, x):
        pass
