"""PoseResNet -- drop-in for reference pose_estimators/pose_resnet.py:17-153.

Same constructor (``num_joints, num_layers=50``), ``state_dict`` keys and NCHW fp32 I/O;
forward = HIP engine.  Depths 18/34 do not work in the reference either (its BasicBlock
lacks ``expansion``) and raise here.
"""
from __future__ import annotations

from . import graph as _graph
from . import params as _params
from .pose_estimator import PoseEstimator

BN_MOMENTUM = 0.1


class PoseResNet(PoseEstimator):
    def __init__(self, num_joints, num_layers=50):
        super().__init__(num_joints=num_joints)
        self.deconv_with_bias = False
        self._graph = _graph.build_pose_resnet(num_joints, num_layers)
        self._holders = _params.attach_parameters(self, self._graph)
        # backbone: torch defaults; deconv head + final layer: N(0, 0.001) (pose_resnet.py:48-67)
        _params.init_torch_default_(self._holders, skip_prefix=("deconv_layers", "final_layer"))
        _params.init_normal_(self._holders, std=0.001, only_prefix=("deconv_layers", "final_layer"))
        self._runner = None

    def forward(self, x):
        from ..engine import run_network

        return run_network(self, x)
