"""PoseHighResolutionNet -- drop-in for reference pose_estimators/hrnet.py:293-532.

Same constructor (``num_joints, hrnet_cfg=None``), ``state_dict`` keys, ``.num_joints``,
``.pretrained_layers`` and NCHW fp32 I/O; the forward pass is executed by the HIP
engine (fused conv+BN+residual+ReLU launches over NHWC activations) instead of ~900
framework ops.  There is no CPU path: calling the model on a non-HIP tensor raises.
"""
from __future__ import annotations

import torch

from . import graph as _graph
from . import params as _params
from .config import get_default_configs
from .pose_estimator import PoseEstimator

BN_MOMENTUM = 0.1


class PoseHighResolutionNet(PoseEstimator):
    def __init__(self, num_joints, hrnet_cfg=None):
        super().__init__(num_joints=num_joints)
        self.hrnet_cfg = get_default_configs().HRNET if hrnet_cfg is None else hrnet_cfg
        self._graph = _graph.build_hrnet(num_joints, self.hrnet_cfg)
        self._holders = _params.attach_parameters(self, self._graph)
        self.pretrained_layers = self.hrnet_cfg.PRETRAINED_LAYERS
        # reference default init (hrnet.py:355-368)
        _params.init_normal_(self._holders, std=0.001)
        self._runner = None

    def forward(self, x):
        from ..engine import run_network

        return run_network(self, x)

    def load_pretrained_weights(self, path_to_weights):
        """hrnet.py:503-532: re-init, then load the entries whose first key component is in
        ``pretrained_layers`` (or all, when it starts with '*'), non-strict."""
        _params.init_normal_(self._holders, std=0.001)
        sd = torch.load(path_to_weights, map_location="cpu")
        keep = {
            k: v
            for k, v in sd.items()
            if k.split(".")[0] in self.pretrained_layers or self.pretrained_layers[0] == "*"
        }
        self.load_state_dict(keep, strict=False)
