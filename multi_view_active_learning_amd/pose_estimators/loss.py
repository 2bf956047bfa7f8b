"""Masked heat-map MSE -- drop-in for reference pose_estimators/loss.py:10-24.

``pose_2d_mse`` = sum(where(valid, (h - gt)^2, 0)) / (N * H * W): the divisor omits J
(SURVEY Appendix A.12).  Forward and backward are single fused HIP reductions
(``mval_masked_mse_fwd`` / ``_bwd``) wrapped in an autograd Function.
"""
from __future__ import annotations

import torch

from .. import _lib


class _MaskedMSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, heatmaps, gt, valid, denom):
        h = heatmaps.contiguous()
        g = gt.to(dtype=torch.float32).contiguous()
        lead = h.shape[0] * h.shape[1]
        hw = h.shape[-1] * h.shape[-2]
        if valid is None:
            v = None
        else:
            v = valid.expand(h.shape[0], h.shape[1], 1, 1).reshape(lead).to(torch.uint8).contiguous()
        out = _lib.masked_mse_fwd(h, g, v, lead, hw, denom)
        ctx.save_for_backward(h, g, v if v is not None else torch.empty(0, device=h.device))
        ctx.has_valid = v is not None
        ctx.dims = (lead, hw, denom)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        h, g, v = ctx.saved_tensors
        lead, hw, denom = ctx.dims
        gh = _lib.masked_mse_bwd(h, g, v if ctx.has_valid else None, grad_out.contiguous(), lead, hw, denom)
        return gh, None, None, None


class Pose2DMeanSquaredError:
    def pose_2d_mse(self, heatmaps, gt_heatmaps, joint_valid=None):
        """heatmaps, gt (N, J, H, W); joint_valid broadcastable (N, J, 1, 1) bool/uint8."""
        denom = heatmaps.shape[0] * heatmaps.shape[-1] * heatmaps.shape[-2]
        return _MaskedMSE.apply(heatmaps, gt_heatmaps, joint_valid, float(denom))

    def pose_2d_mse_single_batch(self, heatmap, gt_heatmap):
        """loss.py:22-24: sum((h - gt)^2) / (H * W)."""
        h = heatmap.reshape(1, -1, heatmap.shape[-2], heatmap.shape[-1])
        g = gt_heatmap.reshape(1, -1, gt_heatmap.shape[-2], gt_heatmap.shape[-1])
        return _MaskedMSE.apply(h, g, None, float(heatmap.shape[-1] * heatmap.shape[-2]))
