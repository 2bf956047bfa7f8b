"""Keypoint decode + RANSAC-DLT triangulation -- drop-in for the hot-path part of the
reference's utils/triangulation.py (:168-484).

``triangulation(...)`` keeps the reference signature and result dict for one frame
(utils/triangulation.py:168-233) and is a thin view over ``triangulate_batch`` which
processes a whole batch of frames with three HIP launches (arg-max decode, pairwise
RANSAC + DLT, per-frame reduction) instead of V*J device->host syncs and
J*(C(V,2)+1) LAPACK calls per frame.  float64 geometry, SVD-free (csrc/triangulate.hip).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib


def _device_of(heatmaps):
    if not torch.is_tensor(heatmaps) or not heatmaps.is_cuda:
        raise _lib.MvalError("heatmaps must be a HIP tensor: the hot path has no CPU implementation")
    return heatmaps.device


def _as_valid_u8(valid, shape, device):
    if valid is None:
        return None
    v = torch.as_tensor(valid)
    v = (v != 0).to(torch.uint8).reshape(shape)
    return v.to(device).contiguous()


def triangulate_batch(
    heatmaps,
    proj_matricies,
    stride,
    valid_joints,
    use_soft_argmax=False,
    use_reprojection_xe=False,
    sigma=None,
    n_iters=64,
    reprojection_error_epsilon=5,
    mirror_nonsquare_quirk=True,
    keypoints_2d=None,
):
    """heatmaps (B,V,J,Hh,Wh) f32 HIP tensor, proj (B,V,3,4), valid (B,J) ->
    dict of HIP tensors: keypoints_3d (B,J,3) f64, keypoints_2d (B,V,J,2) i64|f32,
    metric (B,) f64, inlier_count (B,) i32 (-1 where no joint is valid), joint_error,
    joint_inliers (B,J).

    ``mirror_nonsquare_quirk``: the reference splits the flat arg-max index with
    ``shape[2]`` (the map HEIGHT) for both x and y (utils/evaluation.py:25-26).
    True reproduces that bit for bit; False uses the geometrically correct width.

    ``keypoints_2d``: key-points already decoded from these heat-maps (the fused scoring pass,
    ``_lib.score_decode_maps``): the decode launch, i.e. a second read of the heat-maps, is skipped."""
    dev = _device_of(heatmaps)
    if heatmaps.dim() != 5:
        raise ValueError("heatmaps must be (B, V, J, Hh, Wh)")
    b, v, j, hh, wh = heatmaps.shape
    if v < 2:
        raise AssertionError("need at least two views")  # reference: assert len(points) >= 2
    if v * (v - 1) // 2 > n_iters:
        raise NotImplementedError(
            "more view pairs than n_iters: the reference samples pairs from python's global RNG there"
        )
    hm = heatmaps.to(torch.float32).contiguous()
    proj = torch.as_tensor(proj_matricies).to(device=dev, dtype=torch.float64).reshape(b, v, 3, 4).contiguous()
    valid = _as_valid_u8(valid_joints, (b, j), dev)
    if keypoints_2d is not None:
        kp2d = keypoints_2d
    elif use_soft_argmax:
        kp2d = _lib.soft_argmax(hm, b * v * j, hh, wh, float(stride)).reshape(b, v, j, 2)
    else:
        kp2d = _lib.argmax_decode(hm, valid, b, v, j, hh, wh, int(stride), hh if mirror_nonsquare_quirk else wh)
    kp3d, jerr, jinl, metric, inl = _lib.triangulate_ransac(kp2d, proj, valid, b, v, j, float(reprojection_error_epsilon))
    if use_reprojection_xe:
        metric = _lib.reprojection_xe(kp3d, proj, hm, b, v, j, hh, wh, float(sigma))
    return {
        "keypoints_3d": kp3d,
        "keypoints_2d": kp2d,
        "metric": metric,
        "inlier_count": inl,
        "joint_error": jerr,
        "joint_inliers": jinl,
    }


def triangulation(
    heatmaps,
    proj_matricies,
    stride,
    valid_joints,
    use_soft_argmax=False,
    use_reprojection_xe=False,
    sigma=None,
    n_iters=64,
    reprojection_error_epsilon=5,
    direct_optimization=False,
):
    """One frame, reference signature and return types (utils/triangulation.py:168-233):
    heatmaps (V,J,Hh,Wh), proj (V,3,4), valid (J,) -> {"keypoints_3d": ndarray (J,3) f64,
    "keypoints_2d": ndarray (V,J,2) int64|f32, "metric": float, "inlier_count": int}."""
    if direct_optimization:
        raise NotImplementedError("direct_optimization (scipy Huber least-squares, off in every reference call site)")
    r = triangulate_batch(
        heatmaps.unsqueeze(0),
        torch.as_tensor(proj_matricies).unsqueeze(0),
        stride,
        torch.as_tensor(valid_joints).reshape(1, -1),
        use_soft_argmax,
        use_reprojection_xe,
        sigma,
        n_iters,
        reprojection_error_epsilon,
    )
    inl = int(r["inlier_count"][0].item())
    if inl < 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")  # np.min([])
    return {
        "keypoints_3d": r["keypoints_3d"][0].cpu().numpy(),
        "keypoints_2d": r["keypoints_2d"][0].cpu().numpy(),
        "metric": float(r["metric"][0].item()),
        "inlier_count": inl,
    }
