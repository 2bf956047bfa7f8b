"""TEST INFRASTRUCTURE ONLY -- CPU (numpy) restatement of the reference's
keypoint decode + RANSAC-DLT triangulation path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import this module; the product path (``multi_view_active_learning_amd``)
never does and fails loudly when its HIP library is missing.

Pinned against the real reference (imported via ``oracle/ref_harness.py`` in the
build container) by ``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``,
including the reference's own test input ``tests/test_triangulation.py:15-69``.

Every function cites the reference lines it restates (paths relative to
``/root/reference``).
"""
from __future__ import annotations

import itertools

import numpy as np


# --------------------------------------------------------------------------
# keypoint decode
# --------------------------------------------------------------------------
def argmax_decode(heatmaps: np.ndarray, stride: int, valid_joints) -> np.ndarray:
    """utils/evaluation.py:13-30 ``get_scaled_pred_corrdinates``.

    heatmaps (V, J, Hh, Wh) float32 -> (V, J, 2) int64 [x, y].
    Mirrors the reference's quirk: the flat argmax index is split with
    ``shape[2]`` (= Hh) for BOTH modulo and division, ties
    resolve to the lowest flat index, invalid joints give [0, 0].
    """
    heatmaps = np.asarray(heatmaps)
    v, j, hh, wh = heatmaps.shape
    flat = heatmaps.reshape(v, j, hh * wh)
    # torch.argmax treats NaN as the maximum (first NaN wins); mirror that
    nan_mask = np.isnan(flat)
    idx = np.argmax(np.where(nan_mask, np.inf, flat), axis=-1)
    has_nan = nan_mask.any(axis=-1)
    if has_nan.any():
        first_nan = np.argmax(nan_mask, axis=-1)
        idx = np.where(has_nan, first_nan, idx)
    out = np.zeros((v, j, 2), dtype=np.int64)
    out[..., 0] = (idx % hh) * stride
    out[..., 1] = (idx // hh) * stride
    valid = np.asarray(valid_joints).astype(bool).reshape(-1)
    out[:, ~valid, :] = 0
    return out


def spatial_soft_argmax2d(heatmaps: np.ndarray) -> np.ndarray:
    """kornia.spatial_soft_argmax2d(x, temperature=1, normalized_coordinates=False)
    as called at utils/triangulation.py:194-196 and utils/evaluation.py:38.

    PARITY UNPINNED: kornia is a third-party dependency that is neither vendored
    under /root/reference nor installed here (version not pinned by the reference,
    TARGETS:33-50).  This restates its documented definition: softmax over the
    flattened H*W map, expectation of the un-normalised pixel grid
    (x in [0, W-1], y in [0, H-1]); output (..., 2) = (x, y), float32.
    """
    x = np.asarray(heatmaps, dtype=np.float32)
    *lead, h, w = x.shape
    flat = x.reshape(*lead, h * w)
    m = flat.max(axis=-1, keepdims=True)
    e = np.exp(flat - m, dtype=np.float32)
    p = e / e.sum(axis=-1, keepdims=True, dtype=np.float32)
    xs = np.tile(np.arange(w, dtype=np.float32), h)
    ys = np.repeat(np.arange(h, dtype=np.float32), w)
    ex = (p * xs).sum(axis=-1, dtype=np.float32)
    ey = (p * ys).sum(axis=-1, dtype=np.float32)
    return np.stack([ex, ey], axis=-1).astype(np.float32)


def spatial_soft_argmax2d_torch(heatmaps, normalized_coordinates=False, temperature=None):
    """torch-facing wrapper so the reference can call the restatement when
    ``oracle.ref_harness`` installs it as the kornia stand-in."""
    import torch

    assert not normalized_coordinates
    out = spatial_soft_argmax2d(heatmaps.detach().cpu().numpy())
    return torch.from_numpy(out)


# --------------------------------------------------------------------------
# projective helpers
# --------------------------------------------------------------------------
def homogeneous_to_euclidean(points: np.ndarray) -> np.ndarray:
    """utils/triangulation.py:387-399 (w == 0 -> 1 guard)."""
    points = np.asarray(points)
    z = points.T[-1]
    z = np.where(z == 0, np.ones_like(z), z)
    return (points.T[:-1] / z).T


def project(proj_matrix: np.ndarray, points_3d: np.ndarray) -> np.ndarray:
    """utils/triangulation.py:459-477 (numpy branch), points (N,3) -> (N,2)."""
    pts = np.hstack([points_3d, np.ones((len(points_3d), 1))])
    return homogeneous_to_euclidean(pts @ proj_matrix.T)


def triangulate_dlt(proj_matricies: np.ndarray, points: np.ndarray) -> np.ndarray:
    """utils/triangulation.py:341-368: rows x*P[2]-P[0], y*P[2]-P[1]; vh[3]."""
    n = len(proj_matricies)
    a = np.zeros((2 * n, 4))
    for j in range(n):
        a[2 * j] = points[j][0] * proj_matricies[j][2, :] - proj_matricies[j][0, :]
        a[2 * j + 1] = points[j][1] * proj_matricies[j][2, :] - proj_matricies[j][1, :]
    _, _, vh = np.linalg.svd(a, full_matrices=False)
    return homogeneous_to_euclidean(vh[3, :])


def reprojection_errors(kp3d: np.ndarray, points: np.ndarray, proj_matricies: np.ndarray) -> np.ndarray:
    """utils/triangulation.py:371-384: err_v = 1/2 * ||pt_v - pi(P_v X)||."""
    errs = []
    for pt, pm in zip(points, proj_matricies):
        pr = project(pm, kp3d[None, :])
        errs.append(0.5 * np.sqrt(np.sum((pt - pr) ** 2, axis=1))[0])
    return np.asarray(errs)


def triangulate_ransac(proj_matricies, points, n_iters=64, eps=5.0):
    """utils/triangulation.py:260-338 without direct_optimization.

    Deterministic for C(V,2) <= n_iters (V <= 11): lexicographic pairs, first
    strictly-larger inlier set wins, the sampled pair is always an inlier.
    """
    proj_matricies = np.asarray(proj_matricies)
    points = np.asarray(points)
    n_views = len(points)
    assert len(proj_matricies) == n_views and n_views >= 2
    pairs = list(itertools.combinations(range(n_views), 2))
    if len(pairs) > n_iters:
        raise NotImplementedError("V >= 12 consumes python's global RNG in the reference (out of scope)")
    inliers: set = set()
    for pr in pairs:
        pr = list(pr)
        x = triangulate_dlt(proj_matricies[pr], points[pr])
        err = reprojection_errors(x, points, proj_matricies)
        new = set(pr)
        for v in range(n_views):
            if err[v] < eps:
                new.add(v)
        if len(new) > len(inliers):
            inliers = new
    if not inliers:
        inliers = set(range(n_views))
    lst = np.array(sorted(inliers))
    x = triangulate_dlt(proj_matricies[lst], points[lst])
    err = reprojection_errors(x, points[lst], proj_matricies[lst])
    return x, float(np.mean(err)), len(inliers)


def compute_xe(keypoints_3d, proj_matricies, pred_heatmaps, sigma):
    """utils/triangulation.py:236-257: sum over (view, joint) of
    mean((pred - exp(-|grid - kp|^2 / (2 sigma^2)))^2); kp is the reprojection in
    INPUT-pixel units laid on the HEATMAP-sized grid (mirrored as is).  float64
    because the reference's rendered target is float64 (numpy keypoints)."""
    pred = np.asarray(pred_heatmaps)
    _, _, h, w = pred.shape
    gx = np.arange(w, dtype=np.float32)[None, :].astype(np.float64)
    gy = np.arange(h, dtype=np.float32)[:, None].astype(np.float64)
    total = 0.0
    for v, pm in enumerate(proj_matricies):
        kp2d = project(pm, np.asarray(keypoints_3d, dtype=np.float64))
        for j, kp in enumerate(kp2d):
            expo = (gx - kp[0]) ** 2 + (gy - kp[1]) ** 2
            target = np.exp(-expo / (2.0 * sigma**2))
            d = pred[v, j].astype(np.float64) - target
            total = total + np.sum(d * d) / (1 * w * h)
    return total


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
):
    """utils/triangulation.py:168-233."""
    heatmaps = np.asarray(heatmaps)
    pm = np.asarray(proj_matricies)
    valid = np.asarray(valid_joints).astype(bool).reshape(-1)
    n_joints = heatmaps.shape[1]
    if use_soft_argmax:
        kp2d = spatial_soft_argmax2d(heatmaps) * np.float32(stride)
    else:
        kp2d = argmax_decode(heatmaps, stride, valid)
    kp3d = np.zeros((n_joints, 3))
    errs, counts = [], []
    for j in range(n_joints):
        if not valid[j]:
            continue
        x, e, c = triangulate_ransac(pm, kp2d[:, j], n_iters, reprojection_error_epsilon)
        kp3d[j] = x
        errs.append(e)
        counts.append(c)
    if use_reprojection_xe:
        metric = compute_xe(kp3d, pm, heatmaps, sigma)
    else:
        metric = np.mean(errs)
    return {
        "keypoints_3d": kp3d,
        "keypoints_2d": kp2d,
        "metric": metric,
        "inlier_count": np.min(counts),  # ValueError when no joint is valid (reference :231)
    }
