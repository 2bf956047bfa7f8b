"""Deterministic synthetic inputs for parity tests and benchmarks (host logic, numpy only).

There is no dataset or checkpoint on the build/GPU boxes, so every test and the
benchmark run on inputs generated here from a seed with ``numpy.random.default_rng``
(never torch's RNG, so CPU oracle and GPU path see bit-identical inputs):

* images  ~ N(0,1), shape (N, V, 3, H, W)            
* cameras: V cameras on a ring of radius 3000 mm looking at the origin,
  f = 300 px * (H/256), principal point (W/2, H/2), P = K [R|t] in float64
  (the dataset hands float64 projection matrices: reference dataset/dataset.py:195)
* weights: a variance-preserving random ``state_dict`` under the reference's key
  names.  The reference's own init (N(0, 0.001) on every conv,
  hrnet.py:355-368) collapses activations to ~1e-10, so it is useless for numerics
  tests; this recipe keeps every layer O(1) and gives every BatchNorm non-trivial
  running statistics so the BN arithmetic is exercised.
* Gaussian heat-maps (sigma in heat-map pixels) around projected 3-D joints, the
  shape of the dataset's ground truth (reference dataset/dataset.py:196-207).
"""
from __future__ import annotations

import zlib

import numpy as np


def _rng(seed: int, tag: str) -> np.random.Generator:
    return np.random.default_rng([int(seed) & 0x7FFFFFFF, zlib.crc32(tag.encode())])


def images(seed: int, n: int, v: int, h: int, w: int) -> np.ndarray:
    return _rng(seed, "images").standard_normal((n, v, 3, h, w), dtype=np.float32)


def ring_cameras(v: int, h: int, w: int, radius: float = 3000.0, seed: int = 0) -> np.ndarray:
    """(V, 3, 4) float64 projection matrices; small seeded jitter in height so the
    views are not coplanar-symmetric."""
    rng = _rng(seed, "cameras")
    f = 300.0 * (h / 256.0)
    k = np.array([[f, 0.0, w / 2.0], [0.0, f, h / 2.0], [0.0, 0.0, 1.0]])
    out = np.zeros((v, 3, 4))
    for i in range(v):
        ang = 2.0 * np.pi * (i + 0.25 * rng.uniform(-1, 1)) / v
        c = np.array([radius * np.cos(ang), radius * np.sin(ang), 400.0 * rng.uniform(-1, 1)])
        z = -c / np.linalg.norm(c)  # optical axis towards the origin
        up = np.array([0.0, 0.0, 1.0])
        x = np.cross(z, up)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        r = np.stack([x, y, z])
        t = -r @ c
        out[i] = k @ np.concatenate([r, t[:, None]], axis=1)
    return out


def joints_3d(seed: int, n: int, j: int, sigma_mm: float = 300.0) -> np.ndarray:
    """(N, 3, J) float32 -- the dataset's '3d_keypoints' layout [coord, joint]."""
    return (_rng(seed, "joints").standard_normal((n, 3, j)) * sigma_mm).astype(np.float32)


def project(p: np.ndarray, x3: np.ndarray) -> np.ndarray:
    """P (..., 3, 4), X (J, 3) -> (..., J, 2)."""
    xh = np.concatenate([x3, np.ones((x3.shape[0], 1))], axis=1)
    q = np.einsum("...ij,kj->...ki", p, xh)
    return q[..., :2] / q[..., 2:3]


def gaussian_heatmaps(
    seed: int,
    proj: np.ndarray,
    kp3d: np.ndarray,
    hh: int,
    wh: int,
    stride: int,
    sigma: float = 1.0,
    noise: float = 0.02,
    outlier_views: int = 0,
) -> np.ndarray:
    """Heat-maps (N, V, J, Hh, Wh) float32 peaked at the projection of kp3d
    (N, 3, J) through proj (N, V, 3, 4); ``outlier_views`` views per frame get
    their peaks moved to a random place (exercises the RANSAC inlier vote)."""
    rng = _rng(seed, "heatmaps")
    n, v = proj.shape[:2]
    j = kp3d.shape[2]
    ys = np.arange(hh, dtype=np.float32)[:, None]
    xs = np.arange(wh, dtype=np.float32)[None, :]
    out = np.empty((n, v, j, hh, wh), dtype=np.float32)
    for b in range(n):
        p2 = project(proj[b], kp3d[b].T.astype(np.float64)) / stride  # (V, J, 2)
        bad = rng.permutation(v)[:outlier_views]
        for vi in range(v):
            for ji in range(j):
                cx, cy = p2[vi, ji]
                if vi in bad:
                    cx, cy = rng.uniform(2, wh - 3), rng.uniform(2, hh - 3)
                g = np.exp(-((xs - np.float32(cx)) ** 2 + (ys - np.float32(cy)) ** 2) / np.float32(2.0 * sigma**2))
                out[b, vi, ji] = g
    out += noise * rng.standard_normal(out.shape, dtype=np.float32)
    return out


# --------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------
def _is_residual_tail_bn(key: str) -> bool:
    """BatchNorms that close a residual branch (their output is added to the skip)."""
    parts = key.split(".")
    if "branches" in parts and parts[-2] == "bn2":
        return True
    if parts[0].startswith("layer") and parts[-2] == "bn3":
        return True
    return False


def synthetic_state_dict(shapes: dict, seed: int = 0, residual_gain: float = 0.25, fuse_gain: float = 0.2) -> dict:
    """``shapes``: ordered {state_dict key: shape} (from a freshly built model).
    Returns {key: np.ndarray} (float32; ``num_batches_tracked`` int64 zeros).

    conv / deconv weights ~ N(0, sqrt(2 / fan_in)); BN gamma ~ U(0.8, 1.2)
    (x residual_gain on residual tails, x fuse_gain on fuse/downsample paths; tuned so
    HRNet-W32 heat-maps come out with std ~4),
    beta ~ N(0, 0.1), running_mean ~ N(0, 0.1), running_var ~ U(0.6, 1.4);
    biases ~ N(0, 0.1).
    """
    out = {}
    for key, shape in shapes.items():
        rng = _rng(seed, key)
        shape = tuple(shape)
        leaf = key.split(".")[-1]
        if leaf == "num_batches_tracked":
            out[key] = np.zeros(shape, dtype=np.int64)
        elif leaf == "weight" and len(shape) == 4:
            if key.startswith("deconv_layers"):
                # ConvTranspose2d weight is (Cin, Cout, k, k); stride 2 => each output
                # pixel sees k*k/4 taps of every input channel
                fan_in = shape[0] * shape[2] * shape[3] / 4.0
            else:
                fan_in = shape[1] * shape[2] * shape[3]
            std = np.sqrt(2.0 / fan_in)
            if key.startswith("final_layer"):
                std = np.sqrt(1.0 / fan_in)
            out[key] = (rng.standard_normal(shape) * std).astype(np.float32)
        elif leaf == "weight":
            g = rng.uniform(0.8, 1.2, shape)
            if _is_residual_tail_bn(key):
                g = g * residual_gain
            elif "fuse_layers" in key or "downsample" in key:
                g = g * fuse_gain
            out[key] = g.astype(np.float32)
        elif leaf == "bias":
            out[key] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == "running_mean":
            out[key] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == "running_var":
            out[key] = rng.uniform(0.6, 1.4, shape).astype(np.float32)
        else:  # pragma: no cover
            raise KeyError(f"unexpected state_dict entry {key}")
    return out
