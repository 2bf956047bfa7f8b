"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's core-set
(greedy k-center) selector, utils/coreset.py:13-95, and of the one sklearn routine
on its path (``sklearn.metrics.pairwise_distances(..., "euclidean")``, present in
this image as scikit-learn 1.7.2 and pinned through the golden vectors).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import this; the product path never does.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np


def stacked_features(sal_dict, al_dict, root_idx: int) -> np.ndarray:
    """utils/coreset.py:35-47: pool rows first, labeled rows last; each pose
    (J, >=3) -> transpose -> rows 0..2 minus the root joint column -> flatten
    => [x_0..x_{J-1}, y_0.., z_0..] (3J,) float64."""
    poses = list(sal_dict.values()) + list(al_dict.values())
    feats = []
    for pose in poses:
        p = np.array(pose).transpose([1, 0])
        feats.append((p[0:3, :] - p[0:3, root_idx : root_idx + 1]).flatten())
    return np.stack(feats)


def euclidean_expanded(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """sklearn.metrics.pairwise.euclidean_distances for float64 inputs:
    d = -2 x.y^T ; d += |x|^2 ; d += |y|^2 ; max(d, 0) ; sqrt  (that order)."""
    xx = np.einsum("ij,ij->i", x, x)[:, None]
    yy = np.einsum("ij,ij->i", y, y)[None, :]
    d = -2.0 * (x @ y.T)
    d += xx
    d += yy
    np.maximum(d, 0, out=d)
    return np.sqrt(d)


def kcenter_greedy(features: np.ndarray, labeled_idx, n_select: int):
    """utils/coreset.py:49-95 on an explicit feature table.

    Returns (picks, gaps): picked row indices and, per step, top-1 minus top-2 of
    ``min_distances`` (the boundary gap recorded in fixtures so that a parity miss
    can be attributed to a numerical near-tie).  Labeled rows take part in
    min/argmax; nothing is masked; empty labeled set -> np.argmax(None) == 0.
    AssertionError if a labeled row is picked (coreset.py:91).
    """
    features = np.asarray(features, dtype=np.float64)
    labeled_idx = list(labeled_idx)
    min_d = None
    if labeled_idx:
        d = euclidean_expanded(features, features[labeled_idx])
        min_d = np.min(d, axis=1).reshape(-1, 1)
    picks, gaps = [], []
    for _ in range(n_select):
        ind = int(np.argmax(min_d))  # np.argmax(None) == 0
        if min_d is not None:
            top2 = np.partition(min_d.ravel(), -2)[-2:]
            gaps.append(float(top2[1] - top2[0]))
        else:
            gaps.append(float("inf"))
        assert ind not in labeled_idx
        d = euclidean_expanded(features, features[[ind]])
        min_d = d if min_d is None else np.minimum(min_d, d)
        picks.append(ind)
    return picks, gaps


class CoreSet:
    """Call-compatible restatement of utils/coreset.py:13-95 (stdout noise dropped)."""

    def __init__(self, sal_dict, al_dict, joint_root_index, metric="euclidean"):
        assert metric == "euclidean"
        self.sal_dict = OrderedDict(sal_dict)
        self.al_dict = OrderedDict(al_dict)
        self.features = stacked_features(self.sal_dict, self.al_dict, joint_root_index)
        self.sal_keys = list(self.sal_dict.keys())
        self.n_obs = len(sal_dict) + len(al_dict)
        self.al_indices = list(range(len(sal_dict), len(sal_dict) + len(al_dict)))

    def select_batch(self, N, **kwargs):
        picks, self.gaps = kcenter_greedy(self.features, self.al_indices, N)
        return [self.sal_keys[i] for i in picks]
