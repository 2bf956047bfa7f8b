"""Core-set (greedy k-center) selection -- drop-in for reference utils/coreset.py:13-95.

The feature table lives in HBM (float64, transposed for coalesced streaming) and each
greedy step is ONE launch (csrc/kcenter.hip); the reference runs sklearn
``pairwise_distances`` + numpy on the CPU, redundantly on every rank.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from .. import _lib


def get_al_dict_for_coreset(labeled_data):
    """``ActiveLearningDataset.get_al_dict_for_coreset`` (dataset/dataset.py:47-51) as a function of the dataset's
    ``labeled_data`` list: the labeled-set side of the k-center problem, ``{index: (J, >=3) float64 pose}`` (each
    record's ``"3d_keypoints"`` is stored (>=3, J)).  Hand the result to ``CoreSet(sal_dict, al_dict, root)`` or
    ``ActiveLearningStrategy.select_al_guids(..., labeled_dict=...)``."""
    return {idx: np.array(labeled_data[idx]["3d_keypoints"]).transpose([1, 0]) for idx in range(len(labeled_data))}


class CoreSet:
    def __init__(self, sal_dict, al_dict, joint_root_index, metric="euclidean", device=None):
        if metric != "euclidean":
            raise NotImplementedError("only the euclidean metric (the reference never passes another)")
        self.sal_dict = OrderedDict(sal_dict)
        self.al_dict = OrderedDict(al_dict)
        # the reference prints list(al_dict.values())[0] (coreset.py:19) -> IndexError when empty
        list(self.sal_dict.values())[0]
        list(self.al_dict.values())[0]
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.features = self._compute_stacked_features(joint_root_index)
        self.sal_keys = list(self.sal_dict.keys())
        self.name = "kcenter"
        self.metric = metric
        self.min_distances = None
        self.max_distances = None
        self.n_obs = len(sal_dict) + len(al_dict)
        self.al_indices = list(range(len(sal_dict), len(sal_dict) + len(al_dict)))
        self.already_selected = []

    @classmethod
    def from_tensors(cls, pool_pose, labeled_pose, joint_root_index, sal_keys=None):
        """Fast path without python dicts: pool_pose (n,J,>=3), labeled_pose (L,J,>=3) HIP or
        host arrays ([joint][coord] rows, the reference's per-pose layout)."""
        self = cls.__new__(cls)
        pool = torch.as_tensor(pool_pose)
        self.device = pool.device if pool.is_cuda else torch.device("cuda", torch.cuda.current_device())
        lab = torch.as_tensor(labeled_pose)
        f_pool = self._features_of(pool, joint_root_index)
        f_lab = self._features_of(lab, joint_root_index) if lab.shape[0] else f_pool[:0]
        self.features = torch.cat([f_pool, f_lab], dim=0).contiguous()
        n, l = pool.shape[0], lab.shape[0]
        self.sal_keys = list(range(n)) if sal_keys is None else list(sal_keys)
        self.name, self.metric = "kcenter", "euclidean"
        self.min_distances = None
        self.max_distances = None
        self.n_obs = n + l
        self.al_indices = list(range(n, n + l))
        self.already_selected = []
        return self

    def _features_of(self, pose, root_idx):
        pose = pose.to(device=self.device, dtype=torch.float64).contiguous()
        n, j, rows = pose.shape
        return _lib.coreset_features(pose, int(root_idx), n, j, rows)

    def _compute_stacked_features(self, root_idx):
        """coreset.py:35-47; pool rows first, labeled rows last."""
        pool = np.asarray(list(self.sal_dict.values()), dtype=np.float64)
        lab = np.asarray([np.asarray(p, dtype=np.float64) for p in self.al_dict.values()])
        f = [self._features_of(torch.from_numpy(pool), root_idx), self._features_of(torch.from_numpy(lab), root_idx)]
        return torch.cat(f, dim=0).contiguous()

    def _run(self, centers, n_select):
        lab = torch.as_tensor(list(centers), dtype=torch.int64, device=self.device) if len(centers) else None
        picks, md = _lib.kcenter_select(self.features, lab, n_select, self.min_distances)
        self.min_distances = md
        return picks

    def update_distances(self, cluster_centers, only_new=True, reset_dist=False):
        """coreset.py:49-69."""
        if reset_dist:
            self.min_distances = None
        if only_new:
            cluster_centers = [d for d in cluster_centers if d not in self.already_selected]
        if cluster_centers:
            self._run(cluster_centers, 0)

    def select_batch(self, N, **kwargs):
        """coreset.py:71-95: init min-distances against the labeled rows, then N greedy
        steps (arg-max, first maximum on ties; distance to the new centre; minimum) -- all on
        device, N+2 launches, one device->host copy of the N picks."""
        already_selected = self.al_indices
        centers = [d for d in already_selected if d not in self.already_selected]
        picks = self._run(centers, int(N)).cpu().numpy().tolist()
        first_labeled = len(self.sal_keys)
        for ind in picks:
            assert ind < first_labeled and ind not in already_selected  # coreset.py:91
        self.already_selected = already_selected
        self.last_picks = picks
        return [self.sal_keys[i] for i in picks]
