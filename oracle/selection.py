"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the pseudo-label filter of the reference's
``_sal_pseudo_labeling`` (strategy.py:952-1001); pinned by tests/golden/sal_filter.json, which the real
reference produced."""
from __future__ import annotations

import math
import random

import numpy as np


def nearest_center(feat, centers):
    """``KMeans.predict`` (sklearn, present here): argmin_k ||x - c_k||^2, first minimum."""
    feat, centers = np.asarray(feat, dtype=np.float64), np.asarray(centers, dtype=np.float64)
    d = (centers * centers).sum(1)[None, :] - 2.0 * feat @ centers.T
    return d.argmin(1)


def sal_pseudo_label_guids(sal_dict, al_guids, pseudo_label_guids, pseudo_num_frames, inlier_threshold, root,
                           centers=None, num_clusters=None):
    keep = {
        g: m for g, m in sal_dict["sal_metric"].items()
        if g not in al_guids and not math.isnan(m) and g not in pseudo_label_guids and sal_dict["inlier_count"][g] > inlier_threshold
    }
    guids = sorted(keep, key=keep.get)  # ascending reprojection metric, stable
    if centers is None:
        return random.sample(guids[: 2 * pseudo_num_frames], pseudo_num_frames)
    feats = []
    for g in guids:
        kp = np.array(sal_dict["pred_3d_keypoints"][g]).T
        feats.append((kp[0:3, :] - kp[0:3, root : root + 1]).flatten())
    labels = nearest_center(np.asarray(feats), centers) if guids else []
    counter, per, out = [0] * num_clusters, pseudo_num_frames // num_clusters, []
    for g, c in zip(guids, labels):
        if counter[c] < per:
            counter[c] += 1
            out.append(g)
    return out
