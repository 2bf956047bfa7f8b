"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the per-sample uncertainty
scorers (reference strategy.py:1149-1215) and top-N selection (strategy.py:932-949).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import this; the product path never does.

``peak_local_max`` is third-party arithmetic (scikit-image; absent from
/root/reference and from the interpreter the tests run on; the reference's
``indices=True`` keyword pins it to < 0.20).  It is restated here and PINNED
against the real library: the build container carries scikit-image 0.18.3 in its
Anaconda python3.9 tree, ``tests/golden/make_peaks_golden.py`` runs it there and
stores, for 122 maps (noise, quantised, sparse, smooth, plateaus, constant, and
their row-soft-maxed forms), the library's candidate list, the order its argsort
gave them, and its results; ``tests/test_oracle_golden.py`` holds the restatement
to them stage by stage -- candidates equal on all maps, the spacing pass equal on
all maps when fed the library's order, whole results (order included) equal on
every map without tied candidates.  What cannot be pinned is the library's order
among EQUAL intensities: ``np.argsort(-intensities)`` is numpy's unstable
quicksort, so for tied candidates which of two adjacent equal maxima survives
differs between numpy releases; this restatement (and the device kernel) breaks
ties in row-major order.  The arithmetic *around* it (softmax over peak values,
entropy, AVG/STD) is pinned against the real reference by running the reference's
own ``_compute_mpe`` / ``_compute_bsb`` with this function installed as the
``skimage.feature.peak_local_max`` stand-in (oracle/ref_harness.py).
"""
from __future__ import annotations

import heapq
import math

import numpy as np


# --------------------------------------------------------------------------
# scikit-image 0.18/0.19 peak_local_max (2-D, defaults as used by the reference)
# --------------------------------------------------------------------------
def _maximum_filter_constant0(img: np.ndarray, r: int) -> np.ndarray:
    """ndi.maximum_filter(img, footprint=ones((2r+1,2r+1)), mode='constant', cval=0)."""
    h, w = img.shape
    pad = np.zeros((h + 2 * r, w + 2 * r), dtype=img.dtype)
    pad[r : r + h, r : r + w] = img
    out = np.full_like(img, -np.inf)
    for dy in range(2 * r + 1):
        for dx in range(2 * r + 1):
            out = np.maximum(out, pad[dy : dy + h, dx : dx + w])
    return out


def peak_candidates(image, min_distance=1):
    """Candidate maxima in np.nonzero (row-major) order: skimage's _get_peak_mask + _exclude_border for the reference's
    call (threshold_abs = threshold_rel = None -> threshold = image.min(); exclude_border=True -> a border of width
    min_distance removed; footprint ones((2 md + 1,) * 2), maximum filter with constant 0 outside; an all-candidate
    ("trivial") image has none).  Returns (K, 2) int64 (row, col)."""
    image = np.asarray(image)
    r = int(min_distance)
    thr = image.min()
    mx = _maximum_filter_constant0(image, r)
    mask = image == mx
    if np.all(mask):
        mask[:] = False
    mask &= image > thr
    if r > 0:
        mask[:r, :] = False
        mask[-r:, :] = False
        mask[:, :r] = False
        mask[:, -r:] = False
    rows, cols = np.nonzero(mask)
    return np.stack([rows, cols], axis=1).astype(np.int64).reshape(-1, 2)


def ensure_spacing(coord, spacing):
    """skimage._shared.coord.ensure_spacing(coord, spacing, p_norm=inf) on an ORDERED list: walking in that order, a kept
    point rejects every later point at Chebyshev distance < spacing (the library's k-d tree batches give the same set)."""
    coord = np.asarray(coord, dtype=np.int64).reshape(-1, 2)
    keep = []
    rejected = np.zeros(len(coord), dtype=bool)
    for i in range(len(coord)):
        if rejected[i]:
            continue
        keep.append(i)
        if i + 1 < len(coord):
            d = np.max(np.abs(coord[i + 1 :] - coord[i]), axis=1)
            rejected[i + 1 :] |= d < spacing
    return coord[keep]


def peak_local_max(image, min_distance=1, indices=True, num_peaks=np.inf, **unused):
    """skimage.feature.peak_local_max as called at strategy.py:1168-1170,1204-1206: candidates (peak_candidates), sorted by
    descending intensity (the library: np.argsort(-intensities), unstable among ties; here ties keep row-major order),
    thinned by ensure_spacing(min_distance), truncated to num_peaks.  Returns (K, 2) int array of (row, col)."""
    assert indices
    image = np.asarray(image)
    coord = peak_candidates(image, min_distance)
    vals = image[coord[:, 0], coord[:, 1]]
    coord = ensure_spacing(coord[np.argsort(-vals, kind="stable")], int(min_distance))
    if np.isfinite(num_peaks) and len(coord) > num_peaks:
        coord = coord[: int(num_peaks)]
    return coord.astype(np.int64).reshape(-1, 2)


# --------------------------------------------------------------------------
# scorers
# --------------------------------------------------------------------------
def _row_softmax(m: np.ndarray) -> np.ndarray:
    """torch.nn.functional.softmax on a 2-D tensor with implicit dim -> dim=1
    (strategy.py:1185,1202), float32."""
    m = np.asarray(m, dtype=np.float32)
    mx = m.max(axis=1, keepdims=True)
    e = np.exp(m - mx, dtype=np.float32)
    return e / e.sum(axis=1, keepdims=True, dtype=np.float32)


def compute_hps(heatmaps, joint_valid):
    """strategy.py:1178-1187: per (view, valid joint) 1 - max(row_softmax(map))
    as python floats of float32 values."""
    hm = np.asarray(heatmaps, dtype=np.float32)
    valid = np.asarray(joint_valid).astype(bool).reshape(-1)
    out = []
    for v in range(hm.shape[0]):
        for k in range(hm.shape[1]):
            if not valid[k]:
                continue
            p = _row_softmax(hm[v, k])
            out.append(float(np.float32(1) - p.max()))
    return out


def compute_hp(heatmaps, joint_valid, config="AVG"):
    """strategy.py:1188-1193."""
    hps = compute_hps(heatmaps, joint_valid)
    if config == "AVG":
        return sum(hps) / len(hps)
    return np.std(np.array(hps))


def compute_mpes(heatmaps, joint_valid):
    """strategy.py:1160-1176; float32 arithmetic exactly as numpy>=2 evaluates it
    (np.float32 * python float stays float32, python ``sum`` folds left to right)."""
    hm = np.asarray(heatmaps, dtype=np.float32)
    valid = np.asarray(joint_valid).astype(bool).reshape(-1)
    ents = []
    for v in range(hm.shape[0]):
        for k in range(hm.shape[1]):
            if not valid[k]:
                continue
            coords = peak_local_max(hm[v, k], min_distance=2, indices=True)
            peaks = [hm[v, k][c[0]][c[1]] for c in coords]
            probs = np.exp(peaks) / sum(np.exp(peaks))
            ent = sum(-prob * math.log(prob) for prob in probs)
            ents.append(ent)
    return ents


def compute_mpe(heatmaps, joint_valid, config="AVG"):
    """strategy.py:1149-1158."""
    ents = compute_mpes(heatmaps, joint_valid)
    if config == "AVG":
        return sum(ents) / len(ents)
    return np.std(np.array(ents))


def compute_bsbs(heatmaps, joint_valid):
    """strategy.py:1195-1209 (IndexError when a map has fewer than two peaks)."""
    hm = np.asarray(heatmaps, dtype=np.float32)
    valid = np.asarray(joint_valid).astype(bool).reshape(-1)
    out = []
    for v in range(hm.shape[0]):
        for k in range(hm.shape[1]):
            if not valid[k]:
                continue
            p = _row_softmax(hm[v, k])
            coords = peak_local_max(p, min_distance=2, indices=True, num_peaks=2)
            probs = [p[c[0]][c[1]] for c in coords]
            out.append(abs(probs[0] - probs[1]))
    return out


def compute_bsb(heatmaps, joint_valid, config="AVG"):
    """strategy.py:1210-1215."""
    b = compute_bsbs(heatmaps, joint_valid)
    if config == "AVG":
        return sum(b) / len(b)
    return np.std(np.array(b))


def select_top_n(al_metric: dict, n: int):
    """strategy.py:932-949: drop NaNs, heapq.nlargest(n, d, key=d.get) (stable:
    ties keep dict insertion order)."""
    d = {g: m for g, m in al_metric.items() if not math.isnan(m)}
    return heapq.nlargest(n, d, key=d.get)
