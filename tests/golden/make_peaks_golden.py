#!/opt/conda/bin/python3.9
"""Golden vectors of skimage.feature.peak_local_max from the REAL scikit-image 0.18.3 (the release line the reference's
`indices=True` calls at strategy.py:1168-1170, 1204-1206 imply), which this build container carries in its Anaconda
python3.9 tree (/opt/conda).  Run in the build container only:

    /opt/conda/bin/python3.9 tests/golden/make_peaks_golden.py

Writes tests/golden/peaks_skimage.npz: the input maps (float32) and, for each,
  cand   the candidate coordinates in np.nonzero order after the library's _get_peak_mask + _exclude_border,
  order  the permutation its np.argsort(-intensities) applied to them (numpy's default quicksort is NOT stable: among equal
         intensities the order -- and with it which of two adjacent equal maxima survives -- depends on the numpy build; the
         stored order is numpy 1.26's),
  full   peak_local_max(m, min_distance=2, indices=True),  top2: the same with num_peaks=2, in the order returned.  Inputs where the library's own glue decides the result: noise, quantised maps (ties, plateaus),
sparse spikes (also inside the excluded border), smooth bumps, constant maps, and the row-soft-maxed form of each
(what _compute_bsb feeds it).  No scikit-image code is copied: only its outputs are stored."""
import os
import sys

import numpy as np
import skimage
from skimage.feature import peak_local_max
from skimage.feature.peak import _exclude_border, _get_excluded_border_width, _get_peak_mask, _get_threshold

assert skimage.__version__.startswith("0.18"), skimage.__version__
HERE = os.path.dirname(os.path.abspath(__file__))


def row_softmax(m):
    m = m.astype(np.float32)
    e = np.exp(m - m.max(axis=1, keepdims=True), dtype=np.float32)
    return (e / e.sum(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32)


def make(kind, hh, wh, rng):
    yy, xx = np.mgrid[0:hh, 0:wh]
    if kind == "noise":
        return rng.standard_normal((hh, wh)).astype(np.float32)
    if kind == "quantised":
        return np.round(rng.standard_normal((hh, wh)) * 1.5).astype(np.float32)
    if kind == "sparse":
        m = np.zeros((hh, wh), np.float32)
        for _ in range(int(rng.integers(1, 6))):
            m[rng.integers(0, hh), rng.integers(0, wh)] = float(rng.integers(1, 3))
        return m
    if kind == "smooth":
        m = np.zeros((hh, wh), np.float64)
        for _ in range(int(rng.integers(1, 5))):
            cy, cx, a = rng.uniform(0, hh), rng.uniform(0, wh), rng.uniform(0.2, 1.0)
            m += a * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * rng.uniform(1.0, 3.0) ** 2))
        return m.astype(np.float32)
    if kind == "plateau":  # flat-topped blobs: whole regions of equal maxima, adjacent equal peaks
        m = np.zeros((hh, wh), np.float32)
        for _ in range(int(rng.integers(1, 4))):
            y0, x0 = int(rng.integers(0, hh - 3)), int(rng.integers(0, wh - 3))
            m[y0 : y0 + int(rng.integers(1, 5)), x0 : x0 + int(rng.integers(1, 6))] = float(rng.integers(1, 3))
        return m
    if kind == "constant":
        return np.full((hh, wh), 0.25, np.float32)
    raise ValueError(kind)


maps, names = [], []
rng = np.random.default_rng(20211130)
for (hh, wh), seeds in (((64, 64), 1), ((64, 48), 1), ((96, 72), 1), ((16, 20), 3), ((24, 32), 3), ((6, 7), 2)):
    for kind in ("noise", "quantised", "sparse", "smooth", "plateau", "constant"):
        for s in range(seeds if kind != "constant" else 1):
            m = make(kind, hh, wh, rng)
            maps.append(m)
            names.append(f"{kind}_{hh}x{wh}_{s}")
            maps.append(row_softmax(m))
            names.append(f"{kind}_{hh}x{wh}_{s}_rowsoftmax")

out = {"names": np.array(names), "skimage_version": np.array(skimage.__version__), "numpy_version": np.array(np.__version__)}
for i, m in enumerate(maps):
    full = np.asarray(peak_local_max(m, min_distance=2, indices=True)).reshape(-1, 2).astype(np.int32)
    top2 = np.asarray(peak_local_max(m, min_distance=2, indices=True, num_peaks=2)).reshape(-1, 2).astype(np.int32)
    mask = _get_peak_mask(m, np.ones((5, 5), dtype=bool), _get_threshold(m, None, None))
    mask = _exclude_border(mask, _get_excluded_border_width(m, 2, True))
    coord = np.nonzero(mask)
    out[f"map{i}"] = m
    out[f"cand{i}"] = np.transpose(coord).astype(np.int32).reshape(-1, 2)
    out[f"order{i}"] = np.argsort(-m[coord]).astype(np.int32)
    out[f"full{i}"] = full
    out[f"top2_{i}"] = top2
np.savez_compressed(os.path.join(HERE, "peaks_skimage.npz"), **out)
print(f"{len(maps)} maps -> peaks_skimage.npz ({os.path.getsize(os.path.join(HERE, 'peaks_skimage.npz')) / 1024:.0f} KiB), scikit-image {skimage.__version__}")
