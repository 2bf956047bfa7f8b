"""CPU: pin the oracle restatement (oracle/*.py) against the golden vectors captured
from the real reference (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

import cases
from oracle import coreset as ocoreset
from oracle import geometry, models, scoring

G = os.path.join(os.path.dirname(__file__), "golden")


def test_reference_own_test_vector():
    proj, hm, valid, stride = cases.reference_test_input()
    z = np.load(os.path.join(G, "triangulation_reftest.npz"))
    r = geometry.triangulation(hm, proj, stride, valid)
    assert r["keypoints_2d"].dtype == np.int64
    np.testing.assert_array_equal(r["keypoints_2d"], z["keypoints_2d"])
    np.testing.assert_array_equal(r["keypoints_2d"][0, 0], [88, 88])
    np.testing.assert_allclose(r["keypoints_3d"], z["keypoints_3d"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(r["keypoints_3d"][0], [-37.22307725, -125.64283665, -23.89420967], atol=1e-7)
    assert abs(r["metric"] - float(z["metric"])) < 1e-12
    assert r["inlier_count"] == int(z["inlier_count"]) == 3


@pytest.mark.parametrize("name", list(cases.triangulation_cases()))
def test_triangulation_synth(name):
    c = cases.triangulation_cases()[name]
    z = np.load(os.path.join(G, "triangulation_synth.npz"))
    hm, proj, valid = cases.build_triangulation_case(c)
    for b in range(hm.shape[0]):
        r = geometry.triangulation(hm[b], proj[b], c["stride"], valid[b])
        np.testing.assert_array_equal(r["keypoints_2d"], z[name + "/keypoints_2d"][b])
        np.testing.assert_allclose(r["keypoints_3d"], z[name + "/keypoints_3d"][b], rtol=0, atol=1e-6)
        np.testing.assert_allclose(r["metric"], z[name + "/metric"][b], rtol=1e-9)
        assert r["inlier_count"] == z[name + "/inlier_count"][b]


def test_nonsquare_argmax_quirk():
    # peak at (row 10, col 7) of a 64x48 map -> (x, y) = (487 % 64, 487 // 64) * stride
    hm = np.zeros((1, 1, 64, 48), dtype=np.float32)
    hm[0, 0, 10, 7] = 1
    kp = geometry.argmax_decode(hm, 4, [True])
    np.testing.assert_array_equal(kp[0, 0], [39 * 4, 7 * 4])


@pytest.mark.parametrize("name", list(cases.xe_cases()))
def test_xe(name):
    c = cases.xe_cases()[name]
    z = np.load(os.path.join(G, "triangulation_xe.npz"))
    hm, proj, valid = cases.build_triangulation_case(c)
    for b in range(hm.shape[0]):
        r = geometry.triangulation(hm[b], proj[b], c["stride"], valid[b], False, True, c["sigma"])
        np.testing.assert_allclose(r["metric"], z[name + "/metric"][b], rtol=1e-9)


@pytest.mark.parametrize("name", list(cases.scoring_cases()))
def test_scoring(name):
    c = cases.scoring_cases()[name]
    z = np.load(os.path.join(G, "scoring.npz"))
    hm, valid = cases.build_scoring_case(c)
    fns = dict(HP=scoring.compute_hp, MPE=scoring.compute_mpe, BSB=scoring.compute_bsb)
    for kind, fn in fns.items():
        for cfg in ("AVG", "STD"):
            want = z[f"{name}/{kind}_{cfg}"]
            dt = str(z[f"{name}/{kind}_{cfg}_dtype"])
            for b in range(hm.shape[0]):
                got = torch.tensor(fn(hm[b], valid[b], cfg))
                assert str(got.dtype) == dt
                # HP: torch's vectorised CPU softmax sums in a different order than numpy
                tol = 2e-6 if kind in ("HP", "BSB") else 0.0
                assert abs(float(got) - want[b]) <= tol * max(1.0, abs(want[b])), (kind, cfg, float(got), want[b])


def test_peak_local_max_known_answers():
    # single interior peak -> one peak, entropy 0
    m = np.zeros((16, 16), dtype=np.float32)
    m[8, 8] = 1
    np.testing.assert_array_equal(scoring.peak_local_max(m, min_distance=2), [[8, 8]])
    assert scoring.compute_mpe(m[None, None], [True]) == 0
    # two equal far-apart peaks -> ln 2
    m[3, 12] = 1
    assert abs(scoring.compute_mpe(m[None, None], [True]) - np.log(2)) < 1e-6
    # border peaks (within min_distance of the edge) are excluded
    b = np.zeros((16, 16), dtype=np.float32)
    b[1, 8] = 1
    assert len(scoring.peak_local_max(b, min_distance=2)) == 0
    # constant image: nothing
    assert len(scoring.peak_local_max(np.ones((8, 8), np.float32), min_distance=2)) == 0
    # plateau of two adjacent equal maxima: the second is rejected (distance 1 < 2) ...
    p = np.zeros((16, 16), dtype=np.float32)
    p[8, 8] = p[8, 9] = 1
    np.testing.assert_array_equal(scoring.peak_local_max(p, min_distance=2), [[8, 8]])
    # ... but equal maxima exactly min_distance apart are both kept
    q = np.zeros((16, 16), dtype=np.float32)
    q[8, 8] = q[8, 10] = 1
    assert len(scoring.peak_local_max(q, min_distance=2)) == 2
    # num_peaks keeps the highest
    r = np.zeros((16, 16), dtype=np.float32)
    r[4, 4], r[10, 10], r[4, 10] = 1, 3, 2
    np.testing.assert_array_equal(scoring.peak_local_max(r, min_distance=2, num_peaks=2), [[10, 10], [4, 10]])


def test_maximum_filter_restatement_vs_scipy():
    """scikit-image's peak_local_max is built on scipy.ndimage.maximum_filter, and scipy IS in the image: the hand-written
    5 x 5 constant-border maximum filter of the restatement equals it bit for bit -- on noise, on plateaus (ties), on maps
    whose maxima sit in the border, on negative maps (where the constant-0 border wins) and on non-square maps.  What
    stays unverified of peak_local_max is skimage's glue around the filter (threshold = image.min(), exclude_border,
    the descending sort and ensure_spacing), covered by known-answer tests only."""
    from scipy import ndimage

    rng = np.random.default_rng(0)
    maps = [rng.standard_normal((64, 64)).astype(np.float32), rng.standard_normal((64, 48)).astype(np.float32),
            rng.standard_normal((96, 72)).astype(np.float32) - 3.0,                       # all negative: the border's zeros win
            np.round(rng.standard_normal((32, 32)) * 2).astype(np.float32),               # plateaus / ties
            np.zeros((16, 16), np.float32), np.full((9, 7), -1.0, np.float32)]
    edge = np.zeros((20, 20), np.float32)
    edge[0, 0] = edge[19, 5] = edge[7, 19] = 5.0
    maps.append(edge)
    for m in maps:
        for r in (1, 2):
            want = ndimage.maximum_filter(m, footprint=np.ones((2 * r + 1, 2 * r + 1)), mode="constant", cval=0.0)
            np.testing.assert_array_equal(scoring._maximum_filter_constant0(m, r), want)


def test_soft_argmax_independent_formulation():
    """The soft-arg-max restatement against a second formulation that shares no code with it (torch.softmax over the
    flattened map, torch.meshgrid pixel grid, float64), plus two properties of the definition: a shifted map shifts the
    expectation by the shift (for mass that stays inside), and a dominant peak pins it.  kornia's own conventions that
    remain unverified: pixel-centre grid origin (0 .. W-1, not 0.5-offset), (x, y) output order, temperature 1."""
    import torch

    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 5, 24, 40)).astype(np.float32) * 3
    got = geometry.spatial_soft_argmax2d(x)
    t = torch.from_numpy(x).double()
    p = torch.softmax(t.reshape(3, 5, -1), dim=-1).reshape(3, 5, 24, 40)
    ys, xs = torch.meshgrid(torch.arange(24, dtype=torch.float64), torch.arange(40, dtype=torch.float64), indexing="ij")
    want = torch.stack([(p * xs).sum((-1, -2)), (p * ys).sum((-1, -2))], dim=-1).numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-4)  # float32 sums over 960 pixels
    # translation: a blob well inside the map, moved by (dx, dy)
    yy, xx = np.mgrid[0:32, 0:32]
    blob = lambda cx, cy: (-((xx - cx) ** 2 + (yy - cy) ** 2) / 4.0).astype(np.float32)[None, None]
    a, b = geometry.spatial_soft_argmax2d(blob(10, 12))[0, 0], geometry.spatial_soft_argmax2d(blob(17, 9))[0, 0]
    np.testing.assert_allclose(b - a, [7, -3], atol=1e-4)
    np.testing.assert_allclose(a, [10, 12], atol=1e-4)


def test_soft_argmax_vs_scipy_center_of_mass():
    """A third-party implementation of the same expectation: scipy.ndimage.center_of_mass of the soft-maxed map is
    (sum y p, sum x p) / sum p -- the soft-arg-max in (row, col) order; kornia returns (x, y) in pixels (unverified: kornia
    itself is nowhere in the image)."""
    from scipy import ndimage
    from scipy.special import softmax

    rng = np.random.default_rng(2)
    x = (rng.standard_normal((2, 3, 20, 28)) * 4).astype(np.float32)
    got = geometry.spatial_soft_argmax2d(x)
    for b in range(2):
        for j in range(3):
            p = softmax(x[b, j].astype(np.float64).ravel()).reshape(20, 28)
            cy, cx = ndimage.center_of_mass(p)
            np.testing.assert_allclose(got[b, j], [cx, cy], rtol=0, atol=2e-4)


def test_soft_argmax_known_answers():
    m = np.full((1, 1, 8, 12), -1e4, dtype=np.float32)
    m[0, 0, 5, 9] = 0
    np.testing.assert_allclose(geometry.spatial_soft_argmax2d(m)[0, 0], [9, 5], atol=1e-6)
    u = np.zeros((1, 1, 8, 12), dtype=np.float32)
    np.testing.assert_allclose(geometry.spatial_soft_argmax2d(u)[0, 0], [5.5, 3.5], atol=1e-5)


@pytest.mark.parametrize("name", list(cases.coreset_cases()))
def test_coreset(name):
    c = cases.coreset_cases()[name]
    z = np.load(os.path.join(G, "coreset.npz"))
    sal, al = cases.build_coreset_case(c)
    cs = ocoreset.CoreSet(sal, al, c["root"])
    keys = cs.select_batch(c["select"])
    idx = {k: i for i, k in enumerate(sal)}
    np.testing.assert_array_equal([idx[k] for k in keys], z[name + "/picks"])
    assert min(z[name + "/gaps"]) > 1e-6  # fixtures are not near-ties


def test_coreset_reference_degenerate_case():
    z = np.load(os.path.join(G, "coreset.npz"))
    sal = {i: [[0, 1, 2] for _ in range(19)] for i in range(20)}
    al = {i: [[0, 1, 2] for _ in range(19)] for i in range(5)}
    picks = ocoreset.CoreSet(sal, al, 2).select_batch(5)
    np.testing.assert_array_equal(picks, z["reftest/picks"])  # duplicates: [0]*5


def _sd(c):
    return {k: torch.from_numpy(v) for k, v in cases.model_state_dict(c).items()}


@pytest.mark.parametrize("name", ["w32", "r50", "w32_small"])
def test_models_eval(name):
    c = cases.model_cases()[name]
    z = np.load(os.path.join(G, "models.npz"))
    x = torch.from_numpy(cases.model_input(c))
    sd = _sd(c)
    with torch.no_grad():
        if c["arch"] == "resnet50":
            y = models.pose_resnet_forward(sd, x)
        else:
            y = models.hrnet_forward(sd, x, models.HRNET_W32)
    y = y.numpy()
    np.testing.assert_allclose(y[0], z[name + "/heatmaps0"], rtol=0, atol=2e-5)
    flat = y.reshape(y.shape[0], y.shape[1], -1)
    np.testing.assert_array_equal(flat.argmax(-1), z[name + "/argmax"])


def test_selection_order():
    d = {"a": 1.0, "b": 3.0, "c": 3.0, "d": float("nan"), "e": 2.0}
    assert scoring.select_top_n(d, 3) == ["b", "c", "e"]


def test_sal_dict_fixture_is_consistent():
    with open(os.path.join(G, "sal_dict.json")) as f:
        z = json.load(f)
    for name, c in cases.sal_cases().items():
        e = z[name]
        assert list(e["al_metric"]) == ["0-0", "1-3", "7-100", "8-103"]
        assert scoring.select_top_n(e["al_metric"], c["select"]) == e["nlargest"]


@pytest.mark.parametrize("name", list(cases.pck_cases()))
def test_pck_restatement_vs_reference_golden(name):
    """3-D PCK / PCKh (utils/evaluation.py:121-195): the oracle's vectorised float32 restatement gives the
    reference's fractions EXACTLY (integer counts), including distances that land on a threshold."""
    from oracle import models

    z = np.load(os.path.join(G, "pck.npz"))
    c = cases.pck_cases()[name]
    pred, gt, valid = cases.pck_arrays(c)
    for tag, thr in (("pck", (1, 2, 3, 4, 5)), ("pck_wide", (10, 25, 50, 100, 150))):
        got = np.asarray([models.compute_3d_pck(pred, gt, valid, t, c["j"]) for t in thr])
        np.testing.assert_array_equal(got, z[f"{name}/{tag}"])
    got = np.asarray([models.compute_3d_pckh(pred, gt, float(t), c["j"]) for t in z[f"{name}/pckh_thresholds"]])
    np.testing.assert_array_equal(got, z[f"{name}/pckh"])


@pytest.mark.parametrize("name", list(cases.preprocess_cases()))
def test_preprocess_restatement_vs_reference_golden(name):
    """prepare_single_view (dataset/dataset.py:158-220) run by the REAL reference on an in-memory PNG
    (tests/golden/preprocess.npz) against the oracle restatement: every field identical -- the LANCZOS
    resize bit for bit, projections and Gaussian heat-maps to the last float."""
    from oracle import preprocess as opp

    z = np.load(os.path.join(G, "preprocess.npz"))
    c = cases.preprocess_cases()[name]
    img, kp3d, cam = cases.preprocess_inputs(c)
    r = opp.prepare_view(img, c["box"], cam, kp3d, c["scale"], c["in_w"], c["in_h"], c["stride"], c["sigma"])
    for k in ("images", "square_box", "2d_after_crop", "proj_matrices", "2d_keypoints", "gt_heatmap"):
        np.testing.assert_array_equal(r[k], z[f"{name}/{k}"], err_msg=k)


def test_lanczos_restatement_vs_pillow():
    """The third-party resampler the reference calls (PIL.Image.resize, LANCZOS) is present here: the
    restatement is pinned against it directly, down- and up-scaling, odd sizes."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image

    from oracle import preprocess as opp

    rng = np.random.default_rng(0)
    for (h, w, ow, oh) in [(300, 300, 256, 256), (123, 97, 64, 48), (80, 80, 96, 96), (517, 333, 128, 96)]:
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(img).resize((ow, oh), resample=Image.LANCZOS))
        np.testing.assert_array_equal(opp.resize_lanczos_u8(img, ow, oh), want)


@pytest.mark.parametrize("name", list(cases.sal_filter_cases()))
def test_sal_filter_restatement_vs_reference_golden(name):
    """Pseudo-label filter of _sal_pseudo_labeling (strategy.py:952-1001), cluster-balanced and random.sample
    variants, against what the reference itself selected."""
    import random

    from oracle import selection

    with open(os.path.join(G, "sal_filter.json")) as f:
        want = json.load(f)[name]
    c = cases.sal_filter_cases()[name]
    sal, done = cases.sal_filter_inputs(c)
    random.seed(c["seed"])
    got = selection.sal_pseudo_label_guids(sal, want["al_guids"], done, c["pseudo_num"], c["thr"], 2,
                                           want["centers"] if c["use_clusters"] else None, c["clusters"])
    assert got == want["sal_guids"]


def test_peak_local_max_vs_real_scikit_image():
    """The peak_local_max restatement against the REAL scikit-image 0.18.3 (tests/golden/peaks_skimage.npz, written by
    tests/golden/make_peaks_golden.py under the build container's Anaconda python3.9), stage by stage on 122 maps: the
    candidate list equals the library's on every map; the spacing pass equals the library's on every map when fed the
    library's own order; the whole result (order included, also with num_peaks=2) equals it on every map whose candidates
    have no equal intensities.  With ties the library's order is numpy's unstable argsort (differs between numpy
    releases); the restatement's row-major tie order still yields the same NUMBER of peaks wherever no two tied
    candidates are closer than the spacing."""
    z = np.load(os.path.join(G, "peaks_skimage.npz"))
    names = [str(v) for v in z["names"]]
    assert str(z["skimage_version"]).startswith("0.18") and len(names) >= 100
    tie_free = 0
    for i, nm in enumerate(names):
        m = z[f"map{i}"]
        cand = scoring.peak_candidates(m, 2)
        np.testing.assert_array_equal(cand, z[f"cand{i}"], err_msg=nm)
        lib = scoring.ensure_spacing(z[f"cand{i}"][z[f"order{i}"]], 2)
        np.testing.assert_array_equal(lib, z[f"full{i}"], err_msg=nm)
        np.testing.assert_array_equal(lib[:2], z[f"top2_{i}"], err_msg=nm)
        vals = m[cand[:, 0], cand[:, 1]]
        if len(np.unique(vals)) == len(vals):
            tie_free += 1
            np.testing.assert_array_equal(scoring.peak_local_max(m, min_distance=2), z[f"full{i}"], err_msg=nm)
            np.testing.assert_array_equal(scoring.peak_local_max(m, min_distance=2, num_peaks=2), z[f"top2_{i}"], err_msg=nm)
    assert tie_free >= 60  # every noise / smooth map and their soft-maxed forms without underflow plateaus

