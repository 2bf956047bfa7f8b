"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path, called through the
C-ABI via ctypes, against (a) the golden vectors captured from the real reference and
(b) the CPU oracle restatement on the same seeded inputs."""
import json
import math
import os

import numpy as np
import pytest
import torch

import cases
from oracle import coreset as ocoreset
from oracle import geometry, models, scoring

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from multi_view_active_learning_amd import _lib

    _lib.lib()  # fail loudly when the extension is missing
    return torch.device("cuda:0")


# ---------------------------------------------------------------------------------
def test_reference_own_test_vector(dev):
    from multi_view_active_learning_amd.utils.triangulation import triangulation

    proj, hm, valid, stride = cases.reference_test_input()
    z = np.load(os.path.join(G, "triangulation_reftest.npz"))
    r = triangulation(torch.from_numpy(hm).to(dev), torch.from_numpy(proj), stride, torch.from_numpy(valid))
    assert r["keypoints_2d"].dtype == np.int64 and r["keypoints_3d"].dtype == np.float64
    np.testing.assert_array_equal(r["keypoints_2d"], z["keypoints_2d"])
    # tolerance: 1e-6 mm vs the reference's LAPACK SVD (budget 1e-3 mm end to end)
    np.testing.assert_allclose(r["keypoints_3d"], z["keypoints_3d"], rtol=0, atol=1e-6)
    assert abs(r["metric"] - float(z["metric"])) < 1e-9 * float(z["metric"])
    assert r["inlier_count"] == 3 and isinstance(r["inlier_count"], int) and isinstance(r["metric"], float)


@pytest.mark.parametrize("name", list(cases.triangulation_cases()))
def test_triangulation_vs_reference_golden(dev, name):
    from multi_view_active_learning_amd.utils.triangulation import triangulate_batch

    c = cases.triangulation_cases()[name]
    z = np.load(os.path.join(G, "triangulation_synth.npz"))
    hm, proj, valid = cases.build_triangulation_case(c)
    r = triangulate_batch(torch.from_numpy(hm).to(dev), torch.from_numpy(proj), c["stride"], torch.from_numpy(valid))
    np.testing.assert_array_equal(r["keypoints_2d"].cpu().numpy(), z[name + "/keypoints_2d"])
    np.testing.assert_array_equal(r["inlier_count"].cpu().numpy(), z[name + "/inlier_count"])
    # 1e-6 mm absolute (budget 1e-3 mm) + 1e-9 relative: the non-square decode quirk makes some
    # rays near-parallel and puts points at ~1e7 mm, where LAPACK itself is only that accurate
    np.testing.assert_allclose(r["keypoints_3d"].cpu().numpy(), z[name + "/keypoints_3d"], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(r["metric"].cpu().numpy(), z[name + "/metric"], rtol=1e-9)


def test_triangulation_noise_free_known_answer(dev):
    """Analytic KAT: exact projections of known 3-D points triangulate back to them."""
    from multi_view_active_learning_amd import synth
    from multi_view_active_learning_amd import _lib

    v, j, b = 4, 19, 5
    proj = np.stack([synth.ring_cameras(v, 256, 256, seed=s) for s in range(b)])
    x = synth.joints_3d(3, b, j).astype(np.float64)  # (b,3,j)
    kp2 = np.ascontiguousarray(np.stack([synth.project(proj[i], x[i].T) for i in range(b)]), dtype=np.float32)  # (b,v,j,2)
    k3, _, inl, metric, cnt = _lib.triangulate_ransac(
        torch.from_numpy(kp2).to(dev), torch.from_numpy(proj).to(dev), None, b, v, j, 5.0)
    np.testing.assert_allclose(k3.cpu().numpy(), x.transpose(0, 2, 1), rtol=0, atol=5e-2)  # f32 pixel rounding
    assert int(cnt.min()) == v and float(metric.max()) < 1e-2


@pytest.mark.parametrize("name", list(cases.xe_cases()))
def test_xe_vs_reference_golden(dev, name):
    from multi_view_active_learning_amd.utils.triangulation import triangulate_batch

    c = cases.xe_cases()[name]
    z = np.load(os.path.join(G, "triangulation_xe.npz"))
    hm, proj, valid = cases.build_triangulation_case(c)
    r = triangulate_batch(torch.from_numpy(hm).to(dev), torch.from_numpy(proj), c["stride"], torch.from_numpy(valid),
                          False, True, c["sigma"])
    np.testing.assert_allclose(r["metric"].cpu().numpy(), z[name + "/metric"], rtol=1e-9)


def test_argmax_ties_nan_nonsquare_invalid(dev):
    from multi_view_active_learning_amd.utils.evaluation import get_scaled_pred_corrdinates

    hm = np.zeros((2, 3, 64, 48), dtype=np.float32)
    hm[0, 0, 10, 7] = 1.0  # non-square quirk: (487 % 64, 487 // 64)
    hm[0, 1, 5, 5] = hm[0, 1, 30, 2] = 2.0  # tie -> lowest flat index
    hm[1, 0, 20, 20] = np.nan  # NaN is the maximum, like torch.argmax
    hm[1, 0, 40, 1] = 9.0
    hm[1, 2, 63, 47] = 1.0
    got = get_scaled_pred_corrdinates(torch.from_numpy(hm).to(dev), 4, 3, torch.tensor([1, 1, 1]))
    want = geometry.argmax_decode(hm, 4, [True] * 3)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(got[0, 0], [39 * 4, 7 * 4])
    got = get_scaled_pred_corrdinates(torch.from_numpy(hm).to(dev), 4, 3, torch.tensor([1, 0, 1]))
    np.testing.assert_array_equal(got[:, 1], 0)
    # all -inf and constant maps -> index 0
    flat = np.full((1, 2, 8, 8), -np.inf, dtype=np.float32)
    flat[0, 1] = 3.0
    got = get_scaled_pred_corrdinates(torch.from_numpy(flat).to(dev), 4, 2, torch.ones(2))
    np.testing.assert_array_equal(got, 0)


def test_soft_argmax_vs_oracle(dev):
    from multi_view_active_learning_amd import _lib

    rng = np.random.default_rng(5)
    hm = (rng.standard_normal((3, 4, 5, 32, 24)) * 3).astype(np.float32)
    got = _lib.soft_argmax(torch.from_numpy(hm).to(dev), 60, 32, 24, 4.0).cpu().numpy().reshape(3, 4, 5, 2)
    want = geometry.spatial_soft_argmax2d(hm) * np.float32(4)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-4)  # fp32 sums in a different order


# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(cases.scoring_cases()))
def test_scoring_vs_reference_golden(dev, name):
    from multi_view_active_learning_amd.strategy import score_heatmaps_batch

    c = cases.scoring_cases()[name]
    z = np.load(os.path.join(G, "scoring.npz"))
    hm, valid = cases.build_scoring_case(c)
    t = torch.from_numpy(hm).to(dev)
    for kind in ("HP", "MPE", "BSB"):
        for cfg in ("AVG", "STD"):
            out, per_map, n_peaks, _ = score_heatmaps_batch(kind, cfg, t, torch.from_numpy(valid))
            got = out.cpu().numpy()
            want = z[f"{name}/{kind}_{cfg}"]
            # fp32 exp / sum order differ between torch-CPU, numpy and the device: few ulp
            np.testing.assert_allclose(got, want, rtol=3e-6, atol=1e-7, err_msg=f"{kind}_{cfg}")


def test_scoring_per_map_vs_oracle(dev):
    from multi_view_active_learning_amd import _lib

    c = cases.scoring_cases()["noise_nonsq"]
    hm, valid = cases.build_scoring_case(c)
    b, v, j, hh, wh = hm.shape
    t = torch.from_numpy(hm).to(dev)
    allv = np.ones(j, bool)
    for kind, fn in ((_lib.SCORE_HP, scoring.compute_hps), (_lib.SCORE_MPE, scoring.compute_mpes), (_lib.SCORE_BSB, scoring.compute_bsbs)):
        per, cnt = _lib.score_maps(kind, t, b * v * j, hh, wh)
        per = per.cpu().numpy().reshape(b, v * j)
        for bi in range(b):
            want = np.asarray(fn(hm[bi], allv), dtype=np.float64)
            # BSB is a difference of two fp32 probabilities: 2-3 ulp(0.1..1) absolute
            np.testing.assert_allclose(per[bi], want, rtol=3e-6, atol=4e-7 if kind == _lib.SCORE_BSB else 1e-7)
        if kind == _lib.SCORE_MPE:
            cnt = cnt.cpu().numpy().reshape(b, v, j)
            want_n = [[len(scoring.peak_local_max(hm[0, vi, ji], min_distance=2)) for ji in range(j)] for vi in range(v)]
            np.testing.assert_array_equal(cnt[0], want_n)


@pytest.mark.parametrize("hw", [(64, 64), (64, 48), (96, 72)], ids=lambda s: "%dx%d" % s)
def test_mpe_bsb_fuzz_vs_oracle(dev, hw):
    """Seeded fuzz of the device MPE / BSB statistics against the CPU restatement over the inputs where peak_local_max's
    glue decides the result: quantised maps (plateaus and exact ties between peaks), sparse maps (fewer than two peaks,
    peaks in the excluded border), smooth maps (a handful of Gaussian bumps) and plain noise, at the heat-map sizes of
    C2 (64 x 64), C1 (64 x 48) and C4 / C5 (96 x 72).  Peak counts must agree exactly; BSB is compared only where a map
    has two peaks (the reference raises IndexError otherwise, the device returns NaN).  Sparse maps make the row-softmaxed map one wide
    plateau, whose candidate list overflows the first pass: the rescue pass of csrc/scoring.hip must give the exact count."""
    from hypothesis import HealthCheck, given, seed, settings
    from hypothesis import strategies as st
    from multi_view_active_learning_amd import _lib

    hh, wh = hw
    yy, xx = np.mgrid[0:hh, 0:wh]

    def make(kind, s):
        rng = np.random.default_rng(s)
        if kind == 0:  # noise
            return rng.standard_normal((hh, wh)).astype(np.float32)
        if kind == 1:  # plateaus and ties: a few levels only
            return np.round(rng.standard_normal((hh, wh)) * 1.5).astype(np.float32)
        if kind == 2:  # sparse spikes (0, 1 or 2 of them, some in the 2-pixel border), possibly equal
            m = np.zeros((hh, wh), np.float32)
            for _ in range(int(rng.integers(0, 4))):
                m[rng.integers(0, hh), rng.integers(0, wh)] = float(rng.integers(1, 3))
            return m
        m = np.zeros((hh, wh), np.float64)  # smooth bumps, like real heat-maps
        for _ in range(int(rng.integers(1, 5))):
            cy, cx, a = rng.uniform(0, hh), rng.uniform(0, wh), rng.uniform(0.2, 1.0)
            m += a * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * rng.uniform(1.0, 3.0) ** 2))
        return m.astype(np.float32)

    @seed(1234)
    @settings(max_examples=12, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(st.lists(st.tuples(st.integers(0, 3), st.integers(0, 2**20)), min_size=8, max_size=8))
    def run(specs):
        maps = np.stack([make(k, s) for k, s in specs])
        t = torch.from_numpy(maps).to(dev)
        mpe, cnt = _lib.score_maps(_lib.SCORE_MPE, t, len(maps), hh, wh)
        bsb, cnt2 = _lib.score_maps(_lib.SCORE_BSB, t, len(maps), hh, wh)
        mpe, cnt, bsb, cnt2 = mpe.cpu().numpy(), cnt.cpu().numpy(), bsb.cpu().numpy(), cnt2.cpu().numpy()
        for i, m in enumerate(maps):
            peaks = scoring.peak_local_max(m, min_distance=2)
            assert cnt[i] == len(peaks), (specs[i], cnt[i], len(peaks))
            want = scoring.compute_mpes(m[None, None], [True])[0] if len(peaks) else 0.0
            np.testing.assert_allclose(mpe[i], want, rtol=3e-6, atol=2e-7, err_msg=str(specs[i]))
            p = scoring._row_softmax(m)
            pk = scoring.peak_local_max(p, min_distance=2, num_peaks=2)
            # BSB's count is of ALL peaks of the row-softmaxed map.  Where that map has plateaus (everything but noise), which
            # pixels tie depends on the last bit of exp / sum, which differs between numpy and the device: only "are there
            # two peaks" -- the one thing the reference's result depends on -- is compared there.
            n_all = len(scoring.peak_local_max(p, min_distance=2))
            if specs[i][0] == 0:
                assert cnt2[i] == n_all, (specs[i], cnt2[i], n_all)
            assert (cnt2[i] >= 2) == (len(pk) == 2), (specs[i], cnt2[i], n_all)
            if len(pk) == 2:
                np.testing.assert_allclose(bsb[i], abs(p[pk[0][0], pk[0][1]] - p[pk[1][0], pk[1][1]]), rtol=3e-6, atol=4e-7, err_msg=str(specs[i]))

    run()


def test_mpe_bsb_vs_real_scikit_image_goldens(dev):
    """The device MPE / BSB statistics against peaks found by the REAL scikit-image 0.18.3 (tests/golden/peaks_skimage.npz):
    on every golden map whose candidates have no equal intensities (with ties the library's own order is numpy's unstable
    argsort) the device's peak count equals the library's and its entropy equals the reference's formula over the library's
    peaks; BSB likewise from the library's top two peaks of the row-soft-maxed map."""
    from multi_view_active_learning_amd import _lib

    z = np.load(os.path.join(G, "peaks_skimage.npz"))
    names = [str(v) for v in z["names"]]
    idx = {nm: i for i, nm in enumerate(names)}

    def tie_free(i):
        m, c = z[f"map{i}"], z[f"cand{i}"]
        v = m[c[:, 0], c[:, 1]]
        return len(np.unique(v)) == len(v)

    by_shape = {}
    for i, nm in enumerate(names):
        if nm.endswith("_rowsoftmax") or not tie_free(i):
            continue
        by_shape.setdefault(z[f"map{i}"].shape, []).append(i)
    checked_mpe = checked_bsb = 0
    for (hh, wh), ids in by_shape.items():
        if hh < 8 or wh < 8:
            continue
        maps = np.stack([z[f"map{i}"] for i in ids])
        t = torch.from_numpy(maps).to(dev)
        mpe, cnt = _lib.score_maps(_lib.SCORE_MPE, t, len(ids), hh, wh)
        bsb, cnt2 = _lib.score_maps(_lib.SCORE_BSB, t, len(ids), hh, wh)
        mpe, cnt, bsb, cnt2 = mpe.cpu().numpy(), cnt.cpu().numpy(), bsb.cpu().numpy(), cnt2.cpu().numpy()
        for k, i in enumerate(ids):
            m, full = z[f"map{i}"], z[f"full{i}"]
            assert cnt[k] == len(full), (names[i], cnt[k], len(full))
            if len(full):  # strategy.py:1171-1175 over the library's peaks, in its order
                peaks = [m[r][c] for r, c in full]
                probs = np.exp(peaks) / sum(np.exp(peaks))
                want = sum(-p * math.log(p) for p in probs)
                np.testing.assert_allclose(mpe[k], want, rtol=3e-6, atol=2e-7, err_msg=names[i])
                checked_mpe += 1
            j = idx[names[i] + "_rowsoftmax"]
            if tie_free(j) and len(z[f"top2_{j}"]) == 2:  # strategy.py:1202-1208 over the library's two highest peaks
                p, (a, b) = z[f"map{j}"], z[f"top2_{j}"]
                np.testing.assert_allclose(bsb[k], abs(p[a[0]][a[1]] - p[b[0]][b[1]]), rtol=3e-5, atol=4e-7, err_msg=names[i])
                assert cnt2[k] >= 2
                checked_bsb += 1
    assert checked_mpe >= 20 and checked_bsb >= 10, (checked_mpe, checked_bsb)


def test_mpe_candidate_list_sizes_and_tie_paths(dev):
    """csrc/scoring.hip sorts lists of at most 256 candidates by rank, longer ones with the bitonic network, and redoes maps with
    more than 512 in the rescue pass; equal values far apart must not start the spacing pass, an equal ADJACENT pixel that is no
    candidate must start it and reject nothing.  Constructed 96 x 72 maps (peaks on a lattice, distinct values unless stated)
    against the CPU restatement: counts exact, entropies within the float32 tolerance of the other MPE tests."""
    from multi_view_active_learning_amd import _lib

    hh, wh = 96, 72
    rng = np.random.default_rng(5)

    def lattice(step):
        m = np.full((hh, wh), -1.0, np.float32)
        ys, xs = np.arange(2, hh - 2, step), np.arange(2, wh - 2, step)
        vals = rng.permutation(len(ys) * len(xs)).astype(np.float32) / (len(ys) * len(xs)) + 0.25
        m[np.ix_(ys, xs)] = vals.reshape(len(ys), len(xs))
        return m, len(ys) * len(xs)

    maps, want_n = [], []
    for step in (3, 4, 6):  # 713 (rescue pass), 391 (bitonic, first pass), 192 (rank sort) peaks
        m, n = lattice(step)
        maps.append(m); want_n.append(n)
    m, n = lattice(6)  # two pairs of equal values far apart: ties, nothing adjacent
    m[8, 8] = m[50, 44]
    m[14, 20] = m[62, 32]
    maps.append(m); want_n.append(n)
    m, n = lattice(6)  # as above, and one peak with an equal neighbour that a larger value three pixels away keeps from being a candidate
    m[8, 8] = m[50, 44]
    m[20, 21] = m[20, 20]
    m[20, 23] = 4.0
    maps.append(m); want_n.append(n + 1)
    m, n = lattice(6)  # a two-pixel plateau: one of the two survives
    m[8, 8] = m[50, 44]
    m[20, 21] = m[20, 20]
    maps.append(m); want_n.append(n)
    assert want_n[0] > 512 and 256 < want_n[1] <= 512 and want_n[2] <= 256
    maps = np.stack(maps)
    mpe, cnt = _lib.score_maps(_lib.SCORE_MPE, torch.from_numpy(maps).to(dev), len(maps), hh, wh)
    mpe, cnt = mpe.cpu().numpy(), cnt.cpu().numpy()
    for i, m in enumerate(maps):
        peaks = scoring.peak_local_max(m, min_distance=2)
        assert len(peaks) == want_n[i], (i, len(peaks), want_n[i])
        assert cnt[i] == want_n[i], (i, cnt[i], want_n[i])
        np.testing.assert_allclose(mpe[i], scoring.compute_mpes(m[None, None], [True])[0], rtol=3e-6, atol=2e-7, err_msg=str(i))


def test_peak_known_answers_on_device(dev):
    from multi_view_active_learning_amd import _lib

    maps = np.zeros((5, 16, 16), dtype=np.float32)
    maps[0, 8, 8] = 1  # single peak -> H = 0
    maps[1, 8, 8] = maps[1, 3, 12] = 1  # two equal peaks -> ln 2
    maps[2, 1, 8] = 1  # border peak excluded -> no peaks -> 0
    maps[3, 8, 8] = maps[3, 8, 9] = 1  # adjacent plateau -> one peak
    maps[4, 8, 8] = maps[4, 8, 10] = 1  # plateau at distance 2 -> both kept
    per, cnt = _lib.score_maps(_lib.SCORE_MPE, torch.from_numpy(maps).to(dev), 5, 16, 16)
    np.testing.assert_array_equal(cnt.cpu().numpy(), [1, 2, 0, 1, 2])
    np.testing.assert_allclose(per.cpu().numpy(), [0, np.log(2), 0, 0, np.log(2)], atol=1e-6)
    hp, _ = _lib.score_maps(_lib.SCORE_HP, torch.zeros((1, 64, 64), device=dev), 1, 64, 64)
    assert abs(float(hp[0]) - (1 - 1 / 64)) < 1e-7  # row-wise softmax


# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(cases.coreset_cases()))
def test_coreset_vs_reference_golden(dev, name):
    from multi_view_active_learning_amd.utils.coreset import CoreSet

    c = cases.coreset_cases()[name]
    z = np.load(os.path.join(G, "coreset.npz"))
    sal, al = cases.build_coreset_case(c)
    cs = CoreSet(sal, al, c["root"])
    keys = cs.select_batch(c["select"])
    np.testing.assert_array_equal(cs.last_picks, z[name + "/picks"])  # bit-exact selected indices
    assert keys == [list(sal)[i] for i in z[name + "/picks"]]
    md = cs.min_distances.cpu().numpy()
    # distances agree to 1e-9 relative; the self-distance of a picked row is sqrt of pure
    # rounding noise (0 here, up to ~2e-4 from BLAS in the reference), hence the absolute term
    np.testing.assert_allclose(md[:: max(1, cs.n_obs // 64)], z[name + "/final_min_distances"], rtol=1e-9, atol=1e-3)


def test_coreset_degenerate_duplicates_and_tensor_path(dev):
    from multi_view_active_learning_amd.utils.coreset import CoreSet

    sal = {i: [[0, 1, 2] for _ in range(19)] for i in range(20)}
    al = {i: [[0, 1, 2] for _ in range(19)] for i in range(5)}
    assert CoreSet(sal, al, 2).select_batch(5) == [0, 0, 0, 0, 0]  # reference tests/test_coreset.py
    with pytest.raises(IndexError):
        CoreSet(sal, {}, 2)
    c = cases.coreset_cases()["n1000_l200_j42"]
    pool, lab = cases.coreset_arrays(c)
    z = np.load(os.path.join(G, "coreset.npz"))
    cs = CoreSet.from_tensors(torch.from_numpy(pool).to(dev), torch.from_numpy(lab).to(dev), c["root"])
    np.testing.assert_array_equal(cs.select_batch(c["select"]), z["n1000_l200_j42/picks"])
    # continuation: a second call keeps the running minimum (coreset.py:59-62)
    more = cs.select_batch(3)
    o = ocoreset.kcenter_greedy(cs.features.cpu().numpy(), cs.al_indices, c["select"] + 3)[0]
    assert more == o[-3:]


# ---------------------------------------------------------------------------------
def test_masked_mse_and_mkpe_vs_oracle(dev):
    from multi_view_active_learning_amd.pose_estimators.loss import Pose2DMeanSquaredError
    from multi_view_active_learning_amd.utils.evaluation import compute_mkpe, mkpe_per_sample

    rng = np.random.default_rng(7)
    h = rng.standard_normal((6, 19, 64, 48)).astype(np.float32)
    g = rng.standard_normal((6, 19, 64, 48)).astype(np.float32)
    valid = rng.uniform(size=(6, 19, 1, 1)) > 0.3
    ht = torch.from_numpy(h).to(dev).requires_grad_(True)
    loss = Pose2DMeanSquaredError().pose_2d_mse(ht, torch.from_numpy(g).to(dev), torch.from_numpy(valid).to(dev))
    loss.backward()
    hc = torch.from_numpy(h).requires_grad_(True)
    want = models.pose_2d_mse(hc, torch.from_numpy(g), torch.from_numpy(valid))
    want.backward()
    assert abs(loss.item() - want.item()) <= 2e-6 * abs(want.item())
    np.testing.assert_allclose(ht.grad.cpu().numpy(), hc.grad.numpy(), rtol=1e-6, atol=1e-12)
    l1 = Pose2DMeanSquaredError().pose_2d_mse_single_batch(torch.from_numpy(h[0, :1]).to(dev), torch.from_numpy(g[0, :1]).to(dev))
    assert abs(l1.item() - float(((h[0, :1] - g[0, :1]) ** 2).sum() / (64 * 48))) < 1e-4
    # MKPE
    s, j = 7, 19
    pred = rng.standard_normal((s, j, 3)).astype(np.float32) * 100
    gt = rng.standard_normal((s, 4, j)).astype(np.float32) * 100
    vj = (rng.uniform(size=(s, j)) > 0.2).astype(np.float32)
    vj[:, 0] = 1
    got = compute_mkpe([torch.from_numpy(p).to(dev) for p in pred], [torch.from_numpy(x).to(dev) for x in gt],
                       [torch.from_numpy(x).to(dev) for x in vj])
    want = models.compute_mkpe([torch.from_numpy(p) for p in pred], [torch.from_numpy(x) for x in gt], [torch.from_numpy(x) for x in vj])
    assert abs(got.item() - want.item()) <= 2e-6 * abs(want.item()) or (np.isnan(got.item()) and np.isnan(want.item()))
    per = mkpe_per_sample(torch.from_numpy(pred).to(dev), torch.from_numpy(gt).to(dev), torch.from_numpy(vj).to(dev)).cpu().numpy()
    for i in range(s):
        w = models.compute_mkpe([torch.from_numpy(pred[i])], [torch.from_numpy(gt[i])], [torch.from_numpy(vj[i])]).item()
        assert (np.isnan(per[i]) and np.isnan(w)) or abs(per[i] - w) <= 2e-6 * abs(w)


# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(cases.sal_cases()))
def test_sal_dict_vs_reference_golden(dev, name):
    """_compute_sal_dict + selection: same five dicts, same key order, same picks as the
    reference run captured in tests/golden/sal_dict.json."""
    from multi_view_active_learning_amd.config import get_default_configs
    from multi_view_active_learning_amd.strategy import ActiveLearningStrategy

    with open(os.path.join(G, "sal_dict.json")) as f:
        want = json.load(f)[name]
    c = cases.sal_cases()[name]
    cfg = get_default_configs()
    cfg.AL.STRATEGY = c["strategy"]
    cfg.POSE_ESTIMATOR.STRIDE = c["stride"]
    cfg.AL.USE_REPROJECTION_XE = c.get("xe", False)
    cfg.AL.REPROJECTION_SIGMA = c.get("sigma", 1.0)
    loader, heatmaps = cases.build_sal_loader(c)
    it = iter(heatmaps)

    def fake_model(images):
        return torch.from_numpy(next(it)).to(dev)

    tl = [{k: torch.from_numpy(v) for k, v in dp.items()} for dp in loader]
    sal = ActiveLearningStrategy(cfg)._compute_sal_dict(tl, fake_model)
    for field in ("al_metric", "sal_metric", "inlier_count", "mkpe", "pred_3d_keypoints"):
        assert list(sal[field]) == list(want[field]), field  # key order = gather order
    for g in want["al_metric"]:
        tol = 0 if c["strategy"] == "CORESET" else 3e-6
        assert abs(sal["al_metric"][g] - want["al_metric"][g]) <= tol * abs(want["al_metric"][g]) + 1e-12, (g, sal["al_metric"][g], want["al_metric"][g])
        assert abs(sal["sal_metric"][g] - want["sal_metric"][g]) <= 1e-6 * abs(want["sal_metric"][g])
        assert sal["inlier_count"][g] == want["inlier_count"][g]
        a, b_ = sal["mkpe"][g], want["mkpe"][g]
        assert (np.isnan(a) and np.isnan(b_)) or abs(a - b_) <= 1e-5 * abs(b_)
        # fp32-rounded keypoints: 1e-3 mm budget, typically identical
        np.testing.assert_allclose(sal["pred_3d_keypoints"][g], want["pred_3d_keypoints"][g], rtol=0, atol=1e-3)
    st = ActiveLearningStrategy(cfg)
    if c["strategy"] != "CORESET":
        assert st.select_al_guids(sal, c["select"]) == want["nlargest"]


@pytest.mark.parametrize("name", list(cases.pck_cases()))
def test_pck3d_vs_reference_golden(dev, name):
    """3-D PCK / PCKh counters on the device against the reference's golden fractions: exact equality (the
    kernel rounds every float32 operation like torch does; counts are integers)."""
    from multi_view_active_learning_amd.utils import evaluation

    z = np.load(os.path.join(G, "pck.npz"))
    c = cases.pck_cases()[name]
    pred, gt, valid = cases.pck_arrays(c)
    pl = [torch.from_numpy(p).to(dev) for p in pred]
    gl = [torch.from_numpy(g).to(dev) for g in gt]
    vl = [torch.from_numpy(v).to(dev) for v in valid]
    for tag, thr in (("pck", (1, 2, 3, 4, 5)), ("pck_wide", (10, 25, 50, 100, 150))):
        t, got = evaluation.compute_3d_pck_figure(pl, gl, vl, c["j"], thresholds=thr)
        assert t == thr
        np.testing.assert_array_equal(np.asarray(got), z[f"{name}/{tag}"])
    t, got = evaluation.compute_3d_pckh_figure(pl, gl, c["j"])
    np.testing.assert_array_equal(np.asarray(got), z[f"{name}/pckh"])
    # single-threshold entry points
    assert evaluation.compute_3d_pck(pl, gl, vl, 3, c["j"]) == list(z[f"{name}/pck"][2])
    assert evaluation.compute_3d_pckh(pl, gl, 0.5, c["j"]) == list(z[f"{name}/pckh"][4])
    # a joint that is never valid: the reference divides by zero
    v0 = [v.clone() for v in vl]
    for v in v0:
        v[1] = 0
    with pytest.raises(ZeroDivisionError):
        evaluation.compute_3d_pck(pl, gl, v0, 3, c["j"])


def test_evaluate_all_vs_oracle(dev):
    """_evaluate_all's result dict (MKPE + 3-D PCK at 1..5 mm + PCKh, strategy.py:584-649) over a fake loader
    against the oracle chain: triangulation restatement -> float32 rounding -> metric restatements."""
    from multi_view_active_learning_amd.config import get_default_configs
    from multi_view_active_learning_amd.strategy import ActiveLearningStrategy

    name = next(iter(cases.sal_cases()))
    c = cases.sal_cases()[name]
    cfg = get_default_configs()
    cfg.POSE_ESTIMATOR.STRIDE = c["stride"]
    loader, heatmaps = cases.build_sal_loader(c)
    it = iter(heatmaps)
    tl = [{k: torch.from_numpy(v) for k, v in dp.items()} for dp in loader]
    got = ActiveLearningStrategy(cfg).evaluate_all(tl, lambda images: torch.from_numpy(next(it)).to(dev))
    preds, gts, valids = [], [], []
    for dp, hm in zip(loader, heatmaps):
        b = dp["proj_matrices"].shape[0]
        hm = hm.reshape((b, -1) + hm.shape[1:])
        for i in range(b):
            r = geometry.triangulation(hm[i], dp["proj_matrices"][i], c["stride"], dp["joint_valid"][i].astype(bool))
            preds.append(torch.from_numpy(r["keypoints_3d"].astype(np.float32)))
            gts.append(torch.from_numpy(dp["3d_keypoints"][i].astype(np.float32)))
            valids.append(torch.from_numpy(dp["joint_valid"][i].astype(np.float32)))
    j = preds[0].shape[0]
    want_mkpe = models.compute_mkpe(preds, gts, valids).item()
    assert abs(got["mkpe"] - want_mkpe) <= 1e-5 * abs(want_mkpe)
    assert got["thresholds"] == (1, 2, 3, 4, 5)
    for t, row in zip(got["thresholds"], got["pcks"]):
        assert row == models.compute_3d_pck(preds, gts, valids, t, j)
    for t, row in zip(got["pckh_thresholds"], got["pckh_pcks"]):
        assert row == models.compute_3d_pckh(preds, gts, t, j)


@pytest.mark.parametrize("name", list(cases.preprocess_cases()))
def test_prepare_views_vs_reference_golden(dev, name):
    """Device input pipeline against the reference's prepare_single_view outputs: the LANCZOS-resized,
    normalised image bit for bit; host-side camera fields identical; Gaussian heat-maps within one float32
    ulp (device exp vs torch's)."""
    from multi_view_active_learning_amd.utils import preprocess

    z = np.load(os.path.join(G, "preprocess.npz"))
    c = cases.preprocess_cases()[name]
    img, kp3d, cam = cases.preprocess_inputs(c)
    r = preprocess.prepare_views([torch.from_numpy(img).to(dev)], [c["box"]], [cam], kp3d, c["scale"], c["in_w"], c["in_h"],
                                 c["stride"], c["sigma"])
    np.testing.assert_array_equal(r["images"][0].cpu().numpy(), z[f"{name}/images"])
    for k in ("square_box", "2d_after_crop", "proj_matrices", "2d_keypoints"):
        np.testing.assert_array_equal(r[k][0], z[f"{name}/{k}"], err_msg=k)
    np.testing.assert_allclose(r["gt_heatmap"][0].cpu().numpy(), z[f"{name}/gt_heatmap"], rtol=1.2e-7, atol=1e-45)


def test_prepare_views_batch_and_oracle(dev):
    """Several views with different raw sizes and boxes in ONE call, against the oracle restatement (larger
    sizes than the fixtures: 640 x 480 -> 256 x 256, a box that leaves the image on two sides)."""
    from oracle import preprocess as opp
    from multi_view_active_learning_amd.utils import preprocess

    rng = np.random.default_rng(9)
    imgs, boxes = [], []
    for (h0, w0, box) in [(480, 640, (100, 50, 500, 430)), (360, 360, (-40, -30, 250, 300)), (200, 300, (10, 20, 120, 140))]:
        imgs.append(rng.integers(0, 256, size=(h0, w0, 3), dtype=np.uint8))
        boxes.append(opp.scale_bbox(opp.get_square_bbox(box), 1.1))
    got = preprocess.resize_views([torch.from_numpy(i).to(dev) for i in imgs], boxes, 256, 256).cpu().numpy()
    for i, (im, b) in enumerate(zip(imgs, boxes)):
        crop = opp.crop_zero_fill(im[..., ::-1], b)
        want = (opp.resize_lanczos_u8(crop, 256, 256) / 255.0 - opp.IMAGENET_MEAN) / opp.IMAGENET_STD
        np.testing.assert_array_equal(got[i], want.transpose(2, 0, 1).astype(np.float32))


@pytest.mark.parametrize("name", list(cases.sal_filter_cases()))
def test_sal_filter_vs_reference_golden(dev, name):
    """select_al_guids + select_sal_guids against the reference's _sal_pseudo_labeling: same AL picks, same
    pseudo-labelled guids in the same order (cluster assignment on the device)."""
    import random

    from multi_view_active_learning_amd.config import get_default_configs
    from multi_view_active_learning_amd.strategy import ActiveLearningStrategy

    with open(os.path.join(G, "sal_filter.json")) as f:
        want = json.load(f)[name]
    c = cases.sal_filter_cases()[name]
    sal, done = cases.sal_filter_inputs(c)
    cfg = get_default_configs()
    cfg.AL.STRATEGY = "HP"
    cfg.SAL.INLIER_THRESHOLD = c["thr"]
    cfg.SAL.NUM_CLUSTERS = c["clusters"]
    st = ActiveLearningStrategy(cfg)
    al = st.select_al_guids(sal, c["al_num"])
    assert al == want["al_guids"]
    random.seed(c["seed"])
    got = st.select_sal_guids(sal, al, done, c["pseudo_num"], want["centers"] if c["use_clusters"] else None, device=dev)
    assert got == want["sal_guids"]


@pytest.mark.parametrize("shape", [(2, 4, 19, 64, 64), (1, 8, 19, 96, 72), (3, 2, 5, 17, 23)], ids=lambda s: "x".join(map(str, s)))
def test_fused_score_decode_equals_separate_passes(dev, shape):
    """mval_score_decode_maps (ONE staged read per heat-map: statistic + hard arg-max) against mval_score_maps and
    mval_argmax_decode (two reads): identical bits, including ties, NaN maps, invalid joints and the reference's
    non-square index split."""
    from multi_view_active_learning_amd import _lib

    b, v, j, hh, wh = shape
    rng = np.random.default_rng(hh * 7 + wh)
    hm = rng.standard_normal(shape).astype(np.float32) * 0.3
    hm[0, 0, 0] = 0.25                      # a constant map: all ties -> index 0, no peak above the minimum
    hm[0, 1, 1, 3, 4] = np.nan              # NaN is the arg-max
    hm[-1, -1, 2, 5:7, 6:8] = 9.0           # a 2 x 2 plateau: first index wins, plateau spacing in the peak list
    valid = np.ones((b, j), dtype=np.uint8)
    valid[0, 3] = 0
    t = torch.from_numpy(hm).to(dev)
    vd = torch.from_numpy(valid).to(dev)
    for kind in (_lib.SCORE_HP, _lib.SCORE_MPE, _lib.SCORE_BSB):
        for split in (hh, wh):
            stat, cnt, kp = _lib.score_decode_maps(kind, t, vd, b, v, j, hh, wh, 4, split)
            stat0, cnt0 = _lib.score_maps(kind, t, b * v * j, hh, wh)
            kp0 = _lib.argmax_decode(t, vd, b, v, j, hh, wh, 4, split)
            assert torch.equal(kp, kp0)
            assert torch.equal(cnt, cnt0)
            np.testing.assert_array_equal(stat.cpu().numpy(), stat0.cpu().numpy())  # (NaN == NaN here)
    assert kp[0, :, 3].abs().sum().item() == 0  # invalid joint -> (0, 0)
