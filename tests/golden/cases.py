"""Shared definitions of the golden-vector cases.

Both ``make_golden.py`` (runs the real reference, build container only) and the test
suite (runs the oracle restatement / the HIP path) build their INPUTS from these
seeded definitions, so the committed fixtures hold expected outputs only.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np

from multi_view_active_learning_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------
# triangulation
# --------------------------------------------------------------------------
def reference_test_input():
    """Input of the reference's own test (tests/test_triangulation.py:15-69), captured
    as DATA by make_golden.capture_reference_test_input() into reftest_input.npz."""
    z = np.load(os.path.join(HERE, "reftest_input.npz"))
    return z["proj"], z["heatmaps"], z["valid"], int(z["stride"])


def triangulation_cases():
    return OrderedDict(
        v4_square=dict(seed=11, b=3, v=4, j=19, h=256, w=256, stride=4, noise=0.02, outliers=0, invalid=()),
        v4_outlier=dict(seed=12, b=3, v=4, j=19, h=256, w=256, stride=4, noise=0.05, outliers=1, invalid=(3, 17)),
        v8_nonsquare=dict(seed=13, b=2, v=8, j=19, h=384, w=288, stride=4, noise=0.02, outliers=2, invalid=(0,)),
        v2_nonsquare=dict(seed=14, b=2, v=2, j=19, h=256, w=192, stride=4, noise=0.02, outliers=0, invalid=()),
        v8_j42=dict(seed=15, b=1, v=8, j=42, h=256, w=256, stride=4, noise=0.05, outliers=3, invalid=(40, 41)),
        v3_noise=dict(seed=16, b=2, v=3, j=7, h=128, w=128, stride=8, noise=1.0, outliers=0, invalid=()),
    )


def xe_cases():
    return OrderedDict(
        xe_v4=dict(seed=21, b=2, v=4, j=19, h=256, w=256, stride=4, noise=0.02, outliers=0, invalid=(5,), sigma=1.0),
        xe_v3_nonsq=dict(seed=22, b=1, v=3, j=6, h=128, w=96, stride=4, noise=0.1, outliers=1, invalid=(), sigma=2.0),
    )


def build_triangulation_case(c):
    """-> heatmaps (B,V,J,Hh,Wh) f32, proj (B,V,3,4) f64, valid (B,J) bool."""
    b, v, j = c["b"], c["v"], c["j"]
    hh, wh = c["h"] // c["stride"], c["w"] // c["stride"]
    proj = np.stack([synth.ring_cameras(v, c["h"], c["w"], seed=c["seed"] + 100 * i) for i in range(b)])
    kp3d = synth.joints_3d(c["seed"], b, j)
    hm = synth.gaussian_heatmaps(c["seed"], proj, kp3d, hh, wh, c["stride"], 1.0, c["noise"], c["outliers"])
    valid = np.ones((b, j), dtype=bool)
    for k in c["invalid"]:
        valid[:, k] = False
    return hm, proj, valid


# --------------------------------------------------------------------------
# scoring
# --------------------------------------------------------------------------
def scoring_cases():
    return OrderedDict(
        gauss_64=dict(seed=31, b=2, v=4, j=19, hh=64, wh=64, kind="gauss", invalid=(2,)),
        noise_nonsq=dict(seed=32, b=2, v=2, j=5, hh=32, wh=24, kind="noise", invalid=()),
        noise_96x72=dict(seed=33, b=1, v=8, j=19, hh=96, wh=72, kind="noise", invalid=(7, 8)),
    )


def build_scoring_case(c):
    rng = np.random.default_rng(c["seed"])
    b, v, j, hh, wh = c["b"], c["v"], c["j"], c["hh"], c["wh"]
    hm = rng.standard_normal((b, v, j, hh, wh)).astype(np.float32)
    if c["kind"] == "gauss":
        ys = np.arange(hh, dtype=np.float32)[:, None]
        xs = np.arange(wh, dtype=np.float32)[None, :]
        hm *= np.float32(0.05)
        for idx in np.ndindex(b, v, j):
            for _ in range(int(rng.integers(1, 4))):
                cx, cy = rng.uniform(3, wh - 4), rng.uniform(3, hh - 4)
                a = np.float32(rng.uniform(0.3, 1.0))
                hm[idx] += a * np.exp(-((xs - np.float32(cx)) ** 2 + (ys - np.float32(cy)) ** 2) / np.float32(2.0))
    valid = np.ones((b, j), dtype=bool)
    for k in c["invalid"]:
        valid[:, k] = False
    return hm, valid


# --------------------------------------------------------------------------
# _compute_sal_dict
# --------------------------------------------------------------------------
def sal_cases():
    base = dict(seed=41, nbatch=2, b=2, v=4, j=19, h=256, w=256, stride=4, noise=0.05, outliers=1, select=2)
    return OrderedDict(
        hp=dict(base, strategy="HP"),
        triangulation=dict(base, strategy="TRIANGULATION", seed=42),
        coreset=dict(base, strategy="CORESET", seed=43),
        mpe=dict(base, strategy="MPE", seed=44),
        bsb=dict(base, strategy="BSB", seed=45),
        tri_xe=dict(base, strategy="TRIANGULATION", seed=46, xe=True, sigma=1.5),
    )


def build_sal_loader(c):
    """-> (list of batch dicts with numpy values, list of heat-map batches (B*V,J,Hh,Wh))."""
    loader, hms = [], []
    hh, wh = c["h"] // c["stride"], c["w"] // c["stride"]
    for i in range(c["nbatch"]):
        seed = c["seed"] * 10 + i
        b, v, j = c["b"], c["v"], c["j"]
        proj = np.stack([synth.ring_cameras(v, c["h"], c["w"], seed=seed + 100 * k) for k in range(b)])
        kp3d = synth.joints_3d(seed, b, j)
        hm = synth.gaussian_heatmaps(seed, proj, kp3d, hh, wh, c["stride"], 1.0, c["noise"], c["outliers"])
        valid = np.ones((b, j), dtype=np.float32)
        valid[:, (3 + i) % j] = 0
        gt = np.concatenate([kp3d, np.ones((b, 1, j), dtype=np.float32)], axis=1)  # (B, 4, J): x,y,z,conf
        loader.append(
            dict(
                images=np.zeros((b, v, 3, 8, 8), dtype=np.float32),
                pose=np.arange(b, dtype=np.int64) + 7 * i,
                frame_id=np.arange(b, dtype=np.int64) * 3 + 100 * i,
                proj_matrices=proj,
                joint_valid=valid,
                **{"3d_keypoints": gt},
            )
        )
        hms.append(hm.reshape(b * v, j, hh, wh))
    return loader, hms


# --------------------------------------------------------------------------
# core-set
# --------------------------------------------------------------------------
def coreset_cases():
    return OrderedDict(
        n64_l5_j19=dict(seed=51, n=64, l=5, j=19, root=2, select=10),
        n1000_l1_j19=dict(seed=52, n=1000, l=1, j=19, root=2, select=20),
        n1000_l200_j42=dict(seed=53, n=1000, l=200, j=42, root=21, select=20),
        n50000_l200_j19=dict(seed=54, n=50000, l=200, j=19, root=2, select=100),
    )


def coreset_arrays(c):
    """pool (n, J, 3) float32 values (what the gathered fp32 predictions are),
    labeled (l, J, 4) float64 (dataset '3d_keypoints'.T rows: x,y,z,conf)."""
    rng = np.random.default_rng(c["seed"])
    pool = (rng.standard_normal((c["n"], c["j"], 3)) * 300.0).astype(np.float32)
    lab = np.concatenate(
        [rng.standard_normal((c["l"], c["j"], 3)) * 300.0, np.ones((c["l"], c["j"], 1))], axis=2
    ).astype(np.float64)
    return pool, lab


def build_coreset_case(c):
    """-> (sal_dict, al_dict) as the reference sees them: pool values are python lists
    of fp32-valued floats (strategy.py:1141-1143), labeled values float64 arrays."""
    pool, lab = coreset_arrays(c)
    sal = OrderedDict(("%d-%d" % (i // 100, i % 100), pool[i].tolist()) for i in range(c["n"]))
    al = OrderedDict((i, lab[i]) for i in range(c["l"]))
    return sal, al


# --------------------------------------------------------------------------
# models
# --------------------------------------------------------------------------
def model_cases():
    return OrderedDict(
        w32=dict(arch="hrnet_w32", seed=0, n=2, h=256, w=256, j=19),
        w48=dict(arch="hrnet_w48", seed=0, n=1, h=384, w=288, j=19),
        r50=dict(arch="resnet50", seed=0, n=2, h=256, w=192, j=19),
        w32_small=dict(arch="hrnet_w32", seed=1, n=3, h=64, w=96, j=5),
    )


def train_cases():
    return OrderedDict(
        w32_train=dict(
            arch="hrnet_w32", seed=2, n=4, h=128, w=128, j=19,
            grad_keys=["conv1.weight", "layer1.0.conv2.weight", "stage3.1.branches.2.0.conv1.weight",
                       "stage4.2.fuse_layers.0.3.1.weight", "stage2.0.fuse_layers.1.0.0.0.weight",
                       "final_layer.weight", "final_layer.bias"],
            bn_keys=["bn1", "stage2.0.branches.1.3.bn2", "stage4.1.fuse_layers.2.0.0.1"],
        ),
        r50_train=dict(
            arch="resnet50", seed=3, n=2, h=128, w=96, j=19,
            grad_keys=["conv1.weight", "layer2.0.downsample.0.weight", "layer4.2.conv3.weight",
                       "deconv_layers.0.weight", "deconv_layers.7.weight", "final_layer.bias"],
            bn_keys=["bn1", "layer3.0.downsample.1", "deconv_layers.4"],
        ),
    )


def product_model(c):
    """The build's model for a case (parameter tree only; no compute here)."""
    from multi_view_active_learning_amd.pose_estimators import PoseHighResolutionNet, PoseResNet, hrnet_w48

    if c["arch"] == "hrnet_w32":
        return PoseHighResolutionNet(c["j"])
    if c["arch"] == "hrnet_w48":
        return PoseHighResolutionNet(c["j"], hrnet_cfg=hrnet_w48())
    return PoseResNet(c["j"])


def model_state_dict(c):
    shapes = product_model(c)._graph.param_shapes()
    return synth.synthetic_state_dict(shapes, c["seed"])


def model_input(c):
    return synth.images(c["seed"], c["n"], 1, c["h"], c["w"]).reshape(c["n"], 3, c["h"], c["w"])


def train_input(c):
    x = model_input(c)
    rng = np.random.default_rng(c["seed"] + 1000)
    hh, wh = c["h"] // 4, c["w"] // 4
    ys = np.arange(hh, dtype=np.float32)[:, None]
    xs = np.arange(wh, dtype=np.float32)[None, :]
    gt = np.zeros((c["n"], c["j"], hh, wh), dtype=np.float32)
    for idx in np.ndindex(c["n"], c["j"]):
        cx, cy = rng.uniform(0, wh - 1), rng.uniform(0, hh - 1)
        gt[idx] = np.exp(-((xs - np.float32(cx)) ** 2 + (ys - np.float32(cy)) ** 2) / np.float32(2.0))
    valid = rng.uniform(size=(c["n"], c["j"])) > 0.2
    return x, gt, valid


def build_reference_model(ns, c):
    """Real reference module with the case's synthetic weights (make_golden only)."""
    import torch

    from multi_view_active_learning_amd.pose_estimators import hrnet_w48

    if c["arch"] == "hrnet_w32":
        m = ns.hrnet.PoseHighResolutionNet(c["j"])
    elif c["arch"] == "hrnet_w48":
        m = ns.hrnet.PoseHighResolutionNet(c["j"], hrnet_cfg=hrnet_w48())
    else:
        m = ns.pose_resnet.PoseResNet(c["j"])
    sd = {k: torch.from_numpy(v) for k, v in model_state_dict(c).items()}
    m.load_state_dict(sd, strict=True)
    return m


def pck_cases():
    return OrderedDict(
        small=dict(seed=11, s=37, j=19, noise=2.0, p_valid=0.8),
        panoptic=dict(seed=12, s=256, j=19, noise=40.0, p_valid=0.95),
        ties=dict(seed=13, s=64, j=5, noise=0.0, p_valid=1.0),  # integer offsets: distances land ON thresholds
    )


def pck_arrays(c):
    """pred (S,J,3) f32, gt (S,4,J) f32 (rows x,y,z,confidence like the panoptic labels), valid (S,J) f32."""
    rng = np.random.default_rng(c["seed"])
    gt3 = (rng.standard_normal((c["s"], 3, c["j"])) * 300.0).astype(np.float32)
    gt = np.concatenate([gt3, np.ones((c["s"], 1, c["j"]), np.float32)], axis=1)
    if c["noise"] > 0:
        pred = gt3.transpose(0, 2, 1) + (rng.standard_normal((c["s"], c["j"], 3)) * c["noise"]).astype(np.float32)
    else:
        gt[:, :3] = np.round(gt[:, :3])
        pred = gt[:, :3].transpose(0, 2, 1) + rng.integers(-3, 4, size=(c["s"], c["j"], 3)).astype(np.float32)
    valid = (rng.uniform(size=(c["s"], c["j"])) < c["p_valid"]).astype(np.float32)
    valid[0] = 1.0  # every joint valid at least once (the reference divides by the valid count)
    return np.ascontiguousarray(pred.astype(np.float32)), gt, valid


def preprocess_cases():
    """raw image size (h0, w0), detection box (left, top, right, bottom), SCALE_BBOX, network input (w, h),
    heat-map stride, sigma, distortion or not."""
    return OrderedDict(
        inside=dict(seed=21, h0=240, w0=320, box=(60, 30, 200, 190), scale=1.0, in_w=64, in_h=64, stride=4, sigma=1.0, dist=False),
        outside=dict(seed=22, h0=180, w0=260, box=(-20, 40, 150, 230), scale=1.2, in_w=64, in_h=48, stride=4, sigma=1.0, dist=True),
        upscale=dict(seed=23, h0=90, w0=90, box=(20, 25, 60, 70), scale=1.0, in_w=96, in_h=96, stride=4, sigma=2.0, dist=False),
        wide=dict(seed=24, h0=300, w0=500, box=(100, 120, 420, 200), scale=1.0, in_w=128, in_h=128, stride=4, sigma=1.0, dist=False),
    )


def preprocess_inputs(c):
    """Smooth-plus-noise RGB image (so that resampling matters), a camera looking at the joints, J = 19."""
    rng = np.random.default_rng(c["seed"])
    yy, xx = np.mgrid[0 : c["h0"], 0 : c["w0"]]
    img = np.stack([127 + 120 * np.sin(xx / 9.0 + c["seed"]), 127 + 120 * np.cos(yy / 7.0), (3 * xx + 5 * yy) % 256], -1)
    img = np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)
    kp3d = np.concatenate([rng.normal(0, 200, (3, 19)), np.ones((1, 19))], 0)  # (4, J): x, y, z, confidence
    cam = dict(R=np.eye(3).tolist(), t=[[0.0], [0.0], [2500.0]],
               K=[[600.0, 0.0, c["w0"] / 2.0], [0.0, 600.0, c["h0"] / 2.0], [0.0, 0.0, 1.0]],
               dist=[0.05, -0.01, 0.001, 0.002, 0.0] if c["dist"] else None)
    return img, kp3d, cam


def sal_filter_cases():
    return OrderedDict(
        clusters=dict(seed=31, n=300, j=19, al_num=20, pseudo_num=40, clusters=10, thr=7, use_clusters=True, pseudo_done=15),
        random=dict(seed=32, n=200, j=19, al_num=10, pseudo_num=30, clusters=10, thr=7, use_clusters=False, pseudo_done=5),
        few=dict(seed=33, n=60, j=19, al_num=5, pseudo_num=20, clusters=4, thr=9, use_clusters=True, pseudo_done=0),
    )


def sal_filter_inputs(c):
    """A synthetic sal_dict as _compute_sal_dict returns it (guid -> python floats / ints / J x 3 lists of
    float32-valued floats), with NaNs, plus the guids that are already pseudo-labelled."""
    rng = np.random.default_rng(c["seed"])
    guids = ["%d-%d" % (int(p), int(f)) for p, f in zip(rng.integers(0, 4, c["n"]), rng.permutation(c["n"]))]
    al = rng.uniform(0, 1, c["n"])
    al[rng.uniform(size=c["n"]) < 0.05] = np.nan
    salm = rng.uniform(0.5, 30, c["n"]).astype(np.float32).astype(np.float64)
    salm[rng.uniform(size=c["n"]) < 0.05] = np.nan
    inl = rng.integers(3, 13, c["n"])
    modes = rng.normal(0, 300, (c["clusters"], c["j"], 3))
    kp = (modes[rng.integers(0, c["clusters"], c["n"])] + rng.normal(0, 40, (c["n"], c["j"], 3))).astype(np.float32)
    d = {
        "al_metric": OrderedDict((g, float(v)) for g, v in zip(guids, al)),
        "sal_metric": OrderedDict((g, float(v)) for g, v in zip(guids, salm)),
        "inlier_count": OrderedDict((g, int(v)) for g, v in zip(guids, inl)),
        "pred_3d_keypoints": OrderedDict((g, k.astype(np.float64).tolist()) for g, k in zip(guids, kp)),
        "mkpe": OrderedDict((g, 0.0) for g in guids),
    }
    done = [guids[i] for i in rng.permutation(c["n"])[: c["pseudo_done"]]]
    return d, done
