"""TEST INFRASTRUCTURE ONLY -- stock-PyTorch (CPU, fp32) functional restatement of
the reference's two heatmap networks, driven purely by a ``state_dict`` with the
reference's key names.

* ``hrnet_forward``      restates pose_estimators/hrnet.py:468-501 (+ :36-52 BasicBlock,
  :75-95 Bottleneck, :269-287 module forward/fuse, :370-413 transitions).
* ``pose_resnet_forward`` restates pose_estimators/pose_resnet.py:139-153 (+ :211-231
  Bottleneck, :107-137 deconv head).
* ``pose_2d_mse``        restates pose_estimators/loss.py:14-20.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module.  It is pinned against the real reference modules (same weights
via ``load_state_dict``) by tests/golden/make_golden.py and tests/test_oracle_models.py
(the latter runs only where /root/reference exists); golden heatmaps are committed
under tests/golden/.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_MOMENTUM = 0.1  # hrnet.py:16, pose_resnet.py:14 (the un-named BNs use torch's default, also 0.1)
BN_EPS = 1e-5

HRNET_W32 = dict(
    stage2=dict(modules=1, branches=2, blocks=4, channels=(32, 64)),
    stage3=dict(modules=4, branches=3, blocks=4, channels=(32, 64, 128)),
    stage4=dict(modules=3, branches=4, blocks=4, channels=(32, 64, 128, 256)),
)
HRNET_W48 = dict(
    stage2=dict(modules=1, branches=2, blocks=4, channels=(48, 96)),
    stage3=dict(modules=4, branches=3, blocks=4, channels=(48, 96, 192)),
    stage4=dict(modules=3, branches=4, blocks=4, channels=(48, 96, 192, 384)),
)


def _conv(sd, x, key, stride=1):
    w = sd[key + ".weight"]
    return F.conv2d(x, w, sd.get(key + ".bias"), stride=stride, padding=w.shape[-1] // 2)


def _bn(sd, x, key, training):
    return F.batch_norm(
        x,
        sd[key + ".running_mean"],
        sd[key + ".running_var"],
        sd[key + ".weight"],
        sd[key + ".bias"],
        training,
        BN_MOMENTUM,
        BN_EPS,
    )


def _basic_block(sd, x, p, training):
    y = F.relu(_bn(sd, _conv(sd, x, p + ".conv1"), p + ".bn1", training))
    y = _bn(sd, _conv(sd, y, p + ".conv2"), p + ".bn2", training)
    return F.relu(y + x)


def _bottleneck(sd, x, p, training, stride=1):
    y = F.relu(_bn(sd, _conv(sd, x, p + ".conv1"), p + ".bn1", training))
    y = F.relu(_bn(sd, _conv(sd, y, p + ".conv2", stride), p + ".bn2", training))
    y = _bn(sd, _conv(sd, y, p + ".conv3"), p + ".bn3", training)
    if (p + ".downsample.0.weight") in sd:
        x = _bn(sd, _conv(sd, x, p + ".downsample.0", stride), p + ".downsample.1", training)
    return F.relu(y + x)


def _hr_module(sd, xs, p, nb, blocks, n_out, training):
    xs = list(xs)
    for b in range(nb):
        for k in range(blocks):
            xs[b] = _basic_block(sd, xs[b], f"{p}.branches.{b}.{k}", training)
    outs = []
    for i in range(n_out):
        y = None
        for j in range(nb):
            if j == i:
                t = xs[j]
            elif j > i:
                q = f"{p}.fuse_layers.{i}.{j}"
                t = _bn(sd, _conv(sd, xs[j], q + ".0"), q + ".1", training)
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
            else:
                t = xs[j]
                for k in range(i - j):
                    q = f"{p}.fuse_layers.{i}.{j}.{k}"
                    t = _bn(sd, _conv(sd, t, q + ".0", 2), q + ".1", training)
                    if k != i - j - 1:
                        t = F.relu(t)
            y = t if y is None else y + t
        outs.append(F.relu(y))
    return outs


def hrnet_forward(sd, x, arch=HRNET_W32, training=False):
    """x (N,3,H,W) fp32 -> (N,J,H/4,W/4).  ``sd`` maps reference key names to tensors
    (running stats are updated in place when ``training``)."""
    x = F.relu(_bn(sd, _conv(sd, x, "conv1", 2), "bn1", training))
    x = F.relu(_bn(sd, _conv(sd, x, "conv2", 2), "bn2", training))
    for k in range(4):
        x = _bottleneck(sd, x, f"layer1.{k}", training)
    ys = [x]
    for s, name in enumerate(("stage2", "stage3", "stage4")):
        cfg = arch[name]
        nb = cfg["branches"]
        t = f"transition{s + 1}"
        xs = []
        for i in range(nb):
            if i < len(ys):
                if (f"{t}.{i}.0.weight") in sd:
                    xs.append(F.relu(_bn(sd, _conv(sd, ys[i], f"{t}.{i}.0"), f"{t}.{i}.1", training)))
                else:
                    xs.append(ys[i])
            else:
                z = ys[-1]
                for j in range(i + 1 - len(ys)):
                    z = F.relu(_bn(sd, _conv(sd, z, f"{t}.{i}.{j}.0", 2), f"{t}.{i}.{j}.1", training))
                xs.append(z)
        for m in range(cfg["modules"]):
            last = name == "stage4" and m == cfg["modules"] - 1
            xs = _hr_module(sd, xs, f"{name}.{m}", nb, cfg["blocks"], 1 if last else nb, training)
        ys = xs
    return _conv(sd, ys[0], "final_layer")


def pose_resnet_forward(sd, x, layers=(3, 4, 6, 3), training=False):
    """PoseResNet-50/101/152 (bottleneck variants), pose_resnet.py:139-153."""
    x = F.relu(_bn(sd, F.conv2d(x, sd["conv1.weight"], None, 2, 3), "bn1", training))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, n in enumerate(layers):
        for k in range(n):
            stride = 2 if (li > 0 and k == 0) else 1
            x = _bottleneck(sd, x, f"layer{li + 1}.{k}", training, stride)
    for d in range(3):
        x = F.conv_transpose2d(x, sd[f"deconv_layers.{3 * d}.weight"], None, stride=2, padding=1)
        x = F.relu(_bn(sd, x, f"deconv_layers.{3 * d + 1}", training))
    return _conv(sd, x, "final_layer")


def pose_2d_mse(heatmaps, gt_heatmaps, joint_valid=None):
    """loss.py:14-20: sum(where(valid,(h-g)^2,0)) / (N*H*W)  (J is NOT in the divisor)."""
    loss = (heatmaps - gt_heatmaps) ** 2
    if joint_valid is not None:
        loss = torch.where(joint_valid.bool(), loss, torch.zeros_like(loss))
    return torch.sum(loss) / (heatmaps.shape[0] * heatmaps.shape[-1] * heatmaps.shape[-2])


def compute_mkpe(pred_list, gt_list, valid_list):
    """utils/evaluation.py:198-208: pred (J,3), gt (>=3,J), valid (J,)."""
    kpe = torch.zeros_like(valid_list[0]).float()
    count = torch.zeros_like(valid_list[0])
    for pred, gt, valid in zip(pred_list, gt_list, valid_list):
        d = torch.square(pred.permute([1, 0]) - gt[:3, :])
        d = torch.where(valid.bool(), d, torch.zeros_like(d))
        kpe = kpe + torch.sqrt(torch.sum(d, dim=0))
        count = count + valid
    return torch.mean(kpe / count)


def _dist3_f32(pred, gt):
    """float32 distances with the reference's operation order: sqrt(((dx^2 + dy^2) + dz^2)), pred (S,J,3),
    gt (S,>=3,J) -> (S,J) (utils/evaluation.py:162-166,188-192: 0-d tensor arithmetic, each op rounded)."""
    import numpy as np

    p = np.asarray(pred, dtype=np.float32)
    g = np.asarray(gt, dtype=np.float32)
    dx = p[:, :, 0] - g[:, 0, :]
    dy = p[:, :, 1] - g[:, 1, :]
    dz = p[:, :, 2] - g[:, 2, :]
    return np.sqrt((dx * dx + dy * dy) + dz * dz, dtype=np.float32)


def compute_3d_pck(pred_3d_labels, gt_3d_labels, valid_joints, threshold_mm, num_keypoints):
    """utils/evaluation.py:177-195: per joint, the fraction of VALID samples with distance < threshold_mm."""
    import numpy as np

    d = _dist3_f32(np.stack([np.asarray(p) for p in pred_3d_labels]), np.stack([np.asarray(g) for g in gt_3d_labels]))
    v = np.stack([np.asarray(x) for x in valid_joints]).astype(bool)[:, :num_keypoints]
    hit = (d[:, :num_keypoints].astype(np.float64) < float(threshold_mm)) & v
    return [int(k) / int(c) for k, c in zip(hit.sum(0), v.sum(0))]  # ZeroDivisionError if a joint is never valid


def compute_3d_pckh(pred_3d_labels, gt_3d_labels, threshold, num_keypoints):
    """utils/evaluation.py:150-174: threshold x the gt distance between joints 0 and 1 (float32 product), every
    joint of every sample counted."""
    import numpy as np

    g = np.stack([np.asarray(x, dtype=np.float32) for x in gt_3d_labels])
    d = _dist3_f32(np.stack([np.asarray(p) for p in pred_3d_labels]), g)
    hx, hy, hz = g[:, 0, 0] - g[:, 0, 1], g[:, 1, 0] - g[:, 1, 1], g[:, 2, 0] - g[:, 2, 1]
    head = np.sqrt((hx * hx + hy * hy) + hz * hz, dtype=np.float32) * np.float32(threshold)
    hit = d[:, :num_keypoints] < head[:, None]
    return [int(k) / len(g) for k in hit.sum(0)]
