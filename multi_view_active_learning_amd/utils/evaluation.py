"""Keypoint decode + MKPE -- drop-in for the hot-path part of the reference's
utils/evaluation.py (:13-58 arg-max decode, :198-208 compute_mkpe)."""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib


def get_scaled_pred_corrdinates(pred_map, stride, num_keypoints, valid_joints):
    """utils/evaluation.py:13-30 (name kept, typo included): (V,J,Hh,Wh) HIP tensor ->
    numpy int64 (V, num_keypoints, 2) [x, y], one launch instead of V*J argmax + 2*V*J
    ``.item()`` syncs.  Mirrors the ``shape[2]`` split quirk."""
    v, j, hh, wh = pred_map.shape
    valid = torch.as_tensor(valid_joints)
    vu8 = (valid != 0).to(torch.uint8).reshape(1, -1)[:, :j].to(pred_map.device).contiguous()
    out = _lib.argmax_decode(pred_map.to(torch.float32).contiguous(), vu8, 1, v, j, hh, wh, int(stride), hh)
    return out[0, :, :num_keypoints].cpu().numpy()


def get_pred_coordinates(pred_map, bbox, num_keypoints, use_softargmax=False):
    """utils/evaluation.py:33-58: coordinates scaled by the (square) bounding box."""
    n, j, hh, wh = pred_map.shape
    hm = pred_map.to(torch.float32).contiguous()
    bbox = torch.as_tensor(bbox, dtype=torch.float32, device=pred_map.device).reshape(n, 4)
    if use_softargmax:
        coords = _lib.soft_argmax(hm, n * j, hh, wh, 1.0).reshape(n, j, 2)
        scale = (bbox[:, 3] - bbox[:, 1]) / (1.0 * wh)
        return coords * scale[:, None, None]
    idx = _lib.argmax_decode(hm, None, 1, n, j, hh, wh, 1, hh)[0]  # (n, j, 2): (idx % hh, idx // hh)
    sx = (bbox[:, 3] - bbox[:, 1]) / (1.0 * wh)
    sy = (bbox[:, 2] - bbox[:, 0]) / (1.0 * hh)
    x = idx[..., 0].to(torch.float32) * sx[:, None]
    y = idx[..., 1].to(torch.float32) * sy[:, None]
    xy = torch.stack([x, y], dim=-1)[:, :num_keypoints]
    return [[[xy[b, k, 0], xy[b, k, 1]] for k in range(xy.shape[1])] for b in range(n)]


def compute_mkpe(pred_3d_labels, gt_3d_labels, valid_joints):
    """utils/evaluation.py:198-208: lists of pred (J,3), gt (>=3,J), valid (J,) -> 0-d tensor."""
    pred = torch.stack([torch.as_tensor(p) for p in pred_3d_labels]).to(torch.float32)
    gt = torch.stack([torch.as_tensor(g) for g in gt_3d_labels]).to(torch.float32)
    valid = torch.stack([torch.as_tensor(v) for v in valid_joints]).to(torch.float32)
    if not pred.is_cuda:
        raise _lib.MvalError("compute_mkpe: inputs must be HIP tensors")
    s, j, _ = pred.shape
    out, _ = _lib.mkpe(pred.contiguous(), gt.contiguous(), valid.contiguous(), s, j, gt.shape[1])
    return out


def mkpe_per_sample(pred, gt, valid):
    """Batched form of the per-sample call at strategy.py:1134: pred (S,J,3), gt (S,>=3,J),
    valid (S,J) HIP tensors -> (S,) f32 (NaN when a joint of the sample is invalid, as in
    the reference where 0/0 enters the mean)."""
    s, j, _ = pred.shape
    _, per = _lib.mkpe(
        pred.to(torch.float32).contiguous(), gt.to(torch.float32).contiguous(), valid.to(torch.float32).contiguous(),
        s, j, gt.shape[1],
    )
    return per


def _stack3(pred_3d_labels, gt_3d_labels, valid_joints=None):
    pred = torch.stack([torch.as_tensor(p) for p in pred_3d_labels]).to(torch.float32)
    gt = torch.stack([torch.as_tensor(g) for g in gt_3d_labels]).to(torch.float32)
    if not pred.is_cuda:
        raise _lib.MvalError("3-D PCK: inputs must be HIP tensors")
    valid = None
    if valid_joints is not None:
        valid = torch.stack([torch.as_tensor(v) for v in valid_joints]).to(pred.device, torch.float32).contiguous()
    return pred.contiguous(), gt.to(pred.device).contiguous(), valid


def _fractions(hits, counts, num_keypoints):
    """[hits / count per joint] per threshold as Python floats (int / int, like the reference's k / c)."""
    h, c = hits.cpu().tolist(), counts.cpu().tolist()
    return [[row[k] / c[k] for k in range(num_keypoints)] for row in h]  # ZeroDivisionError as in the reference


def compute_3d_pck_figure(pred_3d_labels, gt_3d_labels, valid_joints, num_keypoints, thresholds=(1, 2, 3, 4, 5)):
    """utils/evaluation.py:134-147: (thresholds, [per-joint PCK list per threshold]) -- ONE kernel launch for all
    thresholds instead of S x J x T ``.item()`` round trips."""
    pred, gt, valid = _stack3(pred_3d_labels, gt_3d_labels, valid_joints)
    hits, counts = _lib.pck3d(pred, gt, valid, thresholds, 0)
    return thresholds, _fractions(hits, counts, num_keypoints)


def compute_3d_pck(pred_3d_labels, gt_3d_labels, valid_joints, threshold_mm, num_keypoints):
    """utils/evaluation.py:177-195."""
    return compute_3d_pck_figure(pred_3d_labels, gt_3d_labels, valid_joints, num_keypoints, (threshold_mm,))[1][0]


def compute_3d_pckh_figure(pred_3d_labels, gt_3d_labels, num_keypoints,
                           thresholds=(0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0)):
    """utils/evaluation.py:121-131."""
    pred, gt, _ = _stack3(pred_3d_labels, gt_3d_labels)
    hits, counts = _lib.pck3d(pred, gt, None, thresholds, 1)
    return thresholds, _fractions(hits, counts, num_keypoints)


def compute_3d_pckh(pred_3d_labels, gt_3d_labels, threshold, num_keypoints):
    """utils/evaluation.py:150-174."""
    return compute_3d_pckh_figure(pred_3d_labels, gt_3d_labels, num_keypoints, (threshold,))[1][0]
