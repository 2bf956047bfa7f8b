"""Per-view input pipeline on the device: the reference's ``ActiveLearningDataset.prepare_single_view``
(dataset/dataset.py:158-220; "val"/"test" split, i.e. without RandAugment) for a batch of views.

Host (numpy float64, the reference's own arithmetic): square / scaled box (utils/triangulation.py:96-134),
camera update after crop and resize (:44-67), projection of the 3-D joints (:153-165, 433-484).
Device (csrc/preprocess.hip): BGR flip, zero-filled crop, PIL LANCZOS resize (bit-exact with Pillow's
8-bit resampler), ImageNet normalisation, Gaussian ground-truth heat-maps.  JPEG decode stays outside.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib


def get_square_bbox(bbox):
    """utils/triangulation.py:96-118."""
    left, upper, right, lower = bbox
    width, height = right - left, lower - upper
    if width > height:
        y_center = (upper + lower) // 2
        upper = y_center - width // 2
        lower = upper + width
    else:
        x_center = (left + right) // 2
        left = x_center - height // 2
        right = left + height
    return left, upper, right, lower


def scale_bbox(bbox, scale):
    """utils/triangulation.py:121-134."""
    left, upper, right, lower = bbox
    width, height = right - left, lower - upper
    x_center, y_center = (right + left) // 2, (lower + upper) // 2
    new_width, new_height = int(scale * width), int(scale * height)
    new_left = x_center - new_width // 2
    new_upper = y_center - new_height // 2
    return new_left, new_upper, new_left + new_width, new_upper + new_height


def _project(K, R, t, dist, X):
    """project_3d_points_with_camera (utils/triangulation.py:153-165): X (N, 3) -> (N, 2) float64."""
    if dist is not None:
        Kd = np.asarray(dist, dtype=np.float64).flatten()
        x = np.asarray(R.dot(X.T) + t)
        x[0:2, :] = x[0:2, :] / x[2, :]
        r = x[0, :] * x[0, :] + x[1, :] * x[1, :]
        x[0, :] = (x[0, :] * (1 + Kd[0] * r + Kd[1] * r * r + Kd[4] * r * r * r) + 2 * Kd[2] * x[0, :] * x[1, :]
                   + Kd[3] * (r + 2 * x[0, :] * x[0, :]))
        x[1, :] = (x[1, :] * (1 + Kd[0] * r + Kd[1] * r * r + Kd[4] * r * r * r) + 2 * Kd[3] * x[0, :] * x[1, :]
                   + Kd[2] * (r + 2 * x[1, :] * x[1, :]))
        x[0, :] = K[0, 0] * x[0, :] + K[0, 1] * x[1, :] + K[0, 2]
        x[1, :] = K[1, 0] * x[0, :] + K[1, 1] * x[1, :] + K[1, 2]
        return x.T[:, :2]
    h = np.concatenate([X, np.ones((X.shape[0], 1))], axis=1) @ K.dot(np.hstack([R, t])).T
    w = h[:, 2:3].copy()
    w[w == 0] = 1.0
    return h[:, :2] / w


class _ViewDesc(C.Structure):
    """include/mval_hip.h: struct mval_view_desc."""

    _fields_ = [("img", C.c_void_p), ("h0", C.c_int32), ("w0", C.c_int32), ("left", C.c_int32), ("top", C.c_int32),
                ("right", C.c_int32), ("bottom", C.c_int32), ("tmp_off", C.c_int64)]


# Descriptor uploads of resize_views: a small ring of pinned host buffers per device, copied with non_blocking=True on the current
# stream (a pageable copy would stall the host until the stream drains -- a batch pipeline then loses the overlap of the next
# batch's launches with the GPU's work); a slot is reused only after the event recorded behind its last copy has completed.
_DESC_RING = {}


def _upload_descs(raw: bytes, dev):
    key = dev.index
    ring = _DESC_RING.setdefault(key, {"k": 0, "slots": [None] * 4})
    i = ring["k"] % 4
    ring["k"] += 1
    slot = ring["slots"][i]
    if slot is None or slot[0].numel() < len(raw):
        slot = [torch.empty(max(len(raw), 16384), dtype=torch.uint8).pin_memory(), torch.cuda.Event()]
        ring["slots"][i] = slot
    else:
        slot[1].synchronize()
    slot[0][: len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
    out = slot[0][: len(raw)].to(dev, non_blocking=True)
    slot[1].record(torch.cuda.current_stream(dev))
    return out


def resize_views(images, boxes, in_w, in_h):
    """Pixel half of prepare_single_view for a list of decoded RGB images (uint8 HIP tensors (h0, w0, 3)) and
    their SQUARE boxes: (V, 3, in_h, in_w) float32 on the device (BGR order, ImageNet-normalised)."""
    if not images or any((not torch.is_tensor(im)) or (not im.is_cuda) or im.dtype != torch.uint8 for im in images):
        raise _lib.MvalError("resize_views: images must be uint8 HIP tensors (h0, w0, 3)")
    dev = images[0].device
    images = [im.contiguous() for im in images]
    descs = (_ViewDesc * len(images))()
    rows = 0
    for d, im, b in zip(descs, images, boxes):
        left, top, right, bottom = (int(v) for v in b)
        if right <= left or bottom <= top:
            raise ValueError("resize_views: empty box %r" % (b,))
        d.img, d.h0, d.w0 = im.data_ptr(), im.shape[0], im.shape[1]
        d.left, d.top, d.right, d.bottom = left, top, right, bottom
        d.tmp_off = rows * in_w * 3
        rows += bottom - top
    lib = _lib.lib()
    lib.mval_prepare_views_workspace_bytes.restype = C.c_size_t
    ws = torch.empty(int(lib.mval_prepare_views_workspace_bytes(C.c_int(len(images)), C.c_int64(rows), C.c_int(in_w), C.c_int(in_h))),
                     dtype=torch.uint8, device=dev)
    dd = _upload_descs(bytes(descs), dev)
    out = torch.empty((len(images), 3, in_h, in_w), dtype=torch.float32, device=dev)
    _lib._check(
        lib.mval_prepare_views(_lib._p(dd), C.c_int(len(images)), C.c_int(max(d.bottom - d.top for d in descs)),
                               C.c_int(max(d.right - d.left for d in descs)), C.c_int(in_w), C.c_int(in_h), _lib._p(out),
                               _lib._p(ws), _lib._stream()),
        "mval_prepare_views")
    return out


def gt_heatmaps(pt, sigma, h, w):
    """dataset.py:198-207: pt (..., 2) float64 heat-map-pixel coordinates (HIP tensor) -> (..., h, w) float32."""
    pt = pt.to(torch.float64).contiguous()
    if not pt.is_cuda:
        raise _lib.MvalError("gt_heatmaps: points must be a HIP tensor")
    n = pt.numel() // 2
    out = torch.empty(tuple(pt.shape[:-1]) + (h, w), dtype=torch.float32, device=pt.device)
    _lib._check(_lib.lib().mval_gt_heatmaps(_lib._p(pt), C.c_int64(n), C.c_double(float(sigma)), C.c_int(h), C.c_int(w),
                                            _lib._p(out), _lib._stream()), "mval_gt_heatmaps")
    return out


def prepare_views(images, boxes, cameras, kp_3d, scale_bbox_factor, in_w, in_h, gt_stride, sigma):
    """Batch of views of one or more frames: images (list of uint8 HIP tensors (h0, w0, 3) RGB), boxes (list of
    (left, top, right, bottom)), cameras (list of dicts R, t, K, dist), kp_3d ((>=3, J) array, or one per
    view).  Returns the reference's per-view entries stacked over views: images (V,3,H,W) and gt_heatmap
    (V,J,h,w) on the device; proj_matrices (V,3,4) float64, 2d_keypoints / 2d_after_crop (V,J,2) float32,
    square_box (V,4) float32 on the host."""
    v = len(images)
    kps = kp_3d if isinstance(kp_3d, (list, tuple)) else [kp_3d] * v
    sq, proj, pts, pts_crop = [], [], [], []
    for box, cam, kp in zip(boxes, cameras, kps):
        bbox = scale_bbox(get_square_bbox(tuple(int(b) for b in box)), scale_bbox_factor)
        K = np.array(cam["K"], dtype=np.float64).copy()
        R = np.array(cam["R"], dtype=np.float64)
        t = np.array(cam["t"], dtype=np.float64).reshape(3, 1)
        K[0, 2], K[1, 2] = K[0, 2] - bbox[0], K[1, 2] - bbox[1]            # Camera.update_after_crop
        skel = np.array(np.asarray(kp).transpose([1, 0]))[:, :3].astype(np.float64)
        pts_crop.append(_project(K, R, t, cam.get("dist"), skel))
        height, width = bbox[3] - bbox[1], bbox[2] - bbox[0]               # image shape before the resize
        fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]                # Camera.update_after_resize
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fx * (in_w / width), fy * (in_h / height), cx * (in_w / width), cy * (in_h / height)
        proj.append(K.dot(np.hstack([R, t])))
        pts.append(_project(K, R, t, cam.get("dist"), skel))
        sq.append(bbox)
    out_images = resize_views(images, sq, in_w, in_h)
    pt = np.stack(pts)
    hm = gt_heatmaps(torch.from_numpy(pt / gt_stride).to(out_images.device), sigma, in_h // gt_stride, in_w // gt_stride)
    return {
        "images": out_images,
        "gt_heatmap": hm,
        "proj_matrices": np.stack(proj),
        "2d_keypoints": pt.astype(np.float32),
        "2d_after_crop": np.stack(pts_crop).astype(np.float32),
        "square_box": np.asarray(sq, dtype=np.float32),
    }
