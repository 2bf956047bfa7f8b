"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's per-view input pipeline
(SURVEY 8(f) item 2): ``dataset/dataset.py:158-220`` ``prepare_single_view`` -- BGR flip, square /
scaled box (``utils/triangulation.py:96-134``), zero-filled crop (``:77-93``), PIL LANCZOS resize to
the network input size (``dataset.py:208-211``), ImageNet normalisation (``:137-145``) and the
Gaussian ground-truth heat-maps (``dataset.py:198-207``).

The resize restates the published algorithm of Pillow's ``ImagingResample`` for 8-bit images
(third-party dependency, present in this container as Pillow 12.2.0; pinned by
tests/test_oracle_golden.py against ``PIL.Image.resize`` itself): separable two-pass filter,
horizontal first, float64 Lanczos-3 weights normalised per output pixel, converted to 22-bit fixed
point, integer accumulation from 2^21, ``>> 22`` and clamp to 0..255 after EACH pass.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN = np.array([0.485, 0.456, 0.406])
IMAGENET_STD = np.array([0.229, 0.224, 0.225])


def _lanczos(x: float) -> float:
    if -3.0 <= x < 3.0:
        if x == 0.0:
            return 1.0
        a, b = x * math.pi, x / 3.0 * math.pi
        return (math.sin(a) / a) * (math.sin(b) / b)
    return 0.0


def lanczos_coeffs(in_size: int, out_size: int):
    """Pillow ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the box (0, in_size): returns
    (bounds (out,2) int [first source index, count], kk (out, ksize) int32)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One resample pass along `axis` (0 = rows / vertical, 1 = columns / horizontal) of an (h, w, c) u8 image."""
    src = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.empty((bounds.shape[0],) + src.shape[1:], dtype=np.int64)
    for o in range(bounds.shape[0]):
        lo, n = bounds[o]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        if n:
            acc = acc + np.tensordot(kk[o, :n], src[lo : lo + n], axes=(0, 0))
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def resize_lanczos_u8(img, out_w: int, out_h: int):
    """``Image.fromarray(img).resize((out_w, out_h), Image.LANCZOS)`` for an (h, w, 3) uint8 image."""
    h, w = img.shape[:2]
    out = img
    if out_w != w:
        out = _pass(out, *lanczos_coeffs(w, out_w), axis=1)
    if out_h != h:
        out = _pass(out, *lanczos_coeffs(h, out_h), axis=0)
    return out


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


def crop_zero_fill(img, bbox):
    """utils/triangulation.py:77-93 (PIL ``crop``: the box may leave the image; missing parts are 0)."""
    left, upper, right, lower = (int(v) for v in bbox)
    h, w = img.shape[:2]
    out = np.zeros((lower - upper, right - left, img.shape[2]), dtype=img.dtype)
    y0, y1, x0, x1 = max(upper, 0), min(lower, h), max(left, 0), min(right, w)
    if y1 > y0 and x1 > x0:
        out[y0 - upper : y1 - upper, x0 - left : x1 - left] = img[y0:y1, x0:x1]
    return out


def prepare_image(raw_rgb, box, scale_box, out_w, out_h):
    """The image half of ``prepare_single_view`` (dataset.py:158-181,208-218, without augmentation):
    raw decoded RGB (H0, W0, 3) uint8 + detection box -> (3, out_h, out_w) float32 and the square box."""
    image = raw_rgb[..., ::-1]
    bbox = scale_bbox(get_square_bbox(tuple(int(v) for v in box)), scale_box)
    image = crop_zero_fill(image, bbox)
    image = resize_lanczos_u8(image, out_w, out_h)
    image = (image / 255.0 - IMAGENET_MEAN) / IMAGENET_STD
    return np.ascontiguousarray(image.transpose(2, 0, 1)).astype(np.float32), bbox


def gt_heatmaps(pt, sigma, h, w):
    """dataset.py:198-207: pt (J, 2) float64 in heat-map pixels -> (J, h, w) float32; the exponent and the
    exp are float64 (float32 grid minus float64 labels promotes), the result is cast to float32."""
    pt = np.asarray(pt, dtype=np.float64)
    gx = np.arange(w, dtype=np.float32).astype(np.float64)[None, None, :]
    gy = np.arange(h, dtype=np.float32).astype(np.float64)[None, :, None]
    e = (gx - pt[:, 0, None, None]) ** 2 + (gy - pt[:, 1, None, None]) ** 2
    return np.exp(-e / (2.0 * (sigma**2))).astype(np.float32)


def project_points(K, R, t, dist, pts3d):
    """utils/triangulation.py:153-165,433-484: pts3d (N, 3) -> (N, 2) float64; with distortion the OpenCV
    radial/tangential model (``Kd = [k1, k2, p1, p2, k3]``), otherwise ``K [R|t]`` and a w-divide."""
    K, R, t = np.asarray(K, dtype=np.float64), np.asarray(R, dtype=np.float64), np.asarray(t, dtype=np.float64).reshape(3, 1)
    X = np.asarray(pts3d, dtype=np.float64)
    if dist is not None:
        Kd = np.asarray(dist, dtype=np.float64).flatten()
        x = R.dot(X.T) + t
        x[0:2, :] = x[0:2, :] / x[2, :]
        r = x[0, :] * x[0, :] + x[1, :] * x[1, :]
        x0 = (x[0, :] * (1 + Kd[0] * r + Kd[1] * r * r + Kd[4] * r * r * r) + 2 * Kd[2] * x[0, :] * x[1, :]
              + Kd[3] * (r + 2 * x[0, :] * x[0, :]))
        x[0, :] = x0  # the reference overwrites x[0] before using it for x[1]
        x1 = (x[1, :] * (1 + Kd[0] * r + Kd[1] * r * r + Kd[4] * r * r * r) + 2 * Kd[3] * x[0, :] * x[1, :]
              + Kd[2] * (r + 2 * x[1, :] * x[1, :]))
        x[1, :] = x1
        x[0, :] = K[0, 0] * x[0, :] + K[0, 1] * x[1, :] + K[0, 2]
        x[1, :] = K[1, 0] * x[0, :] + K[1, 1] * x[1, :] + K[1, 2]
        return x.T[:, :2]
    P = K.dot(np.hstack([R, t]))
    h = np.concatenate([X, np.ones((X.shape[0], 1))], axis=1) @ P.T
    w = h[:, 2:3].copy()
    w[w == 0] = 1.0  # _homogeneous_to_euclidean's zero guard (utils/triangulation.py:387-405)
    return h[:, :2] / w


def prepare_view(raw_rgb, box, camera, kp_3d, scale_box, in_w, in_h, gt_stride, sigma):
    """``prepare_single_view`` (dataset/dataset.py:158-220) for the "val"/"test" split (no augmentation):
    returns the same per-view entries as the reference."""
    images, bbox = prepare_image(raw_rgb, box, scale_box, in_w, in_h)
    K = np.array(camera["K"], dtype=np.float64).copy()
    K[0, 2] -= bbox[0]  # Camera.update_after_crop (utils/triangulation.py:44-52)
    K[1, 2] -= bbox[1]
    skel = np.array(np.asarray(kp_3d).transpose([1, 0]))[:, :3]
    pt_crop = project_points(K, camera["R"], camera["t"], camera["dist"], skel)
    h0, w0 = bbox[3] - bbox[1], bbox[2] - bbox[0]  # image_shape_before_resize = the crop's (h, w)
    K[0, 0] *= in_w / w0  # Camera.update_after_resize (:54-67)
    K[1, 1] *= in_h / h0
    K[0, 2] *= in_w / w0
    K[1, 2] *= in_h / h0
    R, t = np.asarray(camera["R"], dtype=np.float64), np.asarray(camera["t"], dtype=np.float64).reshape(3, 1)
    pt = project_points(K, R, t, camera["dist"], skel)
    return {
        "images": images,
        "square_box": np.asarray(bbox, dtype=np.float32),
        "2d_after_crop": pt_crop.astype(np.float32),
        "proj_matrices": K.dot(np.hstack([R, t])),
        "2d_keypoints": pt.astype(np.float32),
        "gt_heatmap": gt_heatmaps(pt / gt_stride, sigma, in_h // gt_stride, in_w // gt_stride),
    }
