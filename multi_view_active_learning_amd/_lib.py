"""ctypes binding of the C-ABI shared library ``csrc/libmval_hip.so`` (include/mval_hip.h).

The library is the product: every hot-path entry point of this package ends in one of
these calls.  There is no CPU fallback -- if the library has not been built
(``python -c 'import __graft_entry__ as g; g.build()'``) or an input is not a HIP tensor,
the call raises.  Device pointers are borrowed from torch tensors that the caller keeps
alive; launches go to torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MVAL_LIB_TAG=<tag>: a variant library built by `MVAL_BUILD_TAG=<tag> MVAL_EXTRA_CFLAGS=... python -m ...build` -- measurement A/B only)
LIB_PATH = os.path.join(_HERE, "csrc", "libmval_hip" + ("_" + os.environ["MVAL_LIB_TAG"] if os.environ.get("MVAL_LIB_TAG") else "") + ".so")

_lock = threading.Lock()
_lib = None


class MvalError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raises if it is missing."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise MvalError(
                        f"{LIB_PATH} not found: the HIP extension is not built. "
                        "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                        "There is no CPU fallback for the hot path."
                    )
                l = C.CDLL(LIB_PATH)
                l.mval_last_error.restype = C.c_char_p
                l.mval_kcenter_workspace_bytes.restype = C.c_size_t
                l.mval_net_create.restype = C.c_void_p
                l.mval_packed_weight_floats.restype = C.c_size_t
                l.mval_op_flops.restype = C.c_double
                _lib = l
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = lib().mval_last_error()
        raise MvalError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _same_device(*tensors):
    """Launches go to the CURRENT device's current stream: every tensor of a call must live there (a tensor on
    another GPU would be addressed from the wrong device's stream)."""
    cur = torch.cuda.current_device()
    for t in tensors:
        if t is not None and torch.is_tensor(t) and t.is_cuda and t.device.index != cur:
            raise MvalError(f"tensor on cuda:{t.device.index} but the current device is cuda:{cur}: "
                            "call torch.cuda.set_device() (one process per GPU) or wrap the call in torch.cuda.device()")


def _p(t):
    """Device pointer of a contiguous HIP tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise MvalError("hot-path tensors must live on the HIP device (no CPU path)")
    if not t.is_contiguous():
        raise MvalError("hot-path tensors must be contiguous")
    _same_device(t)
    return C.c_void_p(t.data_ptr())


def _req(t, dtype, name):
    if t.dtype != dtype:
        raise MvalError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


# --------------------------------------------------------------------------
# keypoint decode / triangulation
# --------------------------------------------------------------------------
# Decode from the heat-map layer's epilogue: a forward whose final kernel kept arg-max keys of every map
# (mval_net_forward_keys) remembers them for the tensor it returned; argmax_decode of THAT tensor (any reshape of it,
# unmodified) then needs no second read of the heat-maps.  MVAL_EPILOGUE_DECODE=0 switches the mechanism off.
ARGMAX_SLOTS = 128  # MVAL_ARGMAX_SLOTS (include/mval_hip.h): partial keys per map
_ARGMAX_KEYS = {}  # id(network output) -> (weak reference to it, keys [n_images, ARGMAX_SLOTS, joints] int64 bit patterns)


def epilogue_decode_enabled():
    return os.environ.get("MVAL_EPILOGUE_DECODE", "1") != "0"


def forget_argmax_keys(t):
    """Drop the remembered keys of a network output: for callers that write into it through its data pointer (any kernel of
    this package launched on it in place) -- torch's version counter, which argmax_keys_of relies on, does not see such writes."""
    base = t._base if t._base is not None else t
    _ARGMAX_KEYS.pop(id(base), None)


def remember_argmax_keys(out, keys):
    k = id(out)
    _ARGMAX_KEYS[k] = (weakref.ref(out, lambda _r, k=k: _ARGMAX_KEYS.pop(k, None)), keys)


def argmax_keys_of(hm):
    """The keys of the network output `hm` is (a full view of), or None: unknown tensor, a slice, or written to since."""
    base = hm._base if hm._base is not None else hm
    e = _ARGMAX_KEYS.get(id(base))
    if e is None or e[0]() is not base:
        return None
    try:
        if base._version != 0:
            return None
    except RuntimeError:  # inference-mode tensors track no version: nothing tells whether the maps were written to since
        return None
    if hm.numel() != base.numel() or hm.data_ptr() != base.data_ptr() or not hm.is_contiguous() or hm.dtype != torch.float32:
        return None
    return e[1]


def argmax_from_keys(keys, valid, b, v, j, stride, split_width):
    out = torch.empty((b, v, j, 2), dtype=torch.int64, device=keys.device)
    _check(
        lib().mval_argmax_from_keys(_p(_req(keys, torch.int64, "argmax keys")), _p(valid), _p(out), C.c_int(b), C.c_int(v), C.c_int(j),
                                    C.c_int(stride), C.c_int(split_width), _stream()),
        "mval_argmax_from_keys",
    )
    return out


def argmax_decode(hm, valid, b, v, j, hh, wh, stride, split_width):
    keys = argmax_keys_of(hm) if epilogue_decode_enabled() else None
    # the keys are laid out [image][slot][joint of the PLAN]: only the caller's factorisation that matches it may use them
    # (b = n * J, v = 1, j = 1 has the same element count and would index them wrongly: ADVICE round 3)
    if keys is not None and tuple(keys.shape) == (b * v, ARGMAX_SLOTS, j) and hm.numel() == b * v * j * hh * wh:
        return argmax_from_keys(keys, valid, b, v, j, stride, split_width)
    out = torch.empty((b, v, j, 2), dtype=torch.int64, device=hm.device)
    _check(
        lib().mval_argmax_decode(
            _p(_req(hm, torch.float32, "heatmaps")), _p(valid), _p(out),
            C.c_int(b), C.c_int(v), C.c_int(j), C.c_int(hh), C.c_int(wh),
            C.c_int(stride), C.c_int(split_width), _stream(),
        ),
        "mval_argmax_decode",
    )
    return out


def soft_argmax(hm, n_maps, hh, wh, scale):
    out = torch.empty((n_maps, 2), dtype=torch.float32, device=hm.device)
    _check(
        lib().mval_soft_argmax(
            _p(_req(hm, torch.float32, "heatmaps")), _p(out), C.c_longlong(n_maps),
            C.c_int(hh), C.c_int(wh), C.c_float(scale), _stream(),
        ),
        "mval_soft_argmax",
    )
    return out


def triangulate_ransac(kp2d, proj, valid, b, v, j, eps):
    dev = proj.device
    kp_is_f32 = 1 if kp2d.dtype == torch.float32 else 0
    if not kp_is_f32:
        _req(kp2d, torch.int64, "keypoints_2d")
    kp3d = torch.empty((b, j, 3), dtype=torch.float64, device=dev)
    jerr = torch.empty((b, j), dtype=torch.float64, device=dev)
    jinl = torch.empty((b, j), dtype=torch.int32, device=dev)
    metric = torch.empty((b,), dtype=torch.float64, device=dev)
    inl = torch.empty((b,), dtype=torch.int32, device=dev)
    _check(
        lib().mval_triangulate_ransac(
            _p(kp2d), C.c_int(kp_is_f32), _p(_req(proj, torch.float64, "proj_matricies")), _p(valid),
            _p(kp3d), _p(jerr), _p(jinl), _p(metric), _p(inl),
            C.c_int(b), C.c_int(v), C.c_int(j), C.c_double(eps), _stream(),
        ),
        "mval_triangulate_ransac",
    )
    return kp3d, jerr, jinl, metric, inl


def reprojection_xe(kp3d, proj, hm, b, v, j, hh, wh, sigma):
    out = torch.empty((b,), dtype=torch.float64, device=hm.device)
    ws = torch.empty((b * v * j,), dtype=torch.float64, device=hm.device)
    _check(
        lib().mval_reprojection_xe(
            _p(_req(kp3d, torch.float64, "keypoints_3d")), _p(_req(proj, torch.float64, "proj")),
            _p(_req(hm, torch.float32, "heatmaps")), _p(out), _p(ws),
            C.c_int(b), C.c_int(v), C.c_int(j), C.c_int(hh), C.c_int(wh), C.c_double(sigma), _stream(),
        ),
        "mval_reprojection_xe",
    )
    return out


# --------------------------------------------------------------------------
# uncertainty scorers
# --------------------------------------------------------------------------
SCORE_HP, SCORE_MPE, SCORE_BSB = 0, 1, 2
REDUCE_AVG_F64, REDUCE_AVG_F32, REDUCE_STD_F64, REDUCE_STD_F32 = 0, 1, 2, 3


def score_maps(kind, hm, n_maps, hh, wh):
    """Per-map statistic (float32) + per-map peak count (int32)."""
    out = torch.empty((n_maps,), dtype=torch.float32, device=hm.device)
    cnt = torch.empty((n_maps,), dtype=torch.int32, device=hm.device)
    _check(
        lib().mval_score_maps(
            C.c_int(kind), _p(_req(hm, torch.float32, "heatmaps")), _p(out), _p(cnt),
            C.c_longlong(n_maps), C.c_int(hh), C.c_int(wh), _stream(),
        ),
        "mval_score_maps",
    )
    return out, cnt


def score_decode_maps(kind, hm, valid, b, v, j, hh, wh, stride, split_width):
    """Per-map statistic + peak count + hard arg-max key-points (B,V,J,2) int64 from ONE read of the heat-maps."""
    n_maps = b * v * j
    out = torch.empty((n_maps,), dtype=torch.float32, device=hm.device)
    cnt = torch.empty((n_maps,), dtype=torch.int32, device=hm.device)
    kp = torch.empty((b, v, j, 2), dtype=torch.int64, device=hm.device)
    _check(
        lib().mval_score_decode_maps(
            C.c_int(kind), _p(_req(hm, torch.float32, "heatmaps")), _p(valid), _p(out), _p(cnt), _p(kp),
            C.c_int(b), C.c_int(v), C.c_int(j), C.c_int(hh), C.c_int(wh), C.c_int(stride), C.c_int(split_width), _stream(),
        ),
        "mval_score_decode_maps",
    )
    return out, cnt, kp


def score_reduce(per_map, valid, b, v, j, mode):
    out = torch.empty((b,), dtype=torch.float64, device=per_map.device)
    _check(
        lib().mval_score_reduce(
            _p(_req(per_map, torch.float32, "per_map")), _p(valid), _p(out),
            C.c_int(b), C.c_int(v), C.c_int(j), C.c_int(mode), _stream(),
        ),
        "mval_score_reduce",
    )
    return out


# --------------------------------------------------------------------------
# loss / metric
# --------------------------------------------------------------------------
def masked_mse_fwd(h, g, valid, lead, hw, denom):
    out = torch.zeros((), dtype=torch.float32, device=h.device)
    ws = torch.empty((2048,), dtype=torch.float64, device=h.device)
    _check(
        lib().mval_masked_mse_fwd(
            _p(_req(h, torch.float32, "heatmaps")), _p(_req(g, torch.float32, "gt")), _p(valid), _p(out), _p(ws),
            C.c_longlong(lead), C.c_longlong(hw), C.c_double(denom), _stream(),
        ),
        "mval_masked_mse_fwd",
    )
    return out


def masked_mse_bwd(h, g, valid, grad_out, lead, hw, denom):
    gh = torch.empty_like(h)
    _check(
        lib().mval_masked_mse_bwd(
            _p(h), _p(g), _p(valid), _p(_req(grad_out, torch.float32, "grad")), _p(gh),
            C.c_longlong(lead), C.c_longlong(hw), C.c_double(denom), _stream(),
        ),
        "mval_masked_mse_bwd",
    )
    return gh


def mkpe(pred, gt, valid, s, j, gt_rows):
    """pred (S,J,3) f32, gt (S,gt_rows,J) f32, valid (S,J) f32 -> scalar f32 (+ per-sample (S,))."""
    out = torch.empty((), dtype=torch.float32, device=pred.device)
    per = torch.empty((s,), dtype=torch.float32, device=pred.device)
    _check(
        lib().mval_mkpe(
            _p(_req(pred, torch.float32, "pred")), _p(_req(gt, torch.float32, "gt")),
            _p(_req(valid, torch.float32, "valid")), _p(out), _p(per),
            C.c_longlong(s), C.c_int(j), C.c_int(gt_rows), _stream(),
        ),
        "mval_mkpe",
    )
    return out, per


def pck3d(pred, gt, valid, thresholds, mode):
    """3-D PCK (mode 0, mm thresholds over valid joints) / PCKh (mode 1) counters: pred (S,J,3) f32, gt (S,R,J) f32,
    valid (S,J) f32 or None -> hits (T,J) int64, counts (J,) int64 (device)."""
    s, j, _ = pred.shape
    thr = torch.as_tensor(list(thresholds), dtype=torch.float64).to(pred.device)
    hits = torch.empty((thr.numel(), j), dtype=torch.int64, device=pred.device)
    counts = torch.empty((j,), dtype=torch.int64, device=pred.device)
    _check(
        lib().mval_pck3d(
            _p(_req(pred, torch.float32, "pred")), _p(_req(gt, torch.float32, "gt")),
            _p(_req(valid, torch.float32, "valid")) if valid is not None else _p(None), _p(thr), C.c_int(thr.numel()),
            C.c_int(mode), _p(hits), _p(counts), C.c_longlong(s), C.c_int(j), C.c_int(gt.shape[1]), _stream(),
        ),
        "mval_pck3d",
    )
    return hits, counts


# --------------------------------------------------------------------------
# core-set (k-center greedy)
# --------------------------------------------------------------------------
def kcenter_workspace_bytes(n_obs: int, d: int) -> int:
    return int(lib().mval_kcenter_workspace_bytes(C.c_longlong(n_obs), C.c_int(d)))


def kcenter_select(feat, labeled_idx, n_select, min_dist=None):
    """feat (n_obs, D) f64; labeled_idx (L,) int64 or None.  Returns (picks int64 (n_select,),
    min_dist f64 (n_obs,))."""
    n_obs, d = feat.shape
    dev = feat.device
    picks = torch.empty((n_select,), dtype=torch.int64, device=dev)
    md = torch.empty((n_obs,), dtype=torch.float64, device=dev) if min_dist is None else min_dist
    norms = torch.empty((n_obs,), dtype=torch.float64, device=dev)
    ws = torch.empty((kcenter_workspace_bytes(n_obs, d) // 8 + 1,), dtype=torch.float64, device=dev)
    nl = 0 if labeled_idx is None else int(labeled_idx.numel())
    _check(
        lib().mval_kcenter_select(
            _p(_req(feat, torch.float64, "features")), C.c_longlong(n_obs), C.c_int(d),
            _p(labeled_idx if nl else None), C.c_longlong(nl), C.c_int(n_select),
            C.c_int(0 if min_dist is None else 1),
            _p(norms), _p(md), _p(picks), _p(ws), _stream(),
        ),
        "mval_kcenter_select",
    )
    return picks, md


def nearest_center(feat, centers):
    """feat (n, D) f64, centers (K, D) f64 (device) -> (n,) int32 labels."""
    n, d = feat.shape
    out = torch.empty((n,), dtype=torch.int32, device=feat.device)
    _check(
        lib().mval_nearest_center(_p(_req(feat, torch.float64, "feat")), _p(_req(centers, torch.float64, "centers")),
                                  C.c_longlong(n), C.c_int(d), C.c_int(centers.shape[0]), _p(out), _stream()),
        "mval_nearest_center",
    )
    return out


def coreset_features(pose, root_idx, n, j, rows):
    """pose (n, j, rows>=3) f64 [joint, coord] -> (n, 3j) f64 root-relative, coord-major."""
    out = torch.empty((n, 3 * j), dtype=torch.float64, device=pose.device)
    _check(
        lib().mval_coreset_features(
            _p(_req(pose, torch.float64, "pose")), _p(out), C.c_longlong(n), C.c_int(j), C.c_int(rows),
            C.c_int(root_idx), _stream(),
        ),
        "mval_coreset_features",
    )
    return out
