"""On-disk formats of an active-learning experiment, so that a run can be resumed across
implementations: the per-iteration guid lists / score dictionaries and the training checkpoints.

Everything here is host-side file handling that mirrors what the reference writes and reads:

* ``<LOG_DIR>/<EXPR_NAME>/SAMPLED-GUID-ITER-<i>``  one line, ``json.dumps(list of guids)`` (strategy.py:127-134)
* ``<LOG_DIR>/<EXPR_NAME>/SAL-GUID-ITER-<i>``      same, the pseudo-labelled guids            (strategy.py:112-118)
* ``<LOG_DIR>/<EXPR_NAME>/SAL-DICT-ITER-<i>``      ``json.dumps`` of the five per-guid dicts  (strategy.py:119-125)
* ``<LOG_DIR>/<EXPR_NAME>/ITER-<i>/checkpoints/CKPT-*.pth``  ``torch.save({"epoch", "global_step", "state_dict",
  "optimizer"})`` (strategy.py:681-711), read back with ``ckpt["state_dict"]`` and ``strict=True``
  (strategy.py:714-721,147-151,883-888).

Readers take the first line of a guid file, as ``restore_dataset`` does (strategy.py:314-337).
"""
from __future__ import annotations

import io
import json
import os

import numpy as np
import torch

SAMPLED_GUID = "SAMPLED-GUID-ITER-%d"
SAL_GUID = "SAL-GUID-ITER-%d"
SAL_DICT = "SAL-DICT-ITER-%d"
FINAL_CKPT = "CKPT-FINAL.pth"
SAL_DICT_KEYS = ("al_metric", "sal_metric", "inlier_count", "pred_3d_keypoints", "mkpe")


def _experiment_file(log_dir, expr_name, pattern, iteration):
    return os.path.join(log_dir, expr_name, pattern % iteration)


def sampled_guid_path(log_dir, expr_name, iteration):
    return _experiment_file(log_dir, expr_name, SAMPLED_GUID, iteration)


def sal_guid_path(log_dir, expr_name, iteration):
    return _experiment_file(log_dir, expr_name, SAL_GUID, iteration)


def sal_dict_path(log_dir, expr_name, iteration):
    return _experiment_file(log_dir, expr_name, SAL_DICT, iteration)


def checkpoints_dir(log_dir, expr_name, iteration=None):
    """``_prepare_experiment`` (strategy.py:651-679) puts checkpoints under ``<experiment>/checkpoints``; an AL
    iteration trains under ``EXPR_NAME + "/ITER-%d"`` (strategy.py:246-248)."""
    if iteration is not None:
        expr_name = expr_name + "/ITER-%d" % iteration
    return os.path.join(log_dir, expr_name, "checkpoints")


def checkpoint_name(epoch, global_step, mkpe=None):
    """Default name (strategy.py:690-691) or the evaluation snapshot's (strategy.py:499-500: step, then MKPE)."""
    if mkpe is not None:
        return "CKPT-E%d-MKPE%.2f.pth" % (global_step, mkpe)
    return "CKPT-E%d-S%d.pth" % (epoch, global_step)


def _write_text(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(text)
    return path


def write_sampled_guids(log_dir, expr_name, iteration, al_guids):
    return _write_text(sampled_guid_path(log_dir, expr_name, iteration), json.dumps(list(al_guids)))


def write_sal_guids(log_dir, expr_name, iteration, sal_guids):
    return _write_text(sal_guid_path(log_dir, expr_name, iteration), json.dumps(list(sal_guids)))


def write_sal_dict(log_dir, expr_name, iteration, sal_dict):
    """The five dicts in the reference's key order; values are python floats / nested lists already
    (``tables_to_sal_dict``), so ``json.dumps`` gives the reference's text (NaN is written as ``NaN``)."""
    missing = [k for k in SAL_DICT_KEYS if k not in sal_dict]
    if missing:
        raise KeyError("sal_dict lacks %s" % ", ".join(missing))
    return _write_text(sal_dict_path(log_dir, expr_name, iteration), json.dumps(sal_dict))


def write_iteration(log_dir, expr_name, iteration, al_guids, sal_guids=None, sal_dict=None):
    """What rank 0 leaves behind after ``sample_next_batch`` (strategy.py:54-135): iteration 0 (random seed batch)
    writes only the sampled guids; later iterations also the score dictionaries and -- when the pseudo-label
    filter kept anything -- the pseudo-labelled guids."""
    out = {}
    if iteration != 0 and sal_dict is not None:
        if sal_guids is not None and len(sal_guids) != 0:
            out["sal_guids"] = write_sal_guids(log_dir, expr_name, iteration, sal_guids)
        out["sal_dict"] = write_sal_dict(log_dir, expr_name, iteration, sal_dict)
    out["al_guids"] = write_sampled_guids(log_dir, expr_name, iteration, al_guids)
    return out


def read_guids(path):
    with open(path, "r") as f:
        return json.loads(f.readline())


def read_sal_dict(path):
    with open(path, "r") as f:
        return json.loads(f.read())


def restore_guids(log_dir, expr_name, iteration, expr_type="AL"):
    """``restore_dataset`` (strategy.py:314-337) without the dataset: the guid lists to label, one per finished
    iteration 0..iteration-1, and -- for a SAL experiment past its first iteration -- the pseudo-labelled guids of
    iteration-1 (else None)."""
    labeled = [read_guids(sampled_guid_path(log_dir, expr_name, i)) for i in range(0, iteration)]
    pseudo = None
    if expr_type == "SAL" and iteration > 1:
        pseudo = read_guids(sal_guid_path(log_dir, expr_name, iteration - 1))
    return labeled, pseudo


def read_cluster_features(path, joint_root_index):
    """The SAL cluster file (strategy.py:38-52): JSON ``{guid: (>=3, J) pose}`` -> the (n, 3J) root-relative rows the
    reference fits its KMeans on (same feature as the k-center one, utils/coreset.py:35-47)."""
    with open(path, "r") as f:
        clusters = json.load(f)
    rows = []
    for guid in clusters:
        kp = np.array(clusters[guid])
        rows.append((kp[0:3, :] - kp[0:3, joint_root_index : joint_root_index + 1]).flatten())
    return np.asarray(rows)


# ---- checkpoints ----------------------------------------------------------------------------------------------


def save_checkpoint(directory, epoch, global_step, pose_estimator, optimizer, ckpt_name=None):
    """``_save_checkpoints`` (strategy.py:681-711): an existing file of that name is replaced; returns the path.
    ``pose_estimator`` is saved as it is handed over -- wrapped in DistributedDataParallel its keys carry the
    ``module.`` prefix, exactly like the reference's (workflow.py:133)."""
    if ckpt_name is None:
        ckpt_name = checkpoint_name(epoch, global_step)
    path = os.path.join(directory, ckpt_name)
    os.makedirs(directory, exist_ok=True)
    if os.path.isfile(path):
        os.remove(path)
    with open(path, "wb") as f:
        torch.save(
            {
                "epoch": epoch,
                "global_step": global_step,
                "state_dict": pose_estimator.state_dict(),
                "optimizer": optimizer.state_dict(),
            },
            f,
        )
    return path


def load_checkpoint(path, map_location="cpu"):
    with open(path, "rb") as f:
        return torch.load(io.BytesIO(f.read()), map_location=map_location)


def _match_ddp_prefix(state_dict, module):
    """Checkpoints written from a DistributedDataParallel wrapper have ``module.``-prefixed keys; make them fit
    the module they are loaded into (wrapped or bare).  The reference always loads into the wrapped model."""
    want = any(k.startswith("module.") for k in module.state_dict())
    have = bool(state_dict) and all(k.startswith("module.") for k in state_dict)
    if have and not want:
        return {k[len("module.") :]: v for k, v in state_dict.items()}
    if want and not have:
        return {"module." + k: v for k, v in state_dict.items()}
    return state_dict


def restore_checkpoint(path, pose_estimator, optimizer=None):
    """Load ``state_dict`` strictly (strategy.py:719,888) and, when given, the optimizer state; returns
    ``(epoch, global_step)``."""
    ckpt = load_checkpoint(path)
    pose_estimator.load_state_dict(_match_ddp_prefix(ckpt["state_dict"], pose_estimator), strict=True)
    if optimizer is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    return ckpt["epoch"], ckpt["global_step"]


def load_weights(pose_estimator, restore_from="", init_weight="", estimator_type="HRNET"):
    """``_load_weights`` (strategy.py:713-743).  ``restore_from``: a checkpoint written by ``save_checkpoint`` (or by
    the reference), loaded strictly.  ``init_weight``: a bare ImageNet ``state_dict``; PoseResNet drops the
    ``final_layer`` entries, HRNet keeps the entries whose first name component is in ``pretrained_layers`` (or all for
    ``"*"``); both load non-strictly.  Returns "restored" / "initialized" / "scratch"."""
    if restore_from:
        restore_checkpoint(restore_from, pose_estimator)
        return "restored"
    if init_weight:
        pretrained = torch.load(init_weight, map_location="cpu")
        inner = pose_estimator.module if hasattr(pose_estimator, "module") else pose_estimator
        if estimator_type == "POSE_RESNET":
            pretrained = {k: v for k, v in pretrained.items() if k not in ("final_layer.weight", "final_layer.bias")}
        elif estimator_type == "HRNET":
            layers = inner.pretrained_layers
            pretrained = {k: v for k, v in pretrained.items() if k.split(".")[0] in layers or layers[0] == "*"}
        else:
            raise ValueError("unknown estimator type %r" % (estimator_type,))
        pose_estimator.load_state_dict(_match_ddp_prefix(pretrained, pose_estimator), strict=False)
        return "initialized"
    return "scratch"
