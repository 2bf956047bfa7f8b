"""TEST INFRASTRUCTURE ONLY -- loader for the *real* reference (this container only).

Imports facebookresearch/multi_view_active_learning from ``/root/reference`` on
CPU so that golden vectors can be generated (``tests/golden/make_golden.py``)
and the numpy/torch restatement in ``oracle/`` can be pinned against it.

Nothing here travels to the GPU box: ``/root/reference`` does not exist there
and no test marked ``gpu``, ``smoke()`` or ``bench.py`` imports this module.

Six of the reference's imports are absent from the image (colorlog, yacs,
kornia, iopath, skimage, tensorboard); they are replaced by inert stand-ins
that are never on an arithmetic path, with two documented exceptions that the
caller may opt into:

* ``skimage.feature.peak_local_max`` -> ``oracle.scoring.peak_local_max``
  (our restatement of scikit-image 0.18/0.19 semantics, itself pinned on vectors of the
  real scikit-image 0.18.3: tests/golden/make_peaks_golden.py).  This lets the
  reference's own ``_compute_mpe`` / ``_compute_bsb`` arithmetic run
  (``strategy.py:1149-1215``) under the interpreter that has torch but no scikit-image.
* ``kornia.spatial_soft_argmax2d`` -> ``oracle.geometry.spatial_soft_argmax2d``.

The second is flagged "parity unpinned" in DESIGN.md (kornia is nowhere in the image).
"""
from __future__ import annotations

import os
import sys
import tempfile
import types

REFERENCE_ROOT = "/root/reference"

_loaded = {}


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pose_estimators"))


class _AttrDict(dict):
    """10-line stand-in for yacs.config.CfgNode (attribute access + clone)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        out = _AttrDict()
        for k, v in self.items():
            out[k] = v.clone() if isinstance(v, _AttrDict) else (list(v) if isinstance(v, list) else v)
        return out


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load(third_party_restatements: bool = True):
    """Import the reference's library modules; returns a namespace of them."""
    if _loaded:
        return _loaded["ns"]
    if not available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference
    import logging

    import torch

    here = os.path.dirname(os.path.abspath(__file__))
    repo = os.path.dirname(here)
    if repo not in sys.path:
        sys.path.insert(0, repo)

    _mod("colorlog", basicConfig=lambda *a, **k: logging.basicConfig(level=logging.WARNING))
    _mod("yacs")
    _mod("yacs.config", CfgNode=_AttrDict)
    soft = None
    plm = None
    if third_party_restatements:
        from oracle import geometry as _g
        from oracle import scoring as _s

        soft = _g.spatial_soft_argmax2d_torch
        plm = _s.peak_local_max
    _mod("kornia", spatial_soft_argmax2d=soft)
    _mod("iopath")
    _mod("iopath.common")

    class PathManager:  # never used on the arithmetic path
        def open(self, *a, **k):
            return open(*a, **k)

    _mod("iopath.common.file_io", PathManager=PathManager)
    _mod("skimage")
    _mod("skimage.feature", peak_local_max=plm)
    _mod("tensorboard")
    import torch.utils  # noqa: F401

    tb = _mod("torch.utils.tensorboard", summary_writer=None)
    torch.utils.tensorboard = tb

    # the reference uses generic top-level names (utils, config, dataset, ...)
    sys.path.insert(0, REFERENCE_ROOT)
    # strategy.py / triangulation.py hard-code .cuda(); identity on CPU
    torch.Tensor.cuda = lambda self, *a, **k: self
    logging.disable(logging.INFO)

    import config as ref_config  # type: ignore
    import strategy as ref_strategy  # type: ignore
    from pose_estimators import hrnet as ref_hrnet  # type: ignore
    from pose_estimators import loss as ref_loss  # type: ignore
    from pose_estimators import pose_resnet as ref_pose_resnet  # type: ignore
    from utils import coreset as ref_coreset  # type: ignore
    from utils import evaluation as ref_evaluation  # type: ignore
    from utils import triangulation as ref_triangulation  # type: ignore

    ns = types.SimpleNamespace(
        config=ref_config,
        strategy=ref_strategy,
        hrnet=ref_hrnet,
        pose_resnet=ref_pose_resnet,
        loss=ref_loss,
        coreset=ref_coreset,
        evaluation=ref_evaluation,
        triangulation=ref_triangulation,
    )
    _loaded["ns"] = ns
    return ns


def init_single_rank_gloo():
    """1-rank gloo group so that the reference's per-sample all_gathers run."""
    import torch.distributed as dist

    if not dist.is_initialized():
        f = tempfile.NamedTemporaryFile(prefix="mval_ref_sync_", delete=False)
        f.close()
        os.unlink(f.name)
        dist.init_process_group("gloo", rank=0, world_size=1, init_method="file://" + f.name)


def make_strategy(al_strategy: str, **overrides):
    """ActiveLearningStrategy with default cfg, NUM_GPUS=1 (strategy.py:28)."""
    ns = load()
    cfg = ns.config.get_default_configs()
    cfg.NUM_GPUS = 1
    cfg.AL.STRATEGY = al_strategy
    for k, v in overrides.items():
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    init_single_rank_gloo()
    return ns.strategy.ActiveLearningStrategy(cfg)
