"""Hot-path entry points of the reference's ``ActiveLearningStrategy`` (strategy.py):

  _compute_batch_heatmap :772-782     _compute_batch_loss :762-770
  _compute_mpe/_compute_hp/_compute_bsb :1149-1215
  _compute_sal_dict :1004-1147        selection part of _sal_pseudo_labeling :932-949
  train_step (inner loop body :460-487)   _evaluate_all core :597-636

Same names, arguments and result types, so that the reference's orchestration code
(experiment dirs, checkpoints, TensorBoard -- out of scope here) can call them unchanged.
What changes is the execution model: a batch of frames is scored by a handful of HIP
launches with NO per-sample device->host synchronisation and NO per-sample collective;
ranks exchange ONE packed table per scoring pass (two RCCL collectives: a size exchange, then the data gather) instead of the
reference's 8 tiny all_gathers per sample (strategy.py:1106-1114).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from heapq import nlargest

import numpy as np
import torch

from . import _lib
from .pose_estimators.loss import Pose2DMeanSquaredError
from .utils import coreset, evaluation, triangulation

_KIND = {"HP": _lib.SCORE_HP, "MPE": _lib.SCORE_MPE, "BSB": _lib.SCORE_BSB}


def _reduce_mode(kind: str, config: str) -> int:
    """Precision/order of the reference's aggregation: HP values are python
    floats (float64 sum), MPE/BSB values are np.float32 (float32 sum)."""
    if config == "AVG":
        return _lib.REDUCE_AVG_F64 if kind == "HP" else _lib.REDUCE_AVG_F32
    if config == "STD":
        return _lib.REDUCE_STD_F64 if kind == "HP" else _lib.REDUCE_STD_F32
    raise NotImplementedError("AL.%s_CONFIG should be either AVG or STD." % kind)


def score_heatmaps_batch(kind: str, config: str, heatmaps, joint_valid):
    """heatmaps (B,V,J,Hh,Wh) f32 HIP, joint_valid (B,J) -> (B,) float64 HIP tensor holding
    the value the reference's ``_compute_{hp,mpe,bsb}`` returns for each frame (exactly
    representable float32 where the reference's result is float32)."""
    b, v, j, hh, wh = heatmaps.shape
    hm = heatmaps.to(torch.float32).contiguous()
    per_map, n_peaks = _lib.score_maps(_KIND[kind], hm, b * v * j, hh, wh)
    valid = (torch.as_tensor(joint_valid) != 0).to(torch.uint8).reshape(b, j).to(hm.device).contiguous()
    out = _lib.score_reduce(per_map, valid, b, v, j, _reduce_mode(kind, config))
    return out, per_map.reshape(b, v, j), n_peaks.reshape(b, v, j), valid


def score_decode_heatmaps_batch(kind: str, config: str, heatmaps, joint_valid, stride, mirror_nonsquare_quirk=True):
    """``score_heatmaps_batch`` + the hard arg-max decode of ``triangulate_batch`` from ONE read of the heat-maps
    (csrc/scoring.hip, DECODE): what a pool-scoring pass needs per map (strategy.py:1027-1090 reads every heat-map in
    get_scaled_pred_corrdinates AND in _compute_{hp,mpe,bsb}).  Returns (out, per_map, n_peaks, valid, keypoints_2d)."""
    b, v, j, hh, wh = heatmaps.shape
    hm = heatmaps.to(torch.float32).contiguous()
    valid = (torch.as_tensor(joint_valid) != 0).to(torch.uint8).reshape(b, j).to(hm.device).contiguous()
    per_map, n_peaks, kp2d = _lib.score_decode_maps(_KIND[kind], hm, valid, b, v, j, hh, wh, int(stride),
                                                    hh if mirror_nonsquare_quirk else wh)
    out = _lib.score_reduce(per_map, valid, b, v, j, _reduce_mode(kind, config))
    return out, per_map.reshape(b, v, j), n_peaks.reshape(b, v, j), valid, kp2d


def tables_to_sal_dict(per_rank, batch_sizes, sal_dict=None):
    """Packed per-rank tables [pose, frame_id, al_metric, sal_metric, inlier_count, mkpe,
    keypoints_3d(3J)] -> the reference's five dicts, inserted in its gather order
    (for batch: for sample: for rank; strategy.py:1024,1036,1115-1145).

    ``batch_sizes``: one list of batch sizes PER RANK (a single flat list = every rank has that
    structure, the DistributedSampler case).  Ranks may be ragged (a short last shard, a short
    last batch): the walk covers the longest structure and skips what a rank does not have, so
    every rank builds the identical dict from the identical gathered inputs."""
    if sal_dict is None:
        sal_dict = {k: OrderedDict() for k in ("al_metric", "sal_metric", "inlier_count", "pred_3d_keypoints", "mkpe")}
    j = (per_rank[0].shape[1] - 6) // 3
    if len(batch_sizes) == 0 or np.isscalar(batch_sizes[0]):
        batch_sizes = [list(batch_sizes)] * len(per_rank)
    if len(batch_sizes) != len(per_rank):
        raise ValueError("tables_to_sal_dict: one batch-size list per rank expected")
    sizes = [[int(x) for x in b] for b in batch_sizes]
    for r, (tab, b) in enumerate(zip(per_rank, sizes)):
        if sum(b) != tab.shape[0]:
            raise ValueError("tables_to_sal_dict: rank %d has %d rows but batch sizes sum to %d" % (r, tab.shape[0], sum(b)))
    from .parallel import reference_gather_order

    for r, row in reference_gather_order(sizes):
        e = per_rank[r][row]
        guid = "%s-%s" % (int(e[0]), int(e[1]))
        sal_dict["sal_metric"][guid] = float(e[3])
        sal_dict["inlier_count"][guid] = float(e[4])
        sal_dict["pred_3d_keypoints"][guid] = e[6:].reshape(j, 3).tolist()
        sal_dict["al_metric"][guid] = float(e[2])
        sal_dict["mkpe"][guid] = float(e[5])
    return sal_dict


_SMALL_KEYS = ("proj_matrices", "joint_valid", "3d_keypoints", "pose", "frame_id")


def _stage_batch(dp):
    """The batch's SMALL tensors (cameras, validity, ground truth, ids) as device tensors, uploaded on the caller's stream BEFORE the
    network is launched, through pinned staging (torch's caching host allocator) with non_blocking copies.  Why: score_batch /
    evaluate_mkpe run inside the side-stream context (parallel.PostStream), and a pageable host -> device copy issued there blocks the
    host until the side stream -- which has just waited for this batch's network -- drains, so the NEXT batch's network could not be
    enqueued meanwhile.  Returns a shallow copy of ``dp``; host-only runs (no HIP device) get ``dp`` back."""
    if not torch.cuda.is_available() or not isinstance(dp, dict):
        return dp
    dev = torch.device("cuda", torch.cuda.current_device())
    out = dict(dp)
    for k in _SMALL_KEYS:
        if k not in dp:
            continue
        t = torch.as_tensor(dp[k])
        if not t.is_cuda:
            t = t.contiguous()
            t = (t if t.is_pinned() else t.pin_memory()).to(dev, non_blocking=True)
        out[k] = t
    return out


class ActiveLearningStrategy:
    def __init__(self, al_cfg):
        self.al_cfg = al_cfg
        self.num_joints = al_cfg.DATA.NUM_JOINTS
        self.joint_root_index = 2 if al_cfg.DATA.TYPE == "panoptic" else 21  # strategy.py:34-37
        self.loss = Pose2DMeanSquaredError()

    # ---- model glue ---------------------------------------------------------------
    @staticmethod
    def _compute_batch_heatmap(pose_estimator, data):
        """strategy.py:772-782: (B,V,C,H,W) -> (B*V,C,H,W), frame-major / view-minor."""
        images = data["images"].cuda()
        c, h, w = images.shape[2], images.shape[3], images.shape[4]
        return pose_estimator(images.reshape([-1, c, h, w]))

    @staticmethod
    def _compute_batch_loss(gt_heatmap, heatmaps, per_view_joint_valid, loss):
        """strategy.py:762-770."""
        _, _, j, h, w = gt_heatmap.shape
        return loss.pose_2d_mse(
            heatmaps, gt_heatmap.reshape([-1, j, h, w]), per_view_joint_valid.reshape([-1, j, 1, 1])
        )

    def train_step(self, pose_estimator, optimizer, data, lr_scheduler=None):
        """Body of the training inner loop (strategy.py:460-487): zero_grad -> heat-maps ->
        masked MSE -> host guard (NaN / inf / > LOSS_CLIP_VALUE skips the step)
        -> backward -> optimizer step -> scheduler step.  Returns (loss_value, stepped)."""
        optimizer.zero_grad()
        heatmaps = self._compute_batch_heatmap(pose_estimator, data)
        gt = data["gt_heatmap"].cuda()
        pv = data["per_view_joint_valid"].to(torch.uint8).cuda()
        batch_loss = self._compute_batch_loss(gt, heatmaps, pv, self.loss)
        value = batch_loss.data.item()
        ok = not (math.isnan(value) or math.isinf(value) or value > self.al_cfg.TRAIN.LOSS_CLIP_VALUE)
        if ok:
            batch_loss.backward()
            optimizer.step()
        if lr_scheduler is not None:
            lr_scheduler.step()
        return value, ok

    # ---- per-frame scorers (reference signatures) ---------------------------------------
    def _score_one(self, kind, config, heatmaps, joint_valid):
        out, _, n_peaks, valid = score_heatmaps_batch(kind, config, heatmaps.unsqueeze(0), torch.as_tensor(joint_valid).reshape(1, -1))
        if kind == "BSB":
            bad = (n_peaks[0] < 2) & (valid[0].bool()[None, :])
            if bool(bad.any().item()):
                raise IndexError("list index out of range")  # strategy.py:1208 with < 2 peaks
        if bool((n_peaks < 0).any().item()):
            raise _lib.MvalError("peak list overflow (> 2048 local maxima in one heat-map)")
        v = out[0].item()
        if config == "AVG":
            return float(v)
        return np.float64(v) if kind == "HP" else np.float32(v)

    def _compute_mpe(self, heatmaps, joint_valid):
        return self._score_one("MPE", self.al_cfg.AL.MPE_CONFIG, heatmaps, joint_valid)

    def _compute_hp(self, heatmaps, joint_valid):
        return self._score_one("HP", self.al_cfg.AL.HP_CONFIG, heatmaps, joint_valid)

    def _compute_bsb(self, heatmaps, joint_valid):
        return self._score_one("BSB", self.al_cfg.AL.BSB_CONFIG, heatmaps, joint_valid)

    # ---- pool scoring ---------------------------------------------------------------
    def score_batch(self, heatmaps, dp):
        """Everything the reference does per sample inside _compute_sal_dict
        (strategy.py:1036-1094,1134), for a whole batch, on device.  heatmaps (B*V,J,h,w).
        Returns a (B, 6 + 3J) float64 HIP table:
        [pose, frame_id, al_metric, sal_metric, inlier_count, mkpe, keypoints_3d(3J)]."""
        cfg = self.al_cfg
        dev = heatmaps.device
        pose = torch.as_tensor(dp["pose"]).reshape(-1)
        b = pose.shape[0]
        _, j, hh, wh = heatmaps.shape
        hm = heatmaps.reshape(b, -1, j, hh, wh)
        joint_valid = torch.as_tensor(dp["joint_valid"]).reshape(b, j)
        strat = cfg.AL.STRATEGY
        scored = None
        if strat in ("MPE", "HP", "BSB") and not cfg.AL.USE_SOFTARGMAX:
            # one read of every heat-map for both the uncertainty statistic and the key-point decode
            scored = score_decode_heatmaps_batch(strat, getattr(cfg.AL, strat + "_CONFIG"), hm, joint_valid, cfg.POSE_ESTIMATOR.STRIDE)
        r = triangulation.triangulate_batch(
            hm, dp["proj_matrices"], cfg.POSE_ESTIMATOR.STRIDE, joint_valid,
            cfg.AL.USE_SOFTARGMAX, cfg.AL.USE_REPROJECTION_XE, cfg.AL.REPROJECTION_SIGMA,
            keypoints_2d=None if scored is None else scored[4],
        )
        pred32 = r["keypoints_3d"].to(torch.float32)  # torch.Tensor(results["keypoints_3d"]) (:1046)
        sal_metric = r["metric"].to(torch.float32).to(torch.float64)  # torch.Tensor([metric]) (:1061)
        if strat == "RANDOM":
            al = torch.cat([torch.rand(1) for _ in range(b)]).to(torch.float64).to(dev)
        elif strat == "TRIANGULATION":
            al = r["metric"].to(torch.float64)  # torch.tensor([np.float64]) keeps float64 (:1075)
        elif strat in ("MPE", "HP", "BSB"):
            conf = getattr(cfg.AL, strat + "_CONFIG")
            al, _, n_peaks, valid = scored[:4] if scored is not None else score_heatmaps_batch(strat, conf, hm, joint_valid)
            self._pending_checks.append((strat, n_peaks, valid))
            if conf == "AVG" or strat != "HP":
                al = al.to(torch.float32).to(torch.float64)  # torch.tensor(python float) is float32
        elif strat == "CORESET":
            al = torch.zeros(b, dtype=torch.float64, device=dev)
        else:
            raise NotImplementedError()
        gt = torch.as_tensor(dp["3d_keypoints"]).to(dev)
        mk = evaluation.mkpe_per_sample(pred32, gt, joint_valid.to(dev))
        self._pending_checks.append(("INLIER", r["inlier_count"], None))
        table = torch.cat(
            [
                pose.to(dev, torch.float64)[:, None],
                torch.as_tensor(dp["frame_id"]).reshape(b).to(dev, torch.float64)[:, None],
                al[:, None],
                sal_metric[:, None],
                r["inlier_count"].to(torch.float64)[:, None],
                mk.to(torch.float64)[:, None],
                pred32.to(torch.float64).reshape(b, 3 * j),
            ],
            dim=1,
        )
        return table

    def _compute_sal_dict(self, data_loader, pose_estimator):
        """strategy.py:1004-1147.  Same five OrderedDicts keyed "<pose>-<frame_id>", filled in
        the reference's gather order (batch -> sample -> rank), but with one device->host
        copy and (when torch.distributed is initialised) one all_gather for the whole pass."""
        sal_dict = {
            "al_metric": OrderedDict(),
            "sal_metric": OrderedDict(),
            "inlier_count": OrderedDict(),
            "pred_3d_keypoints": OrderedDict(),
            "mkpe": OrderedDict(),
        }
        self._pending_checks = []
        tables, sizes = [], []
        from .parallel import PostStream, world

        post = PostStream()  # a batch's decode / scoring / triangulation runs beside the next batch's network
        with torch.no_grad():
            for dp in data_loader:
                dp = _stage_batch(dp)  # (small tensors to the device before the network launch: no host-blocking copy on the side stream)
                heatmaps = self._compute_batch_heatmap(pose_estimator, dp)
                with post.batch(heatmaps, _lib.argmax_keys_of(heatmaps) if torch.is_tensor(heatmaps) and heatmaps.is_cuda else None, dp):
                    t = self.score_batch(heatmaps, dp)
                tables.append(t)
                sizes.append(t.shape[0])
        post.join()

        if not tables:
            from .parallel import _collectives_on

            if not _collectives_on():
                return sal_dict
            # an empty shard still takes part in the pass's collective (the other ranks would hang otherwise)
            j = self.num_joints
            dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
            tables = [torch.zeros((0, 6 + 3 * j), dtype=torch.float64, device=dev)]
        # the reference's per-sample error behaviour is checked once per pass -- AFTER the collectives (a rank that raised
        # before them would leave the others waiting); the flag travels with the size exchange
        err = None
        try:
            self._raise_deferred_errors()
        except Exception as e:  # noqa: BLE001
            err = e
        local = torch.cat(tables, dim=0)
        from .parallel import gather_tables

        per_rank, per_rank_sizes = gather_tables(local, sizes, error_flag=int(err is not None))  # host arrays + every rank's batch sizes
        if err is not None:
            raise err
        return tables_to_sal_dict(per_rank, per_rank_sizes, sal_dict)

    def _raise_deferred_errors(self):
        """Error behaviour of the reference's per-sample loop, checked once per pass."""
        for kind, t, valid in self._pending_checks:
            if kind == "INLIER":
                if bool((t < 0).any().item()):
                    raise ValueError("zero-size array to reduction operation minimum which has no identity")
            else:
                if bool((t < 0).any().item()):
                    raise _lib.MvalError("peak list overflow (> 2048 local maxima in one heat-map)")
                if kind == "BSB" and bool(((t < 2) & valid.bool()[:, None, :]).any().item()):
                    raise IndexError("list index out of range")
        self._pending_checks = []

    def select_al_guids(self, sal_dict, al_num_frames, labeled_dict=None):
        """Selection part of _sal_pseudo_labeling (strategy.py:932-949): NaN filter, then
        CORESET -> CoreSet(pred_3d_keypoints, labeled, root).select_batch(N), else
        heapq.nlargest(N, ...) (stable: ties keep gather order)."""
        al_metric_dict = {g: m for g, m in sal_dict["al_metric"].items() if not math.isnan(m)}
        if self.al_cfg.AL.STRATEGY == "CORESET":
            cs = coreset.CoreSet(sal_dict["pred_3d_keypoints"], labeled_dict, self.joint_root_index)
            return cs.select_batch(al_num_frames)
        return nlargest(al_num_frames, al_metric_dict, key=al_metric_dict.get)

    def select_sal_guids(self, sal_dict, al_guids, pseudo_label_guids, pseudo_num_frames, kmeans_centers=None,
                         device=None):
        """Pseudo-label filter of _sal_pseudo_labeling (strategy.py:952-1001): keep frames that were not
        sampled by AL, are not pseudo-labelled yet, have a finite sal_metric and MORE than SAL.INLIER_THRESHOLD
        inliers; sort ascending by sal_metric (stable); then either fill SAL.NUM_CLUSTERS pose clusters with
        pseudo_num_frames // NUM_CLUSTERS frames each (cluster of a frame = nearest of ``kmeans_centers``,
        (K, 3J) float64, to its root-relative pose -- ONE device launch for all candidates instead of a
        ``kmeans.predict`` per frame) or, without clusters, ``random.sample`` of the best 2 N (python's
        global RNG, exactly like the reference)."""
        import random

        al = set(al_guids)
        done = set(pseudo_label_guids)
        thr = self.al_cfg.SAL.INLIER_THRESHOLD
        sal_metric_dict = {
            g: m for g, m in sal_dict["sal_metric"].items()
            if g not in al and not math.isnan(m) and g not in done and sal_dict["inlier_count"][g] > thr
        }
        sal_guids = sorted(sal_metric_dict, key=sal_metric_dict.get)
        if kmeans_centers is None:
            return random.sample(sal_guids[: 2 * pseudo_num_frames], pseudo_num_frames)
        k = self.al_cfg.SAL.NUM_CLUSTERS
        if not sal_guids:
            return []
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        pose = torch.as_tensor(np.asarray([sal_dict["pred_3d_keypoints"][g] for g in sal_guids], dtype=np.float64)).to(device)
        n, j, rows = pose.shape
        feat = _lib.coreset_features(pose.contiguous(), self.joint_root_index, n, j, rows)
        centers = torch.as_tensor(np.asarray(kmeans_centers, dtype=np.float64)).to(device).contiguous()
        labels = _lib.nearest_center(feat, centers).cpu().tolist()
        counter = [0] * k
        per_cluster_count = pseudo_num_frames // k
        out = []
        for g, c in zip(sal_guids, labels):
            if counter[c] < per_cluster_count:
                counter[c] += 1
                out.append(g)
        return out

    # ---- evaluation core (strategy.py:597-636) ----------------------------------------
    def evaluate_mkpe(self, data_loader, pose_estimator):
        """_evaluate_all's MKPE path: heat-maps -> hard arg-max triangulation -> MPJPE over the
        whole loader (one size exchange + one packed data gather instead of 3 all_gathers per sample)."""
        preds, gts, valids, sizes = [], [], [], []
        from .parallel import PostStream

        post = PostStream()  # (decode + triangulation of a batch beside the next batch's network)
        with torch.no_grad():
            for dp in data_loader:
                dp = _stage_batch(dp)
                hm = self._compute_batch_heatmap(pose_estimator, dp)
                b = torch.as_tensor(dp["pose"]).reshape(-1).shape[0] if "pose" in dp else dp["images"].shape[0]
                _, j, hh, wh = hm.shape
                jv = torch.as_tensor(dp["joint_valid"]).reshape(b, j)
                with post.batch(hm, _lib.argmax_keys_of(hm) if hm.is_cuda else None, dp, jv):
                    r = triangulation.triangulate_batch(hm.reshape(b, -1, j, hh, wh), dp["proj_matrices"],
                                                        self.al_cfg.POSE_ESTIMATOR.STRIDE, jv)
                    preds.append(r["keypoints_3d"].to(torch.float32))
                    gts.append(torch.as_tensor(dp["3d_keypoints"]).to(hm.device, torch.float32))
                    valids.append(jv.to(hm.device, torch.float32))
                sizes.append(b)
        post.join()
        from .parallel import all_gather_reference_order as gather

        j = self.num_joints if not preds else preds[0].shape[1]
        rows = 3 if not gts else gts[0].shape[1]
        if preds:
            # ONE packed table per pass: [pred 3J | gt rows*J | valid J] per sample
            local = torch.cat([torch.cat(preds).reshape(-1, 3 * j), torch.cat(gts).reshape(-1, rows * j),
                               torch.cat(valids).reshape(-1, j)], dim=1)
        else:  # an empty shard still takes part in the collective
            dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
            local = torch.zeros((0, (4 + rows) * j), dtype=torch.float32, device=dev)
        # the reference appends its per-sample all_gathers batch by batch, sample by sample, rank by rank
        # (strategy.py:600-636): same row order here, so the float32 sample-order sums of compute_mkpe match
        table = gather(local, sizes)
        pred = table[:, : 3 * j].reshape(-1, j, 3)
        gt = table[:, 3 * j : (3 + rows) * j].reshape(-1, rows, j)
        valid = table[:, (3 + rows) * j :]
        out, _ = _lib.mkpe(pred.contiguous(), gt.contiguous(), valid.contiguous(), pred.shape[0], pred.shape[1], gt.shape[1])
        self._eval_tables = (pred, gt, valid)
        return out

    def evaluate_all(self, data_loader, pose_estimator):
        """_evaluate_all (strategy.py:584-649): MKPE + 3-D PCK at 1..5 mm (+ PCKh for panoptic data) over the
        whole loader; returns the reference's result dict."""
        mkpe = self.evaluate_mkpe(data_loader, pose_estimator)
        pred, gt, valid = self._eval_tables
        j = pred.shape[1]
        thresholds, pcks = evaluation.compute_3d_pck_figure(pred, gt, valid, j)
        results = {"mkpe": mkpe.item(), "thresholds": thresholds, "pcks": pcks}
        if self.al_cfg.DATA.TYPE == "panoptic":
            t, p = evaluation.compute_3d_pckh_figure(pred, gt, j)
            results["pckh_thresholds"], results["pckh_pcks"] = t, p
        return results
