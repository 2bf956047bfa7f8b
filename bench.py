#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: frames x views / s, heat-map -> triangulated 3-D.

    python bench.py --gpus N --steps K --warmup W

N > 1 without torchrun's RANK in the environment: this process starts the N ranks itself (as CHILD processes,
``python -m torch.distributed.run --nproc-per-node N bench.py ...``, before anything here touches the GPU) and relays
rank 0's line; under torchrun it is one of the ranks.  Fewer than N devices, or WORLD_SIZE != N, is an error.

Workload (BASELINE.json configs[1], "c2"): HRNet-W32, 4 views, 256x256, batch 32 frames
(128 images per step and per GPU), forward-only heat-maps + hard arg-max decode + pairwise
RANSAC-DLT triangulation + reprojection metric.  Random synthetic weights / frames / cameras
(multi_view_active_learning_amd/synth.py); inputs are resident in HBM before the timed region.

One step = one pass of the hot path over one batch.  Frames are independent, so ranks shard
frames with NO data-path collective (weak scaling: per-GPU work is fixed).

Prints ONE JSON line on rank 0 -- at most 4 096 characters (compact_line; round 5's 21.9 KB line fell out of the driver's stdout tail) -- with the
driver's contract fields plus
  roofline     the dominant kernel family by time: {kernel (a label), bound, achieved, peak, unit, frac, traffic, seconds_in_kernel_per_step}.
               For C2 the fused 3x3 stride-1 convs on the fp16 matrix cores: algorithmic conv FLOPs of its launches in one step / the time
               spent in them (hipEvents around every launch on the launch stream, mval_net_forward_timed) against the peak of the split
               in use (three fp16 MFMA products per fp32 product: 2500 / 3 = 833 TFLOP/s); traffic = HBM bytes per launch from the committed
               rocprofv3 counter passes of this command (profiles/r06);
  exact_modes  ms per step of the same workload with the bit-faithful conv kernels (bf3: exact 3-way bf16 split; fp32: exact-fp32 MFMA) and
               with the fp32-activation fp16 split (h2), 20 steps each, outside the headline's timed region;
  companions   (default c2 run at one GPU only) {ms_per_step, value, kernel, bound, frac} of BASELINE configs[2] (C3 training step), configs[3]'s
               slice (C4) and C2 from uint8 host crops, each run by a child process of this command after the headline (>= 2 s timed each);
  cpu_baseline the CPU oracle (stock torch fp32 HRNet-W32 + numpy RANSAC-DLT restatement, oracle/) timed on the host on a bounded sample of
               the same workload; "parity_sample" compares the HIP path with the oracle on that sample's first frames (heat-map max error,
               2-D key-point equality, MPJPE and worst-joint delta in mm) -- outside every timed region.
Everything else the run measured -- the other kernel families each against its own bound, pass counts, notes, per-rank attribution, the
companions' own records -- goes to the detail record (--detail-out, default bench_detail.json beside this script).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32-input MFMA
PEAK_HBM_GBPS = 8000.0  # same guide: HBM3E
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: dense bf16 / fp16 MFMA (the 5 PF headline includes 2:1 sparsity)
SPLIT_PRODUCTS = {"p2": 3, "h2": 3, "bf3": 6}  # MFMA products per algorithmic product of the 16-bit splits
PARITY_UNPINNED = ["soft_argmax (kornia absent)", "MPE / BSB peak_local_max: pinned on scikit-image 0.18.3 vectors except the order among equal intensities "
                   "(numpy's unstable argsort)"]
# `dtype` of the line = the arithmetic the convs compute in (a label, <= 120 characters); the long form goes to the detail record
DTYPE_LABEL = {"fp32": "f32 (exact-fp32 MFMA)",
               "bf3": "f32 (convs: exact 3-way bf16 split, 6 bf16 MFMA products per fp32 product, fp32 accumulate)",
               "h2": "f32 (convs: scaled 2-way fp16 split, 3 fp16 MFMA products per fp32 product, fp32 accumulate)",
               "p2": "f32 (fp16 (h,l)-pair activations+weights: 3 fp16 MFMA products per fp32 product, fp32 accumulate)"}
DTYPE_NOTE = {"fp32": "v_mfma_f32_16x16x4_f32 everywhere",
              "bf3": "fp32 values as exact 3-way bf16 splits on the bf16 MFMA, fp32 accumulate",
              "h2": "fp32 values as scaled 2-way fp16 splits on the fp16 MFMA, fp32 accumulate; measured at the exact-fp32 MFMA chain's error",
              "p2": "activations held as scaled fp16 (h, l) pairs = 22-bit significands, weights likewise; convs: 3 fp16 MFMA products per fp32 "
                    "product, lo x lo dropped, fp32 accumulate; measured below the exact-fp32 MFMA chain's error"}
FLOP_PER_IMAGE = {"hrnet_w32_256": 20.387e9}  # conv FLOPs (2*MAC) per frame x view

WORKLOADS = {
    # name: (arch, views, H, W, frames per step, joints, train?)
    # BASELINE.json configs[0], the reference's own CPU-runnable plumbing case (8 images per step:
    # launch-latency bound on a GPU; "c1x16" is the same network at 64 frames per step)
    "c1": dict(arch="resnet50", v=2, h=256, w=192, frames=4, j=19, train=False,
               desc="PoseResNet-50 2-view 256x192 batch-4 forward heat-maps + arg-max + RANSAC-DLT triangulation"),
    "c1x16": dict(arch="resnet50", v=2, h=256, w=192, frames=64, j=19, train=False,
                  desc="PoseResNet-50 2-view 256x192 batch-64 forward heat-maps + arg-max + RANSAC-DLT triangulation"),
    "c2": dict(arch="hrnet_w32", v=4, h=256, w=256, frames=32, j=19, train=False,
               desc="HRNet-W32 4-view 256x256 batch-32 forward heat-maps + arg-max + RANSAC-DLT triangulation"),
    # BASELINE.json configs[2]: same shapes, one training step (train-mode BN forward, masked MSE,
    # backward, Adam) -- selectable for measurement; the driver's default line stays c2
    "c3": dict(arch="hrnet_w32", v=4, h=256, w=256, frames=32, j=19, train=True,
               desc="HRNet-W32 4-view 256x256 batch-32 training step (masked-MSE backward + Adam)"),
    # BASELINE.json configs[3]/[4] per-GPU slice: HRNet-W48, 8 views, 384x288; one step = 8 frames
    # (64 images) of the unlabeled pool: heat-maps + arg-max + RANSAC-DLT + MPE entropy scoring
    "c4": dict(arch="hrnet_w48", v=8, h=384, w=288, frames=8, j=19, train=False, score="MPE",
               desc="HRNet-W48 8-view 384x288 pool scoring: heat-maps + triangulation + MPE entropy, 8 frames/step"),
    # BASELINE.json configs[4] shape: core-set selection pass over a FIXED pool sharded across the
    # ranks (strong scaling): per-rank heat-maps + triangulation, ONE RCCL all_gather of the fp32
    # 3-D predictions, then the replicated k-center (100 picks against 200 labeled poses)
    "c5": dict(arch="hrnet_w48", v=8, h=384, w=288, frames=8, j=19, train=False, pool=256, labeled=200, picks=100,
               desc="HRNet-W48 8-view 384x288 core-set pass: 256-frame pool sharded over ranks, all_gather of "
                    "3-D predictions, k-center select 100"),
}
FLOP_PER_IMAGE["hrnet_w48_384x288"] = 70.615e9


def build_model(arch, j, dev, seed=0):
    from multi_view_active_learning_amd import synth
    from multi_view_active_learning_amd.pose_estimators import PoseHighResolutionNet, PoseResNet, hrnet_w48

    if arch == "hrnet_w32":
        m = PoseHighResolutionNet(j)
    elif arch == "hrnet_w48":
        m = PoseHighResolutionNet(j, hrnet_cfg=hrnet_w48())
    else:
        m = PoseResNet(j)
    sd = synth.synthetic_state_dict(m._graph.param_shapes(), seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.to(dev).eval(), sd


def cpu_baseline(wl, sd_np, seconds_target=20.0):
    """CPU oracle on a bounded sample: frames of the same shape until ~seconds_target."""
    from multi_view_active_learning_amd import synth
    from oracle import geometry, models

    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    v, h, w, j = wl["v"], wl["h"], wl["w"], wl["j"]
    frames_per_call = max(2, 32 // v)  # >= 32 images per call: small calls starve a many-core host
    imgs = torch.from_numpy(synth.images(123, frames_per_call, v, h, w)).reshape(-1, 3, h, w)
    proj = np.stack([synth.ring_cameras(v, h, w, seed=s) for s in range(frames_per_call)])
    valid = np.ones(j, dtype=bool)
    done, t0 = 0, time.perf_counter()
    first = None
    with torch.no_grad():
        if wl["arch"] == "resnet50":
            fwd = lambda x: models.pose_resnet_forward(sd, x)
        else:
            arch = models.HRNET_W48 if wl["arch"] == "hrnet_w48" else models.HRNET_W32
            fwd = lambda x: models.hrnet_forward(sd, x, arch)
        fwd(imgs[:v])  # warm-up (page-in, thread pool)
        # the fastest thread count for this host (oversubscription made 128 threads slower than 16 in round 1)
        ncpu = os.cpu_count() or 1
        tried = {}
        for nt in sorted({t for t in (16, 32, 64, 128) if t <= ncpu} | {min(ncpu, 8)}):
            torch.set_num_threads(nt)
            fwd(imgs[:v])
            t0 = time.perf_counter()
            fwd(imgs)  # (the call size of the timed loop below: the two figures are comparable)
            tried[nt] = time.perf_counter() - t0
        best = min(tried, key=tried.get)
        torch.set_num_threads(best)
        t0 = time.perf_counter()
        while True:
            hm = fwd(imgs).numpy().reshape(frames_per_call, v, j, h // 4, w // 4)
            res = [geometry.triangulation(hm[b], proj[b], 4, valid) for b in range(frames_per_call)]
            if first is None:
                first = dict(heatmaps=hm, keypoints_3d=np.stack([r["keypoints_3d"] for r in res]),
                             keypoints_2d=np.stack([r["keypoints_2d"] for r in res]))
            done += frames_per_call
            el = time.perf_counter() - t0
            if el > seconds_target or done >= 64:
                break
    return dict(
        _check=dict(images=imgs, proj=proj, **first),
        value=done * v / el, unit="frames*views/s", cores=torch.get_num_threads(), kind="port", host_cpu=host_cpu_model(),
        host_logical_cpus=os.cpu_count(),
        sample=f"{done} frames x {v} views of the same workload, {el:.1f} s, torch-CPU fp32 net + numpy RANSAC-DLT",
        sample_note=f"{done * v} images, stock torch fp32 {wl['arch']} + numpy RANSAC-DLT (oracle/), {torch.get_num_threads()} threads "
                    f"(best of {sorted(tried)}; {frames_per_call * v} images per call)",
        threads_tried={str(k): round(frames_per_call * v / t, 2) for k, t in tried.items()},
        threads_tried_note="frames*views/s of the network alone per thread count, same images per call as `value` (which adds the "
                           "numpy RANSAC-DLT)",
    )


def cpu_baseline_train(wl, sd_np, seconds_target=20.0):
    """CPU oracle training step (stock torch fp32 HRNet-W32 in train mode + masked MSE + autograd backward, the
    reference's inner loop strategy.py:460-487 without the optimizer) on a bounded sample of the same shapes."""
    from multi_view_active_learning_amd import synth
    from oracle import models

    v, h, w, j = wl["v"], wl["h"], wl["w"], wl["j"]
    frames_per_call = 2
    x = torch.from_numpy(synth.images(123, frames_per_call, v, h, w)).reshape(-1, 3, h, w)
    gt = torch.rand(x.shape[0], j, h // 4, w // 4, generator=torch.Generator().manual_seed(11))
    arch = models.HRNET_W48 if wl["arch"] == "hrnet_w48" else models.HRNET_W32
    check = {}

    def step():
        sd = {k: torch.from_numpy(a).clone().requires_grad_(a.dtype == np.float32 and "running" not in k) for k, a in sd_np.items()}
        hm = models.hrnet_forward(sd, x, arch, training=True)
        loss = models.pose_2d_mse(hm, gt)
        loss.backward()
        if not check:  # the parity sample of the line: this step's loss, heat-maps and two gradients, compared with the HIP path by main()
            check.update(images=x, gt=gt, loss=float(loss.detach()), heatmaps=hm.detach().numpy().copy(),
                         grads={k: sd[k].grad.numpy().copy() for k in ("final_layer.weight", "final_layer.bias", "conv1.weight")})

    ncpu = os.cpu_count() or 1
    tried = {}
    for nt in sorted({t for t in (16, 32, 64) if t <= ncpu} | {min(ncpu, 8)}):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        step()
        tried[nt] = time.perf_counter() - t0
    best = min(tried, key=tried.get)
    torch.set_num_threads(best)
    done, t0 = 0, time.perf_counter()
    while True:
        step()
        done += frames_per_call
        el = time.perf_counter() - t0
        if el > seconds_target or done >= 16:
            break
    return dict(_check=check, value=done * v / el, unit="frames*views/s", cores=best, kind="port", host_cpu=host_cpu_model(),
                host_logical_cpus=os.cpu_count(),
                sample=f"{done} frames x {v} views of the same shapes, {el:.1f} s, torch-CPU fp32 fwd + MSE + autograd bwd",
                sample_note=f"stock torch fp32 train-mode forward + masked MSE + autograd backward (oracle/), {best} threads "
                            f"(best of {sorted(tried)}); no optimizer step",
                threads_tried={str(k): round(frames_per_call * v / t, 2) for k, t in tried.items()})


def train_bn_bytes(plan, n):
    """Algorithmic bytes per step of the BatchNorm kernels of a training plan, op by op from the plan's own flags (engine_train.TrainPlan,
    csrc/net_train.hip): every tensor pass a kernel makes counts 4 bytes per element (fp32 NHWC and the fp16 (h, l) planes alike), mask
    bytes 1 byte per 4 elements.  Returns bytes and, per kernel, the pass counts in units of (tensor passes summed over the ops)."""
    z_el = o_el = 0.0
    apply_b = bwd_b = stats_b = 0.0
    ap = dict(read_z=0, read_res=0, write_fp32=0, write_planes=0, write_mask=0)
    bp = dict(reduce_read_gout=0, reduce_read_z=0, reduce_read_out_or_mask=0, reduce_rw_res_grads=0, apply_read_g=0, apply_read_z=0,
              apply_read_out_or_mask=0, apply_write_fp32=0, apply_write_planes=0)
    n_bn = 0
    # (round 6) ops whose apply runs inside their reader's staging (z_out) move no apply bytes; ops whose backward reduction was kept by their
    # reader's data-gradient epilogue (the op in front of one with MVAL_TRAIN_BSUM) move no reduction bytes
    presummed = {i - 1 for i, o in enumerate(plan.ops) if o.p2_flags & 4096}
    for idx, o in enumerate(plan.ops):
        if not o.has_bn:
            continue
        n_bn += 1
        op = o.op
        ze = float(n) * op.hout * op.wout * op.cout
        oe = float(n) * (op.hout << op.up) * (op.wout << op.up) * op.cout
        z_el += ze
        nres = int(op.res1_off >= 0) + int(op.res2_off >= 0)
        o_el += oe * (1 + nres)
        # forward apply
        p2_only = bool(o.p2_flags & 2)
        if not o.z_out:
            apply_b += 4 * ze + 4 * oe * nres + (0 if p2_only else 4 * oe) + (4 * oe if o.out_p2_off > 0 else 0) + (oe / 4 if o.mask_off > 0 else 0)
            ap["read_z"] += 1; ap["read_res"] += nres; ap["write_fp32"] += int(not p2_only); ap["write_planes"] += int(o.out_p2_off > 0)
            ap["write_mask"] += int(o.mask_off > 0)
        # backward
        fused = not (o.p2_flags & 64) and op.up == 0 and op.cout % 4 == 0
        first = o.first_touch >> 1
        acc_res = sum(1 for k_, off in ((1, o.gres1_off), (2, o.gres2_off)) if off >= 0 and not (first & k_))  # read-modify-write
        st_res = sum(1 for k_, off in ((1, o.gres1_off), (2, o.gres2_off)) if off >= 0)
        if fused:
            mode_ = 0 if not op.relu else (3 if o.mask_off > 0 else 1) if nres else 2
            mask_b = {0: 0.0, 1: 4 * oe, 2: 0.0, 3: oe / 4}[mode_]
            from_slot = st_res > 0 and bool(first & 3)  # apply re-reads the residual slot this op stored the masked gradient in: no mask
            dz_p2 = bool(o.p2_flags & 4) and o.gin_off >= 0
            w32 = not (dz_p2 and (o.p2_flags & 8))
            # (a pre-summed op with residuals: the apply pass masks from the kept bits and scatters the residual gradients itself)
            red = 4 * oe * (acc_res + st_res) if idx in presummed else 4 * oe + 4 * ze + mask_b + 4 * oe * (acc_res + st_res)
            if idx in presummed:
                from_slot = False
            bwd_b += red + 4 * oe + 4 * ze + (0 if from_slot else mask_b) + (4 * ze if w32 else 0) + (4 * ze if dz_p2 else 0)
            if idx not in presummed:
                bp["reduce_read_gout"] += 1; bp["reduce_read_z"] += 1; bp["reduce_read_out_or_mask"] += mask_b / (4 * oe)
                bp["reduce_rw_res_grads"] += acc_res + st_res
            bp["apply_read_g"] += 1; bp["apply_read_z"] += 1; bp["apply_read_out_or_mask"] += 0 if from_slot else mask_b / (4 * oe)
            bp["apply_write_fp32"] += int(w32); bp["apply_write_planes"] += int(dz_p2)
        else:  # round-3 pair: reduce (gout at the upsampled size [, out], z, residual gradients, masked / window-summed copy to gz), apply in place
            bwd_b += 4 * oe * (1 + int(bool(op.relu))) + 4 * ze + 4 * oe * (acc_res + st_res) + 4 * ze + 3 * 4 * ze
            bp["reduce_read_gout"] += oe / ze; bp["reduce_read_z"] += 1; bp["reduce_read_out_or_mask"] += int(bool(op.relu)) * oe / ze
            bp["reduce_rw_res_grads"] += (acc_res + st_res) * oe / ze
            bp["apply_read_g"] += 2; bp["apply_read_z"] += 1; bp["apply_write_fp32"] += 1  # (+ the reduce's write of gz counted under apply_read_g)
    # ops whose conv epilogue keeps no statistics partials (3-channel stem: in_nchw) still read z once
    for o in plan.ops:
        if o.has_bn and (o.op.in_nchw or (o.p2_flags & 128)):
            stats_b += 4.0 * n * o.op.hout * o.op.wout * o.op.cout
    rnd = lambda d: {k_: round(float(v_), 1) for k_, v_ in d.items()}
    return dict(apply=apply_b, bwd=bwd_b, stats=stats_b, apply_passes=rnd(ap), bwd_passes=rnd(bp), n_bn=n_bn,
                applies_in_reader=sum(int(o.z_out) for o in plan.ops), reductions_in_reader_dgrad=len(presummed),
                apply_r3=4.0 * (z_el + o_el), bwd_r3=2.0 * 4.0 * o_el + 4.0 * 4.0 * z_el)


def train_rooflines(model, step, frames, v, mode):
    """Per-kernel-family rooflines of the training step: mval_train_timing brackets every launch group with hipEvents
    (measurement mode, outside the timed region); conv families against the matrix-core peak of the split in use,
    the BatchNorm streams against HBM with their algorithmic bytes."""
    import ctypes as C

    from multi_view_active_learning_amd import _lib

    lib = _lib.lib()
    plan = next(iter(model._train_plans.values()))
    ms = (C.c_float * 6)()
    reps = 3
    _lib._check(lib.mval_train_timing(ms), "mval_train_timing")
    try:  # (the library holds a pointer into `ms` while armed)
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
    finally:
        _lib._check(lib.mval_train_timing(None), "mval_train_timing")
    t = [float(x) * 1e-3 / reps for x in ms]
    n = frames * v
    fl_fwd = sum(float(lib.mval_op_flops(C.byref(o.op), C.c_int(n))) for o in plan.ops)
    fl_dgrad = sum(float(lib.mval_op_flops(C.byref(o.op), C.c_int(n))) for o in plan.ops if o.gin_off >= 0)
    peak = PEAK_BF16_MFMA_TFLOPS / SPLIT_PRODUCTS.get(mode, 6) if mode != "fp32" else PEAK_FP32_MFMA_TFLOPS
    note = (f"dense 16-bit MFMA peak 2500 TFLOP/s / {SPLIT_PRODUCTS.get(mode, 6)} products (layers the fp16 split does not cover run "
            "the bf16x3 or exact-fp32 kernels: their time is in, the peak is the default split's)")
    bn = train_bn_bytes(plan, n)
    # HBM bytes per step by kernel from the committed rocprofv3 counter passes of THIS workload (tools/collect_profiles.sh: separate
    # --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE x2 gfx950 correction; Infinity-Cache hits are counted by those counters)
    pmc, pmc_src = {}, None
    for r_ in ("r06", "r05", "r04"):
        p_ = os.path.join(ROOT, "profiles", r_, f"bench_c3_{r_.replace('0', '')}_summary.json")
        if os.path.exists(p_):
            try:
                with open(p_) as f:
                    d = json.load(f)
                steps_traced = max(1, int(d.get("launch_counts", {}).get("adam_step_kernel", 0)))
                for row in d["hbm_traffic_by_instantiation"]:
                    pmc[row["kernel"]] = row["launches"] * row["total_bytes"] / steps_traced
                pmc_src = os.path.relpath(p_, ROOT)
                break
            except (OSError, KeyError, ValueError):
                pmc = {}

    def traffic_of(prefixes):
        v_ = sum(b for k_, b in pmc.items() if k_.startswith(prefixes))
        return round(v_) if v_ else None

    def mf(name, f, tt, prefixes=()):
        return dict(bound="mfma", achieved=round(f / tt / 1e12, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(f / tt / 1e12 / peak, 4),
                    traffic=traffic_of(prefixes) if prefixes else None, peak_note=note, seconds_in_kernel_per_step=round(tt, 6), flops_per_step=f, **_kn(name))

    def hb(name, b, tt, prefixes=(), **extra):
        return dict(bound="hbm", achieved=round(b / tt / 1e9, 1), peak=PEAK_HBM_GBPS, unit="GB/s", frac=round(b / tt / 1e9 / PEAK_HBM_GBPS, 4),
                    traffic=traffic_of(prefixes) if prefixes else None, traffic_source=pmc_src, **_kn(name),
                    peak_note="HBM3E ~8 TB/s (guide); ~6.3 TB/s achievable", seconds_in_kernel_per_step=round(tt, 6), bytes_per_step=b, **extra)

    fams = [
        mf("conv forward (conv_p2_kernel<..., EPI 3> over fp16-pair activations -- raw fp32 NHWC z + the batch-statistics sums of "
           "its persistent workgroups; conv_split_kernel / conv_mfma_kernel for the shapes P2 does not cover)", fl_fwd, t[0]),
        dict(bound="latency", achieved=None, peak=None, unit=None, frac=None, traffic=None,
             **_kn("BatchNorm statistics (finalize of the conv epilogues' float64 partials -> mean / invstd / running stats: no pass over z; "
                   "ops whose conv keeps no partials -- the stem -- run bn_stats_partial over z)"),
             peak_note="launch- / latency-bound: one small dependent launch per BatchNorm", launches_per_step=bn["n_bn"],
             seconds_in_kernel_per_step=round(t[1], 6), bytes_per_step=bn["stats"]),
        hb("BatchNorm apply (bn_apply_fwd[_p2]: z [+ residuals] -> normalise + residuals + ReLU (+ upsample) -> out as P2 planes and / or fp32 "
           "[+ ReLU mask bytes])", bn["apply"], t[2], ("bn_apply_fwd",), passes=bn["apply_passes"],
           bytes_round3_count=bn["apply_r3"], frac_round3_count=round(bn["apply_r3"] / t[2] / 1e9 / PEAK_HBM_GBPS, 4)),
        hb("BatchNorm backward (bn_bwd_reduce2 + finalize + bn_bwd_apply2[_p2]; bn_bwd_reduce + bn_bwd_apply for up-sampled terms / the "
           "final layer: gout, z [, mask] -> residual gradients, dgamma, dbeta, dz as P2 planes and / or fp32)", bn["bwd"], t[3], ("bn_bwd",),
           passes=bn["bwd_passes"], bytes_round3_count=bn["bwd_r3"], frac_round3_count=round(bn["bwd_r3"] / t[3] / 1e9 / PEAK_HBM_GBPS, 4)),
        mf("weight gradient (conv_wgrad_bf3_kernel split-K, both operands staged from the P2 planes, + float64 slab reduction)", fl_fwd, t[4],
           ("conv_wgrad", "wgrad_reduce", "slab_reduce")),
        mf("data gradient (conv_p2_kernel<..., EPI 3> accumulating into the fp32 gradient slot; stride 2: four 2x2 parity convs)", fl_dgrad, t[5]),
    ]
    for f_ in fams:
        if f_["bound"] == "hbm":
            f_["bytes_note"] = ("bytes_per_step = what THIS round's kernels read and write, summed op by op from the plan (`passes`: the same in units "
                                "of one fp32 tensor pass at the op's output size, per kernel); bytes_round3_count / frac_round3_count = the byte "
                                "count round 3's kernels moved, for comparison across rounds only; traffic = rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE "
                                "per step from traffic_source (counts Infinity-Cache hits)")
    fams.sort(key=lambda f: -f["seconds_in_kernel_per_step"])
    top = fams[0]
    top["other_kernels"] = fams[1:]
    top["sum_of_families_ms"] = round(sum(t) * 1e3, 3)
    return top


LINE_LIMIT = 4096  # the driver keeps an 8 000-character tail of stdout: the line it parses stays well inside it


def _short(s, n):
    """A label for the line: the text before the first explanatory parenthesis, at most n characters."""
    if s is None:
        return None
    s = str(s).split(" (")[0].split("; ")[0].strip()
    return s if len(s) <= n else s[: n - 1].rstrip() + "~"


def _num(x, nd=4):
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, (float, np.floating)):
        x = float(x)
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return round(x, nd) if abs(x) >= 1e-3 or x == 0 else float(f"{x:.3e}")
    return x


_ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "seconds_in_kernel_per_step")


def _kn(desc):
    """A kernel family's description -> {"kernel": its name (what stands before the first parenthesis), "kernel_note": the prose}: JSON values
    stay labels, the explanation travels beside them in the detail record."""
    desc = str(desc)
    i = desc.find(" (")
    if i < 0:
        return {"kernel": desc}
    note = desc[i + 1:].strip()
    depth, closes_at = 0, -1
    for k, ch in enumerate(note):  # strip the outer parentheses only when ONE group spans the whole note
        depth += ch == "("
        depth -= ch == ")"
        if depth == 0:
            closes_at = k
            break
    return {"kernel": desc[:i].strip(), "kernel_note": note[1:-1] if closes_at == len(note) - 1 else note}


def _roof_line(roof):
    if not roof:
        return None
    r = {"kernel": _short(roof.get("kernel"), 60)}
    r.update({k: _num(roof.get(k)) for k in _ROOF_KEYS})
    r["seconds_in_kernel_per_step"] = _num(roof.get("seconds_in_kernel_per_step"), 6)
    for k in ("launches_per_step", "avg_launch_us"):
        if roof.get(k) is not None:
            r[k] = _num(roof[k], 2)
    return r


def compact_line(out, detail_path=None):
    """The ONE line of stdout: the driver's contract fields, the dominant kernel's roofline, the CPU baseline with its parity
    sample, the bit-strict companions' times and one {ms_per_step, value, frac} triple per companion workload -- flat, numbers
    and short labels only, at most LINE_LIMIT characters, strict JSON (no NaN / Infinity).  Everything else the run measured
    (other kernels, pass counts, notes, per-rank attribution) is the detail record (--detail-out)."""
    cfg = out.get("config") or {}
    line = {k: _num(out.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "timed_repeats",
                                          "higher_is_better", "scaling", "vs_baseline")}
    line["dtype"] = str(out.get("dtype"))[:120]
    line["data"] = _short(out.get("data"), 40)
    c = {"workload": str(cfg.get("workload", ""))[:160]}
    for k in ("frames_per_step_per_gpu", "views", "images_per_step_per_gpu", "pool_frames", "frames_per_batch"):
        if k in cfg:
            c[k] = cfg[k]
    if "parallelism" in cfg:
        c["parallelism"] = _short(cfg["parallelism"], 60)
    line["config"] = c
    line["roofline"] = _roof_line(out.get("roofline"))
    cpu = out.get("cpu_baseline")
    if cpu:
        cb = {k: _num(cpu.get(k), 3) for k in ("value", "unit", "cores", "kind")}
        cb["host_cpu"] = _short(cpu.get("host_cpu"), 48)
        cb["sample"] = str(cpu.get("sample"))[:110]
        ps = cpu.get("parity_sample")
        if ps:
            flat = {}
            for k, v_ in ps.items():
                if isinstance(v_, dict):  # (the training sample's three gradient errors: the worst one)
                    vals = [x for x in v_.values() if isinstance(x, (int, float))]
                    if vals:
                        flat[k + "_max"] = _num(max(vals))
                elif isinstance(v_, (int, float, bool)):
                    flat[k] = _num(v_)
            cb["parity_sample"] = flat
        line["cpu_baseline"] = cb
    em = out.get("exact_modes")
    if em:
        line["exact_modes"] = {k: (_num(v_, 3) if isinstance(v_, (int, float)) else "failed") for k, v_ in em.items() if k != "note"}
    comp = out.get("companions")
    if comp:
        cc = {}
        for name, d in comp.items():
            if not isinstance(d, dict):
                continue
            if "error" in d:
                cc[name] = {"error": str(d["error"])[:80]}
            else:
                rf = d.get("roofline") or {}
                cc[name] = {"ms_per_step": _num(d.get("ms_per_step"), 3), "value": _num(d.get("value"), 1),
                            "kernel": _short(rf.get("kernel"), 40), "bound": rf.get("bound"), "frac": _num(rf.get("frac"))}
                if rf.get("whole_step_tflops_3x_forward") is not None:  # (training: 3 x forward conv FLOPs / step time)
                    cc[name]["step_tflops_3x_fwd"] = _num(rf["whole_step_tflops_3x_forward"], 1)
                if isinstance(d.get("cpu_baseline"), dict):
                    cc[name]["cpu_value"] = _num(d["cpu_baseline"].get("value"), 2)
        line["companions"] = cc
    att = out.get("attribution")
    if att:
        a = {}
        pr = att.get("per_rank_s") or {}
        a["per_rank_s"] = {k: pr.get(k) for k in ("min", "mean", "max") if k in pr}
        if len(pr.get("all", [])) <= 8:
            a["per_rank_s"]["all"] = pr.get("all")
        for k in ("compute_s", "gather_s", "select_s", "allreduce_exposed_s"):
            if k in att:
                a[k] = att[k]
        line["attribution"] = a
    if out.get("input_inclusive"):
        line["input_inclusive"] = {"host_to_device_GBps": out["input_inclusive"].get("host_to_device_GBps")}
    if out.get("picks_crc") is not None:
        line["picks_crc"] = out["picks_crc"]
    if detail_path:
        line["detail"] = detail_path
    # the limit is a contract, not a hope: drop the optional groups, least important first, until the line fits
    for drop in (None, "input_inclusive", "attribution", "companions", "exact_modes", "detail"):
        if drop:
            line.pop(drop, None)
        s = json.dumps(line, allow_nan=False, separators=(", ", ": "))
        if len(s) <= LINE_LIMIT:
            return s
    raise RuntimeError(f"bench line is {len(s)} characters with every optional group dropped")


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def path_kernels(dev, frames, v, j, hh, wh):
    """The non-conv kernels of the path at this workload's sizes, each against its own bound: events
    on torch's current stream, which is the stream the wrappers launch on."""
    from multi_view_active_learning_amd import _lib, synth
    from multi_view_active_learning_amd.utils.coreset import CoreSet

    def timed(fn, reps=30):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    def hbm(name, t, nbytes, note):
        return dict(bound="hbm", achieved=round(nbytes / t / 1e9, 1), peak=PEAK_HBM_GBPS, unit="GB/s",
                    frac=round(nbytes / t / 1e9 / PEAK_HBM_GBPS, 4), traffic=None, peak_note=note, **_kn(name),
                    launches_per_step=1, avg_launch_us=round(t * 1e6, 2), bytes_per_step=nbytes,
                    seconds_in_kernel_per_step=round(t, 7))

    out = []
    hm = torch.rand(frames, v, j, hh, wh, device=dev)
    valid = torch.ones(frames, j, dtype=torch.uint8, device=dev)
    n_maps, map_bytes = frames * v * j, frames * v * j * hh * wh * 4.0
    keys = torch.zeros((frames * v, _lib.ARGMAX_SLOTS, j), dtype=torch.int64, device=dev)
    t = timed(lambda: _lib.argmax_from_keys(keys, valid, frames, v, j, 4, hh))
    out.append(hbm("argmax_from_keys_kernel (key-points from the arg-max keys the heat-map layer's epilogue kept: the step's "
                   "decode, SURVEY 8(f1); argmax_decode_kernel, the decode from a read of the maps, serves tensors that are not the network's "
                   "own output and is not on this path any more)", t, n_maps * (8.0 * _lib.ARGMAX_SLOTS + 16.0),
                   "the maps' rows of partial keys in, two int64 out per map: launch-latency bound"))
    for kind, name in ((_lib.SCORE_HP, "HP"), (_lib.SCORE_MPE, "MPE"), (_lib.SCORE_BSB, "BSB")):
        t = timed(lambda: _lib.score_decode_maps(kind, hm, valid, frames, v, j, hh, wh, 4, hh))
        out.append(hbm(f"score_maps_kernel<{name}, decode> (per-map uncertainty statistic AND hard arg-max key-point from one "
                       "staged read: the scoring pass's only pass over the heat-maps)", t, map_bytes,
                       "one read of the heat-maps (uniform-noise maps: ~160 local maxima each to sort for MPE / BSB; "
                       "trained heat-maps have a handful)"))
    # soft-arg-max (utils/triangulation.py:191-200 use_soft_argmax=True, utils/evaluation.py:38): one read of the maps; at this
    # workload's map size and at C4's (8 frames x 8 views x 19 maps of 96 x 72)
    for (f_, v_, h_, w_, tag) in ((frames, v, hh, wh, "this workload"), (8, 8, 96, 72, "C4 slice: 8 frames x 8 views, 96x72 maps")):
        hm_s = hm if (f_, v_, h_, w_) == (frames, v, hh, wh) else torch.rand(f_, v_, j, h_, w_, device=dev)
        n_s = f_ * v_ * j
        t = timed(lambda: _lib.soft_argmax(hm_s, n_s, h_, w_, 4.0))
        out.append(hbm(f"soft_argmax_kernel ({n_s} maps of {h_}x{w_}; {tag}; parity unpinned: kornia absent)", t, n_s * h_ * w_ * 4.0,
                       "one read of the heat-maps (soft-max over the whole map, expectation of the pixel grid), two floats out per map"))
    proj = torch.from_numpy(np.stack([synth.ring_cameras(v, hh * 4, wh * 4, seed=s) for s in range(frames)])).to(dev)
    kp = _lib.score_decode_maps(_lib.SCORE_HP, hm, valid, frames, v, j, hh, wh, 4, hh)[2]
    t = timed(lambda: _lib.triangulate_ransac(kp, proj, valid, frames, v, j, 4.0))
    out.append(dict(bound="latency", achieved=round(frames * j / t), peak=None, unit="problems/s", frac=None, traffic=None,
                    **_kn("ransac_dlt_kernel (pairwise RANSAC + DLT + reprojection error, float64)"),
                    peak_note=f"{frames * j} (frame, joint) problems x {v * (v - 1) // 2} pairs per launch: "
                              "a few waves per CU, latency-bound; bytes are negligible",
                    launches_per_step=1, avg_launch_us=round(t * 1e6, 2), seconds_in_kernel_per_step=round(t, 7)))
    # core-set k-center at the BASELINE pool size (config.py:43-44: 200 labelled + 50 000 pool, 100 picks; D = 3J)
    rng = np.random.default_rng(0)
    pool = torch.from_numpy(rng.standard_normal((50000, j, 3)) * 300.0).to(dev)
    lab = torch.from_numpy(rng.standard_normal((200, j, 3)) * 300.0).to(dev)
    cs = CoreSet.from_tensors(pool, lab, 2)
    t = timed(lambda: cs.select_batch(100), reps=5)
    step_bytes = 50200 * (3 * j) * 8.0 + 2 * 50200 * 8.0
    e = hbm("kcenter_step_kernel (greedy k-center, 100 picks over 50 200 x 57 float64 features)", t / 100.0, step_bytes,
            "per greedy step: features + min-distance read/write; 23.7 MB sits in MALL/L2, so the step is launch-latency bound")
    e["launches_per_step"] = 103
    e["select_batch_100_ms"] = round(t * 1e3, 3)
    out.append(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 150 (c2: ~2.5 s timed), 1 for whole-pool passes")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--pool", type=int, default=None,
                    help="c4 / c5: frames of the fixed pool sharded over the ranks (BASELINE: 50000); one step = one whole pass")
    ap.add_argument("--frames-per-batch", type=int, default=None,
                    help="c4 / c5 pool passes: frames (x 8 views) per network launch (default 8: 64 images)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rccl-world-1", action="store_true",
                    help="with --gpus 1: initialise the nccl (= RCCL) process group at world size 1 and keep the pool passes' "
                         "collectives on (MVAL_DIST_NO_SHORTCUT=1): the single-GPU rehearsal of the multi-GPU pass")
    ap.add_argument("--no-exact-modes", action="store_true", help="skip the companion timings of the other conv kernel families")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--with-input", action="store_true",
                    help="inference workloads: every batch starts from uint8 camera crops in PINNED HOST memory (2h x 2w pixels per view), "
                         "uploaded on a copy stream one batch ahead and turned into the network's input by the device input pipeline "
                         "(mval_prepare_views: crop + PIL-LANCZOS resize + normalise, dataset/dataset.py:158-220).  Reported as its own "
                         "line; never the headline `value` contract (inputs resident in HBM)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run a batch's decode / scoring / triangulation on the network's stream instead of the side stream "
                         "(parallel.PostStream: by default it overlaps the next batch's network)")
    ap.add_argument("--adam", default="mval", choices=("mval", "torch", "fused"),
                    help="c3: the optimizer -- optim.Adam (one launch, default), torch.optim.Adam (foreach) or torch.optim.Adam(fused=True)")
    ap.add_argument("--no-rooflines", action="store_true",
                    help="profiling passes (tools/collect_profiles.sh): skip the per-kernel measurement runs after the timed region, so that a "
                         "rocprofv3 trace of this command holds the workload's steps only")
    ap.add_argument("--min-timed-seconds", type=float, default=2.0,
                    help="the K-step region is repeated until it is at least this long (counter passes use 0: K steps exactly)")
    ap.add_argument("--no-companions", action="store_true",
                    help="default c2 run: skip the C3 / C4 companion lines (child processes after the headline)")
    ap.add_argument("--shared-device", action="store_true",
                    help="TEST ONLY (tests/test_gpu_distributed.py): the N ranks all use cuda:0 and talk over gloo -- the rehearsal of this "
                         "script's N > 1 code (self-launch, barriers, rank reductions, attribution, DDP) on a one-GPU box.  RCCL refuses two "
                         "ranks on one device; the collectives' payloads are the same device tensors.  Not a measurement: the line says so")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "bench_detail.json"),
                    help="file for the run's full record (every kernel family, notes, pass counts, per-rank attribution, the companions' own "
                         "records); stdout carries only the compact line (<= 4 KB).  '' = do not write it")
    args = ap.parse_args()
    pool_pass = args.pool is not None or WORKLOADS[args.workload].get("pool") is not None
    if args.steps is None:
        args.steps = 1 if (args.pool or 0) > 4096 else 3 if pool_pass else 150 if args.workload == "c2" else 40
    if args.warmup is None:
        args.warmup = 1 if pool_pass else 5

    if args.gpus > 1 and "RANK" not in os.environ:
        # Self-launch: N child ranks, one per GPU, before this process has touched the GPU (device_count() does
        # not initialise it); the children print, this process only relays their exit code.
        import socket
        import subprocess

        have = torch.cuda.device_count()
        if have < (1 if args.shared_device else args.gpus):
            sys.exit(f"bench.py --gpus {args.gpus}: needs {args.gpus} devices, this node has {have}")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.shared_device else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                 "(or without torchrun: bench.py starts its own ranks)")
    if torch.cuda.device_count() <= local_rank:
        sys.exit(f"bench.py: rank {rank} needs device {local_rank}, this node has {torch.cuda.device_count()}")
    if world > 1 or args.rccl_world_1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29640")
        if args.rccl_world_1:
            os.environ["MVAL_DIST_NO_SHORTCUT"] = "1"
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.shared_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from multi_view_active_learning_amd import _lib, synth
    from multi_view_active_learning_amd.engine import ALGO_MFMA, ALGO_MFMA_BF3, ALGO_MFMA_H2, ALGO_MFMA_P2, _conv_mode, _plan_for
    from multi_view_active_learning_amd.utils.triangulation import triangulate_batch

    _lib.lib()  # fail loudly if the HIP extension is missing
    wl = dict(WORKLOADS[args.workload])
    if args.pool is not None:
        if args.workload not in ("c4", "c5"):
            sys.exit("--pool applies to the pool passes c4 / c5")
        wl["pool"] = args.pool
    if args.frames_per_batch:
        if not (args.pool is not None or wl.get("pool")):
            sys.exit("--frames-per-batch applies to the pool passes")
        wl["frames"] = args.frames_per_batch
    v, h, w, j, frames = wl["v"], wl["h"], wl["w"], wl["j"], wl["frames"]
    model, sd_np = build_model(wl["arch"], j, dev)
    # (--shared-device: every rank holds the SAME resident batch, so that a pool pass's content -- the batch repeated -- does not depend on
    # the number of ranks and the picks of an N-rank rehearsal can be compared with the one-rank run)
    crank = 0 if args.shared_device else rank
    images = torch.from_numpy(synth.images(1000 + crank, frames, v, h, w)).to(dev).reshape(frames * v, 3, h, w)
    proj = torch.from_numpy(np.stack([synth.ring_cameras(v, h, w, seed=crank * 1000 + s) for s in range(frames)])).to(dev)
    valid = torch.ones(frames, j, dtype=torch.uint8, device=dev)

    train = wl["train"]
    if train:
        from multi_view_active_learning_amd.pose_estimators import Pose2DMeanSquaredError

        model.train()
        net = model
        if world > 1 or args.rccl_world_1:
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], broadcast_buffers=True)
        # strategy.py:405-407 builds torch.optim.Adam([{"params": ..., "lr": LR}]): optim.Adam is that class with its step() as ONE launch
        # (csrc/optim.hip).  --adam torch / fused: torch's default (foreach) / fused implementation of the same update.
        from multi_view_active_learning_amd.optim import Adam as MvalAdam

        adam_kind = args.adam
        if adam_kind == "mval":
            opt = MvalAdam([{"params": model.parameters(), "lr": 1e-3}])
        else:
            opt = torch.optim.Adam([{"params": model.parameters(), "lr": 1e-3}], **({"fused": True} if adam_kind == "fused" else {}))
        loss_fn = Pose2DMeanSquaredError()
        gt = torch.rand(frames * v, j, h // 4, w // 4, device=dev)
        pv = torch.ones(frames * v, j, 1, 1, dtype=torch.uint8, device=dev)

    def coreset_pass():
        """One whole selection pass over the fixed pool (this rank's shard re-uses the resident batch
        as synthetic content for every local batch)."""
        from multi_view_active_learning_amd import parallel
        from multi_view_active_learning_amd.utils.coreset import CoreSet

        lo, hi = parallel.shard_range(wl["pool"], rank, world)
        if pass_t.get("on"):
            pass_t["_t0"] = time.perf_counter()
        preds = []
        for f0 in range(lo, hi, frames):
            nb = min(frames, hi - f0)
            hm = model(feed.next_images(nb * v) if feed else images[: nb * v])
            with post.batch(hm, _lib.argmax_keys_of(hm)):
                preds.append(triangulate_batch(hm.reshape(nb, v, j, h // 4, w // 4), proj[:nb], 4, valid[:nb])["keypoints_3d"].to(torch.float32))
        post.join()
        local = torch.cat(preds) if preds else torch.zeros((0, j, 3), device=dev)
        if pass_t.get("on"):
            torch.cuda.synchronize()  # (the shard's compute is done here: what follows is gather + selection)
            pass_t["compute_s"] += time.perf_counter() - pass_t.pop("_t0")
        pool = parallel.all_gather_cat(local)  # one size exchange + one data gather: (pool, J, 3) fp32 over xGMI
        ts = time.perf_counter()
        cs = CoreSet.from_tensors(pool, labeled_pose, 2)
        picks = cs.select_batch(wl["picks"])
        if pass_t.get("on"):
            torch.cuda.synchronize()
            pass_t["select_s"] += time.perf_counter() - ts
        return {"keypoints_3d": pool, "picks": picks}

    def scoring_pass():
        """BASELINE configs[3]: the entropy-scoring pass over the whole fixed pool.  Each rank scores its shard batch
        by batch (heat-maps, triangulation, MPE -- the per-batch body of _compute_sal_dict), then one size exchange and ONE packed
        all_gather of the (frames, 6 + 3J) result tables and the nlargest selection on every rank."""
        from multi_view_active_learning_amd import parallel
        from multi_view_active_learning_amd.strategy import score_decode_heatmaps_batch

        lo, hi = parallel.shard_range(wl["pool"], rank, world)
        if pass_t.get("on"):
            pass_t["_t0"] = time.perf_counter()
        tables = []
        for f0 in range(lo, hi, frames):
            nb = min(frames, hi - f0)
            hm0 = model(feed.next_images(nb * v) if feed else images[: nb * v])
            with post.batch(hm0):
                hm = hm0.reshape(nb, v, j, h // 4, w // 4)
                # one read of the heat-maps: uncertainty statistic + arg-max key-points (as ActiveLearningStrategy.score_batch)
                sc = score_decode_heatmaps_batch(wl["score"], "AVG", hm, valid[:nb], 4)
                al = sc[0]
                r = triangulate_batch(hm, proj[:nb], 4, valid[:nb], keypoints_2d=sc[4])
                fid = torch.arange(f0, f0 + nb, device=dev, dtype=torch.float64)
                tables.append(torch.cat([torch.zeros_like(fid)[:, None], fid[:, None], al[:, None], r["metric"][:, None],
                                         r["inlier_count"].to(torch.float64)[:, None], torch.zeros_like(fid)[:, None],
                                         r["keypoints_3d"].to(torch.float32).to(torch.float64).reshape(nb, 3 * j)], dim=1))
        post.join()
        local = torch.cat(tables) if tables else torch.zeros((0, 6 + 3 * j), dtype=torch.float64, device=dev)
        if pass_t.get("on"):
            torch.cuda.synchronize()  # (the shard's compute is done here: what follows is gather + selection)
            pass_t["compute_s"] += time.perf_counter() - pass_t.pop("_t0")
        table = parallel.all_gather_cat(local)  # the pass's two collectives: sizes (3 int64 per rank), then the data (12.6 MB at 50 k frames)
        ts = time.perf_counter()
        top = torch.topk(table[:, 2], min(100, table.shape[0])).indices  # AL.ITER_AMOUNT = 100 (config.py:44)
        if pass_t.get("on"):
            torch.cuda.synchronize()
            pass_t["select_s"] += time.perf_counter() - ts
        return {"keypoints_3d": table[:, 6:], "picks": top}

    # attribution of a pool pass (N >= 1): this rank's compute, the collectives (parallel.timers()), the selection -- host
    # wall clock with a device synchronisation at the three boundaries of every pass (three syncs per 20 - 150 s pass)
    pass_t = {"on": bool(wl.get("pool")), "compute_s": 0.0, "select_s": 0.0}
    from multi_view_active_learning_amd.parallel import PostStream

    post = PostStream(enabled=not args.no_overlap)

    # ---- --with-input: frames arrive as uint8 crops from pinned host memory, one batch ahead on a copy stream -----------------
    feed = None
    if args.with_input:
        if train:
            sys.exit("--with-input applies to the inference workloads")
        from multi_view_active_learning_amd.utils import preprocess

        class _Feed:
            """Two device buffers of raw uint8 crops (frames * v, 2h, 2w, 3); upload of batch i + 1 on the copy stream while
            batch i is resized and run; the SAME synthetic crops every batch (content does not change the work)."""

            def __init__(self):
                n = frames * v
                self.side = 2 * max(h, w)  # square source crops (the reference crops a square box, then resizes to (w, h))
                g = torch.Generator().manual_seed(7 + rank)
                self.host = torch.randint(0, 256, (n, self.side, self.side, 3), dtype=torch.uint8, generator=g).pin_memory()
                self.dev = [torch.empty_like(self.host, device=dev) for _ in range(2)]
                self.copy = torch.cuda.Stream(device=dev)
                self.ready = [torch.cuda.Event(), torch.cuda.Event()]
                self.free = [torch.cuda.Event(), torch.cuda.Event()]
                self.k = 0
                self.boxes = [(0, 0, self.side, self.side)] * n
                self.bytes = self.host.numel()
                self.out = [None, None]
                self._upload(0)

            def _upload(self, b):
                # the whole input stage of a batch on the copy stream, one batch ahead of the network: H->D copy, then the crop + LANCZOS
                # resize + normalisation (mval_prepare_views) into the batch's own output tensor -- as the reference's loader workers run
                # prepare_single_view ahead of the model (dataset/dataset.py:158-220).  (Until the second half of round 5 the resize ran
                # on the network's stream, in front of every forward: 11.1 vs 10.0 ms per step.)
                with torch.cuda.stream(self.copy):
                    self.copy.wait_event(self.free[b])  # (the network that read this batch's tensors last is enqueued and done; unrecorded at first: no wait)
                    self.dev[b].copy_(self.host, non_blocking=True)
                    self.out[b] = preprocess.resize_views(list(self.dev[b].unbind(0)), self.boxes, w, h)
                    self.ready[b].record(self.copy)

            def next_images(self, nb_images):
                b = self.k & 1
                cur = torch.cuda.current_stream(dev)
                if self.k:
                    self.free[b ^ 1].record(cur)  # (the previous batch's network is enqueued on this stream by now)
                cur.wait_event(self.ready[b])
                x = self.out[b]
                x.record_stream(cur)  # (allocated on the copy stream, read by the network's)
                self._upload(b ^ 1)  # the next batch travels and is resized while this one runs
                self.k += 1
                return x[:nb_images]

        feed = _Feed()
    if wl.get("picks"):
        labeled_pose = torch.from_numpy(np.random.default_rng(5).standard_normal((wl["labeled"], j, 3)) * 300.0).to(dev)

    def step():
        if train:
            opt.zero_grad()
            loss = loss_fn.pose_2d_mse(net(images), gt, pv)
            loss.backward()
            opt.step()
            return {"keypoints_3d": loss.detach().reshape(1)}
        if wl.get("pool"):
            return coreset_pass() if wl.get("picks") else scoring_pass()
        hm0 = model(feed.next_images(frames * v) if feed else images)
        # the batch's decode / scoring / triangulation goes to the side stream (parallel.PostStream, as the product's pass loops
        # do: strategy._compute_sal_dict / evaluate_mkpe): the next step's network overlaps it; sync() joins every stream
        with post.batch(hm0, _lib.argmax_keys_of(hm0)):
            hm = hm0.reshape(frames, v, j, h // 4, w // 4)
            if wl.get("score"):  # one read of the heat-maps for the statistic and the key-points (score_batch's path)
                from multi_view_active_learning_amd.strategy import score_decode_heatmaps_batch

                sc = score_decode_heatmaps_batch(wl["score"], "AVG", hm, valid, 4)
                r = triangulate_batch(hm, proj, 4, valid, keypoints_2d=sc[4])
                r["al_metric"] = sc[0]
                return r
            return triangulate_batch(hm, proj, 4, valid)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # The timed region is K steps, repeated until it is at least ~2 s long (the driver's --steps 20 is 0.25 s of this
    # workload: clocks and caches have not settled); every repeat runs exactly K steps, all of them are timed.
    repeats = 1
    with torch.set_grad_enabled(train):
        if args.warmup == 0 and wl.get("pool"):  # (--warmup 0 on a pool pass: build the plan on one batch instead of a whole pass)
            model(images)
            r = None
        for _ in range(args.warmup if wl.get("pool") else max(args.warmup, 1)):
            r = step()
        sync()
        if not wl.get("pool"):
            t0 = time.perf_counter()
            r = step()
            sync()
            est = max(time.perf_counter() - t0, 1e-4) * args.steps
            repeats = int(min(64, max(1, np.ceil(args.min_timed_seconds / est))))
            if world > 1:  # every rank runs the same number of steps
                t = torch.tensor([repeats], dtype=torch.int64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                repeats = int(t.item())
        sync()
        from multi_view_active_learning_amd import parallel as _par

        _par.reset_timers(on=bool(wl.get("pool")))
        pass_t.update(compute_s=0.0, select_s=0.0)
        t0 = time.perf_counter()
        for _ in range(args.steps * repeats):
            r = step()
        torch.cuda.synchronize()
        el_rank = (time.perf_counter() - t0) / repeats  # this rank's own time, before it waits for the others
        sync()
        el = (time.perf_counter() - t0) / repeats
        _par.reset_timers(on=False)
    attribution = None
    if world > 1 or args.rccl_world_1 or wl.get("pool"):
        # every rank reports before rank 0 prints: per-rank time of the timed region, and for the pool passes its split
        mine = torch.tensor([el_rank, pass_t["compute_s"] / repeats, _par.timers()["gather_s"] / repeats, pass_t["select_s"] / repeats],
                            dtype=torch.float64, device=dev)
        if world > 1 or args.rccl_world_1:
            allr = torch.empty(world * 4, dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(allr, mine)
        else:
            allr = mine
        allr = allr.cpu().reshape(world, 4).numpy()
        attribution = {"per_rank_s": {"min": round(float(allr[:, 0].min()), 4), "mean": round(float(allr[:, 0].mean()), 4),
                                      "max": round(float(allr[:, 0].max()), 4), "all": [round(float(x), 4) for x in allr[:, 0]]},
                       "note": "seconds of the K-step timed region per rank (before the closing barrier); value uses the max over ranks incl. the barrier"}
        if wl.get("pool"):
            attribution.update(compute_s={"min": round(float(allr[:, 1].min()), 4), "max": round(float(allr[:, 1].max()), 4)},
                               gather_s=round(float(allr[:, 2].max()), 4), select_s=round(float(allr[:, 3].max()), 4),
                               split_note="per pass-set: shard compute (heat-maps + decode + triangulation [+ scoring]) / the two collectives "
                                          "(incl. waiting for the slowest rank) / selection (k-center or top-k, replicated)")
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if train and (world > 1 or args.rccl_world_1) and isinstance(net, torch.nn.parallel.DistributedDataParallel):
        # exposed all-reduce: the same step with the gradient synchronisation switched off (no_sync) against the normal one
        def timed_steps(k, ctx):
            sync()
            t1 = time.perf_counter()
            with ctx():
                for _ in range(k):
                    step()
            sync()
            return (time.perf_counter() - t1) / k
        import contextlib

        with torch.enable_grad():
            t_sync = timed_steps(10, contextlib.nullcontext)
            t_nosync = timed_steps(10, net.no_sync)
        tt = torch.tensor([t_sync, t_nosync], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        attribution["allreduce_exposed_s"] = round(float(tt[0] - tt[1]), 5)
        attribution["allreduce_note"] = (f"step with DDP's bucketed RCCL all-reduce {float(tt[0]) * 1e3:.2f} ms vs under no_sync() {float(tt[1]) * 1e3:.2f} ms "
                                         "(10 steps each, max over ranks): what the segmented backward does not hide")
    assert torch.isfinite(r["keypoints_3d"]).all()

    # ---- roofline of the dominant kernel family (outside the timed region) ------------------------
    roof = None
    rank_roof = -1 if args.no_rooflines else 0
    if rank == rank_roof and train:
        with torch.enable_grad():
            roof = train_rooflines(model, step, frames, v, _conv_mode())
        # (SURVEY 8d convention for the whole step: 3 x the forward conv FLOPs over the step's wall time)
        roof["whole_step_tflops_3x_forward"] = round(3.0 * FLOP_PER_IMAGE["hrnet_w32_256"] * frames * v / (el / args.steps) / 1e12, 2)
    if rank == rank_roof and not train:
        plan = _plan_for(model, images)
        with torch.no_grad():
            ms_acc, reps = None, 3
            for _ in range(reps):
                _, ms, flops = plan.forward_timed(images)
                ms_acc = ms if ms_acc is None else ms_acc + ms
        ms = ms_acc / reps

        def family(mask, name, peak, peak_note):
            n = int(mask.sum())
            if n == 0:
                return None
            t = float(ms[mask].sum()) * 1e-3
            f = float(flops[mask].sum())
            return dict(bound="mfma", achieved=round(f / t / 1e12, 2), peak=peak, unit="TFLOP/s",
                        frac=round(f / t / 1e12 / peak, 4), traffic=None, peak_note=peak_note, **_kn(name),
                        launches_per_step=n, avg_launch_us=round(t / n * 1e6, 2), flops_per_step=f,
                        seconds_in_kernel_per_step=round(t, 6))

        def hbm_family(mask, name):
            """HBM-bound operators: algorithmic bytes (input + output + residual reads, fp32) / time."""
            n = int(mask.sum())
            if n == 0:
                return None
            t = float(ms[mask].sum()) * 1e-3
            nimg = plan.n
            b = 0.0
            for o, m_ in zip(plan.ops, mask):
                if m_ and o.kind == 7:  # fused up-sampling terms: every term's input at its own resolution, partial sum in, sum out
                    b += 4.0 * nimg * (sum((o.hout >> o.t_up[t]) * (o.wout >> o.t_up[t]) * o.t_cin[t] for t in range(o.n_terms)) + 2 * o.hout * o.wout * o.cout)
                elif m_:
                    outp = (o.hout << o.up) * (o.wout << o.up) * o.cout
                    # (a fused Bottleneck whose residual is its own input reads that tensor once, algorithmically)
                    b += 4.0 * nimg * (o.hin * o.win * o.cin + outp * (1 + (o.res1_off >= 0 and o.res1_off != o.in_off) + (o.res2_off >= 0)))
            return dict(bound="hbm", achieved=round(b / t / 1e9, 1), peak=PEAK_HBM_GBPS, unit="GB/s",
                        frac=round(b / t / 1e9 / PEAK_HBM_GBPS, 4), traffic=None, **_kn(name),
                        peak_note="HBM3E ~8 TB/s (guide); ~6.3 TB/s achievable",
                        launches_per_step=n, avg_launch_us=round(t / n * 1e6, 2), bytes_per_step=b,
                        seconds_in_kernel_per_step=round(t, 6))

        conv = np.asarray([o.kind not in (1, 4) for o in plan.ops])  # (not max-pool, not the fp32 -> P2 format change)
        f32 = np.asarray([o.algo == ALGO_MFMA for o in plan.ops]) & conv
        k3 = np.asarray([o.k == 3 for o in plan.ops])
        s1 = np.asarray([o.stride == 1 for o in plan.ops])
        stem = np.asarray([o.kind == 0 and o.in_nchw == 1 for o in plan.ops])
        bneck = np.asarray([o.kind == 5 for o in plan.ops])
        stem2 = np.asarray([o.kind == 6 for o in plan.ops])
        fuseup = np.asarray([o.kind == 7 for o in plan.ops])
        fams = []
        # the 256-channel Bottleneck of this plan (in + out, residual = input): algorithmic GB and GFLOP per launch, for its note below
        bn_ops = [(o, f_) for o, f_ in zip(plan.ops, flops) if o.kind == 5 and o.cin == 256]
        bneck_gb = (4.0 * plan.n * (bn_ops[0][0].hin * bn_ops[0][0].win * 256 * 2) / 1e9) if bn_ops else 0.0
        bneck_gf = (bn_ops[0][1] / 1e9) if bn_ops else 0.0
        split_any = np.zeros(len(plan.ops), dtype=bool)
        for algo, pl, what in ((ALGO_MFMA_P2, "p2", "activations kept as fp16 (h, l) plane pairs in HBM, 3 x v_mfma_f32_16x16x32_f16 per 32-deep step"),
                               (ALGO_MFMA_H2, "h2", "fp32 values as scaled 2-way fp16 splits, 3 x v_mfma_f32_16x16x32_f16 per 32-deep step"),
                               (ALGO_MFMA_BF3, "bf3", "fp32 values as exact 3-way bf16 splits, 6 x v_mfma_f32_16x16x32_bf16 per 32-deep step")):
            m_ = np.asarray([o.algo == algo for o in plan.ops]) & conv
            split_any |= m_
            peak = PEAK_BF16_MFMA_TFLOPS / SPLIT_PRODUCTS[pl]
            note = f"dense 16-bit MFMA peak 2500 TFLOP/s / {SPLIT_PRODUCTS[pl]} MFMA products per algorithmic product"
            fams += [
                family(m_ & k3 & s1, ("conv_p2_kernel<3, 1, ...>" + (" + conv_block_p2_kernel<C> (whole BasicBlocks: two 3x3 convs, BNs, residual, "
                                                                      "ReLUs in one launch)" if any(o.kind == 3 for o in plan.ops) else "")
                                      if pl == "p2" else f"conv_split_kernel<{2 if pl == 'h2' else 3}, 3, 1, ...>")
                                     + (" + conv_block_kernel<C> (whole BasicBlocks: two 3x3 convs, BNs, residual, ReLUs in one launch)"
                                        if any(o.kind == 3 for o in plan.ops) and pl == "h2" else "")
                                     + f" (fused 3x3 stride-1 conv+BN+residual+ReLU; {what}, fp32 accumulate)", peak, note),
                family(m_ & stem2, "conv_stem_p2_kernel (both stride-2 stem convs in one launch, both on the matrix cores: 3 -> 64 as one 32-deep step over an im2col "
                                   "gather, 64 -> 64 from LDS; fp32 NCHW image in, P2 planes out; FLOPs of both convs against the split-MFMA peak)", peak, note),
                family(m_ & k3 & ~s1 & ~stem2, ("conv_p2_kernel<3, 2, ...>" if pl == "p2" else f"conv_split_kernel<{2 if pl == 'h2' else 3}, 3, 2, ...>") + " (same, stride 2)", peak, note),
                hbm_family(m_ & bneck, "conv_bneck_p2_kernel<CIN> (whole Bottlenecks of layer1 in one launch: 1x1 -> 3x3 -> 1x1 convs, BNs, residual, "
                                       "ReLUs; the 64-channel intermediates never leave the CU; HBM is the tighter of its two bounds: "
                                       f"{bneck_gb:.2f} GB / 8 TB/s = {bneck_gb / 8e3 * 1e6:.0f} us vs {bneck_gf:.0f} GFLOP x 3 / 2500 TFLOP/s = "
                                       f"{bneck_gf * 3 / 2500e3 * 1e6:.0f} us per launch of this step's {plan.n} images)"),
                hbm_family(m_ & fuseup, "conv_fuse_up_p2_kernel<C> (the two or three up-sampling 1x1 terms of a fuse-layer output added to the partial sum in one "
                                        "launch: the sum is read once and written once)"),
                hbm_family(m_ & ~k3 & ~bneck & ~fuseup, ("conv_p2_kernel<1, 1, ...>" if pl == "p2" else f"conv_split_kernel<{2 if pl == 'h2' else 3}, 1, 1, ...>") + " (fused 1x1 conv+BN+residual+ReLU(+upsample): "
                                     "channel GEMMs of the bottleneck blocks and fuse up-paths; 2x2 parity convs of transposed convs)"),
            ]
        fams += [
            family(f32, "conv_mfma_kernel (exact-fp32 v_mfma_f32_16x16x4_f32: heat-map layer and shapes the split kernels "
                        "do not cover)", PEAK_FP32_MFMA_TFLOPS, "dense fp32-input MFMA peak"),
            hbm_family(stem, "conv_stem_kernel (3-channel NCHW stem conv, VALU)"),
        ]
        fams = sorted([f for f in fams if f], key=lambda f: -f["seconds_in_kernel_per_step"])
        roof = fams[0]
        roof["other_kernels"] = fams[1:] + path_kernels(dev, frames, v, j, h // 4, w // 4)
        # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside
        # the process, so this is the committed rocprofv3 measurement of THIS command (separate
        # --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 gfx950 correction; tools/pmc_summary.py)
        summary = next((p_ for p_ in (os.path.join("profiles", r_, f"bench_c2_{_conv_mode()}_summary.json") for r_ in ("r06", "r05", "r04", "r03"))
                        if os.path.exists(os.path.join(ROOT, p_))), os.path.join("profiles", "r06", f"bench_c2_{_conv_mode()}_summary.json"))
        try:
            with open(os.path.join(ROOT, summary)) as f:
                pre = (roof["kernel"].split(" ...>")[0],) + (("conv_block_kernel",) if "conv_block_kernel" in roof["kernel"] else ()) + \
                      (("conv_block_p2_kernel",) if "conv_block_p2_kernel" in roof["kernel"] else ())
                rows = [r for r in json.load(f)["hbm_traffic_by_instantiation"] if r["kernel"].startswith(pre)]
            if args.workload == "c2" and rows:
                nl = sum(r["launches"] for r in rows)
                roof["traffic"] = round(sum(r["launches"] * r["total_bytes"] for r in rows) / nl)
                roof["traffic_source"] = (summary + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, "
                                          "launch-weighted mean over this kernel's instantiations)")
        except (OSError, KeyError, ValueError):
            pass
        allc = split_any | f32
        roof["frac_of_bf16x3_peak"] = round(roof["achieved"] / (PEAK_BF16_MFMA_TFLOPS / 6.0), 4) if roof.get("unit") == "TFLOP/s" else None
        roof["frac_note"] = ("frac is against the peak of the split in use (fp16x2: 2500 / 3 = 833 TFLOP/s); frac_of_bf16x3_peak is the same "
                             "achieved rate against round 1's 416.7 TFLOP/s, for comparison across rounds")
        roof["all_conv_tflops"] = round(float(flops[allc].sum()) / (float(ms[allc].sum()) * 1e-3) / 1e12, 2)
        roof["all_conv_frac_of_fp32_mfma_peak"] = round(roof["all_conv_tflops"] / PEAK_FP32_MFMA_TFLOPS, 4)
        if getattr(plan, "p2", False):
            roof["p2_bound_slack_log2"] = round(plan.p2_slack_log2(), 2)
            roof["p2_bound_slack_note"] = ("log2 of the largest (a-priori output bound / actual max |x|) over the plan's P2 activations and images; "
                                           "above 15 the engine hands over to the h2 plan (engine.P2_MAX_SLACK_LOG2)")
        roof["whole_forward_ms"] = round(float(ms.sum()), 3)
        roof["non_mfma_ms"] = round(float(ms[~allc].sum()), 3)

    if rank == 0:
        total_units = world * frames * v * args.steps
        if wl.get("pool"):
            total_units = wl["pool"] * v * args.steps
        out = {
            "metric": ("frames*views/sec (heatmap->triangulated 3D) PoseResNet-50 2-view 256x192" if wl["arch"] == "resnet50" else
                       "frames*views/sec (training step) HRNet-W32 4-view 256x256" if train else
                       "frames*views/sec (pool scoring) HRNet-W48 8-view 384x288" if wl.get("score") else
                       "frames*views/sec (core-set selection pass over a fixed pool) HRNet-W48 8-view 384x288" if wl.get("pool") else
                       "frames*views/sec (heatmap->triangulated 3D) HRNet-W32 4-view 256x256"),
            "value": round(total_units / el, 2),
            "unit": "frames*views/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(el / args.steps * 1e3, 3),
            "timed_repeats": repeats,  # the K-step region was timed this many times back to back (>= 2 s in all); value is per K
            "higher_is_better": True,
            "scaling": "strong" if wl.get("pool") else "weak",
            "vs_baseline": None,
            "dtype": DTYPE_LABEL[_conv_mode()],
            "dtype_note": DTYPE_NOTE[_conv_mode()],
            "data": "synthetic (random variance-preserving weights, N(0,1) frames, ring cameras)",
            "config": ({"workload": wl["desc"].replace("256-frame pool", f"{wl['pool']}-frame pool")
                                   + ("" if wl.get("picks") else f"; one step = one pass over a {wl['pool']}-frame pool sharded over the ranks"),
                        "pool_frames": wl["pool"], "frames_per_batch": frames, "views": v,
                        "parallelism": f"pool sharded x{world} by frames, one size exchange + one data gather per pass (RCCL), selection replicated"}
                       if wl.get("pool") else
                       {"workload": wl["desc"], "frames_per_step_per_gpu": frames, "views": v,
                        "images_per_step_per_gpu": frames * v, "parallelism": f"frame-sharded x{world}, no collective",
                        "post_stage": ("side stream (a batch's decode + triangulation overlaps the next batch's network; the timed region ends with "
                                       "a device-wide synchronisation)" if post.enabled else "network's stream")}),
            "roofline": roof,
            "parity_unpinned": PARITY_UNPINNED,
        }
        if args.shared_device:
            out["config"]["parallelism"] = f"REHEARSAL: {world} ranks share cuda:0 over gloo; " + out["config"]["parallelism"]
        if wl.get("pool") and r is not None and "picks" in r:
            import zlib

            pk = r["picks"]
            pk = pk.cpu().tolist() if torch.is_tensor(pk) else [x if isinstance(x, (int, str)) else int(x) for x in pk]
            out["picks_crc"] = zlib.crc32(json.dumps(pk).encode())  # (the selection of the last pass: equal across rank counts on equal content)
        if not train:
            try:  # (how the timed steps enqueued the network: engine.InferencePlan._graph_wanted)
                ip = _plan_for(model, images)
                out["config"]["network_launch"] = ("captured hipGraph replayed per batch (multi-stream plan; the batch is copied into the graph's input "
                                                   "tensor and the heat-maps out of its output tensor inside the timed region)" if getattr(ip, "_graph", None) is not None
                                                   else "eager launches")
            except Exception:
                pass
        if train:
            tp = next(iter(model._train_plans.values()), None)
            out["p2_bound_slack"] = None if tp is None else tp.p2_slack
            out["p2_bound_slack_note"] = ("training plan's probe (engine_train.TrainPlan: first step of the plan, then every 1 024): per kind (act = P2 activation planes, "
                                          "dz = the BatchNorm backward's P2 planes) log2 of the largest a-priori bound / actual max |x| over the tensors and the largest "
                                          "fraction of a tensor's non-zero values below 2^-3 scaled; past 2^15 (or 0.5 of an activation tensor) the model's next steps "
                                          "run the h2 training kernels")
            out["config"]["training_passes"] = (
                "one stream" if tp is None or tp.n_lanes <= 1 else
                f"{tp.n_lanes} lanes (MVAL_TRAIN_LANES mode {os.environ.get('MVAL_TRAIN_LANES', '3')}: HRNet's branches on separate streams; mode 3 = no joins at the "
                "backward's phase changes; bit-identical to one stream; per-kernel rooflines are timed on one stream)")
            if tp is not None:
                out["config"]["bn_in_conv"] = {"applies_in_reader_staging": int(sum(int(t.z_out) for t in tp.ops)),
                                               "reductions_in_reader_dgrad_epilogue": int(sum(int(t.p2_flags & 4096 != 0) for t in tp.ops))}
            out["config"]["optimizer"] = {"mval": "multi_view_active_learning_amd.optim.Adam (torch.optim.Adam subclass, step = one mval_adam_step launch)",
                                          "torch": "torch.optim.Adam (foreach)", "fused": "torch.optim.Adam(fused=True)"}[adam_kind]
        if feed is not None:
            out["input_inclusive"] = {
                "note": "NOT the headline contract: every batch starts from uint8 crops in pinned host memory -- H->D copy and mval_prepare_views "
                        "(crop, PIL-LANCZOS resize, normalise) one batch ahead on a copy stream, then the network",
                "source_crop_px": [feed.side, feed.side], "bytes_uploaded_per_batch": feed.bytes,
                "host_to_device_GBps": round(feed.bytes * (-(-(wl["pool"] // max(world, 1)) // frames) if wl.get("pool") else 1) * args.steps / el / 1e9, 2)}
            out["metric"] += " [input-inclusive: --with-input]"
        if attribution is not None:
            out["attribution"] = attribution
        if not train and not wl.get("pool") and world == 1 and not args.no_exact_modes:
            # the same step with the other conv kernel families (plans are cached per mode), outside the timed region
            exact = {}
            headline_mode = _conv_mode()
            had_env = "MVAL_CONV" in os.environ
            try:  # (a failure here must neither leave the mode changed nor lose the headline measured above)
                for mode_ in ("h2", "bf3", "fp32"):
                    if mode_ == headline_mode:
                        continue
                    os.environ["MVAL_CONV"] = mode_
                    try:
                        with torch.no_grad():
                            for _ in range(3):
                                step()
                            sync()
                            t0 = time.perf_counter()
                            for _ in range(20):
                                step()
                            sync()
                        exact[mode_] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
                    except Exception as e:  # noqa: BLE001
                        exact[mode_] = f"failed: {type(e).__name__}: {e}"
                    # this companion's plan (and its arena) is not needed again
                    for key in [k for k in getattr(model, "_plans", {}) if mode_ in k]:
                        model._plans.pop(key, None)
                    torch.cuda.empty_cache()
            finally:
                if had_env:
                    os.environ["MVAL_CONV"] = headline_mode
                else:
                    os.environ.pop("MVAL_CONV", None)
            exact["note"] = ("ms per step; h2 = round 2's kernels (fp32 NHWC activations, split while staging), bf3 = exact 3-way "
                             "bf16 split (six MFMA products), fp32 = exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) everywhere")
            out["exact_modes"] = exact
        if not args.no_cpu_baseline and train and world == 1:
            cpu = cpu_baseline_train(wl, sd_np, args.cpu_seconds)
            # the training step's parity sample: the HIP path (a fresh model from the same synthetic weights, train mode) on the oracle's
            # sample -- loss, heat-maps and three gradients against the CPU oracle's autograd (outside every timed region)
            chk = cpu.pop("_check")
            m2, _ = build_model(wl["arch"], j, dev)
            m2.train()
            with torch.enable_grad():
                hm2 = m2(chk["images"].to(dev))
                l2 = loss_fn.pose_2d_mse(hm2, chk["gt"].to(dev), torch.ones(hm2.shape[0], j, 1, 1, dtype=torch.uint8, device=dev))
                l2.backward()
            named = dict(m2.named_parameters())
            rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b) + 1e-30))
            cpu["parity_sample"] = {
                "images": int(hm2.shape[0]), "loss_hip": float(l2.detach()), "loss_oracle": chk["loss"],
                "loss_rel_err": abs(float(l2.detach()) - chk["loss"]) / abs(chk["loss"]),
                "heatmap_max_abs_err": float(np.abs(hm2.detach().cpu().numpy() - chk["heatmaps"]).max()),
                "grad_rel_l2_err": {k: rel(named[k].grad.cpu().numpy(), g_) for k, g_ in chk["grads"].items()},
                "note": "one train-mode forward + masked MSE + backward of the same weights on the oracle's sample (batch statistics over these "
                        "images); torch-CPU fp32 itself sits 1e-3 .. 3e-2 from a float64 run on the early layers' gradients (tests/test_gpu_train.py)",
            }
            del m2
            out["cpu_baseline"] = cpu
        if not args.no_cpu_baseline and not train and world == 1:  # rank 0 at N = 1 only
            cpu = cpu_baseline(wl, sd_np, args.cpu_seconds)
            # BASELINE.json's "MPJPE vs ref": the HIP path on the oracle's first sample (outside every timed region)
            chk = cpu.pop("_check")
            with torch.no_grad():
                hm = model(chk["images"].to(dev)).reshape(-1, v, j, h // 4, w // 4)
                got = triangulate_batch(hm, torch.from_numpy(chk["proj"]).to(dev), 4,
                                        torch.ones(hm.shape[0], j, dtype=torch.uint8, device=dev))
            delta = np.linalg.norm(got["keypoints_3d"].cpu().numpy() - chk["keypoints_3d"], axis=-1)
            cpu["parity_sample"] = {
                "frames": int(hm.shape[0]),
                "heatmap_max_abs_err": float(np.abs(hm.cpu().numpy() - chk["heatmaps"]).max()),
                "keypoints_2d_equal": bool(np.array_equal(got["keypoints_2d"].cpu().numpy(), chk["keypoints_2d"])),
                "mpjpe_vs_oracle_mm": float(delta.mean()),
                "max_joint_delta_mm": float(delta.max()),
            }
            out["cpu_baseline"] = cpu
        if args.workload == "c2" and world == 1 and not args.no_companions and not args.rccl_world_1:
            # BASELINE configs[2] / [3] under the same (driver) clock: child processes of this command, run after the headline's
            # timed region and measurements; each times >= 2 s of its own step.  A failure is recorded, not raised.
            import subprocess
            import tempfile

            comp = {}
            _CK = ("kernel", "bound", "achieved", "peak", "unit", "frac", "seconds_in_kernel_per_step", "traffic", "traffic_source", "bytes_per_step",
                   "flops_per_step", "passes", "bytes_round3_count", "frac_round3_count", "launches_per_step", "avg_launch_us",
                   "whole_step_tflops_3x_forward")
            for name, extra in (("c3", ["--steps", "20", "--warmup", "3", "--cpu-seconds", str(min(args.cpu_seconds, 12.0))]),
                                ("c4", ["--steps", "100", "--warmup", "3", "--no-cpu-baseline"]),
                                # the headline's workload from uint8 crops in pinned host memory (H->D copy + device input pipeline in the step): never `value`
                                ("c2_with_input", ["--steps", "100", "--warmup", "5", "--no-cpu-baseline", "--with-input", "--no-rooflines"])):
                fd, child_detail = tempfile.mkstemp(prefix=f"mval_bench_{name}_", suffix=".json")
                os.close(fd)
                cmd = [sys.executable, os.path.abspath(__file__), "--workload", name.split("_")[0], "--no-companions", "--no-exact-modes",
                       "--detail-out", child_detail] + extra
                if args.no_cpu_baseline and "--no-cpu-baseline" not in cmd:
                    cmd.append("--no-cpu-baseline")
                t0 = time.perf_counter()
                try:
                    pr = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
                    d = None
                    if pr.returncode == 0 and os.path.getsize(child_detail) > 0:  # the child's full record (its stdout holds the compact line only)
                        with open(child_detail) as f:
                            d = json.load(f)
                    if d is None:
                        comp[name] = {"error": f"rc {pr.returncode}: {pr.stderr[-300:]}"}
                    else:
                        roof_c = d.get("roofline") or {}
                        comp[name] = {"metric": d["metric"], "ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"],
                                      "steps": d["steps"], "timed_repeats": d["timed_repeats"],
                                      "timed_s": round(d["ms_per_step"] * d["steps"] * d["timed_repeats"] * 1e-3, 2),
                                      "config": d["config"], "dtype": d["dtype"],
                                      "roofline": {k: roof_c.get(k) for k in _CK if k in roof_c},
                                      "families": [{k: f.get(k) for k in _CK if k in f} for f in (roof_c.get("other_kernels") or [])[:6]]}
                        for k in ("input_inclusive", "p2_bound_slack"):
                            if k in d:
                                comp[name][k] = d[k]
                        if "cpu_baseline" in d:
                            comp[name]["cpu_baseline"] = {k: v_ for k, v_ in d["cpu_baseline"].items() if k != "threads_tried_note"}
                except Exception as e:  # noqa: BLE001
                    comp[name] = {"error": f"{type(e).__name__}: {e}"}
                finally:
                    try:
                        os.unlink(child_detail)
                    except OSError:
                        pass
                comp[name]["wall_s"] = round(time.perf_counter() - t0, 1)
            comp["note"] = ("BASELINE configs[2] (C3 training step) and configs[3]'s per-GPU slice (C4: HRNet-W48, 8 views, 384x288, 8 frames + MPE "
                            "scoring) run by this command as child processes after the headline: same box, same driver clock; c2_with_input = the "
                            "headline's workload with every batch starting from uint8 camera crops in pinned host memory (strategy.py:772-782, "
                            "dataset/dataset.py:158-220: H->D copy one batch ahead + mval_prepare_views + network + decode + RANSAC-DLT) -- never `value`")
            out["companions"] = comp
    if world > 1 or args.rccl_world_1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # The JSON line is the LAST thing on stdout: RCCL's version banner sits in the C library's stdio buffer (flushed at exit when stdout is a
        # file or a pipe), so the process group goes first and the C buffers are flushed before the line is written.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        detail_path = None
        if args.detail_out:
            try:
                with open(args.detail_out, "w") as f:
                    json.dump(out, f, indent=1)
                detail_path = os.path.relpath(args.detail_out, ROOT) if os.path.abspath(args.detail_out).startswith(ROOT + os.sep) else args.detail_out
            except OSError as e:  # (a read-only checkout must not cost the line)
                print(f"bench.py: could not write {args.detail_out}: {e}", file=sys.stderr)
        print(compact_line(out, detail_path), flush=True)


if __name__ == "__main__":
    main()
