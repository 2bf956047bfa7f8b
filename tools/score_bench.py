"""Per-kernel times of the post-network stage (decode, scoring, RANSAC-DLT) at the bench workloads' sizes.
   --product-loop [frames=512]: the PRODUCT's pool loop instead -- ActiveLearningStrategy._compute_sal_dict (strategy.py:1004-1147) over a
   loader of HOST tensors (what the reference's DataLoader yields: pageable images (B, V, 3, 384, 288) fp32, float64 cameras), HRNet-W48,
   8 views, 8 frames per batch, MPE scoring: frames x views / s of the loop a user of workflow.py runs, next to bench.py --workload c4
   (which feeds device-resident frames)."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--product-loop" in sys.argv:
    import json, time
    from multi_view_active_learning_amd import synth
    from multi_view_active_learning_amd.config import get_default_configs
    from multi_view_active_learning_amd.pose_estimators import PoseHighResolutionNet, hrnet_w48
    from multi_view_active_learning_amd.strategy import ActiveLearningStrategy
    i = sys.argv.index("--product-loop")
    frames = int(sys.argv[i + 1]) if len(sys.argv) > i + 1 else 512
    b, v, j, h, w = 8, 8, 19, 384, 288
    dev = torch.device("cuda:0")
    m = PoseHighResolutionNet(j, hrnet_cfg=hrnet_w48())
    m.load_state_dict({k: torch.from_numpy(a) for k, a in synth.synthetic_state_dict(m._graph.param_shapes(), 0).items()}, strict=True)
    m = m.to(dev).eval()
    cfg = get_default_configs()
    cfg.AL.STRATEGY = "MPE"
    cfg.POSE_ESTIMATOR.STRIDE = 4
    st = ActiveLearningStrategy(cfg)
    rng = np.random.default_rng(0)
    base = torch.from_numpy(synth.images(5, b, v, h, w))
    def loader(pinned):
        for i0 in range(0, frames, b):
            img = base.pin_memory() if pinned else base.clone()  # (a DataLoader hands over a fresh tensor per batch)
            yield {"images": img, "proj_matrices": torch.from_numpy(np.stack([synth.ring_cameras(v, h, w, seed=i0 + s) for s in range(b)])),
                   "joint_valid": torch.ones(b, j, dtype=torch.uint8), "3d_keypoints": torch.from_numpy(rng.standard_normal((b, 3, j)) * 300.0),
                   "pose": torch.zeros(b, dtype=torch.int64), "frame_id": torch.arange(i0, i0 + b, dtype=torch.int64)}
    out = {}
    for pinned in (False, True):
        batches = list(loader(pinned))  # (built before the clock: the reference's loader workers prepare batches ahead of the model)
        st._compute_sal_dict(batches[:2], m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sal = st._compute_sal_dict(batches, m)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        assert len(sal["al_metric"]) == frames
        out["pinned_host_images" if pinned else "pageable_host_images"] = {"frames_x_views_per_s": round(frames * v / el, 1), "ms_per_batch_of_64_images": round(el / (frames / b) * 1e3, 3)}
    out["workload"] = f"_compute_sal_dict, HRNet-W48 8-view 384x288, MPE, {frames} frames in batches of {b}, host tensors in (21.2 MB of fp32 images per batch over PCIe)"
    print(json.dumps(out))
    sys.exit(0)
from multi_view_active_learning_amd import _lib
from multi_view_active_learning_amd import synth
dev = torch.device("cuda:0")
def proj(f, v, hh, wh):
    return torch.from_numpy(np.stack([synth.ring_cameras(v, hh * 4, wh * 4, seed=s) for s in range(f)])).to(dev)
for (f, v, j, hh, wh) in ((32, 4, 19, 64, 64), (8, 8, 19, 96, 72), (256, 4, 19, 64, 64)):
    hm = torch.rand(f, v, j, hh, wh, device=dev)
    valid = torch.ones(f, j, dtype=torch.uint8, device=dev)
    def timed(fn, reps=50):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps
    b = hm.numel() * 4
    t0 = timed(lambda: _lib.argmax_decode(hm, valid, f, v, j, hh, wh, 4, hh))
    t1 = timed(lambda: _lib.score_decode_maps(_lib.SCORE_HP, hm, valid, f, v, j, hh, wh, 4, hh))
    t2 = timed(lambda: _lib.score_maps(_lib.SCORE_HP, hm, f * v * j, hh, wh))
    t3 = timed(lambda: _lib.score_decode_maps(_lib.SCORE_MPE, hm, valid, f, v, j, hh, wh, 4, hh), 10)
    pm = proj(f, v, hh, wh)
    t4 = timed(lambda: _lib.triangulate_ransac(_lib.argmax_decode(hm, valid, f, v, j, hh, wh, 4, hh), pm, valid, f, v, j, 5.0), 10)
    # heat-maps as a trained network gives them: one or two Gaussian bumps on a small noise floor (a handful of local maxima
    # instead of the ~160 of uniform noise)
    yy, xx = torch.meshgrid(torch.arange(hh, device=dev, dtype=torch.float32), torch.arange(wh, device=dev, dtype=torch.float32), indexing="ij")
    g = torch.Generator(device=dev).manual_seed(1)
    cy = torch.rand(f, v, j, 2, 1, 1, device=dev, generator=g) * hh
    cx = torch.rand(f, v, j, 2, 1, 1, device=dev, generator=g) * wh
    amp = torch.tensor([1.0, 0.4], device=dev).view(1, 1, 1, 2, 1, 1)
    bumps = (amp * torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 2.0 ** 2))).sum(3)
    # (smooth maps: every pixel of a bump's flank is NOT a 5x5 maximum; the floor adds a few shallow ones)
    _, cnt_u = _lib.score_maps(_lib.SCORE_MPE, hm, f * v * j, hh, wh)
    line = f"   MPE+decode {t3*1e6:.1f} us on uniform noise ({cnt_u.float().mean().item():.0f} peaks per map)"
    for floor in (0.002, 0.0):  # a noise floor keeps ~100 shallow maxima per map alive; without it only the bumps remain
        hs = (bumps + floor * torch.rand(f, v, j, hh, wh, device=dev, generator=g)).contiguous()
        t5 = timed(lambda: _lib.score_decode_maps(_lib.SCORE_MPE, hs, valid, f, v, j, hh, wh, 4, hh), 10)
        t6 = timed(lambda: _lib.score_decode_maps(_lib.SCORE_BSB, hs, valid, f, v, j, hh, wh, 4, hh), 10)
        _, cnt = _lib.score_maps(_lib.SCORE_MPE, hs, f * v * j, hh, wh)
        line += (f" | two Gaussian bumps + {floor} noise floor: MPE+decode {t5*1e6:.1f} us, BSB+decode {t6*1e6:.1f} us "
                 f"({cnt.float().mean().item():.1f} peaks per map, {b/t5/1e9:.0f} GB/s)")
    print(line + f" | decode+RANSAC-DLT {t4*1e6:.1f} us")
    print(f"{f}x{v}x{j} maps {hh}x{wh} ({b/1e6:.1f} MB): argmax {t0*1e6:.1f} us {b/t0/1e9:.0f} GB/s | HP+decode {t1*1e6:.1f} us {b/t1/1e9:.0f} GB/s | HP {t2*1e6:.1f} us {b/t2/1e9:.0f} GB/s")
