"""HIP execution engine for the heat-map networks.

``run_network(model, x)`` is what ``PoseHighResolutionNet.forward`` / ``PoseResNet.forward``
call.  Inference (``model.eval()``) compiles the model's layer graph
(pose_estimators/graph.py) into a *plan* for a given (N, H, W):

  * every activation gets a slot in ONE HBM arena (liveness-based reuse; NHWC fp32),
  * every conv gets its weights re-laid-out on device into the MFMA fragment order and its
    eval-mode BatchNorm folded to (scale, shift) in ONE parameter buffer -- re-done only when
    a parameter's version counter changes (load_state_dict, optimizer step),
  * the op list is handed to the C-ABI (``mval_net_create``) once; a forward is then a single
    ``mval_net_forward`` call that enqueues ~300 fused launches on torch's current stream with
    no host synchronisation, no allocation and no framework dispatch (graph-capturable).

Training mode (``model.train()``) runs the same graph through autograd Functions with
train-mode BatchNorm (engine_train.py).  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from .pose_estimators import params as _params

OP_CONV, OP_MAXPOOL, OP_DECONV, OP_BLOCK, OP_TO_P2, OP_BNECK, OP_STEM_P2, OP_FUSE_UP = 0, 1, 2, 3, 4, 5, 6, 7
ALGO_DIRECT, ALGO_MFMA, ALGO_MFMA_BF3, ALGO_MFMA_H2, ALGO_MFMA_P2 = 0, 1, 2, 3, 4
PACK_HWIO, PACK_MFMA16, PACK_MFMA16_BF3, PACK_MFMA16_H2 = 0, 1, 2, 3
AMAX_ROW = 4096
P2_ROW = 512  # include/mval_hip.h: MVAL_P2_ROW (rows of P2 activations)
_PACK_OF = {ALGO_DIRECT: PACK_HWIO, ALGO_MFMA: PACK_MFMA16, ALGO_MFMA_BF3: PACK_MFMA16_BF3, ALGO_MFMA_H2: PACK_MFMA16_H2,
            ALGO_MFMA_P2: PACK_MFMA16_H2}


class MvalOp(C.Structure):
    """include/mval_hip.h: struct mval_op."""

    _fields_ = [
        ("kind", C.c_int32), ("algo", C.c_int32),
        ("k", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("cin", C.c_int32), ("cout", C.c_int32),
        ("hin", C.c_int32), ("win", C.c_int32), ("hout", C.c_int32), ("wout", C.c_int32),
        ("up", C.c_int32), ("relu", C.c_int32), ("in_nchw", C.c_int32), ("out_nchw", C.c_int32),
        ("in_off", C.c_int64), ("out_off", C.c_int64), ("res1_off", C.c_int64), ("res2_off", C.c_int64),
        ("w_off", C.c_int64), ("scale_off", C.c_int64), ("shift_off", C.c_int64),
        ("phase", C.c_int32), ("lane", C.c_int32),
        ("in_amax_off", C.c_int64), ("out_amax_off", C.c_int64),
        ("w2_off", C.c_int64), ("scale2_off", C.c_int64), ("shift2_off", C.c_int64),
        ("bound_off", C.c_int64), ("bound2_off", C.c_int64), ("res1_amax_off", C.c_int64), ("res2_amax_off", C.c_int64),
        ("w3_off", C.c_int64), ("scale3_off", C.c_int64), ("shift3_off", C.c_int64), ("bound3_off", C.c_int64),
        ("n_terms", C.c_int32), ("t_cin", C.c_int32 * 3), ("t_up", C.c_int32 * 3), ("reserved0", C.c_int32),
        ("t_in_off", C.c_int64 * 3), ("t_in_amax_off", C.c_int64 * 3), ("t_w_off", C.c_int64 * 3), ("t_scale_off", C.c_int64 * 3),
        ("t_shift_off", C.c_int64 * 3), ("t_bound_off", C.c_int64 * 3),
    ]


_KIND = {"conv": OP_CONV, "maxpool": OP_MAXPOOL, "deconv": OP_DECONV}


def _align(n, a=64):
    return (n + a - 1) // a * a


def _conv_mode():
    """MVAL_CONV selects the conv kernel family of inference plans:
    p2 (default) -- the h2 arithmetic on "P2" activations (csrc/conv_p2.h): every activation stays in HBM as the pair
                    of fp16 planes the split MFMA consumes, written once by its producer's epilogue, so the consumers
                    stage by copying; plans whose ops the P2 kernels do not all cover (PoseResNet: max-pool,
                    transposed convs) run as h2;
    h2           -- fp32 values as scaled two-way fp16 splits on the fp16 matrix cores (3 MFMA products per
                    32-deep step, fp32 accumulate; the per-tensor power-of-two scales come from max |x| slots the
                    producers keep; measured as accurate as the fp32-MFMA chain);
    bf3          -- exact three-way bf16 splits (6 MFMA products; what training plans always use);
    fp32         -- exact fp32-input MFMA (v_mfma_f32_16x16x4_f32) everywhere."""
    mode = os.environ.get("MVAL_CONV", "p2")
    if mode not in ("p2", "h2", "bf3", "fp32"):
        raise ValueError("MVAL_CONV must be p2, h2, bf3 or fp32")
    return mode


def _mfma_ok(op, in_nchw):
    if os.environ.get("MVAL_FORCE_DIRECT") == "1":
        return False
    if op.kind == "deconv":  # ConvTranspose2d(k4, s2, p1): stride-1 conv over the zero-dilated input
        return op.cin % 16 == 0 and (op.k, op.stride, op.pad) == (4, 2, 1)
    return (op.kind == "conv" and not in_nchw and op.cin % 16 == 0 and op.k in (1, 3)
            and op.stride in (1, 2) and op.pad == op.k // 2)


def _pack_mode(op, pack):
    """mval_pack_conv_weights `transposed` argument: 0 Conv2d weights; ConvTranspose2d weights are read as
    (cin, cout, k, k) -- 1 for the direct kernel, 2 (tap-flipped: the stride-1 conv over the zero-dilated
    input) for the MFMA kernels."""
    if op.kind != "deconv":
        return 0
    # direct kernel: as stored (1); exact-fp32 MFMA: tap-flipped conv over the zero-dilated input (2);
    # split-bf16 MFMA: four 2x2 parity kernels (3)
    return {PACK_HWIO: 1, PACK_MFMA16: 2, PACK_MFMA16_BF3: 3, PACK_MFMA16_H2: 3}[pack]


class InferencePlan:
    def __init__(self, model, n, h, w, device):
        self.model, self.n, self.h, self.w, self.device = model, n, h, w, device
        g = model._graph
        self.graph = g
        # ---- activation geometry, op by op ------------------------------------------------
        dims = {g.input: (h, w)}
        geo = []
        for op in g.ops:
            hin, win = dims[op.src]
            if op.kind == "deconv":
                hout, wout = (hin - 1) * op.stride - 2 * op.pad + op.k, (win - 1) * op.stride - 2 * op.pad + op.k
            else:
                hout, wout = (hin + 2 * op.pad - op.k) // op.stride + 1, (win + 2 * op.pad - op.k) // op.stride + 1
            full = (hout << op.up, wout << op.up)
            for r in (op.res1, op.res2):
                if r is not None and dims[r] != full:
                    raise ValueError(f"input {h}x{w}: branch resolutions do not line up at {op.conv}")
            dims[op.dst] = full
            geo.append((hin, win, hout, wout))
        self.out_hw = dims[g.output]
        self.out_channels = g.acts[g.output].channels
        # ---- arena: liveness-based slot reuse --------------------------------------------------
        last_use = {}
        for i, op in enumerate(g.ops):
            for a in (op.src, op.res1, op.res2):
                if a is not None:
                    last_use[a] = i
        size = {a.id: _align(n * dims[a.id][0] * dims[a.id][1] * a.channels) for a in g.acts if a.id in dims}
        offset, free, top = {}, [], 0  # free: list of (off, size)
        pending, cur_phase = [], g.ops[0].phase if g.ops else 0
        for i, op in enumerate(g.ops):
            if op.phase != cur_phase:
                # lanes of a phase run concurrently on separate streams: a slot released inside
                # the phase may only be re-used after the join at the phase change
                free += pending
                pending, cur_phase = [], op.phase
            if op.dst != g.output:
                need = size[op.dst]
                best = None
                for k, (o, s) in enumerate(free):
                    if s >= need and (best is None or s < free[best][1]):
                        best = k
                if best is None:
                    offset[op.dst] = top
                    top += need
                else:
                    o, s = free.pop(best)
                    offset[op.dst] = o
                    if s > need:
                        free.append((o + need, s - need))
            for a in {op.src, op.res1, op.res2}:
                if a is not None and a != g.input and last_use.get(a) == i:
                    pending.append((offset[a], size[a]))
            # coalesce neighbours
            free.sort()
            merged = []
            for o, s in free:
                if merged and merged[-1][0] + merged[-1][1] == o:
                    merged[-1] = (merged[-1][0], merged[-1][1] + s)
                else:
                    merged.append((o, s))
            free = merged
        # max |x| slots (one float each) behind the activations: what the fp16-split convs scale their input by
        self.amax_base = _align(max(top, 64))
        # (n rows of AMAX_ROW dwords per activation that an fp16-split conv reads; include/mval_hip.h: MVAL_AMAX_ROW)
        self._amax_top = self.amax_base
        amax_slot = {}
        # ---- parameter buffer layout -------------------------------------------------------------
        lib = _lib.lib()
        self.ops = (MvalOp * len(g.ops))()
        self.param_jobs = []  # (op index, packing, w_off, scale_off, shift_off)
        ptop = 0
        # P2 plan (csrc/conv_p2.h): every op but the image stem reads and writes fp16-pair planes; all or nothing
        self.p2 = _conv_mode() == "p2" and os.environ.get("MVAL_FORCE_DIRECT") != "1" and self._p2_covers(lib, g, geo, n)
        mode = "h2" if _conv_mode() == "p2" else _conv_mode()
        row_of = {}  # P2: activation id -> float offset of its n rows
        if self.p2:
            for a in g.acts:
                if a.id in dims and a.id not in (g.input, g.output):
                    row_of[a.id] = self._amax_top
                    self._amax_top += n * P2_ROW
        for i, op in enumerate(g.ops):
            hin, win, hout, wout = geo[i]
            in_nchw = g.acts[op.src].layout == "nchw"
            out_nchw = g.acts[op.dst].layout == "nchw"
            m = self.ops[i]
            m.kind = _KIND[op.kind]
            m.k, m.stride, m.pad, m.cin, m.cout = op.k, op.stride, op.pad, op.cin, op.cout
            m.hin, m.win, m.hout, m.wout = hin, win, hout, wout
            m.up, m.relu, m.in_nchw, m.out_nchw = op.up, int(op.relu), int(in_nchw), int(out_nchw)
            m.algo = ALGO_DIRECT
            if _mfma_ok(op, in_nchw) and lib.mval_op_mfma_supported(C.byref(m), C.c_int(n)):
                m.algo = ALGO_MFMA
                # fp32-accurate 16-bit splits on the matrix cores (fp16x2: 5.3x, bf16x3: 2.67x the fp32-MFMA rate)
                # (the fp16 split wants one image per tile; maps under 8 rows fall back to bf16x3)
                if (op.k in (1, 3) or op.kind == "deconv") and (op.cin % 32 == 0 or op.cin == 48) and op.src != g.input:
                    for split in {"h2": (ALGO_MFMA_H2, ALGO_MFMA_BF3), "bf3": (ALGO_MFMA_BF3,)}.get(mode, ()):
                        if lib.mval_op_algo_supported(C.byref(m), C.c_int(n), C.c_int(split)):
                            m.algo = split
                            break
            if self.p2 and op.src != g.input and op.kind != "maxpool":
                m.algo = ALGO_MFMA_P2
            m.in_off = -1 if op.src == g.input else offset[op.src]
            m.out_off = -1 if op.dst == g.output else offset[op.dst]
            m.res1_off = -1 if op.res1 is None else offset[op.res1]
            m.res2_off = -1 if op.res2 is None else offset[op.res2]
            m.w_off = m.scale_off = m.shift_off = -1
            m.phase, m.lane = op.phase, op.lane
            if m.algo == ALGO_MFMA_H2:
                if op.src not in amax_slot:
                    amax_slot[op.src] = self._amax_top
                    self._amax_top += n * AMAX_ROW
                m.in_amax_off = amax_slot[op.src]
            if op.kind in ("conv", "deconv"):
                pack = _PACK_OF[m.algo]
                nw = int(lib.mval_packed_weight_floats(C.c_int(pack), C.c_int(op.cout), C.c_int(op.cin), C.c_int(op.k)))
                m.w_off = ptop
                ptop += _align(nw)
                m.scale_off = ptop
                ptop += _align(op.cout)
                m.shift_off = ptop
                ptop += _align(op.cout)
                if m.algo == ALGO_MFMA_P2 or (self.p2 and op.src == g.input):  # (the stem's bound: the fused P2 stem needs it)
                    m.bound_off = ptop
                    ptop += 64
                self.param_jobs.append((i, pack, m.w_off, m.scale_off, m.shift_off))
        # producers keep max |x| only for tensors an fp16-split conv reads
        if g.input in amax_slot:
            raise _lib.MvalError("the network input cannot feed an fp16-split conv (no producer to keep its max |x|)")
        for i, op in enumerate(g.ops):
            self.ops[i].out_amax_off = amax_slot.get(op.dst, 0)
        self.graph_ops = self.ops  # one per graph op (what param_jobs index); self.ops becomes the launch list
        if self.p2:
            self.ops = self._p2_launch_list(g, n, dims, offset, row_of)
        else:
            self.ops = self._fuse_blocks(lib, g, n)
        self.arena_floats = _align(self._amax_top)
        self.param_floats = max(ptop, 64)
        self.arena = torch.empty(self.arena_floats, dtype=torch.float32, device=device)
        # rows behind the activations start as zeros: P2 rows need it (a partial slot belongs to ONE producing workgroup, the others
        # must read as zero); the [count, partials ...] rows of the h2 kernels then read "no partials" until their producer has run
        self.arena[self.amax_base :].zero_()
        self.params = torch.zeros(self.param_floats, dtype=torch.float32, device=device)
        self.param_sig = None
        self._graph, self._graph_failed = None, False
        # P2 plans: (offset of the n rows) of every activation kept in HBM, for the bound-vs-actual check (p2_slack)
        self._p2_rows = sorted(set(row_of.values())) if self.p2 else []
        self._slack_pending, self.p2_slack = False, None
        self.net = lib.mval_net_create(self.ops, C.c_int(len(self.ops)))
        if not self.net:
            raise _lib.MvalError("mval_net_create failed: " + lib.mval_last_error().decode())

    @staticmethod
    def _p2_covers(lib, g, geo, n):
        """Every op after the image stem has a P2 kernel (HRNet: yes; PoseResNet's max-pool / transposed convs: no)."""
        if os.environ.get("MVAL_P2", "1") == "0":
            return False
        stems = 0
        stem_dst = None
        for op, (hin, win, hout, wout) in zip(g.ops, geo):
            if op.src == g.input:
                stems += 1
                stem_dst = op.dst
                if op.kind != "conv" or op.cout % 8 or op.res1 is not None or op.res2 is not None or op.up:
                    return False
                continue
            if op.kind == "maxpool":
                # (round 5: PoseResNet, pose_resnet.py:35) a max-pool directly behind the stem runs on the stem's fp32 NHWC output, before
                # the change to planes; anywhere else there is no P2 kernel for it
                if op.src != stem_dst or op.cout % 8 or sum(1 for o in g.ops if stem_dst in (o.src, o.res1, o.res2)) != 1:
                    return False
                continue
            if op.kind not in ("conv", "deconv") or g.acts[op.src].layout == "nchw":
                return False
            # a transposed conv is four parity launches of persistent workgroups: on a few images (BASELINE configs[0]: 8) the h2 plan's
            # one launch with small tiles is ahead (1.55 vs 1.68 ms), from ~32 images on the P2 plan (C1 x 16: 7.2 vs 7.8 ms)
            if op.kind == "deconv" and n < 32 and os.environ.get("MVAL_P2") != "force":
                return False
            # HRNet-W48 (48- / 96-channel branches: half-empty second K chunk; 24 x 18 and 12 x 9 maps) ran SLOWER on P2 than on
            # the h2 kernels through round 3 (C4 24.8 vs 22.4 ms); with round 4's full-width odd tiles (conv_p2.hip OW) and the
            # 48- / 96-channel fused up-paths it is ahead (21.8 vs 22.1 ms).  MVAL_P2_W48=0 keeps such plans on h2.
            if op.cin % 32 and os.environ.get("MVAL_P2_W48", "1") == "0" and os.environ.get("MVAL_P2") != "force":
                return False
            m = MvalOp()
            m.kind, m.algo = _KIND[op.kind], ALGO_MFMA_P2
            m.res1_off = -1 if op.res1 is None else 0
            m.res2_off = -1 if op.res2 is None else 0
            m.k, m.stride, m.pad, m.cin, m.cout = op.k, op.stride, op.pad, op.cin, op.cout
            m.hin, m.win, m.hout, m.wout = hin, win, hout, wout
            m.up, m.relu, m.out_nchw = op.up, int(op.relu), int(g.acts[op.dst].layout == "nchw")
            if not lib.mval_op_algo_supported(C.byref(m), C.c_int(n), C.c_int(ALGO_MFMA_P2)):
                return False
        return stems == 1

    def _p2_launch_list(self, g, n, dims, offset, row_of):
        """Launch list of a P2 plan: the stem conv keeps its fp32 NHWC kernel and writes into a slot of its own, a
        MVAL_OP_TO_P2 launch turns that into planes; every other op is MVAL_ALGO_MFMA_P2 with the rows of its input,
        residuals and output; BasicBlocks of the 32- / 64-channel branches (hrnet.py:36-52) become ONE MVAL_OP_BLOCK
        launch (csrc/conv_block_p2.hip; MVAL_FUSE_BLOCKS=0 keeps the pair: the on-device cross-check)."""
        lib = _lib.lib()
        fuse = os.environ.get("MVAL_FUSE_BLOCKS", "1") != "0"
        # measured (128 images): 32 channels on 64x64 maps 81 us fused vs 2 x 44 us; 64 channels on 32x32 maps 77 us
        # fused vs 2 x 33 us -- the two 64-channel convs are no longer HBM-bound one by one, so only the 32-channel
        # blocks are fused (round 3 measured the fused 64-channel block at 77 us against 2 x 33 for its two P2 convs; 10.38 vs 10.37 ms as a step)
        fuse_c = {32}
        fuse_bneck = fuse and os.environ.get("MVAL_P2_BNECK", "1") != "0"
        fuse_up = fuse and os.environ.get("MVAL_P2_FUSE_UP", "1") != "0"
        uses = {}
        for op in g.ops:
            for a in (op.src, op.res1, op.res2):
                if a is not None:
                    uses[a] = uses.get(a, 0) + 1
        launch, i = [], 0
        while i < len(g.ops):
            op = g.ops[i]
            m = MvalOp()
            C.memmove(C.byref(m), C.byref(self.graph_ops[i]), C.sizeof(MvalOp))
            if op.src == g.input:
                b = g.ops[i + 1] if i + 1 < len(g.ops) else None
                if (fuse and os.environ.get("MVAL_P2_STEM", "1") != "0" and b is not None and op.kind == b.kind == "conv" and op.k == b.k == 3
                        and op.stride == b.stride == 2 and op.pad == b.pad == 1 and op.cin == 3 and op.cout == b.cin == b.cout == 64 and op.bn and b.bn
                        and op.relu and b.relu and b.src == op.dst and uses.get(op.dst, 0) == 1 and b.res1 is None and b.res2 is None
                        and op.up == b.up == 0 and b.dst != g.output and (op.phase, op.lane) == (b.phase, b.lane)):
                    # hrnet.py:303-310: both stride-2 stem convs in ONE launch, the 64-channel half-resolution map never leaves the CU
                    mb = self.graph_ops[i + 1]
                    st = MvalOp()
                    C.memmove(C.byref(st), C.byref(m), C.sizeof(MvalOp))
                    st.kind, st.algo = OP_STEM_P2, ALGO_MFMA_P2
                    st.hout, st.wout = dims[b.dst]
                    st.out_off, st.out_amax_off = mb.out_off, row_of[b.dst]
                    st.in_amax_off = self._amax_top  # rows of the network input: the launch keeps the images' max |x| there
                    st.w2_off, st.scale2_off, st.shift2_off, st.bound2_off = mb.w_off, mb.scale_off, mb.shift_off, mb.bound_off
                    if lib.mval_op_algo_supported(C.byref(st), C.c_int(n), C.c_int(ALGO_MFMA_P2)):
                        self._amax_top += n * P2_ROW
                        launch.append(st)
                        i += 2
                        continue
                ho, wo = dims[op.dst]
                stem_floats = _align(n * ho * wo * op.cout)
                stem_off = self._amax_top  # (behind the rows: only this plan form needs it)
                self._amax_top += stem_floats
                stem_rows = self._amax_top
                self._amax_top += n * AMAX_ROW
                m.out_off, m.out_amax_off = stem_off, stem_rows
                launch.append(m)
                last, src_off, src_rows = op, stem_off, stem_rows
                b = g.ops[i + 1] if i + 1 < len(g.ops) else None
                if b is not None and b.kind == "maxpool" and b.src == op.dst:
                    # pose_resnet.py:35: the max-pool reads the stem's fp32 NHWC output and writes fp32 NHWC (a quarter of the pixels);
                    # the change to planes follows it (the stem's own rows are not needed then)
                    mp = MvalOp()
                    C.memmove(C.byref(mp), C.byref(self.graph_ops[i + 1]), C.sizeof(MvalOp))
                    hp, wp = dims[b.dst]
                    mp.in_off = stem_off
                    mp.out_off = self._amax_top
                    self._amax_top += _align(n * hp * wp * b.cout)
                    mp.out_amax_off = self._amax_top
                    self._amax_top += n * AMAX_ROW
                    m.out_amax_off = 0
                    launch[-1] = m
                    launch.append(mp)
                    last, src_off, src_rows = b, mp.out_off, mp.out_amax_off
                    ho, wo = hp, wp
                    i += 1
                t = MvalOp()
                t.kind, t.algo = OP_TO_P2, ALGO_MFMA_P2
                t.hin, t.win, t.cin, t.hout, t.wout, t.cout = ho, wo, last.cout, ho, wo, last.cout
                t.in_off, t.in_amax_off = src_off, src_rows
                t.out_off, t.out_amax_off = offset[last.dst], row_of[last.dst]
                t.res1_off = t.res2_off = t.w_off = t.scale_off = t.shift_off = -1
                t.phase, t.lane = m.phase, m.lane
                launch.append(t)
                i += 1
                continue
            m.in_amax_off = row_of[op.src]
            m.res1_amax_off = row_of[op.res1] if op.res1 is not None else 0
            m.res2_amax_off = row_of[op.res2] if op.res2 is not None else 0
            m.out_amax_off = row_of.get(op.dst, 0)
            chain = self._up_chain_at(g, i, uses) if fuse_up else None
            if chain is not None:
                # hrnet.py:424-447: the consecutive up-sampling terms of a fuse-layer output in ONE launch; the partial sum is read
                # once and written once instead of once per term
                last = g.ops[chain[-1]]
                fu = MvalOp()
                C.memmove(C.byref(fu), C.byref(m), C.sizeof(MvalOp))
                fu.kind, fu.relu, fu.up = OP_FUSE_UP, int(last.relu), 0
                fu.hin, fu.win = fu.hout, fu.wout = dims[last.dst]
                fu.out_off, fu.out_amax_off = self.graph_ops[chain[-1]].out_off, row_of[last.dst]
                fu.res1_off, fu.res1_amax_off, fu.res2_off, fu.res2_amax_off = m.res1_off, row_of[op.res1], -1, 0
                fu.n_terms = len(chain)
                for j, k in enumerate(chain):
                    gk, ok_ = self.graph_ops[k], g.ops[k]
                    fu.t_cin[j], fu.t_up[j] = ok_.cin, ok_.up
                    fu.t_in_off[j], fu.t_in_amax_off[j] = gk.in_off, row_of[ok_.src]
                    fu.t_w_off[j], fu.t_scale_off[j], fu.t_shift_off[j], fu.t_bound_off[j] = gk.w_off, gk.scale_off, gk.shift_off, gk.bound_off
                if lib.mval_op_algo_supported(C.byref(fu), C.c_int(n), C.c_int(ALGO_MFMA_P2)):
                    launch.append(fu)
                    i = chain[-1] + 1
                    continue
            bn = self._bneck_at(g, i, uses) if fuse_bneck else None
            if bn is not None:
                # hrnet.py:75-95 with 64 planes: conv1x1 -> conv3x3 -> conv1x1 (+ residual) in ONE launch; a downsample branch
                # (1x1 conv of the block's input, emitted between conv2 and conv3) runs first as its own op
                i2, i3, skip = bn
                for k in skip:
                    ms = MvalOp()
                    C.memmove(C.byref(ms), C.byref(self.graph_ops[k]), C.sizeof(MvalOp))
                    ks = g.ops[k]
                    ms.in_amax_off, ms.out_amax_off = row_of[ks.src], row_of[ks.dst]
                    launch.append(ms)
                m2, m3, o3 = self.graph_ops[i2], self.graph_ops[i3], g.ops[i3]
                blk = MvalOp()
                C.memmove(C.byref(blk), C.byref(m), C.sizeof(MvalOp))
                blk.kind, blk.cout, blk.relu = OP_BNECK, o3.cout, 1
                blk.out_off, blk.res1_off, blk.res2_off = m3.out_off, m3.res1_off, -1
                blk.res1_amax_off, blk.res2_amax_off, blk.out_amax_off = row_of[o3.res1], 0, row_of[o3.dst]
                blk.w2_off, blk.scale2_off, blk.shift2_off, blk.bound2_off = m2.w_off, m2.scale_off, m2.shift_off, m2.bound_off
                blk.w3_off, blk.scale3_off, blk.shift3_off, blk.bound3_off = m3.w_off, m3.scale_off, m3.shift_off, m3.bound_off
                if lib.mval_op_algo_supported(C.byref(blk), C.c_int(n), C.c_int(ALGO_MFMA_P2)):
                    launch.append(blk)
                    i = i3 + 1
                    continue
                del launch[len(launch) - len(skip):]
            b = g.ops[i + 1] if i + 1 < len(g.ops) else None
            if (fuse and b is not None and op.kind == b.kind == "conv" and op.k == b.k == 3 and op.stride == b.stride == 1
                    and op.cin == op.cout == b.cin == b.cout and op.cin in fuse_c and op.bn and b.bn and op.relu and b.relu and op.res1 is None
                    and op.res2 is None and op.up == b.up == 0 and b.src == op.dst and b.res1 == op.src and b.res2 is None
                    and uses.get(op.dst, 0) == 1 and op.dst != g.output and (op.phase, op.lane) == (b.phase, b.lane)):
                mb = self.graph_ops[i + 1]
                blk = MvalOp()
                C.memmove(C.byref(blk), C.byref(m), C.sizeof(MvalOp))
                blk.kind = OP_BLOCK
                blk.out_off, blk.res1_off, blk.res2_off = mb.out_off, m.in_off, -1
                blk.out_amax_off = row_of[b.dst]
                blk.w2_off, blk.scale2_off, blk.shift2_off, blk.bound2_off = mb.w_off, mb.scale_off, mb.shift_off, mb.bound_off
                if lib.mval_op_algo_supported(C.byref(blk), C.c_int(n), C.c_int(ALGO_MFMA_P2)):
                    launch.append(blk)
                    i += 2
                    continue
            launch.append(m)
            i += 1
        arr = (MvalOp * len(launch))()
        for k, m in enumerate(launch):
            C.memmove(C.byref(arr[k]), C.byref(m), C.sizeof(MvalOp))
        return arr

    @staticmethod
    def _up_chain_at(g, i, uses):
        """Ops i, i + 1 [, i + 2] are the consecutive up-sampling 1x1 terms of one fuse-layer output (each adds to the previous one's
        result, only the last may have the ReLU) -> their indices, else None."""
        chain = []
        k = i
        while k < len(g.ops) and len(chain) < 3:
            o = g.ops[k]
            if not (o.kind == "conv" and o.k == 1 and o.stride == 1 and o.up >= 1 and o.bn and o.res1 is not None and o.res2 is None
                    and o.cin % 32 == 0 and o.cout in (32, 64, 48, 96) and o.dst != g.output):
                break
            if chain:
                p = g.ops[chain[-1]]
                if not (o.res1 == p.dst and uses.get(p.dst, 0) == 1 and not p.relu and o.cout == p.cout
                        and (o.phase, o.lane) == (p.phase, p.lane)):
                    break
            chain.append(k)
            k += 1
        return chain if len(chain) >= 2 else None

    @staticmethod
    def _bneck_at(g, i, uses):
        """Ops i .. of the graph form a Bottleneck with 64 planes the fused P2 kernel covers -> (index of conv2, index of conv3,
        indices of ops in between that must run first: the downsample branch), else None."""
        a = g.ops[i]
        if not (a.kind == "conv" and a.k == 1 and a.stride == 1 and a.cout == 64 and a.cin in (64, 256) and a.bn and a.relu
                and a.res1 is None and a.res2 is None and a.up == 0 and uses.get(a.dst, 0) == 1 and i + 2 < len(g.ops)):
            return None
        b = g.ops[i + 1]
        if not (b.kind == "conv" and b.k == 3 and b.stride == 1 and b.cin == b.cout == 64 and b.src == a.dst and b.bn and b.relu
                and b.res1 is None and b.res2 is None and b.up == 0 and uses.get(b.dst, 0) == 1):
            return None
        skip, j = [], i + 2
        if g.ops[j].src == a.src and g.ops[j].kind == "conv" and g.ops[j].k == 1 and j + 1 < len(g.ops):  # the downsample branch
            d = g.ops[j]
            if d.res1 is not None or d.res2 is not None or d.up or d.relu or uses.get(d.dst, 0) != 1:
                return None
            skip.append(j)
            j += 1
        c = g.ops[j]
        if not (c.kind == "conv" and c.k == 1 and c.stride == 1 and c.cin == 64 and c.cout == 256 and c.src == b.dst and c.bn and c.relu
                and c.res1 is not None and c.res2 is None and c.up == 0 and c.dst != g.output):
            return None
        if c.res1 != (g.ops[skip[0]].dst if skip else a.src):
            return None
        if len({(o.phase, o.lane) for o in [a, b, c] + [g.ops[k] for k in skip]}) != 1:
            return None
        return i + 1, j, skip

    def _fuse_blocks(self, lib, g, n):
        """Launch list: every BasicBlock of the 32- / 64-channel branches (conv3x3+BN+ReLU -> conv3x3+BN+residual+ReLU,
        hrnet.py:36-52) whose two convs run on the fp16-split kernels becomes ONE MVAL_OP_BLOCK launch
        (csrc/conv_block.hip); everything else is launched op by op.  MVAL_FUSE_BLOCKS=0 keeps the unfused pair (the
        on-device cross-check of the fused kernel, tests/test_gpu_models.py)."""
        fuse = os.environ.get("MVAL_FUSE_BLOCKS", "1") != "0"
        uses = {}
        for op in g.ops:
            for a in (op.src, op.res1, op.res2):
                if a is not None:
                    uses[a] = uses.get(a, 0) + 1
        launch, i = [], 0
        while i < len(g.ops):
            a = g.ops[i]
            b = g.ops[i + 1] if i + 1 < len(g.ops) else None
            ma = self.graph_ops[i]
            if (fuse and b is not None and a.kind == b.kind == "conv" and a.k == b.k == 3 and a.stride == b.stride == 1
                    and a.pad == b.pad == 1 and a.cin == a.cout == b.cin == b.cout and a.cin in (32, 48, 64) and a.bn and b.bn
                    and a.relu and b.relu and a.res1 is None and a.res2 is None and a.up == b.up == 0 and b.src == a.dst
                    and b.res1 == a.src and b.res2 is None and uses.get(a.dst, 0) == 1 and a.dst != g.output
                    and (a.phase, a.lane) == (b.phase, b.lane)
                    and ma.algo == self.graph_ops[i + 1].algo == ALGO_MFMA_H2):
                mb = self.graph_ops[i + 1]
                blk = MvalOp()
                C.memmove(C.byref(blk), C.byref(ma), C.sizeof(MvalOp))
                blk.kind = OP_BLOCK
                blk.out_off, blk.res1_off, blk.res2_off = mb.out_off, ma.in_off, -1
                blk.out_amax_off = mb.out_amax_off
                blk.w2_off, blk.scale2_off, blk.shift2_off = mb.w_off, mb.scale_off, mb.shift_off
                if lib.mval_op_algo_supported(C.byref(blk), C.c_int(n), C.c_int(ALGO_MFMA_H2)):
                    launch.append(blk)
                    i += 2
                    continue
            launch.append(ma)
            i += 1
        arr = (MvalOp * len(launch))()
        for k, m in enumerate(launch):
            C.memmove(C.byref(arr[k]), C.byref(m), C.sizeof(MvalOp))
        return arr

    def __del__(self):
        try:
            if getattr(self, "net", None):
                _lib.lib().mval_net_destroy(C.c_void_p(self.net))
        except Exception:
            pass

    # ---- parameters ---------------------------------------------------------------------
    def _signature(self):
        return _param_signature(self.model)

    def refresh_params(self, force=False):
        sig = self._signature()
        if not force and sig == self.param_sig:
            return
        lib = _lib.lib()
        holders = self.model._holders
        st = _lib._stream()
        base = self.params.data_ptr()
        for i, pack, w_off, s_off, b_off in self.param_jobs:
            op = self.graph.ops[i]
            conv = holders[op.conv]
            w = conv.weight.detach()
            if not w.is_cuda:
                raise _lib.MvalError("model parameters must be on the HIP device (call .cuda())")
            w = w.contiguous()
            _lib._check(
                lib.mval_pack_conv_weights(
                    C.c_int(pack), C.c_int(_pack_mode(op, pack)), C.c_void_p(w.data_ptr()),
                    C.c_void_p(base + 4 * w_off), C.c_int(op.cout), C.c_int(op.cin), C.c_int(op.k), st),
                "mval_pack_conv_weights")
            if op.bn:
                bn = holders[op.bn]
                _lib._check(
                    lib.mval_bn_fold(
                        C.c_void_p(bn.weight.data_ptr()), C.c_void_p(bn.bias.data_ptr()),
                        C.c_void_p(bn.running_mean.data_ptr()), C.c_void_p(bn.running_var.data_ptr()),
                        C.c_float(bn.eps), C.c_void_p(base + 4 * s_off), C.c_void_p(base + 4 * b_off),
                        C.c_int(op.cout), st),
                    "mval_bn_fold")
            else:
                self.params[s_off : s_off + op.cout] = 1.0
                if conv.bias is not None:
                    self.params[b_off : b_off + op.cout] = conv.bias.detach()
                else:
                    self.params[b_off : b_off + op.cout] = 0.0
            gm = self.graph_ops[i]
            if gm.algo == ALGO_MFMA_P2 or (self.p2 and op.src == self.graph.input):
                # [A, B] of the output bound |bn(conv(x))| <= A max|x| + B (csrc/conv_p2.h): A = max_c |scale_c| sum |w_c|
                if op.kind == "deconv":
                    # ConvTranspose2d weights are (cin, cout, 4, 4) and an output pixel of parity (py, px) sees the taps ky in {3 - py, 1 - py},
                    # kx alike (conv_mfma_split.hip pack mode 3): per cout the largest of the four parities' sums
                    wa = w.abs().double()
                    sw = torch.stack([wa[:, :, [3 - py, 1 - py]][:, :, :, [3 - px, 1 - px]].sum(dim=(0, 2, 3)) for py in (0, 1) for px in (0, 1)]).max(dim=0).values
                else:
                    sw = w.abs().double().sum(dim=(1, 2, 3))
                a_ = (sw * self.params[s_off : s_off + op.cout].abs().double()).max() * (1.0 + 1e-6)
                b_ = self.params[b_off : b_off + op.cout].abs().double().max()
                self.params[gm.bound_off : gm.bound_off + 2] = torch.stack([a_, b_]).to(torch.float32)
        self.param_sig = sig
        self._slack_pending = self.p2 and os.environ.get("MVAL_P2_SLACK_CHECK", "1") != "0"

    def p2_slack_log2(self):
        """log2 of the largest (output bound / actual max |x|) over the P2 activations of the LAST forward and its images
        (csrc/conv_p2.h: the scale of a P2 tensor comes from an a-priori bound; values more than ~2^16 below the bound lose
        low-part bits).  Synthetic variance-preserving weights: <= 8 except the fused Bottlenecks' outputs (11 .. 12.6: three
        chained bounds); folded BatchNorm statistics with a wide per-channel spread can compound it further.
        One device -> host copy."""
        if not self._p2_rows:
            return 0.0
        idx = torch.tensor(self._p2_rows, dtype=torch.int64, device=self.device)
        rows = self.arena.view(torch.int32)[(idx[:, None] + torch.arange(self.n * P2_ROW, device=self.device)[None, :])].reshape(-1, P2_ROW)
        amax = rows[:, : P2_ROW // 2].view(torch.float32).max(dim=1).values
        inv = rows[:, P2_ROW - 1 : P2_ROW].view(torch.float32)[:, 0]
        ok = (amax > 0) & (inv > 0) & torch.isfinite(amax)
        if not bool(ok.any()):
            return 0.0
        slack = torch.log2(8192.0 * inv[ok] / amax[ok])  # the bound sits in [2^13, 2^14) of the scaled range
        return float(slack.max())

    def _check_p2_slack(self):
        self._slack_pending = False
        self.p2_slack = self.p2_slack_log2()
        if self.p2_slack > P2_MAX_SLACK_LOG2 and os.environ.get("MVAL_P2", "1") != "force":
            raise P2SlackError(self.p2_slack)

    # ---- run ---------------------------------------------------------------------------------
    def _keys_wanted(self):
        """Decode from the heat-map layer's epilogue: the plan's last kernel also keeps arg-max keys of every
        map it stores (mval_net_forward_keys), unless it runs on a generic kernel or MVAL_EPILOGUE_DECODE=0."""
        return _lib.epilogue_decode_enabled() and bool(_lib.lib().mval_net_keeps_argmax_keys(C.c_void_p(self.net)))

    def _launch(self, x, out, keys=None):
        if keys is not None:
            _lib._check(
                _lib.lib().mval_net_forward_keys(
                    C.c_void_p(self.net), C.c_int(self.n), C.c_void_p(self.arena.data_ptr()),
                    C.c_void_p(self.params.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()),
                    C.c_void_p(keys.data_ptr()), _lib._stream()),
                "mval_net_forward_keys")
            return
        _lib._check(
            _lib.lib().mval_net_forward(
                C.c_void_p(self.net), C.c_int(self.n), C.c_void_p(self.arena.data_ptr()),
                C.c_void_p(self.params.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()),
                _lib._stream()),
            "mval_net_forward")

    def _new_keys(self):
        return torch.empty((self.n, _lib.ARGMAX_SLOTS, self.out_channels), dtype=torch.int64, device=self.device) if self._keys_wanted() else None

    def _graph_wanted(self):
        """Small plans and every multi-stream plan replay a captured hipGraph of their forward (MVAL_GRAPH=0: eager launches, =1: replay
        always).  The reference's default batches are 2
        frames (config.py:67,87): ~300 launches of a few microseconds of work each, i.e. launch-bound -- rounds 1-4 replayed plans of up
        to 32 images only.  Round 5 measured the large ones as well: the multi-stream forward's ~240 launches and its fork / join events
        cost the device less as graph nodes than as host-enqueued packets, also at 128 images -- C2 10.22 -> 10.03 ms, C4 18.14 -> 17.77
        ms per step, twice each on one box (profiles/r05/graph_replay.log), the copy of the batch into the graph's input buffer and of
        the heat-maps out of its output buffer included."""
        mode = os.environ.get("MVAL_GRAPH", "auto")
        if mode == "0" or self._graph_failed:
            return False
        # (single-stream plans -- PoseResNet -- gain nothing from the replay at large batches: C1 x 16 6.32 eager vs 6.36 ms replayed)
        return mode == "1" or self.n * self.h * self.w <= 32 * 256 * 256 or any(int(o.lane) > 0 for o in self.ops)

    def _capture(self, x):
        self._gx = torch.empty_like(x)
        self._gout = torch.empty((self.n, self.out_channels) + tuple(self.out_hw), dtype=torch.float32, device=self.device)
        self._gkeys = self._new_keys()
        self._gx.copy_(x)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):  # warm-up outside the capture (lazy stream / event creation)
            self._launch(self._gx, self._gout, self._gkeys)
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        # (thread-local capture mode: under torch.distributed the RCCL watchdog thread of the process queries its events while this
        # thread captures -- in the default global mode any such call from another thread can invalidate the capture)
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            self._launch(self._gx, self._gout, self._gkeys)
        self._graph = graph

    def forward(self, x):
        self.refresh_params()
        if self._graph_wanted():
            if self._graph is None:
                try:
                    self._capture(x)
                except Exception:  # capture unsupported in this context: fall back to eager launches for good
                    self._graph, self._graph_failed = None, True
                    torch.cuda.synchronize(self.device)
            if self._graph is not None:
                self._gx.copy_(x)
                self._graph.replay()
                if self._slack_pending:
                    self._check_p2_slack()
                out = self._gout.clone()
                if self._gkeys is not None:
                    _lib.remember_argmax_keys(out, self._gkeys.clone())
                return out
        out = torch.empty((self.n, self.out_channels) + tuple(self.out_hw), dtype=torch.float32, device=self.device)
        keys = self._new_keys()
        self._launch(x, out, keys)
        if self._slack_pending:  # first forward with these parameters: is the a-priori bound close enough to the activations?
            self._check_p2_slack()
        if keys is not None:
            _lib.remember_argmax_keys(out, keys)
        return out

    def forward_timed(self, x):
        """Forward with a hipEvent around every op: returns (out, per-op ms list, per-op FLOPs)."""
        import numpy as np

        self.refresh_params()
        out = torch.empty((self.n, self.out_channels) + tuple(self.out_hw), dtype=torch.float32, device=self.device)
        ms = (C.c_float * len(self.ops))()
        _lib._check(
            _lib.lib().mval_net_forward_timed(
                C.c_void_p(self.net), C.c_int(self.n), C.c_void_p(self.arena.data_ptr()),
                C.c_void_p(self.params.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()),
                _lib._stream(), ms),
            "mval_net_forward_timed")
        flops = [float(_lib.lib().mval_op_flops(C.byref(self.ops[i]), C.c_int(self.n))) for i in range(len(self.ops))]
        return out, np.asarray(list(ms), dtype=np.float64), np.asarray(flops)

    def run_op(self, i, x, out):
        """Launch a single GRAPH op, unfused (debug / layer-wise tests)."""
        _lib._check(
            _lib.lib().mval_op_launch(
                C.byref(self.graph_ops[i]), C.c_int(self.n), C.c_void_p(self.arena.data_ptr()),
                C.c_void_p(self.params.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()),
                _lib._stream()),
            "mval_op_launch")


# Every register_parameter / register_buffer / register_module in the process -- `m.bias = nn.Parameter(...)` on a bias that was None, a
# buffer added later, a submodule swapped inside a holder: nn.Module.__setattr__ goes through the same hooks -- moves this epoch; the cached
# (dict, key) slots of _param_signature are rebuilt when it has moved since they were taken (the slots never saw those).
_REG_EPOCH = [0]


def _bump_reg_epoch(*_a):
    _REG_EPOCH[0] += 1
    return None  # (keep the value being registered)


for _reg in ("register_module_parameter_registration_hook", "register_module_buffer_registration_hook", "register_module_module_registration_hook"):
    getattr(torch.nn.modules.module, _reg)(_bump_reg_epoch)


def _param_signature(model):
    """(version counters of every parameter and buffer, data pointers of the parameters): what a plan's packed weights depend on.
    Evaluated on EVERY forward, so the walk over the module tree is done once (nn.Module.parameters() / .buffers() are generators over
    named_modules: 4 400 calls and ~2 ms of host time per HRNet forward -- a third of what a 64-image batch takes on the device) and
    kept as (owning dict, key) slots; a slot is read through its module's own dict, so a parameter REPLACED by assignment
    (`conv.weight = nn.Parameter(...)`, `.to(device)`) is seen like one modified in place.  The slots are dropped when anything was
    registered since (the epoch above), when an entry was deleted (KeyError) and when they belong to another model (a deepcopy)."""
    slots = model.__dict__.get("_sig_slots")
    for _ in range(2):
        if slots is None or slots[2] != _REG_EPOCH[0] or slots[3] != id(model):
            ps, bs = [], []
            for h in model._holders.values():
                for m in h.modules():
                    ps += [(m._parameters, k) for k, v in m._parameters.items() if v is not None]
                    bs += [(m._buffers, k) for k, v in m._buffers.items() if v is not None]
            slots = model.__dict__["_sig_slots"] = (ps, bs, _REG_EPOCH[0], id(model))
        ps, bs = slots[0], slots[1]
        try:
            return tuple(d[k]._version for d, k in ps) + tuple(d[k]._version for d, k in bs) + tuple(d[k].data_ptr() for d, k in ps)
        except (KeyError, AttributeError):  # (an entry deleted or set to None since the walk)
            slots = None
    raise RuntimeError("parameter signature: the module tree changed during the walk")


# bound / actual maximum above which a P2 plan hands over to h2.  The pair (h, l) keeps all 22 significand bits of a value whose
# SCALED magnitude is at least 2^-2 (below, l runs into fp16's subnormal spacing: absolute error 2^-25 scaled); the bound is put
# at 2^13.5, so values a decade below a tensor's maximum keep 22 bits up to a slack of ~2^12 and lose one bit per further
# factor of two: 19 bits (relative 2^-20, the size of the fp32 accumulation error of a 3x3x64 dot product) at 2^15.  Measured on
# the synthetic BASELINE weights (tools/p2_slack.py, 128 images): the fused Bottlenecks' outputs 2^11.3 .. 2^12.6 (three chained
# per-layer bounds), every other tensor <= 2^8.1 -- the arg-max census and the goldens hold there with margin.
P2_MAX_SLACK_LOG2 = 15.0


class P2SlackError(_lib.MvalError):
    """The a-priori output bounds of a P2 plan sit too far above the activations of these parameters (run_network then uses
    the h2 plan: same arithmetic, scales from the exact per-image maxima)."""

    def __init__(self, slack):
        super().__init__(f"P2 plan: output bound 2^{slack:.1f} above the activations' maximum (limit 2^{P2_MAX_SLACK_LOG2:.0f})")
        self.slack = slack


class _conv_mode_as:
    """Temporarily select a conv kernel family (the plan cache is keyed by it).  Not thread-safe: os.environ."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.old = os.environ.get("MVAL_CONV")
        os.environ["MVAL_CONV"] = self.mode

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop("MVAL_CONV", None)
        else:
            os.environ["MVAL_CONV"] = self.old


def _plan_for(model, x):
    n, c, h, w = x.shape
    if c != 3:
        raise ValueError("expected (N, 3, H, W) images")
    cache = model.__dict__.setdefault("_plans", {})
    key = (n, h, w, x.device.index, os.environ.get("MVAL_FORCE_DIRECT") == "1", _conv_mode(), os.environ.get("MVAL_FUSE_BLOCKS", "1"),
           os.environ.get("MVAL_P2", "1"), os.environ.get("MVAL_P2_W48", "1"), os.environ.get("MVAL_P2_BNECK", "1"), os.environ.get("MVAL_P2_STEM", "1"), os.environ.get("MVAL_P2_FUSE_UP", "1"),
           os.environ.get("MVAL_EPILOGUE_DECODE", "1"))
    plan = cache.get(key)
    if plan is None:
        if len(cache) >= 4:  # keep the arena footprint bounded
            cache.pop(next(iter(cache)))
        plan = cache[key] = InferencePlan(model, n, h, w, x.device)
    return plan


def _max_images_per_launch(model, h, w):
    """The conv kernels index activations with 32-bit element offsets: the largest batch whose biggest activation
    (input of any op, or its possibly upsampled output) stays below 2^31 elements."""
    g = model._graph
    dims = {g.input: (h, w)}
    biggest = 3 * h * w
    for op in g.ops:
        hin, win = dims[op.src]
        if op.kind == "deconv":
            hout, wout = (hin - 1) * op.stride - 2 * op.pad + op.k, (win - 1) * op.stride - 2 * op.pad + op.k
        else:
            hout, wout = (hin + 2 * op.pad - op.k) // op.stride + 1, (win + 2 * op.pad - op.k) // op.stride + 1
        dims[op.dst] = (hout << op.up, wout << op.up)
        biggest = max(biggest, hin * win * op.cin, (hout << op.up) * (wout << op.up) * op.cout)
    # P2 plans address their planes with byte offsets below 2^31: 2^29 elements.  Whether a plan ends up on the P2 kernels is
    # InferencePlan._p2_covers' decision (and MVAL_P2=force's): the conservative limit applies whenever the mode allows them
    limit = 2**29 if _conv_mode() == "p2" else 2**31
    return max(1, (limit - 1) // biggest)


def run_network(model, x):
    if not torch.is_tensor(x) or not x.is_cuda:
        raise _lib.MvalError("the heat-map network runs on the HIP device only (no CPU path): pass a .cuda() tensor")
    if x.dtype != torch.float32:
        raise TypeError("expected float32 images")
    _lib._same_device(x, next(iter(model.parameters()), None))
    x = x.contiguous()
    if model.training:
        from .engine_train import run_network_train

        return run_network_train(model, x)
    cap = _max_images_per_launch(model, x.shape[2], x.shape[3])

    def run():
        if x.shape[0] > cap:  # very large batches run as equal slices of one plan size (plus a remainder plan)
            return torch.cat([_plan_for(model, xs).forward(xs) for xs in x.split(cap)])
        return _plan_for(model, x).forward(x)

    # P2 plans whose a-priori bounds sit too far above the activations of the CURRENT parameters (checked on the first
    # forward after every parameter change) hand over to the h2 plan until the parameters change again
    fb = model.__dict__.get("_p2_fallback_sig")
    if fb is not None and _conv_mode() == "p2":
        if _param_signature(model) == fb:
            with _conv_mode_as("h2"):
                return run()
        model.__dict__["_p2_fallback_sig"] = None
    try:
        return run()
    except P2SlackError as e:
        import warnings

        warnings.warn(f"{e}; using the h2 kernels (fp32 activations, exact per-image scales) for these parameters. "
                      "MVAL_P2=force keeps the P2 plan, MVAL_P2_SLACK_CHECK=0 skips the check.", RuntimeWarning, stacklevel=3)
        model.__dict__["_p2_fallback_sig"] = _param_signature(model)
        with _conv_mode_as("h2"):
            return run()
