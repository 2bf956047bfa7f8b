"""Training-mode executor: train-mode BatchNorm forward + full backward of the heat-map
network on the HIP kernels, exposed to PyTorch as ONE autograd node.

Reference behaviour reproduced (strategy.py:460-487): batch statistics per GPU (no
SyncBN), biased variance for normalisation, running stats updated with momentum 0.1 and the
unbiased variance, ``num_batches_tracked`` incremented; gradients for every conv weight, BN
gamma/beta and the final-layer bias.  The optimizer (Adam) and LR schedule stay in PyTorch, and
so does DistributedDataParallel's gradient all-reduce (RCCL): parameters enter the autograd
node as inputs, so DDP's hooks fire as usual.

The plan keeps every operator's raw conv output z and activation (no arena reuse: backward needs
them) and a mirror arena for activation gradients; forward and backward are one C call each
(``mval_train_forward`` / ``mval_train_backward``), with weights (re)packed on device only when a
parameter version changes.  PoseResNet trains through the same plan: max-pool backward routes gradients to the window arg-max,
ConvTranspose2d(k4, s2, p1) runs forward as a conv over the zero-dilated input and backward as a plain
stride-2 conv (data gradient) / a k4 stride-2 weight gradient with the activations' roles swapped.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import torch

from . import _lib
from .engine import (ALGO_DIRECT, ALGO_MFMA, ALGO_MFMA_BF3, ALGO_MFMA_H2, PACK_HWIO, PACK_MFMA16, PACK_MFMA16_BF3,
                     PACK_MFMA16_H2, _KIND, _PACK_OF, MvalOp, _align, _conv_mode, _mfma_ok)

TRAIN_AMAX_ROW = 576  # dwords per magnitude row in training: [count, <= 512 partial maxima], one row per tensor

BN_MOMENTUM = 0.1
BN_EPS = 1e-5
_F = C.POINTER(C.c_float)


class MvalTrainOp(C.Structure):
    """include/mval_hip.h: struct mval_train_op."""

    _fields_ = [
        ("op", MvalOp),
        ("z_off", C.c_int64),
        ("gin_off", C.c_int64), ("gout_off", C.c_int64), ("gres1_off", C.c_int64), ("gres2_off", C.c_int64),
        ("wd_off", C.c_int64),
        ("has_bn", C.c_int32), ("dgrad_algo", C.c_int32), ("first_touch", C.c_int32), ("dgrad_form", C.c_int32),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
        ("mean", C.c_void_p), ("invstd", C.c_void_p),
        ("dweight", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
        ("out_amax_off", C.c_int64), ("gz_amax_off", C.c_int64),
        ("mask_off", C.c_int64),
        ("fwd_p2", C.c_int32), ("p2_flags", C.c_int32),
        ("in_p2_off", C.c_int64), ("in_p2_rows_off", C.c_int64), ("out_p2_off", C.c_int64), ("out_p2_rows_off", C.c_int64),
        ("res1_amax_off", C.c_int64), ("res2_amax_off", C.c_int64),
        ("gz_p2_off", C.c_int64), ("gz_p2_rows_off", C.c_int64),
        ("res1_p2_off", C.c_int64), ("res1_p2_rows_off", C.c_int64), ("res2_p2_off", C.c_int64), ("res2_p2_rows_off", C.c_int64),
        ("zin_rel", C.c_int32), ("z_out", C.c_int32),
    ]


# Bound slack of the P2 training plan (ADVICE round 4 / the inference plan's engine.P2_MAX_SLACK_LOG2): the scale of a P2 tensor comes from an
# a-priori bound -- Samuelson's |bn(z)| <= |gamma| sqrt(M - 1) + |beta| for activations, the same inequality around dgamma / dbeta for dz --
# and values more than ~2^16 below it lose low-part bits.  The plan measures bound / actual maximum of every P2 tensor on its first step and
# every SLACK_EVERY (1 024) steps after (one extra read of every P2 tensor: ~70 ms at the C3 size, < 0.1 % amortised); past the limit the model's later steps run the h2 training kernels
# (exact per-tensor maxima) and a warning says so.  MVAL_TRAIN_SLACK_CHECK=0 switches the probe off, MVAL_TRAIN_P2=force ignores its verdict.
TRAIN_P2_MAX_SLACK_LOG2 = 15.0
# ... or when more than this fraction of a tensor's non-zero values sits below 2^-3 scaled (fewer than 22 significand bits kept): the
# maximum alone does not see a tensor whose gammas spread widely -- ONE large channel sets the scale, the small ones lose the bits.  A
# Gaussian tensor crosses 0.5 at a slack of ~2^14; the BASELINE plan measures < 0.02 (tests/test_gpu_train.py).
TRAIN_P2_MAX_SMALL_FRAC = 0.5
SLACK_EVERY = int(os.environ.get("MVAL_TRAIN_SLACK_EVERY", "1024"))
# switch -> the value an unset variable stands for (the key must tell "unset" from every other setting: MVAL_TRAIN_LANES defaults to mode 3)
_SWITCHES = {"MVAL_TRAIN_P2": "1", "MVAL_TRAIN_P2_WGRAD": "1", "MVAL_TRAIN_P2_DGRAD": "1", "MVAL_TRAIN_P2_RES": "1", "MVAL_TRAIN_EPI_STATS": "1",
             "MVAL_TRAIN_BWD_FUSED": "1", "MVAL_TRAIN_RELU_MASK": "1", "MVAL_TRAIN_DGRAD_PARITY": "1", "MVAL_TRAIN_LANES": "3",
             "MVAL_TRAIN_BN_IN_CONV": "1", "MVAL_TRAIN_BN_BWD_IN_DGRAD": "1", "MVAL_WGRAD_SLAB_ROT": "1",
             "MVAL_TRAIN_WGRAD_BATCH": "0"}
MAX_LANES = 4            # (csrc/conv_common.h MVAL_MAX_LANES)
TRAIN_LANE_FWD, TRAIN_LANE_BWD, TRAIN_LANE_ORD, TRAIN_LANE_FREE = 256, 512, 1024, 2048  # (include/mval_hip.h MVAL_TRAIN_LANE_*)
TRAIN_BSUM = 4096  # (MVAL_TRAIN_BSUM)
TRAIN_WGRAD_DEFER = 8192  # (MVAL_TRAIN_WGRAD_DEFER)

_ARMED = None  # weakref to the plan whose probe rows the library currently points at (one slot per process: csrc/net_train.hip g_probe)


def _switches():
    """The A/B switches a training plan is built under (part of the plan-cache key: a plan never changes its paths after it is built)."""
    return tuple(os.environ.get(k, d) for k, d in _SWITCHES.items())


def lane_flags(g, nl, mode):
    """MVAL_TRAIN_LANE_* bits of every op of graph ``g`` on ``nl`` lanes (pure host logic: TrainPlan._assign_lanes applies them).
    A phase is SHARED if some activation's gradient is written -- data gradient into op.src, residual gradients into op.res1 / op.res2 -- by
    ops of that phase on more than one lane.  FWD: every op on a lane > 0.  BWD: the same ops, in mode "1" only outside shared phases.
    ORD (modes 2, 3): EVERY op of a shared phase, lane 0 included -- its slot writes wait for the slot's previous writer.  FREE (mode 3):
    every op -- the backward joins once per call and an op also waits for the last writer of the slot it reads."""
    writers = {}  # (phase, activation) -> lanes of the ops whose backward writes that activation's gradient
    for op in g.ops:
        for a in (None if op.src == g.input else op.src, op.res1, op.res2):
            if a is not None:
                writers.setdefault((op.phase, a), set()).add(op.lane if op.lane < nl else 0)
    shared = {ph for (ph, _a), lanes in writers.items() if len(lanes) > 1}
    out = []
    for op in g.ops:
        f = 0
        if op.phase in shared and mode != "1":
            f |= TRAIN_LANE_ORD
        if 0 < op.lane < nl:
            f |= TRAIN_LANE_FWD
            if not (op.phase in shared and mode == "1"):
                f |= TRAIN_LANE_BWD
        if mode == "3":
            f |= TRAIN_LANE_FREE
        out.append(f)
    return out


class TrainPlan:
    def __init__(self, model, n, h, w, device, p2=True):
        g = model._graph
        self.model, self.graph, self.n, self.device = model, g, n, device
        self.steps, self.forwards, self.p2_slack, self._probe = 0, 0, None, None
        lib = _lib.lib()
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
        # ---- arenas: every activation and every raw conv output is kept --------------------------
        top = 0
        act_off = {}
        for a in g.acts:
            if a.id in (g.input,) or a.id not in dims:
                continue
            # the graph output also gets a slot: its NHWC gradient lives at the same offset of garena
            act_off[a.id] = top
            top += _align(n * dims[a.id][0] * dims[a.id][1] * a.channels)
        self._act_top = top  # (the gradient arena mirrors the activation slots only)
        z_off = []
        for i, op in enumerate(g.ops):
            if op.bn:
                z_off.append(top)
                top += _align(n * geo[i][2] * geo[i][3] * op.cout)
            else:
                z_off.append(-1)
        # magnitude rows of the fp16-split convs (MVAL_CONV=h2, the default): one per activation a split conv reads,
        # one for the dz scratch
        amax_row = {}
        self.gz_amax_off = top
        top += TRAIN_AMAX_ROW
        self._row_top = top  # rows are handed out below, the arena is sized after the op loop
        # ---- parameters: forward packing, dgrad packing, ones / zeros ---------------------------
        maxc = max(op.cout for op in g.ops)
        self.maxc = maxc
        ptop = 0
        self.ops = (MvalTrainOp * len(g.ops))()
        self.jobs = []
        stat_top = 0
        self.stat_off = []
        gz_max, wsf_max = 0, 0
        # a tensor's magnitude row is written by its producer's BatchNorm apply (or max-pool) only: a conv WITHOUT BatchNorm
        # leaves none, so its consumers may not use the fp16 split (latent today, only final_layer lacks BN)
        keeps_row = {op.dst for op in g.ops if op.bn or op.kind == "maxpool"}
        for i, op in enumerate(g.ops):
            hin, win, hout, wout = geo[i]
            in_nchw = g.acts[op.src].layout == "nchw"
            out_nchw = g.acts[op.dst].layout == "nchw"
            t = self.ops[i]
            m = t.op
            m.kind = _KIND[op.kind]
            m.k, m.stride, m.pad, m.cin, m.cout = op.k, op.stride, op.pad, op.cin, op.cout
            m.hin, m.win, m.hout, m.wout = hin, win, hout, wout
            m.up, m.relu, m.in_nchw, m.out_nchw = op.up, int(op.relu), int(in_nchw), int(out_nchw)
            m.phase, m.lane = op.phase, op.lane
            m.algo = ALGO_DIRECT
            bf3 = _conv_mode() in ("bf3", "h2", "p2")  # (p2 is an inference activation format: training runs its h2 arithmetic)
            h2 = _conv_mode() in ("h2", "p2")  # fp16x2 split (3 products) where it applies, else bf16x3 (6)
            if _mfma_ok(op, in_nchw) and lib.mval_op_mfma_supported(C.byref(m), C.c_int(n)):
                m.algo = ALGO_MFMA
                if bf3 and op.kind == "conv" and op.k in (1, 3) and (op.cin % 32 == 0 or op.cin == 48):
                    for cand in ((ALGO_MFMA_H2, ALGO_MFMA_BF3) if (h2 and op.src != g.input and op.src in keeps_row) else (ALGO_MFMA_BF3,)):
                        if lib.mval_op_algo_supported(C.byref(m), C.c_int(n), C.c_int(cand)):
                            m.algo = cand
                            break
            if m.algo == ALGO_MFMA_H2:
                if op.src not in amax_row:
                    amax_row[op.src] = self._row_top
                    self._row_top += TRAIN_AMAX_ROW
                m.in_amax_off = amax_row[op.src]
            m.in_off = -1 if op.src == g.input else act_off[op.src]
            m.out_off = -1 if op.dst == g.output else act_off[op.dst]
            m.res1_off = -1 if op.res1 is None else act_off[op.res1]
            m.res2_off = -1 if op.res2 is None else act_off[op.res2]
            m.w_off = m.scale_off = m.shift_off = -1
            t.z_off = z_off[i]
            t.has_bn = int(bool(op.bn))
            t.gout_off = act_off[op.dst]
            t.gin_off = -1 if op.src == g.input else act_off[op.src]
            t.gres1_off = -1 if op.res1 is None else act_off[op.res1]
            t.gres2_off = -1 if op.res2 is None else act_off[op.res2]
            t.wd_off = -1
            t.dgrad_algo = ALGO_DIRECT
            if h2 and op.bn:  # dz's magnitude row: the fp16-split data / weight gradients read it
                t.gz_amax_off = self.gz_amax_off
            if op.kind == "maxpool":  # no parameters; backward routes the gradient to the window arg-max
                self.jobs.append((i, None, None))
                self.stat_off.append(stat_top)
                continue
            if op.kind == "deconv" and (m.algo != ALGO_MFMA or not op.bn):
                raise NotImplementedError(f"training a transposed conv needs the MFMA form (k4 s2 p1, cin % 16 == 0) + BN: {op.conv}")
            fpack = _PACK_OF[m.algo]
            nw = int(lib.mval_packed_weight_floats(C.c_int(fpack), C.c_int(op.cout), C.c_int(op.cin), C.c_int(op.k)))
            m.w_off = ptop
            ptop += _align(nw)
            if not op.bn:  # conv + bias
                m.shift_off = ptop
                ptop += _align(op.cout)
            dpack = None
            if op.kind == "deconv":
                # data gradient of ConvTranspose2d = plain stride-2 conv (k4, p1) of dz with the stored
                # (cin, cout, k, k) weights read as Conv2d weights (cout' = cin, cin' = cout)
                if t.gin_off >= 0:
                    t.dgrad_algo = ALGO_MFMA
                    dpack = _PACK_OF[ALGO_MFMA]
                    nd = int(lib.mval_packed_weight_floats(C.c_int(dpack), C.c_int(op.cin), C.c_int(op.cout), C.c_int(op.k)))
                    t.wd_off = ptop
                    ptop += _align(nd)
                self.jobs.append((i, fpack, dpack))
                self.stat_off.append(stat_top)
                stat_top += 2 * _align(op.cout)
                gz_max = max(gz_max, n * hout * wout * op.cout)
                wsf_max = max(wsf_max, int(lib.mval_conv_wgrad_workspace_floats(C.c_int(op.cout), C.c_int(op.cin), C.c_int(op.k))))
                continue
            if t.gin_off >= 0:
                ok = op.cout % 16 == 0 and op.k in (1, 3) and op.stride in (1, 2) and op.pad == op.k // 2
                t.dgrad_algo = ALGO_MFMA if ok else ALGO_DIRECT
                # stride-1 data gradients are plain convs with cin' = cout: bf16x3-split kernel
                if ok and bf3 and (op.cout % 32 == 0 or op.cout == 48) and ((op.k == 3 and op.cin % 16 == 0) or (op.stride == 1 and op.cin % 16 == 0)):
                    t.dgrad_algo = ALGO_MFMA_BF3
                    # stride-1 data gradients of BatchNorm'd convs: fp16x2 (dz's magnitude comes from the BN backward)
                    if h2 and op.bn and op.stride == 1 and op.cin % 16 == 0:
                        d = MvalOp()
                        d.kind, d.k, d.stride, d.pad, d.cin, d.cout = 0, op.k, 1, op.k - 1 - op.pad, op.cout, op.cin
                        d.hin, d.win, d.hout, d.wout = hout, wout, hin, win
                        if lib.mval_op_algo_supported(C.byref(d), C.c_int(n), C.c_int(ALGO_MFMA_H2)):
                            t.dgrad_algo = ALGO_MFMA_H2
                            t.gz_amax_off = self.gz_amax_off
                # stride-2 3x3 on even sizes: four 2x2 parity convs over dz instead of a 3x3 conv over the zero-dilated dz
                # (2.25x fewer tap-pixels, and the fp16 split applies); weights packed with mode 4 as k = 4
                if bf3 and op.k == 3 and op.stride == 2 and op.pad == 1 and os.environ.get("MVAL_TRAIN_DGRAD_PARITY", "1") != "0":
                    for cand in ((ALGO_MFMA_H2, ALGO_MFMA_BF3) if (h2 and op.bn) else (ALGO_MFMA_BF3,)):
                        if lib.mval_conv_dgrad_parity_supported(C.c_int(n), C.c_int(hin), C.c_int(win), C.c_int(op.cin), C.c_int(hout),
                                                                C.c_int(wout), C.c_int(op.cout), C.c_int(cand)):
                            t.dgrad_algo, t.dgrad_form = cand, 1
                            if cand == ALGO_MFMA_H2:
                                t.gz_amax_off = self.gz_amax_off
                            break
                dpack = _PACK_OF[t.dgrad_algo]
                # the data-gradient conv has cin' = cout, cout' = cin
                nd = int(lib.mval_packed_weight_floats(C.c_int(dpack), C.c_int(op.cin), C.c_int(op.cout), C.c_int(4 if t.dgrad_form == 1 else op.k)))
                t.wd_off = ptop
                ptop += _align(nd)
            self.jobs.append((i, fpack, dpack))
            self.stat_off.append(stat_top)
            stat_top += 2 * _align(op.cout)
            gz_max = max(gz_max, n * hout * wout * op.cout)
            wsf_max = max(wsf_max, int(lib.mval_conv_wgrad_workspace_floats(C.c_int(op.cin), C.c_int(op.cout), C.c_int(op.k))))
        # Round 4: forward convs on the P2 kernels (conv_p2.hip EPI 3) where the op's fp16-split conv qualifies: the producer's
        # BatchNorm apply also writes the activation as P2 planes (mval_bn_apply_fwd_p2); the residuals of such producers need
        # magnitude rows (the P2 scale is an a-priori bound + the residuals' exact maxima).  MVAL_TRAIN_P2=0: round 3's h2 forward.
        P2_ROW = 512
        producer = {op.dst: k for k, op in enumerate(g.ops)}
        p2_act = {}
        self.n_bn_in_conv = 0  # (round 6: ops whose BatchNorm apply runs inside their reader's staging; set below for P2 plans)
        self.p2_rows = []
        if h2 and p2 and os.environ.get("MVAL_TRAIN_P2", "1") != "0":
            for i, op in enumerate(g.ops):
                t = self.ops[i]
                k = producer.get(op.src)
                if not (op.kind == "conv" and op.bn and t.op.algo == ALGO_MFMA_H2 and k is not None and g.ops[k].bn and g.ops[k].cout % 8 == 0
                        and op.cout % 4 == 0):
                    continue
                hin, win, hout, wout = geo[i]
                d = MvalOp()
                d.kind, d.k, d.stride, d.pad, d.cin, d.cout = 0, op.k, op.stride, op.pad, op.cin, op.cout
                d.hin, d.win, d.hout, d.wout = hin, win, hout, wout
                if not lib.mval_op_algo_supported(C.byref(d), C.c_int(n), C.c_int(4)):  # (MVAL_ALGO_MFMA_P2)
                    continue
                if op.src not in p2_act:
                    planes = self._row_top
                    self._row_top += _align(n * hin * win * op.cin)
                    rows = self._row_top
                    self._row_top += _align(n * P2_ROW)
                    p2_act[op.src] = (planes, rows)
                    self.p2_rows.append((rows, n * P2_ROW))
                t.fwd_p2 = 1
                t.in_p2_off, t.in_p2_rows_off = p2_act[op.src]
            for a_, (planes, rows) in p2_act.items():
                k = producer[a_]
                pt, po = self.ops[k], g.ops[k]
                pt.out_p2_off, pt.out_p2_rows_off = planes, rows
                for r_, name in ((po.res1, "res1_amax_off"), (po.res2, "res2_amax_off")):
                    if r_ is not None:
                        if r_ not in amax_row:
                            amax_row[r_] = self._row_top
                            self._row_top += TRAIN_AMAX_ROW
                        setattr(pt, name, amax_row[r_])
            # weight gradients read x from the planes where the split kernel covers the conv; an activation whose EVERY consumer reads the
            # planes (P2 forward conv + P2 weight gradient, no residual use, not the network output) is not written as fp32 at all
            p2w = os.environ.get("MVAL_TRAIN_P2_WGRAD", "1") != "0"
            fused_bwd = os.environ.get("MVAL_TRAIN_BWD_FUSED", "1") != "0"  # (round 3's BatchNorm backward reads `out` and writes fp32 dz only)
            consumers = {}
            for i, op in enumerate(g.ops):
                consumers.setdefault(op.src, []).append(i)
                t = self.ops[i]
                if p2w and t.fwd_p2 and t.gz_amax_off > 0 and lib.mval_conv_wgrad_p2_covers(C.c_int(op.cin), C.c_int(op.cout), C.c_int(op.k), C.c_int(op.stride)):
                    t.p2_flags |= 1
            # a residual can be read from the planes as well (mval_bn_apply_fwd_p2_res) when the op that adds it runs the P2 apply: then
            # a block output whose every reader takes the planes -- the next block's first conv, its weight gradient, the residual add at
            # that block's end -- is P2-only too (MVAL_TRAIN_P2_RES=0: residuals stay fp32 NHWC)
            p2res = os.environ.get("MVAL_TRAIN_P2_RES", "1") != "0"
            res_users = {}
            for i, op in enumerate(g.ops):
                for r_, bit in ((op.res1, 16), (op.res2, 32)):
                    if r_ is not None:
                        res_users.setdefault(r_, []).append((i, bit))
            for a_ in p2_act:
                cons = consumers.get(a_, [])
                po = g.ops[producer[a_]]
                # (its own backward must not need `out` either: a ReLU behind residual adds takes its mask from the mask bytes)
                own_ok = not (po.relu and (po.res1 is not None or po.res2 is not None)) or (os.environ.get("MVAL_TRAIN_RELU_MASK", "1") != "0" and po.up == 0)
                users = res_users.get(a_, [])
                users_ok = all(self.ops[i].out_p2_off > 0 and g.ops[i].bn for i, _ in users) and (p2res or not users)
                if p2w and fused_bwd and own_ok and users_ok and a_ != g.output and cons and all(self.ops[i].fwd_p2 and (self.ops[i].p2_flags & 1) for i in cons):
                    self.ops[producer[a_]].p2_flags |= 2
                    for i, bit in users:
                        t = self.ops[i]
                        t.p2_flags |= bit
                        if bit == 16:
                            t.res1_p2_off, t.res1_p2_rows_off = p2_act[a_]
                        else:
                            t.res2_p2_off, t.res2_p2_rows_off = p2_act[a_]
            # data gradients of stride-1 convs on the P2 kernels: the BatchNorm backward also writes dz as P2 planes into ONE scratch
            # (planes of the largest dz, rows, reduction scratch); MVAL_TRAIN_P2_DGRAD=0: the h2 data gradients.  (Round 5 also ran the
            # four parity convs of the stride-2 data gradients on conv_p2_kernel<2, ...> from the dz planes: 65.47 vs 65.2 ms per C3 step --
            # four persistent launches against the h2 kernel's one: not kept.)
            if os.environ.get("MVAL_TRAIN_P2_DGRAD", "1") != "0" and fused_bwd:
                want = []
                for i, op in enumerate(g.ops):
                    t = self.ops[i]
                    hin, win, hout, wout = geo[i]
                    if not (t.dgrad_algo == ALGO_MFMA_H2 and t.dgrad_form == 0 and op.bn and op.up == 0 and op.stride == 1 and op.cout % 8 == 0 and op.cin % 4 == 0
                            and t.gin_off >= 0):
                        continue
                    d = MvalOp()
                    d.kind, d.k, d.stride, d.pad, d.cin, d.cout = 0, op.k, 1, op.k // 2, op.cout, op.cin
                    d.hin, d.win, d.hout, d.wout = hout, wout, hin, win
                    if op.pad == op.k // 2 and lib.mval_op_algo_supported(C.byref(d), C.c_int(n), C.c_int(4)):
                        want.append(i)
                if want:
                    planes = self._row_top
                    self._row_top += _align(max(n * geo[i][2] * geo[i][3] * g.ops[i].cout for i in want))
                    rows = self._row_top
                    self._row_top += _align(n * P2_ROW + 512 + 64)
                    self.p2_rows.append((rows, n * P2_ROW + 512 + 64))
                    for i in want:
                        self.ops[i].p2_flags |= 4
                        self.ops[i].gz_p2_off, self.ops[i].gz_p2_rows_off = planes, rows
                        op = g.ops[i]
                        # the weight gradient reads dz from the planes too where the split kernel covers the conv: no fp32 copy of dz
                        if (os.environ.get("MVAL_TRAIN_P2_WGRAD", "1") != "0" and op.cout % 8 == 0
                                and lib.mval_conv_wgrad_split_covers(C.c_int(op.cin), C.c_int(op.cout), C.c_int(op.k), C.c_int(op.stride))
                                and ((self.ops[i].p2_flags & 1) or self.ops[i].op.in_amax_off > 0)):
                            self.ops[i].p2_flags |= 8
            # Round 6: BatchNorm apply inside the consumer (include/mval_hip.h mval_train_op.z_out / zin_rel; hrnet.py:36-52: conv1 -> bn1 -> relu ->
            # conv2 of every BasicBlock, the 3x3 of a Bottleneck).  A planes-only ReLU output without residual or upsample whose ONE reader is a
            # 3x3 stride-1 P2 conv on the same lane, with the P2 weight gradient reading dz from planes, is never written: that conv's staging
            # and its weight gradient's staging apply the BatchNorm from the producer's raw z.  MVAL_TRAIN_BN_IN_CONV=0: the separate apply pass.
            self.n_bn_in_conv = 0
            if os.environ.get("MVAL_TRAIN_BN_IN_CONV", "1") != "0":
                lib.mval_conv_p2_inz_supported.restype = C.c_int
                for a_ in p2_act:
                    k = producer[a_]
                    po, pt = g.ops[k], self.ops[k]
                    cons = consumers.get(a_, [])
                    if not ((pt.p2_flags & 2) and po.kind == "conv" and po.bn and po.relu and po.res1 is None and po.res2 is None and po.up == 0
                            and len(cons) == 1 and not res_users.get(a_) and a_ != g.output):
                        continue
                    j = cons[0]
                    co, ct = g.ops[j], self.ops[j]
                    if not (co.kind == "conv" and co.k == 3 and co.stride == 1 and co.pad == 1 and ct.fwd_p2 and (ct.p2_flags & 1) and (ct.p2_flags & 8)
                            and (co.phase, co.lane) == (po.phase, po.lane) and j > k):
                        continue
                    hin, win, _, _ = geo[j]
                    if not lib.mval_conv_p2_inz_supported(C.c_int(co.cin), C.c_int(co.cout), C.c_int(hin), C.c_int(win), C.c_int(n)):
                        continue
                    pt.z_out = 1
                    ct.zin_rel = k - j
                    self.n_bn_in_conv += 1

        # the two BatchNorm A/B switches are the PLAN's decision and travel in p2_flags (bit 6: round 3's backward pair, bit 7: statistics by
        # the separate pass): net_train.hip does not read the environment
        bits = (0 if os.environ.get("MVAL_TRAIN_BWD_FUSED", "1") != "0" else 64) | (0 if os.environ.get("MVAL_TRAIN_EPI_STATS", "1") != "0" else 128)
        for i, op in enumerate(g.ops):  # producers leave max |out| where a split conv will look for it
            self.ops[i].out_amax_off = amax_row.get(op.dst, 0)
            self.ops[i].p2_flags |= bits
        self.uses_p2 = any(t.out_p2_off > 0 or (t.p2_flags & 4) for t in self.ops)
        # ReLU behind residual adds (BasicBlock / Bottleneck outputs, fuse sums at the conv resolution): the forward apply keeps
        # (out > 0) as one byte per float4, the backward reads that instead of `out` (a sixteenth of the bytes)
        if os.environ.get("MVAL_TRAIN_RELU_MASK", "1") != "0":
            for i, op in enumerate(g.ops):
                if op.bn and op.relu and op.up == 0 and (op.res1 is not None or op.res2 is not None) and op.cout % 4 == 0:
                    self.ops[i].mask_off = self._row_top
                    self._row_top += _align((n * geo[i][2] * geo[i][3] * op.cout // 4 + 3) // 4)
        # Round 6, the other direction (MVAL_TRAIN_BSUM; MVAL_TRAIN_BN_BWD_IN_DGRAD=0: the reduction pass): the data gradient of a 3x3 stride-1
        # P2 conv whose input's producer is the op right in front of it in the list is the LAST writer of that producer's output gradient
        # (every other reader sits behind it in the list and writes earlier in the backward), so its epilogue keeps the sums the producer's
        # BatchNorm backward would take from a second read of it (P2Args::bs_z): ReLU producers without residual (mask from z: conv1 of a
        # BasicBlock) and with residuals (mask from the kept bits: conv2 of a block followed by another block; the apply pass then also
        # scatters the residual gradients).  The library checks the rest at launch (same lane, room for the partials).
        self.n_bn_bwd_in_dgrad = 0
        if self.uses_p2 and os.environ.get("MVAL_TRAIN_BN_BWD_IN_DGRAD", "1") != "0" and os.environ.get("MVAL_TRAIN_BWD_FUSED", "1") != "0":
            lib.mval_conv_p2_bsum_supported.restype = C.c_int
            for j in range(1, len(g.ops)):
                co, ct, po, pt = g.ops[j], self.ops[j], g.ops[j - 1], self.ops[j - 1]
                if not (co.kind == "conv" and co.k == 3 and co.stride == 1 and co.pad == 1 and co.src == po.dst and (ct.p2_flags & 4) and ct.gin_off >= 0
                        and po.kind == "conv" and po.bn and po.relu and po.up == 0 and (pt.p2_flags & 4) and pt.gin_off >= 0 and pt.gz_p2_off > 0
                        and (co.phase, co.lane) == (po.phase, po.lane)):
                    continue
                if (po.res1 is not None or po.res2 is not None) and not pt.mask_off > 0:
                    continue
                hin, win, _, _ = geo[j]
                if lib.mval_conv_p2_bsum_supported(C.c_int(co.cout), C.c_int(co.cin), C.c_int(hin), C.c_int(win), C.c_int(n)):
                    ct.p2_flags |= TRAIN_BSUM
                    self.n_bn_bwd_in_dgrad += 1
        if g.input in amax_row:
            raise _lib.MvalError("the network input cannot feed an fp16-split conv")
        self._assign_lanes(g, n, geo)
        self.arena_floats = _align(self._row_top)
        # first writer of every activation-gradient slot (backward order) stores, later ones accumulate;
        # slots nobody writes (activations without a consumer) are the only ones zero-filled
        touched = {g.output}
        for i in range(len(g.ops) - 1, -1, -1):
            op, t = g.ops[i], self.ops[i]
            mask = 0
            for bit, act in ((2, op.res1), (4, op.res2), (1, None if op.src == g.input else op.src)):
                if act is not None and act not in touched:
                    touched.add(act)
                    mask |= bit
            t.first_touch = mask
        self.zero_slots = [(act_off[a], n * dims[a][0] * dims[a][1] * g.acts[a].channels)
                           for a in act_off if a not in touched]
        self.ones_off = ptop
        ptop += _align(maxc)
        self.zeros_off = ptop
        ptop += _align(maxc)
        for t in self.ops:
            if not t.has_bn:
                t.op.scale_off = self.ones_off
        self.param_floats = ptop
        f32 = dict(dtype=torch.float32, device=device)
        self.arena = torch.empty(self.arena_floats, **f32)
        for off, cnt in self.p2_rows:  # P2 rows: unused partial slots must read as zero (csrc/conv_p2.h); the scale slot is rewritten every forward
            self.arena[off : off + cnt].zero_()
        self.garena = torch.empty(max(self._act_top, 64), **f32)
        self.params = torch.zeros(self.param_floats, **f32)
        self.params[self.ones_off : self.ones_off + maxc] = 1.0
        self.stats = torch.zeros(max(stat_top, 64), **f32)
        # (lanes: one slice of every scratch buffer per lane, mval_train_*_lanes)
        self.gz_lane, self.wsf_lane = _align(max(gz_max, 64)), _align(max(wsf_max, 64))
        # (measurement only, profiles/r06 item 2b: MVAL_WGRAD_SLAB_ROT=K gives every lane K slab regions and the library walks them op by op, so an op's
        # slab reduction reads from HBM instead of the Infinity Cache -- what a reduction deferred to the end of a backward segment would do)
        self.slab_rot = max(1, int(os.environ.get("MVAL_WGRAD_SLAB_ROT", "1")))
        self.wsf_region = self.wsf_lane
        self.wsf_lane *= self.slab_rot
        self.gz = torch.empty(self.gz_lane * self.n_lanes, **f32)
        self.wsf = torch.empty(self.wsf_lane * self.n_lanes, **f32)
        # float64 scratch of the BatchNorm reductions; sized so that the forward conv epilogues' per-workgroup statistics
        # partials fit (cout x workgroups x 2, workgroups <= pixels / 32 for every tile the large layers use)
        epi = max((op.cout * ((n * geo[i][2] * geo[i][3] + 31) // 32 + 64) * 2 for i, op in enumerate(g.ops) if op.bn), default=0)
        # (round 6: ... and the data gradients' reduction partials, cout x slots x 3 with slots = persistent workgroups x pixel waves <= 2 304)
        bsum = max((g.ops[i - 1].cout * 2304 * 3 for i, t in enumerate(self.ops) if t.p2_flags & TRAIN_BSUM), default=0)
        self.ws_lane = _align(max(512 * maxc * 2, epi, bsum) + 64)
        self.ws = torch.empty(self.ws_lane * self.n_lanes, dtype=torch.float64, device=device)
        self.sums_lane = _align(2 * maxc + 64)
        self.sums = torch.empty(self.sums_lane * self.n_lanes, **f32)
        self.param_sig = None
        self._pack_ptrs, self._pack_jobs = None, None
        # parameter order of the autograd node: conv.weight [, conv.bias] [, bn.weight, bn.bias] per op
        holders = model._holders
        self.param_list, self.grad_slots = [], []
        gtop = 0
        cum = []  # gradient floats up to and including op i
        for op in g.ops:
            cum.append(gtop)
            if op.kind == "maxpool":
                self.grad_slots.append({})
                continue
            conv = holders[op.conv]
            slots = {"w": gtop}
            self.param_list.append(conv.weight)
            gtop += _align(conv.weight.numel(), 4)
            if getattr(conv, "bias", None) is not None:
                slots["b"] = gtop
                self.param_list.append(conv.bias)
                gtop += _align(op.cout, 4)
            if op.bn:
                bn = holders[op.bn]
                slots["g"], slots["be"] = gtop, gtop + _align(op.cout, 4)
                self.param_list += [bn.weight, bn.bias]
                gtop += 2 * _align(op.cout, 4)
            self.grad_slots.append(slots)
            cum[-1] = gtop
        self.grad_floats = gtop
        # Segments of the op list for the autograd chain (run_network_train): ~equal shares of the gradient bytes, so
        # that DistributedDataParallel sees the gradients of the LAST layers while the backward of the earlier ones is
        # still running and can overlap its bucketed all-reduce (RCCL) with it.  One segment = one autograd node.
        nseg = max(1, min(6, gtop // (4 << 20)))  # (no point in splitting a few MB of gradients)
        bounds = [0]
        for k in range(1, nseg):
            i = next(i for i, c_ in enumerate(cum) if c_ >= k * gtop / nseg) + 1
            if bounds[-1] < i < len(g.ops):
                bounds.append(i)
        bounds.append(len(g.ops))
        self.segments = [(a, b) for a, b in zip(bounds, bounds[1:]) if b > a]
        # Round 6 (MVAL_TRAIN_WGRAD_DEFER; opt-in: MVAL_TRAIN_WGRAD_BATCH=1): the weight gradients' slab reductions of a backward segment as ONE
        # launch per 64 ops at its end, every op's slabs in a region of its own (the workspace becomes one arena that holds the largest segment's
        # regions).  Bit-identical and MEASURED SLOWER -- C3 57.7-58.3 -> 60.8 ms (profiles/r06/wgrad_reduce_batched_ab.log): the per-op reduction
        # reads its 28-56 MB of slabs back from the Infinity Cache right after they were written, into a buffer the next op overwrites there; deferred,
        # ~5 GB of slabs per step go out to HBM and come back (the 293 launches it saves are worth at most 1.5 ms: wgrad_reduce_bounds...log).
        self.wgrad_batch = os.environ.get("MVAL_TRAIN_WGRAD_BATCH", "0") == "1" and self.slab_rot == 1
        if self.wgrad_batch:
            lib.mval_conv_wgrad_workspace_floats.restype = C.c_size_t
            need = [0 if op.kind == "maxpool" else (int(lib.mval_conv_wgrad_workspace_floats(C.c_int(op.cin), C.c_int(op.cout), C.c_int(op.k))) + 63) // 64 * 64
                    for op in g.ops]
            total = max(sum(need[a:b]) for a, b in self.segments)
            if total > self.wsf_lane * self.n_lanes:
                self.wsf_lane = _align((total + self.n_lanes - 1) // self.n_lanes)
                self.wsf = torch.empty(self.wsf_lane * self.n_lanes, dtype=torch.float32, device=device)
            for t in self.ops:
                t.p2_flags |= TRAIN_WGRAD_DEFER
        self.seg_params = []
        for lo, hi in self.segments:
            ps = []
            for op in g.ops[lo:hi]:
                if op.kind == "maxpool":
                    continue
                conv = holders[op.conv]
                ps.append(conv.weight)
                if getattr(conv, "bias", None) is not None:
                    ps.append(conv.bias)
                if op.bn:
                    ps += [holders[op.bn].weight, holders[op.bn].bias]
            self.seg_params.append(ps)
        assert sum(len(p_) for p_ in self.seg_params) == len(self.param_list)
        self.bn_counters = [holders[op.bn].num_batches_tracked for op in g.ops if op.bn]

    # ---- parameters ---------------------------------------------------------------------------
    def _assign_lanes(self, g, n, geo):
        """Lanes of the training passes (csrc/net_train.hip, mval_train_*_lanes): the ops of a phase that sit on different lanes -- HRNet's
        branches, the chains of a fuse layer (graph.py; hrnet.py:199-287) -- are independent, so they run on separate streams.  Forward:
        every op on a lane > 0 (each writes only its own tensors' slots, rows, masks, statistics).  Backward: only in phases where every
        gradient slot has ALL its writers on one lane (the branch bodies; a fuse layer's chains all add into the branches' output
        gradients and stay serial), so each slot sees its first-touch store and its accumulations in the serial order: same bits as
        MVAL_TRAIN_LANES=0.  What the ops of a lane share -- dz's magnitude row, the dz plane scratch, and the buffers the C call slices
        per lane -- exists once per lane."""
        self.n_lanes = 1
        # 0: one stream; 1: lanes in the phases without shared gradient slots only; 2: in all phases; 3 (default): and the backward without
        # joins at the phase changes -- every dependency through the slot events
        mode = os.environ.get("MVAL_TRAIN_LANES", _SWITCHES["MVAL_TRAIN_LANES"])
        if mode == "0" or not g.ops:
            return
        nl = min(MAX_LANES, max(op.lane for op in g.ops) + 1)
        if nl <= 1:
            return
        from .engine import P2_ROW
        flags = lane_flags(g, nl, mode)
        row = {0: self.gz_amax_off}
        scratch = {}
        for i, op in enumerate(g.ops):
            t = self.ops[i]
            t.p2_flags |= flags[i]
            if not (flags[i] & TRAIN_LANE_BWD):
                continue
            if t.gz_amax_off > 0:
                if op.lane not in row:
                    row[op.lane] = self._row_top
                    self._row_top += TRAIN_AMAX_ROW
                t.gz_amax_off = row[op.lane]
            if t.p2_flags & 4:
                scratch.setdefault(op.lane, []).append(i)
        for lane, idx in scratch.items():
            planes = self._row_top
            self._row_top += _align(max(n * geo[i][2] * geo[i][3] * g.ops[i].cout for i in idx))
            rows = self._row_top
            self._row_top += _align(n * P2_ROW + 512 + 64)
            self.p2_rows.append((rows, n * P2_ROW + 512 + 64))
            for i in idx:
                self.ops[i].gz_p2_off, self.ops[i].gz_p2_rows_off = planes, rows
        self.n_lanes = nl

    def _build_pack_table(self, holders, base):
        """Device-side job table of every split-bf16 weight packing of the plan (forward and data-gradient
        forms) for ``mval_pack_bf3_jobs``: rebuilt only when a parameter's storage moves."""
        import numpy as np

        rows, first, blocks = [], [], 0
        for (i, fpack, dpack), op in zip(self.jobs, self.graph.ops):
            if op.kind == "maxpool":
                continue
            t = self.ops[i]
            w = holders[op.conv].weight
            if not (w.is_cuda and w.is_contiguous()):
                self._pack_jobs = None
                return
            fmode, dmode = (2, 0) if op.kind == "deconv" else (0, 2)
            dk = op.k
            if t.dgrad_form == 1:  # four-parity form of a stride-2 data gradient: mode 4, packed as k = 4
                dmode, dk = 4, 4
            todo = []
            if fpack in (PACK_MFMA16_BF3, PACK_MFMA16_H2):  # (bit 8 of the mode: fp16-split packing, mval_pack_split_jobs)
                todo.append((base + 4 * t.op.w_off, fmode | (0x100 if fpack == PACK_MFMA16_H2 else 0), op.cout, op.cin, op.k))
            if dpack in (PACK_MFMA16_BF3, PACK_MFMA16_H2):
                todo.append((base + 4 * t.wd_off, dmode | (0x100 if dpack == PACK_MFMA16_H2 else 0), op.cin, op.cout, dk))
            for dst, mode, cout, cin, kk in todo:
                total = kk * kk * ((cin + 31) // 32) * ((cout + 15) // 16) * 512
                rows.append((w.data_ptr(), dst, mode, cout, cin, kk))
                first.append(blocks)
                blocks += (total + 255) // 256
        if not rows:
            self._pack_jobs = None
            return
        table = np.array(rows, dtype=np.dtype([("w", "<u8"), ("p", "<u8"), ("mode", "<i4"), ("cout", "<i4"), ("cin", "<i4"),
                                               ("k", "<i4")]))
        assert table.dtype.itemsize == 32
        self._pack_jobs = torch.from_numpy(table.view(np.uint8).copy()).to(self.device)
        self._pack_first = torch.tensor(first, dtype=torch.int32, device=self.device)
        self._pack_n, self._pack_blocks = len(rows), blocks

    def _refresh(self):
        holders = self.model._holders
        lib = _lib.lib()
        st = _lib._stream()
        ptrs = tuple(p.data_ptr() for p in self.param_list)
        sig = tuple(p._version for p in self.param_list) + ptrs
        repack = sig != self.param_sig
        base = self.params.data_ptr()
        sbase = self.stats.data_ptr()
        if repack and ptrs != self._pack_ptrs:
            self._build_pack_table(holders, base)
            self._pack_ptrs = ptrs
        if repack and self._pack_jobs is not None:
            # every split-bf16 packing (forward and data-gradient forms) in one launch
            _lib._check(lib.mval_pack_split_jobs(C.c_void_p(self._pack_jobs.data_ptr()), C.c_void_p(self._pack_first.data_ptr()),
                                               C.c_int(self._pack_n), C.c_int(self._pack_blocks), st), "pack (batched)")
        for (i, fpack, dpack), op in zip(self.jobs, self.graph.ops):
            t = self.ops[i]
            if op.kind == "maxpool":
                continue
            conv = holders[op.conv]
            w = conv.weight
            if not w.is_cuda:
                raise _lib.MvalError("model parameters must be on the HIP device (call .cuda())")
            if repack:
                batched = self._pack_jobs is not None
                wp = C.c_void_p(w.detach().contiguous().data_ptr())
                # Conv2d: forward as stored (0), data gradient tap-flipped / channel-swapped (2);
                # ConvTranspose2d: the other way round (forward = conv over the zero-dilated input)
                fmode, dmode = (2, 0) if op.kind == "deconv" else (0, 2)
                dk = op.k
                if t.dgrad_form == 1:
                    dmode, dk = 4, 4
                if not (batched and fpack in (PACK_MFMA16_BF3, PACK_MFMA16_H2)):
                    _lib._check(lib.mval_pack_conv_weights(C.c_int(fpack), C.c_int(fmode), wp, C.c_void_p(base + 4 * t.op.w_off),
                                                           C.c_int(op.cout), C.c_int(op.cin), C.c_int(op.k), st), "pack fwd")
                if dpack is not None and not (batched and dpack in (PACK_MFMA16_BF3, PACK_MFMA16_H2)):
                    _lib._check(lib.mval_pack_conv_weights(C.c_int(dpack), C.c_int(dmode), wp, C.c_void_p(base + 4 * t.wd_off),
                                                           C.c_int(op.cin), C.c_int(op.cout), C.c_int(dk), st), "pack dgrad")
                if getattr(conv, "bias", None) is not None:
                    self.params[t.op.shift_off : t.op.shift_off + op.cout] = conv.bias.detach()
            if op.bn:
                bn = holders[op.bn]
                t.gamma, t.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
                t.running_mean, t.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                t.mean = sbase + 4 * self.stat_off[i]
                t.invstd = sbase + 4 * (self.stat_off[i] + _align(op.cout))
        self.param_sig = sig

    # ---- bound-slack probe ----------------------------------------------------------------------
    def _probe_due(self):
        if not self.uses_p2 or os.environ.get("MVAL_TRAIN_SLACK_CHECK", "1") == "0":
            return False
        # counted in FORWARDS: a plan whose backward never runs (train-mode forwards under no_grad, NaN-skipped steps) probes once, not every time
        return self.forwards == 0 or (SLACK_EVERY > 0 and self.forwards % SLACK_EVERY == 0)

    def _probe_arm(self):
        global _ARMED
        self._probe = torch.zeros(len(self.ops) * 8, dtype=torch.int32, device=self.device)
        _lib._check(_lib.lib().mval_train_p2_probe(C.c_void_p(self._probe.data_ptr()), C.c_int(len(self.ops))), "mval_train_p2_probe")
        _ARMED = weakref.ref(self)

    def _probe_disarm(self):
        """The library holds a raw pointer into ``self._probe`` while armed (csrc/net_train.hip g_probe): it is cleared before the tensor
        can go away -- by the read-out, by this plan's next forward, and when the plan itself is dropped (``__del__``: cache.clear())."""
        global _ARMED
        if self._probe is not None:
            if _ARMED is not None and _ARMED() is self:  # (another plan that armed after this one owns the slot now)
                _lib._check(_lib.lib().mval_train_p2_probe(C.c_void_p(0), C.c_int(0)), "mval_train_p2_probe")
                _ARMED = None
            self._probe = None

    def __del__(self):
        try:
            if self._probe is not None:
                torch.cuda.synchronize(self.device)  # (launches of the armed forward may still be writing the rows)
                self._probe_disarm()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def _probe_read(self):
        """Disarm and turn the measured rows into ``self.p2_slack``: per kind ("act" = output planes of the BatchNorm applies, "dz" = the
        BatchNorm backward's planes) log2 of the largest bound / actual maximum over the plan's tensors, the op it belongs to, and the
        largest fraction of a tensor's non-zero values that sit below 2^-3 scaled (fewer than 22 bits kept)."""
        import numpy as np

        raw = self._probe.cpu().numpy().reshape(len(self.ops), 2, 4)  # (synchronises: the armed step's launches are done)
        self._probe_disarm()
        res = {"step": self.steps}
        for kind, col in (("act", 0), ("dz", 1)):
            r = raw[:, col]
            mx, small, nz = r[:, 1].view(np.float32).astype(np.float64), r[:, 2].view(np.uint32).astype(np.float64), r[:, 3].view(np.uint32).astype(np.float64)
            ok = (nz > 0) & np.isfinite(mx) & (mx > 0)
            if not ok.any():
                res[kind] = None
                continue
            slack = np.where(ok, np.log2(8192.0 / np.where(ok, mx, 1.0)), -np.inf)  # the bound sits in [2^13, 2^14) of the scaled range
            frac = np.where(ok, small / np.maximum(nz, 1.0), 0.0)
            k = int(np.argmax(slack))
            res[kind] = {"tensors": int(ok.sum()), "max_log2": round(float(slack[k]), 2), "max_at": self.graph.ops[k].conv,
                         "median_log2": round(float(np.median(slack[ok])), 2), "max_small_frac": round(float(frac.max()), 5),
                         "max_small_frac_at": self.graph.ops[int(np.argmax(frac))].conv}
        self.p2_slack = res
        worst = max((res[k]["max_log2"] for k in ("act", "dz") if res[k]), default=0.0)
        # (activations only: a dz tensor is mostly tiny values by construction -- the mean terms at masked positions -- whose ABSOLUTE error,
        # <= 2^-25 scaled, is what enters the gradient sums; its fraction is reported, not judged)
        small = res["act"]["max_small_frac"] if res["act"] else 0.0
        if (worst > TRAIN_P2_MAX_SLACK_LOG2 or small > TRAIN_P2_MAX_SMALL_FRAC) and os.environ.get("MVAL_TRAIN_P2", "1") != "force":
            import warnings

            warnings.warn(f"P2 training plan: a-priori bound 2^{worst:.1f} above a tensor's maximum (limit 2^{TRAIN_P2_MAX_SLACK_LOG2:.0f}), {small:.2f} of a "
                          f"tensor's non-zero values below 2^-3 scaled (limit {TRAIN_P2_MAX_SMALL_FRAC}): {res}; "
                          "the next steps of this model run the h2 training kernels (scales from exact maxima).  MVAL_TRAIN_P2=force keeps P2, "
                          "MVAL_TRAIN_SLACK_CHECK=0 skips the probe.", RuntimeWarning, stacklevel=2)
            self.model.__dict__["_train_p2_off"] = True

    def forward(self, x):
        self._refresh()
        if self._probe is not None:  # (an armed step whose backward never ran)
            torch.cuda.current_stream(self.device).synchronize()
            self._probe_disarm()
        if self._probe_due():
            self._probe_arm()
        self.forwards += 1
        # the arena, z buffers and batch statistics of THIS forward are what backward reads: a second train-mode
        # forward of the same plan overwrites them, so backward checks that it pairs with the latest forward
        self.generation = getattr(self, "generation", 0) + 1
        out = torch.empty((self.n, self.out_channels) + tuple(self.out_hw), dtype=torch.float32, device=self.device)
        _lib._check(
            _lib.lib().mval_train_forward_lanes(
                self.ops, C.c_int(len(self.ops)), C.c_int(self.n), C.c_void_p(self.arena.data_ptr()),
                C.c_void_p(self.params.data_ptr()), C.c_int64(self.ones_off), C.c_int64(self.zeros_off),
                C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(self.ws.data_ptr()),
                C.c_int64(self.ws_lane), C.c_int(self.n_lanes), C.c_float(BN_MOMENTUM), C.c_float(BN_EPS), _lib._stream()),
            "mval_train_forward_lanes")
        if self.bn_counters:
            torch._foreach_add_(self.bn_counters, 1)
        return out

    def backward(self, x, gout_nchw):
        """Whole backward in one go (the segmented autograd chain calls the two halves below separately)."""
        self.backward_begin(x, gout_nchw)
        out = []
        for k in range(len(self.segments) - 1, -1, -1):
            out = self.backward_segment(k, x) + out
        return out

    def backward_begin(self, x, gout_nchw):
        """Gradient buffer, the first-touch bookkeeping's zero fills, and the loss gradient into the gradient arena."""
        g = self.graph
        grads = torch.empty(self.grad_floats, dtype=torch.float32, device=self.device)
        self._grads = grads
        gb = grads.data_ptr()
        for t, slots, op in zip(self.ops, self.grad_slots, g.ops):
            if not slots:
                continue
            t.dweight = gb + 4 * slots["w"]
            t.dgamma = gb + 4 * slots["g"] if "g" in slots else None
            t.dbeta = gb + 4 * (slots["be"] if "be" in slots else slots["b"]) if ("be" in slots or "b" in slots) else None
        for off, cnt in self.zero_slots:
            self.garena[off : off + cnt].zero_()
        last = self.ops[len(self.ops) - 1]
        gn = gout_nchw.to(torch.float32).permute(0, 2, 3, 1).contiguous()
        self.garena[last.gout_off : last.gout_off + gn.numel()] = gn.reshape(-1)

    def backward_segment(self, k, x):
        """Backward of the ops [lo, hi) of segment k (segments MUST be run last to first: the gradient arena carries
        the state between them) -> that segment's parameter gradients, in param_list order."""
        g = self.graph
        lo, hi = self.segments[k]
        grads = self._grads
        sub = (MvalTrainOp * (hi - lo)).from_address(C.addressof(self.ops) + lo * C.sizeof(MvalTrainOp))
        if self.slab_rot > 1:
            _lib.lib().mval_train_slab_rotation(C.c_int(self.slab_rot), C.c_int64(self.wsf_region))
        _lib.lib().mval_train_timing_base(C.c_int(lo))  # (measurement mode's per-operator breakdown)
        _lib._check(
            _lib.lib().mval_train_backward_lanes(
                sub, C.c_int(hi - lo), C.c_int(self.n), C.c_void_p(self.arena.data_ptr()),
                C.c_void_p(self.garena.data_ptr()), C.c_void_p(self.params.data_ptr()), C.c_int64(self.ones_off),
                C.c_int64(self.zeros_off), C.c_void_p(x.data_ptr()), C.c_void_p(self.gz.data_ptr()),
                C.c_void_p(self.wsf.data_ptr()), C.c_void_p(self.ws.data_ptr()), C.c_void_p(self.sums.data_ptr()),
                C.c_int(self.n_lanes), C.c_int64(self.gz_lane), C.c_int64(self.wsf_lane), C.c_int64(self.ws_lane), C.c_int64(self.sums_lane),
                _lib._stream()),
            "mval_train_backward_lanes")
        out = []
        holders = self.model._holders
        for slots, op in zip(self.grad_slots[lo:hi], g.ops[lo:hi]):
            if not slots:
                continue
            conv = holders[op.conv]
            out.append(grads[slots["w"] : slots["w"] + conv.weight.numel()].view_as(conv.weight))
            if "b" in slots:
                out.append(grads[slots["b"] : slots["b"] + op.cout])
            if op.bn:
                out.append(grads[slots["g"] : slots["g"] + op.cout])
                out.append(grads[slots["be"] : slots["be"] + op.cout])
        if k == 0:  # the step's last segment
            if self._probe is not None:
                self._probe_read()
            self.steps += 1
        return out


_STALE = ("backward() of a train-mode forward whose saved activations were overwritten by a later train-mode "
          "forward of the same model and input shape (the training plan keeps ONE set of activations): call "
          "backward() before the next forward, as the reference's loop does (strategy.py:470-484), or run the "
          "extra forward under model.eval()")


class _SegFn(torch.autograd.Function):
    """One segment of the training graph as an autograd node.  Segment 0's forward runs the WHOLE forward (one C
    call); the nodes are chained through 1-element tokens and the last one returns the heat-maps, so autograd runs the
    backward segments last to first and hands every segment's parameter gradients to their AccumulateGrad nodes (and
    DDP's hooks) as soon as that segment is done."""

    @staticmethod
    def forward(ctx, carry, plan, k, *params):
        ctx.plan, ctx.k = plan, k
        if k == 0:
            plan._x = carry
            plan._x_version = carry._version  # (an in-place edit of the input between forward and backward must not go unnoticed)
            plan._out = plan.forward(carry)
        ctx.generation = plan.generation
        if k == len(plan.segments) - 1:
            return plan._out
        return torch.zeros(1, dtype=torch.float32, device=plan.device)

    @staticmethod
    def backward(ctx, g):
        plan, k = ctx.plan, ctx.k
        if ctx.generation != plan.generation:
            raise _lib.MvalError(_STALE)
        if plan._x is None or plan._x._version != plan._x_version:
            raise _lib.MvalError("the network input was modified in place (or released) between the train-mode forward and backward()")
        if k == len(plan.segments) - 1:
            plan.backward_begin(plan._x, g.contiguous())
        grads = plan.backward_segment(k, plan._x)
        carry_grad = None if k == 0 else torch.zeros(1, dtype=torch.float32, device=plan.device)
        if k == 0:  # the step is over: do not keep the batch's input / output alive for the life of the model
            plan._x = plan._out = None
        return (carry_grad, None, None, *grads)


def run_network_train(model, x):
    n, c, h, w = x.shape
    cache = model.__dict__.setdefault("_train_plans", {})
    p2 = not model.__dict__.get("_train_p2_off", False)  # (the slack probe's verdict on this model's parameters, TrainPlan._probe_read)
    key = (n, h, w, x.device.index, _conv_mode(), p2) + _switches()  # (the mode and the switches select the plan's kernels and packings)
    plan = cache.get(key)
    if plan is None:
        cache.clear()  # one training geometry at a time: the arenas are large
        plan = cache[key] = TrainPlan(model, n, h, w, x.device, p2=p2)
    carry = x
    for k in range(len(plan.segments)):
        carry = _SegFn.apply(carry, plan, k, *plan.seg_params[k])
    return carry
