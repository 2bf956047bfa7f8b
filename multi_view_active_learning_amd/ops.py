"""Single fused operators through the C-ABI (``mval_op_launch``) -- used by layer-wise tests,
by the training executor and for debugging; the network path uses whole-plan launches
(engine.py)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .engine import (ALGO_DIRECT, ALGO_MFMA, ALGO_MFMA_BF3, ALGO_MFMA_H2, OP_BLOCK, OP_CONV, OP_DECONV, OP_MAXPOOL, PACK_HWIO,
                     PACK_MFMA16, PACK_MFMA16_BF3, AMAX_ROW, P2_ROW, _PACK_OF, MvalOp, _align)


def pack_weights(weight, algo, transposed=False):
    """Conv2d weight (Cout,Cin,k,k) [ConvTranspose2d (Cin,Cout,k,k) if transposed] -> packed."""
    lib = _lib.lib()
    if transposed:
        cin, cout, k, _ = weight.shape
    else:
        cout, cin, k, _ = weight.shape
    pack = _PACK_OF[algo]
    # ConvTranspose2d: as stored for the direct kernel (1), tap-flipped for the MFMA kernels (2), which
    # run it as a stride-1 conv over the zero-dilated input
    mode = 0 if not transposed else {ALGO_DIRECT: 1, ALGO_MFMA: 2, ALGO_MFMA_BF3: 3, ALGO_MFMA_H2: 3}[algo]
    n = int(lib.mval_packed_weight_floats(C.c_int(pack), C.c_int(cout), C.c_int(cin), C.c_int(k)))
    out = torch.empty(n, dtype=torch.float32, device=weight.device)
    w = weight.detach().contiguous()
    _lib._check(
        lib.mval_pack_conv_weights(C.c_int(pack), C.c_int(mode), _lib._p(w), _lib._p(out), C.c_int(cout),
                                   C.c_int(cin), C.c_int(k), _lib._stream()),
        "mval_pack_conv_weights")
    return out


def fused_conv(x, weight, scale, shift, stride=1, pad=None, relu=False, res1=None, res2=None, up=0,
               algo=ALGO_MFMA, in_nchw=False, out_nchw=False, kind=OP_CONV):
    """out = act(((conv(x, w) * scale + shift + res1) + res2)), nearest-upsampled by 2^up.
    x NHWC (N,H,W,Cin) unless in_nchw; returns NHWC (N,Ho,Wo,Cout) unless out_nchw."""
    dev = x.device
    transposed = kind == OP_DECONV
    if kind == OP_MAXPOOL:
        cin = cout = x.shape[3]
        k = weight  # kernel size passed in place of the weight
    elif transposed:
        cin, cout, k, _ = weight.shape
    else:
        cout, cin, k, _ = weight.shape
    if pad is None:
        pad = k // 2
    if in_nchw:
        n, _, hin, win = x.shape
    else:
        n, hin, win, _ = x.shape
    if transposed:
        hout, wout = (hin - 1) * stride - 2 * pad + k, (win - 1) * stride - 2 * pad + k
    else:
        hout, wout = (hin + 2 * pad - k) // stride + 1, (win + 2 * pad - k) // stride + 1
    ho, wo = hout << up, wout << up
    tensors = [x.contiguous().reshape(-1)]
    offs = [0]
    for t in (res1, res2):
        if t is not None:
            offs.append(_align(offs[-1] + tensors[-1].numel()))
            tensors.append(t.contiguous().reshape(-1))
    out_off = _align(offs[-1] + tensors[-1].numel())
    amax_off = _align(out_off + n * ho * wo * cout)  # per-image max |x| slots: n for the input, n for the output
    arena = torch.zeros(amax_off + _align(2 * n * AMAX_ROW), dtype=torch.float32, device=dev)
    for o, t in zip(offs, tensors):
        arena[o : o + t.numel()] = t
    if kind == OP_MAXPOOL:
        params = torch.zeros(64, dtype=torch.float32, device=dev)
        w_off = s_off = b_off = -1
    else:
        pw = pack_weights(weight, algo, transposed)
        s_off = _align(pw.numel())
        b_off = s_off + _align(cout)
        params = torch.zeros(b_off + _align(cout), dtype=torch.float32, device=dev)
        params[: pw.numel()] = pw
        params[s_off : s_off + cout] = scale
        params[b_off : b_off + cout] = shift
        w_off = 0
    m = MvalOp()
    m.kind, m.algo = kind, algo
    m.k, m.stride, m.pad, m.cin, m.cout = k, stride, pad, cin, cout
    m.hin, m.win, m.hout, m.wout = hin, win, hout, wout
    m.up, m.relu, m.in_nchw, m.out_nchw = up, int(relu), int(in_nchw), int(out_nchw)
    m.in_off, m.out_off = 0, out_off
    it = iter(offs[1:])
    m.res1_off = next(it) if res1 is not None else -1
    m.res2_off = next(it) if res2 is not None else -1
    m.w_off, m.scale_off, m.shift_off = w_off, s_off, b_off
    m.out_amax_off = amax_off + n * AMAX_ROW
    if algo == ALGO_MFMA_H2:  # the fp16 split scales every image of its input by that image's max |x|
        m.in_amax_off = amax_off
        _lib._check(_lib.lib().mval_amax(_lib._p(arena), C.c_int64(tensors[0].numel() // n), C.c_int(n),
                                         C.c_void_p(arena.data_ptr() + 4 * amax_off), _lib._stream()), "mval_amax")
    _lib._check(
        _lib.lib().mval_op_launch(C.byref(m), C.c_int(n), _lib._p(arena), _lib._p(params), C.c_void_p(0), C.c_void_p(0),
                                  _lib._stream()),
        "mval_op_launch")
    # (tests: the per-image max |x| the kernel kept = the maximum over the first row[0] partials of each image's row)
    rows = arena[amax_off + n * AMAX_ROW : amax_off + 2 * n * AMAX_ROW].view(torch.int32).reshape(n, AMAX_ROW)
    live = torch.arange(AMAX_ROW - 1, device=dev)[None, :] < rows[:, :1]
    fused_conv.last_out_amax = torch.where(live, rows[:, 1:], torch.zeros_like(rows[:, 1:])).amax(1)
    out = arena[out_off : out_off + n * ho * wo * cout]
    return out.reshape(n, cout, ho, wo) if out_nchw else out.reshape(n, ho, wo, cout)


def fused_basic_block(x, w1, scale1, shift1, w2, scale2, shift2):
    """relu(bn2(conv3x3(relu(bn1(conv3x3(x))))) + x) in ONE launch (MVAL_OP_BLOCK, csrc/conv_block.hip): x NHWC
    (N,H,W,C) with C in {32, 64}, w1 / w2 (C,C,3,3), folded BatchNorms as (scale, shift).  Returns NHWC."""
    dev = x.device
    n, h, w, c = x.shape
    lib = _lib.lib()
    pw1, pw2 = pack_weights(w1, ALGO_MFMA_H2), pack_weights(w2, ALGO_MFMA_H2)
    out_off = _align(x.numel())
    amax_off = _align(out_off + x.numel())
    arena = torch.zeros(amax_off + _align(2 * n * AMAX_ROW), dtype=torch.float32, device=dev)
    arena[: x.numel()] = x.contiguous().reshape(-1)
    offs, top = [], 0
    chunks = []
    for t in (pw1, scale1, shift1, pw2, scale2, shift2):
        offs.append(top)
        chunks.append(t.to(dev, torch.float32).reshape(-1))
        top += _align(chunks[-1].numel())
    params = torch.zeros(top, dtype=torch.float32, device=dev)
    for o, t in zip(offs, chunks):
        params[o : o + t.numel()] = t
    m = MvalOp()
    m.kind, m.algo = OP_BLOCK, ALGO_MFMA_H2
    m.k, m.stride, m.pad, m.cin, m.cout = 3, 1, 1, c, c
    m.hin, m.win, m.hout, m.wout = h, w, h, w
    m.up, m.relu, m.in_nchw, m.out_nchw = 0, 1, 0, 0
    m.in_off, m.out_off, m.res1_off, m.res2_off = 0, out_off, 0, -1
    m.w_off, m.scale_off, m.shift_off, m.w2_off, m.scale2_off, m.shift2_off = offs
    m.in_amax_off, m.out_amax_off = amax_off, amax_off + n * AMAX_ROW
    _lib._check(lib.mval_amax(_lib._p(arena), C.c_int64(h * w * c), C.c_int(n), C.c_void_p(arena.data_ptr() + 4 * amax_off),
                              _lib._stream()), "mval_amax")
    _lib._check(lib.mval_op_launch(C.byref(m), C.c_int(n), _lib._p(arena), _lib._p(params), C.c_void_p(0), C.c_void_p(0),
                                   _lib._stream()), "mval_op_launch")
    rows = arena[amax_off + n * AMAX_ROW : amax_off + 2 * n * AMAX_ROW].view(torch.int32).reshape(n, AMAX_ROW)
    live = torch.arange(AMAX_ROW - 1, device=dev)[None, :] < rows[:, :1]
    fused_basic_block.last_out_amax = torch.where(live, rows[:, 1:], torch.zeros_like(rows[:, 1:])).amax(1)
    return arena[out_off : out_off + x.numel()].reshape(n, h, w, c)


def conv_dgrad(dz, weight, in_hw, stride=1, algo=ALGO_MFMA, accumulate_into=None):
    """Data gradient of ``F.conv2d(x, weight, stride=stride, padding=k//2)``: dz NHWC
    (N,Ho,Wo,Cout), weight (Cout,Cin,k,k) -> dx NHWC (N,H,W,Cin) via the forward kernels on
    flipped / channel-swapped weights (mval_conv_dgrad)."""
    lib = _lib.lib()
    cout, cin, k, _ = weight.shape
    n, ho, wo, _ = dz.shape
    h, w = in_hw
    pack = _PACK_OF[algo]
    nw = int(lib.mval_packed_weight_floats(C.c_int(pack), C.c_int(cin), C.c_int(cout), C.c_int(k)))
    wp = torch.empty(nw, dtype=torch.float32, device=dz.device)
    wt = weight.detach().contiguous()
    _lib._check(lib.mval_pack_conv_weights(C.c_int(pack), C.c_int(2), _lib._p(wt), _lib._p(wp), C.c_int(cin), C.c_int(cout),
                                           C.c_int(k), _lib._stream()), "mval_pack_conv_weights")
    ones = torch.ones(max(cin, cout), dtype=torch.float32, device=dz.device)
    zeros = torch.zeros_like(ones)
    dx = accumulate_into if accumulate_into is not None else torch.empty((n, h, w, cin), dtype=torch.float32, device=dz.device)
    _lib._check(
        lib.mval_conv_dgrad(_lib._p(dz.contiguous()), _lib._p(wp), _lib._p(ones), _lib._p(zeros), _lib._p(dx),
                            C.c_int(int(accumulate_into is not None)), C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(cin),
                            C.c_int(ho), C.c_int(wo), C.c_int(cout), C.c_int(k), C.c_int(stride), C.c_int(k // 2),
                            C.c_int(algo), _lib._stream()),
        "mval_conv_dgrad")
    return dx


def conv_dgrad_parity(dz, weight, in_hw, algo=ALGO_MFMA_BF3, accumulate_into=None, dz_amax_row=None):
    """Data gradient of ``F.conv2d(x, weight, stride=2, padding=1)`` with a 3x3 weight and even input sizes as four 2x2
    parity convs over dz in one launch (mval_conv_dgrad_parity): dz NHWC (N,Ho,Wo,Cout) -> dx NHWC (N,H,W,Cin)."""
    lib = _lib.lib()
    cout, cin, k, _ = weight.shape
    n, ho, wo, _ = dz.shape
    h, w = in_hw
    if k != 3 or not lib.mval_conv_dgrad_parity_supported(C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(cin), C.c_int(ho), C.c_int(wo),
                                                          C.c_int(cout), C.c_int(algo)):
        raise _lib.MvalError(f"conv_dgrad_parity: no parity form for {tuple(weight.shape)} on {h}x{w}")
    pack = _PACK_OF[algo]
    nw = int(lib.mval_packed_weight_floats(C.c_int(pack), C.c_int(cin), C.c_int(cout), C.c_int(4)))
    wp = torch.empty(nw, dtype=torch.float32, device=dz.device)
    wt = weight.detach().contiguous()
    _lib._check(lib.mval_pack_conv_weights(C.c_int(pack), C.c_int(4), _lib._p(wt), _lib._p(wp), C.c_int(cin), C.c_int(cout),
                                           C.c_int(4), _lib._stream()), "mval_pack_conv_weights")
    ones = torch.ones(max(cin, cout), dtype=torch.float32, device=dz.device)
    zeros = torch.zeros_like(ones)
    dx = accumulate_into if accumulate_into is not None else torch.empty((n, h, w, cin), dtype=torch.float32, device=dz.device)
    _lib._check(
        lib.mval_conv_dgrad_parity(_lib._p(dz.contiguous()), _lib._p(wp), _lib._p(ones), _lib._p(zeros), _lib._p(dx),
                                   C.c_int(int(accumulate_into is not None)), C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(cin),
                                   C.c_int(ho), C.c_int(wo), C.c_int(cout), C.c_int(algo),
                                   _lib._p(dz_amax_row) if dz_amax_row is not None else C.c_void_p(None), _lib._stream()),
        "mval_conv_dgrad_parity")
    return dx


def p2_bound(weight, scale, shift, transposed=False):
    """[A, B] of csrc/conv_p2.h: |bn(conv(x))| <= A * max|x| + B with A = max_c |scale_c| * sum |w_c|, B = max_c |shift_c|
    (one float32 rounding of slack each: the kernel's scale leaves a factor 2).  transposed (k4 s2 p1, weights (cin, cout, 4, 4)): an
    output pixel of parity (py, px) sees the taps ky in {3 - py, 1 - py}, kx alike -- the largest parity sum per cout."""
    wa = weight.detach().abs().double()
    if transposed:
        sw = torch.stack([wa[:, :, [3 - py, 1 - py]][:, :, :, [3 - px, 1 - px]].sum(dim=(0, 2, 3)) for py in (0, 1) for px in (0, 1)]).max(dim=0).values
    else:
        sw = wa.sum(dim=(1, 2, 3))
    a = (sw * scale.detach().abs().double()).max() * (1.0 + 1e-6)
    b = shift.detach().abs().double().max()
    return torch.stack([a, b]).to(torch.float32)


def to_p2(x_nhwc):
    """fp32 NHWC -> (P2 planes as a float32-typed buffer of the same element count, rows (n, AMAX_ROW) int32)."""
    lib = _lib.lib()
    n, h, w, c = x_nhwc.shape
    x = x_nhwc.contiguous()
    rows_in = torch.zeros(n * AMAX_ROW, dtype=torch.int32, device=x.device)
    rows = torch.zeros(n * P2_ROW, dtype=torch.int32, device=x.device)
    planes = torch.empty(x.numel(), dtype=torch.float32, device=x.device)
    _lib._check(lib.mval_amax(_lib._p(x), C.c_int64(h * w * c), C.c_int(n), _lib._p(rows_in), _lib._stream()), "mval_amax")
    _lib._check(lib.mval_nhwc_to_p2(_lib._p(x), _lib._p(rows_in), _lib._p(planes), _lib._p(rows), C.c_int(n), C.c_int(h), C.c_int(w),
                                    C.c_int(c), _lib._stream()), "mval_nhwc_to_p2")
    return planes, rows


def from_p2(planes, rows, n, h, w, c):
    out = torch.empty((n, h, w, c), dtype=torch.float32, device=planes.device)
    _lib._check(_lib.lib().mval_p2_to_nhwc(_lib._p(planes), _lib._p(rows), _lib._p(out), C.c_int(n), C.c_int(h), C.c_int(w),
                                           C.c_int(c), _lib._stream()), "mval_p2_to_nhwc")
    return out


class P2Conv:
    """One MVAL_ALGO_MFMA_P2 conv set up once (arena, packed weights, converted inputs) and launched many times:
    layer-wise tests and tools/p2_sweep.py.  x / res1 / res2 are fp32 NHWC (converted to P2 here); result() converts
    the output planes back (or returns the fp32 NCHW heat-maps when out_nchw)."""

    def __init__(self, x, weight, scale, shift, stride=1, relu=False, res1=None, res2=None, up=0, out_nchw=False, transposed=False):
        dev = x.device
        lib = _lib.lib()
        n, hin, win, _ = x.shape
        if transposed:  # ConvTranspose2d(k4, s2, p1) weights (cin, cout, 4, 4): four parity launches (csrc/net.hip)
            cin, cout, k, _ = weight.shape
            stride, pad = 2, 1
            hout, wout = 2 * hin, 2 * win
        else:
            cout, cin, k, _ = weight.shape
            pad = k // 2
            hout, wout = (hin + 2 * pad - k) // stride + 1, (win + 2 * pad - k) // stride + 1
        ho, wo = hout << up, wout << up
        self.shape = (n, ho, wo, cout)
        self.out_nchw = out_nchw
        parts = [to_p2(x)] + [to_p2(t) for t in (res1, res2) if t is not None]
        offs, top = [], 0
        for pl, _ in parts:
            offs.append(top)
            top += _align(pl.numel())
        out_off = top
        top += _align(n * ho * wo * cout)
        row_off = top
        nrows = len(parts) + 1
        self.arena = torch.zeros(top + _align(nrows * n * P2_ROW), dtype=torch.float32, device=dev)
        for i, (pl, rows) in enumerate(parts):
            self.arena[offs[i] : offs[i] + pl.numel()] = pl
            self.arena[row_off + i * n * P2_ROW : row_off + (i + 1) * n * P2_ROW] = rows.view(torch.float32)
        pw = pack_weights(weight, ALGO_MFMA_H2, transposed=transposed)
        s_off = _align(pw.numel())
        b_off = s_off + _align(cout)
        bd_off = b_off + _align(cout)
        self.params = torch.zeros(bd_off + 64, dtype=torch.float32, device=dev)
        self.params[: pw.numel()] = pw
        self.params[s_off : s_off + cout] = scale
        self.params[b_off : b_off + cout] = shift
        self.params[bd_off : bd_off + 2] = p2_bound(weight, scale, shift, transposed).to(dev)
        m = MvalOp()
        m.kind, m.algo = (OP_DECONV if transposed else OP_CONV), 4
        m.k, m.stride, m.pad, m.cin, m.cout = k, stride, pad, cin, cout
        m.hin, m.win, m.hout, m.wout = hin, win, hout, wout
        m.up, m.relu, m.in_nchw, m.out_nchw = up, int(relu), 0, int(out_nchw)
        m.in_off, m.out_off = 0, out_off
        m.in_amax_off = row_off
        it = iter(range(1, len(parts)))
        m.res1_off = m.res2_off = -1
        if res1 is not None:
            i = next(it)
            m.res1_off, m.res1_amax_off = offs[i], row_off + i * n * P2_ROW
        if res2 is not None:
            i = next(it)
            m.res2_off, m.res2_amax_off = offs[i], row_off + i * n * P2_ROW
        m.out_amax_off = row_off + len(parts) * n * P2_ROW
        m.w_off, m.scale_off, m.shift_off, m.bound_off = 0, s_off, b_off, bd_off
        self.op, self.out_off, self.n = m, out_off, n
        if not lib.mval_op_algo_supported(C.byref(m), C.c_int(n), C.c_int(4)):
            raise _lib.MvalError("no P2 kernel for this geometry")

    def launch(self):
        _lib._check(_lib.lib().mval_op_launch(C.byref(self.op), C.c_int(self.n), _lib._p(self.arena), _lib._p(self.params),
                                              C.c_void_p(0), C.c_void_p(0), _lib._stream()), "mval_op_launch")

    def out_rows(self):
        n = self.n
        return self.arena[self.op.out_amax_off : self.op.out_amax_off + n * P2_ROW].view(torch.int32).reshape(n, P2_ROW)

    def result(self):
        n, ho, wo, cout = self.shape
        out = self.arena[self.out_off : self.out_off + n * ho * wo * cout]
        if self.out_nchw:
            return out.reshape(n, cout, ho, wo).clone()
        return from_p2(out, self.out_rows().reshape(-1), n, ho, wo, cout)

    def kept_amax(self):
        return self.out_rows()[:, :256].amax(1).view(torch.float32)  # (non-negative floats order like their bits)


def fused_conv_p2(x, weight, scale, shift, **kw):
    c = P2Conv(x, weight, scale, shift, **kw)
    c.launch()
    fused_conv_p2.last = c
    return c.result()


class P2Block:
    """One fused BasicBlock over P2 activations (MVAL_OP_BLOCK with MVAL_ALGO_MFMA_P2, csrc/conv_block_p2.hip), set up once
    and launched many times: relu(bn2(conv3x3(relu(bn1(conv3x3(x))))) + x), x fp32 NHWC (converted here), C in {32, 64}."""

    def __init__(self, x, w1, scale1, shift1, w2, scale2, shift2):
        dev = x.device
        n, h, w, c = x.shape
        self.shape = (n, h, w, c)
        planes, rows = to_p2(x)
        out_off = _align(planes.numel())
        row_off = out_off + _align(planes.numel())
        self.arena = torch.zeros(row_off + _align(2 * n * P2_ROW), dtype=torch.float32, device=dev)
        self.arena[: planes.numel()] = planes
        self.arena[row_off : row_off + n * P2_ROW] = rows.view(torch.float32)
        chunks = [pack_weights(w1, ALGO_MFMA_H2), scale1, shift1, p2_bound(w1, scale1, shift1), pack_weights(w2, ALGO_MFMA_H2), scale2, shift2,
                  p2_bound(w2, scale2, shift2)]
        offs, top = [], 0
        for t in chunks:
            offs.append(top)
            top += _align(t.numel())
        self.params = torch.zeros(top, dtype=torch.float32, device=dev)
        for o, t in zip(offs, chunks):
            self.params[o : o + t.numel()] = t.to(dev, torch.float32).reshape(-1)
        m = MvalOp()
        m.kind, m.algo = OP_BLOCK, 4
        m.k, m.stride, m.pad, m.cin, m.cout = 3, 1, 1, c, c
        m.hin, m.win, m.hout, m.wout = h, w, h, w
        m.up, m.relu, m.in_nchw, m.out_nchw = 0, 1, 0, 0
        m.in_off, m.out_off, m.res1_off, m.res2_off = 0, out_off, 0, -1
        m.w_off, m.scale_off, m.shift_off, m.bound_off, m.w2_off, m.scale2_off, m.shift2_off, m.bound2_off = offs
        m.in_amax_off, m.out_amax_off = row_off, row_off + n * P2_ROW
        self.op, self.out_off, self.n = m, out_off, n
        if not _lib.lib().mval_op_algo_supported(C.byref(m), C.c_int(n), C.c_int(4)):
            raise _lib.MvalError("no fused P2 BasicBlock kernel for this geometry")

    launch = P2Conv.launch
    out_rows = P2Conv.out_rows
    kept_amax = P2Conv.kept_amax

    def result(self):
        n, h, w, c = self.shape
        return from_p2(self.arena[self.out_off : self.out_off + n * h * w * c], self.out_rows().reshape(-1), n, h, w, c)


def fused_basic_block_p2(x, w1, scale1, shift1, w2, scale2, shift2):
    b = P2Block(x, w1, scale1, shift1, w2, scale2, shift2)
    b.launch()
    fused_basic_block_p2.last = b
    return b.result()


class P2Bneck:
    """One fused Bottleneck over P2 activations (MVAL_OP_BNECK, csrc/conv_bneck_p2.hip), set up once and launched many times:
    relu(bn3(conv1x1(relu(bn2(conv3x3(relu(bn1(conv1x1(x)))))))) + res), x fp32 NHWC with 64 or 256 channels (converted here),
    64 planes, res fp32 NHWC with 256 channels (None: x itself, which then has 256 channels).  convs = three (weight, scale,
    shift) triples."""

    def __init__(self, x, convs, res=None):
        dev = x.device
        n, h, w, cin = x.shape
        self.shape = (n, h, w, 256)
        parts = [to_p2(x)] + ([to_p2(res)] if res is not None else [])
        offs_a, top = [], 0
        for pl, _ in parts:
            offs_a.append(top)
            top += _align(pl.numel())
        out_off = top
        top += _align(n * h * w * 256)
        row_off = top
        self.arena = torch.zeros(top + _align((len(parts) + 1) * n * P2_ROW), dtype=torch.float32, device=dev)
        for i, (pl, rows) in enumerate(parts):
            self.arena[offs_a[i] : offs_a[i] + pl.numel()] = pl
            self.arena[row_off + i * n * P2_ROW : row_off + (i + 1) * n * P2_ROW] = rows.view(torch.float32)
        chunks = []
        for wt, sc, sh in convs:
            chunks += [pack_weights(wt, ALGO_MFMA_H2), sc, sh, p2_bound(wt, sc, sh)]
        offs, top = [], 0
        for t in chunks:
            offs.append(top)
            top += _align(t.numel())
        self.params = torch.zeros(top, dtype=torch.float32, device=dev)
        for o, t in zip(offs, chunks):
            self.params[o : o + t.numel()] = t.to(dev, torch.float32).reshape(-1)
        m = MvalOp()
        m.kind, m.algo = 5, 4  # MVAL_OP_BNECK, MVAL_ALGO_MFMA_P2
        m.k, m.stride, m.pad, m.cin, m.cout = 1, 1, 0, cin, 256
        m.hin, m.win, m.hout, m.wout = h, w, h, w
        m.up, m.relu, m.in_nchw, m.out_nchw = 0, 1, 0, 0
        m.in_off, m.out_off, m.res2_off = 0, out_off, -1
        m.res1_off = offs_a[1] if res is not None else 0
        (m.w_off, m.scale_off, m.shift_off, m.bound_off, m.w2_off, m.scale2_off, m.shift2_off, m.bound2_off,
         m.w3_off, m.scale3_off, m.shift3_off, m.bound3_off) = offs
        m.in_amax_off = row_off
        m.res1_amax_off = row_off + (n * P2_ROW if res is not None else 0)
        m.out_amax_off = row_off + len(parts) * n * P2_ROW
        self.op, self.out_off, self.n = m, out_off, n
        if not _lib.lib().mval_op_algo_supported(C.byref(m), C.c_int(n), C.c_int(4)):
            raise _lib.MvalError("no fused P2 Bottleneck kernel for this geometry")

    launch = P2Conv.launch
    out_rows = P2Conv.out_rows
    kept_amax = P2Conv.kept_amax

    def result(self):
        n, h, w, c = self.shape
        return from_p2(self.arena[self.out_off : self.out_off + n * h * w * c], self.out_rows().reshape(-1), n, h, w, c)


def fused_bottleneck_p2(x, convs, res=None):
    b = P2Bneck(x, convs, res)
    b.launch()
    fused_bottleneck_p2.last = b
    return b.result()


class P2Stem:
    """HRNet's stem in one launch (MVAL_OP_STEM_P2, csrc/conv_stem_p2.hip): relu(bn2(conv3x3 s2(relu(bn1(conv3x3 s2(x)))))) from the
    fp32 NCHW image x (n, 3, h, w) to 64 channels of P2 planes at a quarter of the resolution; result(): fp32 NHWC."""

    def __init__(self, x, w1, scale1, shift1, w2, scale2, shift2):
        dev = x.device
        n, _, h, w = x.shape
        ho, wo = h // 4, w // 4
        self.shape = (n, ho, wo, 64)
        self.x = x.contiguous()
        out_floats = _align(n * ho * wo * 64)
        self.arena = torch.zeros(out_floats + _align(2 * n * P2_ROW), dtype=torch.float32, device=dev)
        chunks = [pack_weights(w1, ALGO_DIRECT), scale1, shift1, p2_bound(w1, scale1, shift1), pack_weights(w2, ALGO_MFMA_H2), scale2, shift2,
                  p2_bound(w2, scale2, shift2)]
        offs, top = [], 0
        for t in chunks:
            offs.append(top)
            top += _align(t.numel())
        self.params = torch.zeros(top, dtype=torch.float32, device=dev)
        for o, t in zip(offs, chunks):
            self.params[o : o + t.numel()] = t.to(dev, torch.float32).reshape(-1)
        m = MvalOp()
        m.kind, m.algo = 6, 4  # MVAL_OP_STEM_P2, MVAL_ALGO_MFMA_P2
        m.k, m.stride, m.pad, m.cin, m.cout = 3, 2, 1, 3, 64
        m.hin, m.win, m.hout, m.wout = h, w, ho, wo
        m.up, m.relu, m.in_nchw, m.out_nchw = 0, 1, 1, 0
        m.in_off, m.out_off, m.res1_off, m.res2_off = -1, 0, -1, -1
        m.w_off, m.scale_off, m.shift_off, m.bound_off, m.w2_off, m.scale2_off, m.shift2_off, m.bound2_off = offs
        m.in_amax_off, m.out_amax_off = out_floats, out_floats + n * P2_ROW
        self.op, self.out_off, self.n = m, 0, n
        if not _lib.lib().mval_op_algo_supported(C.byref(m), C.c_int(n), C.c_int(4)):
            raise _lib.MvalError("no fused P2 stem kernel for this geometry")

    def launch(self):
        _lib._check(_lib.lib().mval_op_launch(C.byref(self.op), C.c_int(self.n), _lib._p(self.arena), _lib._p(self.params),
                                              _lib._p(self.x), C.c_void_p(0), _lib._stream()), "mval_op_launch")

    out_rows = P2Conv.out_rows
    kept_amax = P2Conv.kept_amax

    def result(self):
        n, h, w, c = self.shape
        return from_p2(self.arena[: n * h * w * c], self.out_rows().reshape(-1), n, h, w, c)


def fused_stem_p2(x, w1, scale1, shift1, w2, scale2, shift2):
    b = P2Stem(x, w1, scale1, shift1, w2, scale2, shift2)
    b.launch()
    fused_stem_p2.last = b
    return b.result()


class P2FuseUp:
    """The up-sampling terms of one HRNet fuse-layer output in one launch (MVAL_OP_FUSE_UP, csrc/conv_fuse_up_p2.hip):
    act(((res + up(bn(conv1x1(x_0)))) + up(bn(conv1x1(x_1)))) [+ ...]); res fp32 NHWC (n, h, w, c) with c in {32, 64}; terms = two or
    three (x NHWC at (h >> up, w >> up), weight (c, cin, 1, 1), scale, shift, up)."""

    def __init__(self, res, terms, relu=True):
        dev = res.device
        n, h, w, c = res.shape
        self.shape = (n, h, w, c)
        parts = [to_p2(res)] + [to_p2(t[0]) for t in terms]
        offs_a, top = [], 0
        for pl, _ in parts:
            offs_a.append(top)
            top += _align(pl.numel())
        out_off = top
        top += _align(n * h * w * c)
        row_off = top
        self.arena = torch.zeros(top + _align((len(parts) + 1) * n * P2_ROW), dtype=torch.float32, device=dev)
        for i, (pl, rows) in enumerate(parts):
            self.arena[offs_a[i] : offs_a[i] + pl.numel()] = pl
            self.arena[row_off + i * n * P2_ROW : row_off + (i + 1) * n * P2_ROW] = rows.view(torch.float32)
        chunks = []
        for x, wt, sc, sh, up in terms:
            chunks += [pack_weights(wt, ALGO_MFMA_H2), sc, sh, p2_bound(wt, sc, sh)]
        offs, ptop = [], 0
        for t in chunks:
            offs.append(ptop)
            ptop += _align(t.numel())
        self.params = torch.zeros(ptop, dtype=torch.float32, device=dev)
        for o, t in zip(offs, chunks):
            self.params[o : o + t.numel()] = t.to(dev, torch.float32).reshape(-1)
        m = MvalOp()
        m.kind, m.algo = 7, 4  # MVAL_OP_FUSE_UP, MVAL_ALGO_MFMA_P2
        m.k, m.stride, m.pad, m.cin, m.cout = 1, 1, 0, terms[0][0].shape[-1], c
        m.hin, m.win, m.hout, m.wout = h, w, h, w
        m.up, m.relu, m.in_nchw, m.out_nchw = 0, int(relu), 0, 0
        m.in_off, m.out_off, m.res1_off, m.res2_off = offs_a[1], out_off, 0, -1
        m.res1_amax_off = row_off
        m.out_amax_off = row_off + len(parts) * n * P2_ROW
        m.n_terms = len(terms)
        for j, (x, wt, sc, sh, up) in enumerate(terms):
            m.t_cin[j], m.t_up[j] = x.shape[-1], up
            m.t_in_off[j], m.t_in_amax_off[j] = offs_a[1 + j], row_off + (1 + j) * n * P2_ROW
            m.t_w_off[j], m.t_scale_off[j], m.t_shift_off[j], m.t_bound_off[j] = offs[4 * j : 4 * j + 4]
        self.op, self.out_off, self.n = m, out_off, n
        if not _lib.lib().mval_op_algo_supported(C.byref(m), C.c_int(n), C.c_int(4)):
            raise _lib.MvalError("no fused up-path kernel for this geometry")

    launch = P2Conv.launch
    out_rows = P2Conv.out_rows
    kept_amax = P2Conv.kept_amax

    def result(self):
        n, h, w, c = self.shape
        return from_p2(self.arena[self.out_off : self.out_off + n * h * w * c], self.out_rows().reshape(-1), n, h, w, c)


def fused_up_terms_p2(res, terms, relu=True):
    b = P2FuseUp(res, terms, relu)
    b.launch()
    fused_up_terms_p2.last = b
    return b.result()

