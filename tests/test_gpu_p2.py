"""GPU parity tests of the P2 path (csrc/conv_p2.h): activations kept in HBM as the pair of fp16 planes the fp16-split
MFMA convs consume.  Single operators against float64 torch-CPU (and against the exact-fp32 MFMA chain's error), the
fused BasicBlock, the format itself, and whole plans against the h2 / exact-fp32 plans."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from multi_view_active_learning_amd import _lib

    _lib.lib()
    return torch.device("cuda:0")


def _ref_conv(x, w, scale, shift, stride, relu, res1, res2, up):
    y = F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2)
    y = y * scale[None, :, None, None] + shift[None, :, None, None]
    if up:
        y = F.interpolate(y, scale_factor=2**up, mode="nearest")
    if res1 is not None:
        y = y + res1
    if res2 is not None:
        y = y + res2
    return F.relu(y) if relu else y


# n, cin, cout, h, w, k, stride, relu, res1, res2, up, out_nchw
P2_CASES = [
    (3, 32, 32, 64, 64, 3, 1, True, True, False, 0, False),
    (2, 64, 64, 32, 32, 3, 1, True, False, False, 0, False),
    (2, 128, 128, 16, 16, 3, 1, True, True, False, 0, False),
    (5, 256, 256, 8, 8, 3, 1, True, True, False, 0, False),
    (2, 256, 32, 64, 64, 3, 1, True, False, False, 0, False),
    (2, 32, 64, 64, 64, 3, 2, False, True, True, 0, False),
    (2, 64, 128, 32, 32, 3, 2, True, True, True, 0, False),
    (3, 128, 256, 16, 16, 3, 2, True, True, False, 0, False),
    (2, 64, 32, 32, 32, 1, 1, False, True, False, 1, False),
    (2, 128, 32, 16, 16, 1, 1, True, True, False, 2, False),
    (2, 256, 32, 8, 8, 1, 1, True, True, True, 3, False),
    (2, 64, 256, 64, 64, 1, 1, True, True, False, 0, False),
    # the x4-store / SGPR-soffset hazard showed on exactly these shapes (1x1, 64-wide tiles, several sub-tiles per row)
    (2, 256, 64, 64, 64, 1, 1, True, False, False, 0, False),
    (1, 64, 64, 16, 64, 1, 1, False, False, False, 0, False),
    (2, 32, 19, 64, 64, 1, 1, False, False, False, 0, True),
    # HRNet-W48 at 384 x 288: 48-channel first branch, maps that no tile divides
    (1, 48, 48, 96, 72, 3, 1, True, True, False, 0, False),
    (2, 192, 192, 24, 18, 3, 1, True, True, False, 0, False),
    (3, 384, 384, 12, 9, 3, 1, True, True, False, 0, False),
    (2, 48, 96, 96, 72, 3, 2, True, False, False, 0, False),
    # round 4: full-width odd tiles (3 x 18 / 7 x 9 pixels) -- partial last tiles, two residuals, no ReLU, 96 couts
    (2, 96, 192, 10, 18, 3, 1, False, True, True, 0, False),
    (3, 192, 96, 5, 9, 3, 1, True, False, False, 0, False),
    (1, 64, 64, 23, 9, 3, 1, True, True, False, 0, False),
    (2, 96, 96, 48, 36, 3, 1, True, True, False, 0, False),  # two 18-wide odd tiles per row
    (1, 64, 128, 7, 36, 3, 1, False, False, False, 0, False),
    (2, 192, 384, 24, 18, 3, 2, True, True, True, 0, False),
    (1, 192, 48, 24, 18, 1, 1, False, True, False, 2, False),
    (2, 64, 256, 96, 72, 1, 1, True, True, False, 0, False),
    (2, 48, 19, 96, 72, 1, 1, False, False, False, 0, True),
    # round 5: two / three cout sub-tiles per wave (128 / 256 / 96 / 192 channels): residual granules requested inside the epilogue --
    # both residuals, no ReLU, a batch that leaves the tile walk uneven
    (3, 128, 128, 16, 16, 3, 1, False, True, True, 0, False),
    (7, 256, 256, 8, 8, 3, 1, True, True, True, 0, False),
    (1, 96, 96, 48, 36, 3, 1, False, True, True, 0, False),
    (3, 192, 192, 24, 18, 3, 1, True, True, True, 0, False),
    (2, 128, 128, 32, 48, 3, 1, True, False, False, 0, False),
]


def _make(case):
    n, cin, cout, h, w, k, stride, relu, r1, r2, up, out_nchw = case
    rng = np.random.default_rng(abs(hash(case)) % 2**31)
    x = torch.from_numpy(rng.standard_normal((n, cin, h, w)).astype(np.float32))
    wt = torch.from_numpy((rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (cin * k * k))).astype(np.float32))
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    shift = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1)
    ho = ((h + 2 * (k // 2) - k) // stride + 1) << up
    wo = ((w + 2 * (k // 2) - k) // stride + 1) << up
    res1 = torch.from_numpy(rng.standard_normal((n, cout, ho, wo)).astype(np.float32)) if r1 else None
    res2 = torch.from_numpy(rng.standard_normal((n, cout, ho, wo)).astype(np.float32)) if r2 else None
    return x, wt, scale, shift, res1, res2


@pytest.mark.parametrize("case", P2_CASES, ids=lambda c: "n%d_c%d-%d_%dx%d_k%ds%d_r%d%d%d_u%d_o%d" % tuple(int(v) for v in c))
def test_p2_conv_vs_float64(dev, case):
    from multi_view_active_learning_amd import ops

    n, cin, cout, h, w, k, stride, relu, r1, r2, up, out_nchw = case
    x, wt, scale, shift, res1, res2 = _make(case)
    d = torch.float64
    want = _ref_conv(x.to(d), wt.to(d), scale.to(d), shift.to(d), stride, relu, None if res1 is None else res1.to(d),
                     None if res2 is None else res2.to(d), up)
    nhwc = lambda t: None if t is None else t.permute(0, 2, 3, 1).contiguous().to(dev)
    got = ops.fused_conv_p2(nhwc(x), wt.to(dev), scale.to(dev), shift.to(dev), stride=stride, relu=relu, res1=nhwc(res1), res2=nhwc(res2),
                            up=up, out_nchw=out_nchw)
    got = got.cpu() if out_nchw else got.permute(0, 3, 1, 2).cpu()
    np.testing.assert_allclose(got.numpy(), want.float().numpy(), rtol=1e-4, atol=2e-5)
    rms = (got.double() - want).pow(2).mean().sqrt().item()
    if not out_nchw:
        # every producer keeps the exact per-image max |x| of the fp32 values it then stored as (h, l) pairs
        kept = ops.fused_conv_p2.last.kept_amax().cpu()
        assert torch.allclose(kept, got.abs().amax(dim=(1, 2, 3)), rtol=2.0**-21, atol=0)
    if cin % 32 == 0 and not out_nchw:
        # the bound that makes this a legitimate fp32 path: not less accurate than the EXACT-fp32 MFMA chain
        # (measured: rms 0.62 - 0.85 of it -- the inputs arrive already rounded to 22 bits, the chain's roundings dominate)
        y = ops.fused_conv(nhwc(x), wt.to(dev), scale.to(dev), shift.to(dev), stride=stride, relu=relu, res1=nhwc(res1), res2=nhwc(res2),
                           up=up, algo=ops.ALGO_MFMA).permute(0, 3, 1, 2).cpu()
        rms32 = (y.double() - want).pow(2).mean().sqrt().item()
        assert rms <= 1.25 * rms32 + 1e-8, (rms, rms32)


@pytest.mark.parametrize("shape", [(3, 32, 64, 64), (2, 64, 32, 32), (2, 32, 21, 37), (1, 64, 9, 16), (5, 32, 8, 16), (2, 64, 24, 40), (2, 32, 96, 72)],
                         ids=lambda s: "n%d_c%d_%dx%d" % s)
def test_p2_basic_block_vs_float64(dev, shape):
    """MVAL_OP_BLOCK over P2 activations (hrnet.py:19-52 in one launch) against float64, against the same block as two
    P2 conv launches (not less accurate), and image 0 alone gives the same bits (per-image scales, fixed tiling)."""
    from multi_view_active_learning_amd import ops

    n, c, h, w = shape
    rng = np.random.default_rng(7 + h)
    x = torch.from_numpy(np.maximum(rng.standard_normal((n, c, h, w)), 0).astype(np.float32) * 1.5)
    ws = [torch.from_numpy((rng.standard_normal((c, c, 3, 3)) * np.sqrt(2.0 / (c * 9))).astype(np.float32)) for _ in range(2)]
    sc = [torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)) for _ in range(2)]
    sh = [torch.from_numpy(rng.standard_normal(c).astype(np.float32) * 0.1) for _ in range(2)]
    d = torch.float64
    mid = _ref_conv(x.to(d), ws[0].to(d), sc[0].to(d), sh[0].to(d), 1, True, None, None, 0)
    want = _ref_conv(mid, ws[1].to(d), sc[1].to(d), sh[1].to(d), 1, True, x.to(d), None, 0)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    args = [t.to(dev) for t in (ws[0], sc[0], sh[0], ws[1], sc[1], sh[1])]
    got = ops.fused_basic_block_p2(xd, *args)
    kept = ops.fused_basic_block_p2.last.kept_amax().cpu()
    got = got.permute(0, 3, 1, 2).cpu()
    np.testing.assert_allclose(got.numpy(), want.float().numpy(), rtol=1e-4, atol=2e-5)
    assert torch.allclose(kept, got.abs().amax(dim=(1, 2, 3)), rtol=2.0**-21, atol=0)
    m1 = ops.fused_conv_p2(xd, args[0], args[1], args[2], relu=True)
    two = ops.fused_conv_p2(m1, args[3], args[4], args[5], relu=True, res1=xd).permute(0, 3, 1, 2).cpu()
    rms = lambda y: (y.double() - want).pow(2).mean().sqrt().item()
    assert rms(got) <= 1.25 * rms(two) + 1e-8, (rms(got), rms(two))
    alone = ops.fused_basic_block_p2(xd[:1].contiguous(), *args).permute(0, 3, 1, 2).cpu()
    assert torch.equal(alone[0], got[0])


@pytest.mark.parametrize("shape", [(2, 256, 64, 64, False), (2, 64, 64, 64, True), (3, 256, 21, 37, False), (1, 64, 9, 16, True), (2, 256, 96, 72, False),
                                   (2, 256, 64, 48, True), (5, 64, 8, 16, True)], ids=lambda s: "n%d_c%d_%dx%d_r%d" % s)
def test_p2_bottleneck_vs_float64(dev, shape):
    """MVAL_OP_BNECK over P2 activations (hrnet.py:75-95 with 64 planes in one launch: the blocks of HRNet's layer1) against
    float64, against the same block as three P2 conv launches (not less accurate), kept max |x|, and image 0 alone gives the
    same bits.  The residual is the block's input (256 channels) or another tensor (the downsample branch)."""
    from multi_view_active_learning_amd import ops

    n, cin, h, w, with_res = shape
    rng = np.random.default_rng(11 + h + cin)
    x = torch.from_numpy(np.maximum(rng.standard_normal((n, cin, h, w)), 0).astype(np.float32) * 1.5)
    res = torch.from_numpy(rng.standard_normal((n, 256, h, w)).astype(np.float32)) if with_res else None
    convs = []
    for co, ci, k in ((64, cin, 1), (64, 64, 3), (256, 64, 1)):
        convs.append((torch.from_numpy((rng.standard_normal((co, ci, k, k)) * np.sqrt(2.0 / (ci * k * k))).astype(np.float32)),
                      torch.from_numpy(rng.uniform(0.5, 1.5, co).astype(np.float32)), torch.from_numpy(rng.standard_normal(co).astype(np.float32) * 0.1)))
    d = torch.float64
    t = x.to(d)
    for i, (wt, sc, sh) in enumerate(convs):
        t = _ref_conv(t, wt.to(d), sc.to(d), sh.to(d), 1, True, (x if res is None else res).to(d) if i == 2 else None, None, 0)
    want = t
    nhwc = lambda v: v.permute(0, 2, 3, 1).contiguous().to(dev)
    xd, rd = nhwc(x), None if res is None else nhwc(res)
    cd = [tuple(v.to(dev) for v in c) for c in convs]
    got = ops.fused_bottleneck_p2(xd, cd, rd)
    kept = ops.fused_bottleneck_p2.last.kept_amax().cpu()
    got = got.permute(0, 3, 1, 2).cpu()
    np.testing.assert_allclose(got.numpy(), want.float().numpy(), rtol=1e-4, atol=3e-5)
    assert torch.allclose(kept, got.abs().amax(dim=(1, 2, 3)), rtol=2.0**-21, atol=0)
    m1 = ops.fused_conv_p2(xd, *cd[0], relu=True)
    m2 = ops.fused_conv_p2(m1, *cd[1], relu=True)
    three = ops.fused_conv_p2(m2, *cd[2], relu=True, res1=xd if rd is None else rd).permute(0, 3, 1, 2).cpu()
    rms = lambda y: (y.double() - want).pow(2).mean().sqrt().item()
    assert rms(got) <= 1.25 * rms(three) + 1e-8, (rms(got), rms(three))
    alone = ops.fused_bottleneck_p2(xd[:1].contiguous(), cd, None if rd is None else rd[:1].contiguous()).permute(0, 3, 1, 2).cpu()
    assert torch.equal(alone[0], got[0])


@pytest.mark.parametrize("shape", [(2, 256, 256), (3, 64, 64), (1, 384, 288), (2, 100, 76), (1, 16, 64)], ids=lambda s: "n%d_%dx%d" % s)
def test_p2_stem_vs_float64(dev, shape):
    """MVAL_OP_STEM_P2 (hrnet.py:303-310: both stride-2 stem convs in one launch, fp32 NCHW image -> P2 planes) against
    float64, against the stand-alone fp32 stem kernel followed by a P2 conv launch (not less accurate), kept max |x|, and
    image 0 alone gives the same bits."""
    from multi_view_active_learning_amd import ops

    n, h, w = shape
    rng = np.random.default_rng(5 + h)
    x = torch.from_numpy(rng.standard_normal((n, 3, h, w)).astype(np.float32) * 1.2)
    x[-1] *= 3.0  # (per-image scales)
    w1 = torch.from_numpy((rng.standard_normal((64, 3, 3, 3)) * np.sqrt(2.0 / 27)).astype(np.float32))
    w2 = torch.from_numpy((rng.standard_normal((64, 64, 3, 3)) * np.sqrt(2.0 / 576)).astype(np.float32))
    sc = [torch.from_numpy(rng.uniform(0.5, 1.5, 64).astype(np.float32)) for _ in range(2)]
    sh = [torch.from_numpy(rng.standard_normal(64).astype(np.float32) * 0.1) for _ in range(2)]
    d = torch.float64
    y1 = _ref_conv(x.to(d), w1.to(d), sc[0].to(d), sh[0].to(d), 2, True, None, None, 0)
    want = _ref_conv(y1, w2.to(d), sc[1].to(d), sh[1].to(d), 2, True, None, None, 0)
    args = [t.to(dev) for t in (w1, sc[0], sh[0], w2, sc[1], sh[1])]
    got = ops.fused_stem_p2(x.to(dev), *args)
    kept = ops.fused_stem_p2.last.kept_amax().cpu()
    got = got.permute(0, 3, 1, 2).cpu()
    assert got.shape == want.shape
    np.testing.assert_allclose(got.numpy(), want.float().numpy(), rtol=1e-4, atol=3e-5)
    assert torch.allclose(kept, got.abs().amax(dim=(1, 2, 3)), rtol=2.0**-21, atol=0)
    s1 = ops.fused_conv(x.to(dev), args[0], args[1], args[2], stride=2, relu=True, algo=ops.ALGO_DIRECT, in_nchw=True)
    two = ops.fused_conv_p2(s1, args[3], args[4], args[5], stride=2, relu=True).permute(0, 3, 1, 2).cpu()
    rms = lambda y: (y.double() - want).pow(2).mean().sqrt().item()
    assert rms(got) <= 1.25 * rms(two) + 1e-8, (rms(got), rms(two))
    alone = ops.fused_stem_p2(x[:1].contiguous().to(dev), *args).permute(0, 3, 1, 2).cpu()
    assert torch.equal(alone[0], got[0])


@pytest.mark.parametrize("case", [(2, 32, 64, 64, 3), (2, 32, 64, 64, 2), (3, 64, 32, 32, 2), (1, 32, 96, 72, 3), (2, 32, 24, 40, 2), (1, 64, 16, 8, 2),
                                  # round 4: HRNet-W48's fuse outputs (48 / 96 channels: three / six cout sub-tiles over four waves; 8 x 24 and
                                  # 4 x 36 tiles on the 72- / 36-wide maps, 8 x 32 / 4 x 32 elsewhere)
                                  (2, 48, 96, 72, 3), (1, 48, 96, 72, 2), (2, 96, 48, 36, 2), (1, 48, 32, 64, 2), (1, 96, 24, 40, 2)],
                         ids=lambda c: "n%d_c%d_%dx%d_t%d" % c)
def test_p2_fuse_up_terms_vs_float64(dev, case):
    """MVAL_OP_FUSE_UP (hrnet.py:424-447: the up-sampling 1x1 terms of one fuse-layer output added to the partial sum in one
    launch) against float64, against the chain of P2 conv launches it replaces (not less accurate: the chain rounds every
    partial sum to the pair format), kept max |x|, and image 0 alone gives the same bits."""
    from multi_view_active_learning_amd import _lib, ops

    n, c, h, w, nt = case
    rng = np.random.default_rng(3 + h + nt)
    res = torch.from_numpy(np.maximum(rng.standard_normal((n, c, h, w)), 0).astype(np.float32) * 1.5)
    terms = []
    for j in range(nt):
        up, cin = j + 1, c << (j + 1)
        x = torch.from_numpy(np.maximum(rng.standard_normal((n, cin, h >> up, w >> up)), 0).astype(np.float32))
        wt = torch.from_numpy((rng.standard_normal((c, cin, 1, 1)) * np.sqrt(2.0 / cin)).astype(np.float32))
        terms.append((x, wt, torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)), torch.from_numpy(rng.standard_normal(c).astype(np.float32) * 0.1), up))
    d = torch.float64
    want = res.to(d)
    for j, (x, wt, sc, sh, up) in enumerate(terms):
        want = _ref_conv(x.to(d), wt.to(d), sc.to(d), sh.to(d), 1, j == nt - 1, want, None, up)
    nhwc = lambda v: v.permute(0, 2, 3, 1).contiguous().to(dev)
    td = [(nhwc(x), wt.to(dev), sc.to(dev), sh.to(dev), up) for x, wt, sc, sh, up in terms]
    got = ops.fused_up_terms_p2(nhwc(res), td, relu=True)
    kept = ops.fused_up_terms_p2.last.kept_amax().cpu()
    got = got.permute(0, 3, 1, 2).cpu()
    np.testing.assert_allclose(got.numpy(), want.float().numpy(), rtol=1e-4, atol=3e-5)
    assert torch.allclose(kept, got.abs().amax(dim=(1, 2, 3)), rtol=2.0**-21, atol=0)
    try:  # (the one-term kernels need 8-pixel-wide low-resolution maps: the smallest cases have no chain to compare with)
        acc = nhwc(res)
        for j, (x, wt, sc, sh, up) in enumerate(td):
            acc = ops.fused_conv_p2(x, wt, sc, sh, relu=(j == nt - 1), res1=acc, up=up)
        chain = acc.permute(0, 3, 1, 2).cpu()
        rms = lambda y: (y.double() - want).pow(2).mean().sqrt().item()
        assert rms(got) <= 1.25 * rms(chain) + 1e-8, (rms(got), rms(chain))
    except _lib.MvalError:
        assert min(h, w) >> nt < 8
    alone = ops.fused_up_terms_p2(nhwc(res[:1]), [(x[:1].contiguous(), wt, sc, sh, up) for x, wt, sc, sh, up in td], relu=True).permute(0, 3, 1, 2).cpu()
    assert torch.equal(alone[0], got[0])


def test_p2_format(dev):
    """The planes hold h = RNE_fp16(x 2^s), l = RNE_fp16(x 2^s - h) with 2^s a power of two that puts the image's bound in
    [2^13, 2^14): (h + l) 2^-s reproduces x to 2^-22 relative (or 2^-25 of the scaled unit for tiny values), the row keeps
    the exact maximum and 2^-s; images are scaled independently."""
    from multi_view_active_learning_amd import ops
    from multi_view_active_learning_amd.engine import P2_ROW as AMAX_ROW

    rng = np.random.default_rng(3)
    x = rng.standard_normal((3, 16, 24, 40)).astype(np.float32)
    x[1] *= 1000.0
    x[2] *= 1e-3
    xt = torch.from_numpy(x).to(dev)
    planes, rows = ops.to_p2(xt)
    rows = rows.reshape(3, AMAX_ROW)
    inv = rows[:, AMAX_ROW - 1].view(torch.float32).cpu().numpy()
    amax = rows[:, 0].view(torch.float32).cpu().numpy()
    np.testing.assert_array_equal(amax, np.abs(x).reshape(3, -1).max(1))
    for i in range(3):
        m, e = np.frexp(inv[i])
        assert m == 0.5, "the scale is a power of two"
        assert 2.0**13 <= amax[i] / inv[i] < 2.0**14
    back = ops.from_p2(planes, rows.reshape(-1), 3, 16, 24, 40).cpu().numpy()
    err = np.abs(back - x)
    assert np.all(err <= np.maximum(np.abs(x) * 2.0**-22, inv[:, None, None, None] * 2.0**-25))
    h = planes.view(torch.float16).reshape(3, 2, 5, 16, 24, 8)[:, 0].float().cpu().numpy()  # [n][c8][H][W][8] with C = 40
    assert np.isfinite(h).all() and np.abs(h).max() < 2.0**14


def test_p2_bound_holds(dev):
    """The output scale comes from a bound computed BEFORE the image's maximum exists: A max|x| + B + max|r1| + max|r2|.
    Adversarial case: all-positive weights and inputs (no cancellation: the conv reaches its L1 bound)."""
    from multi_view_active_learning_amd import ops
    from multi_view_active_learning_amd.engine import P2_ROW as AMAX_ROW

    n, c, h, w = 2, 64, 16, 16
    x = torch.full((n, h, w, c), 3.0, device=dev)
    wt = torch.full((c, c, 3, 3), 0.25, device=dev)
    one, sh = torch.full((c,), 2.0, device=dev), torch.full((c,), 5.0, device=dev)
    res = torch.full((n, h, w, c), 7.0, device=dev)
    y = ops.fused_conv_p2(x, wt, one, sh, relu=True, res1=res).cpu()
    want = 3.0 * 0.25 * 9 * 64 * 2.0 + 5.0 + 7.0  # interior pixels: every tap inside the image
    assert float(y.max()) == pytest.approx(want, rel=1e-6)
    c_ = ops.fused_conv_p2.last
    planes = c_.arena[c_.out_off : c_.out_off + y.numel()].view(torch.float16).float()
    assert torch.isfinite(planes).all()
    inv = c_.out_rows()[:, AMAX_ROW - 1].view(torch.float32).cpu()
    assert float((want / inv).max()) < 2.0**14, "the bound put the true maximum below the top of the fp16 range"


def test_p2_results_do_not_depend_on_the_batch(dev):
    from multi_view_active_learning_amd import ops

    case = (4, 64, 64, 32, 32, 3, 1, True, True, False, 0, False)
    x, wt, scale, shift, res1, _ = _make(case)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    run = lambda xs, rs: ops.fused_conv_p2(nhwc(xs), wt.to(dev), scale.to(dev), shift.to(dev), relu=True, res1=nhwc(rs)).cpu()
    whole, again = run(x, res1), run(x, res1)
    assert torch.equal(whole, again), "deterministic (no atomics on the data path, fixed tiling)"
    for i in (0, 3):
        assert torch.equal(run(x[i : i + 1], res1[i : i + 1])[0], whole[i])


def _load(c, dev):
    m = cases.product_model(c)
    sd = {k: torch.from_numpy(v) for k, v in cases.model_state_dict(c).items()}
    m.load_state_dict(sd, strict=True)
    return m.to(dev).eval(), sd


def test_p2_plan_structure(dev, monkeypatch):
    """HRNet and (round 5) PoseResNet plans run on P2 activations end to end (stem -> format change -> P2 ops -> fp32 heat-maps) -- all
    or nothing per plan; MVAL_P2=0 keeps the h2 kernels."""
    from multi_view_active_learning_amd import engine

    monkeypatch.setenv("MVAL_CONV", "p2")
    c = cases.model_cases()["w32"]  # (256 x 256; on 64 x 64 inputs the deep branches' maps fall below the P2 tiles: h2 plan)
    m, _ = _load(c, dev)
    x = torch.from_numpy(cases.model_input(c)).to(dev)
    with torch.no_grad():
        m(x)
    plan = engine._plan_for(m, x)
    assert plan.p2
    kinds = [o.kind for o in plan.ops]
    # the two stem convs are ONE launch from the fp32 image to P2 planes (MVAL_P2_STEM=0: stem, format change, P2 conv)
    assert kinds[0] == engine.OP_STEM_P2 and plan.ops[0].in_off == -1 and engine.OP_TO_P2 not in kinds
    assert all(o.algo == engine.ALGO_MFMA_P2 for o in plan.ops)
    # layer1's four Bottlenecks are one launch each (the first one behind its downsample conv), the 32-channel BasicBlocks too
    bn = [i for i, o in enumerate(plan.ops) if o.kind == engine.OP_BNECK]
    assert len(bn) == 4 and [plan.ops[i].cin for i in bn] == [64, 256, 256, 256]
    assert plan.ops[bn[0] - 1].kind == engine.OP_CONV and plan.ops[bn[0] - 1].cout == 256 and plan.ops[bn[0]].res1_off == plan.ops[bn[0] - 1].out_off
    assert all(plan.ops[i].res1_off == plan.ops[i].in_off for i in bn[1:])
    assert sum(o.kind == engine.OP_BLOCK for o in plan.ops) == 32
    # fuse layers: the up-sampling terms of an output are one launch where there are two or three of them
    fu = [o for o in plan.ops if o.kind == engine.OP_FUSE_UP]
    assert sorted((o.cout, o.n_terms) for o in fu) == sorted([(32, 2)] * 4 + [(32, 3)] * 3 + [(64, 2)] * 2)
    # PoseResNet (round 5): stem -> max-pool (both fp32 NHWC) -> format change -> P2 ops: the first Bottleneck of layer1 fused, 1x1 stride-2
    # downsample convs as stride-1 convs over the sub-sampled view, the three transposed convs as parity launches, fp32 heat-maps out
    r = cases.model_cases()["r50"]
    mr, _ = _load(r, dev)
    xr = torch.from_numpy(cases.model_input(r)).to(dev)
    with torch.no_grad():
        mr(xr)
    assert not engine._plan_for(mr, xr).p2  # (a few images: the h2 plan, whose transposed convs are one launch each)
    monkeypatch.setenv("MVAL_P2", "force")
    with torch.no_grad():
        mr(xr)
    pr = engine._plan_for(mr, xr)
    assert pr.p2
    kr = [o.kind for o in pr.ops]
    assert kr[0] == engine.OP_CONV and pr.ops[0].in_off == -1 and kr[1] == engine.OP_MAXPOOL and kr[2] == engine.OP_TO_P2
    assert all(o.algo == engine.ALGO_MFMA_P2 for o in pr.ops[2:]) and kr.count(engine.OP_DECONV) == 3 and pr.ops[-1].out_nchw == 1
    assert sum(1 for o in pr.ops if o.kind == engine.OP_CONV and o.k == 1 and o.stride == 2) == 3
    monkeypatch.setenv("MVAL_P2", "0")
    with torch.no_grad():
        mr(xr)
    assert not engine._plan_for(mr, xr).p2


@pytest.mark.parametrize("case", [(2, 256, 256, 16, 12, True), (1, 2048, 256, 8, 6, True), (3, 64, 48, 32, 24, False), (2, 32, 32, 8, 16, True), (2, 256, 256, 32, 24, True)],
                         ids=lambda c: "n%d_c%d-%d_%dx%d_r%d" % tuple(int(v) for v in c))
def test_p2_transposed_conv_vs_float64(dev, case):
    """ConvTranspose2d(k4, s2, p1) + BN (+ ReLU) over P2 planes (pose_resnet.py:107-137; csrc/net.hip: four 2 x 2 parity convs over the input
    grid on conv_p2_kernel<2, ...>, each scattered to its parity of the output planes) against float64 torch, not less accurate than the
    h2 parity kernels on the same problem; the four launches leave ONE scale and the exact per-image maximum in the shared rows."""
    from multi_view_active_learning_amd import ops

    n, cin, cout, h, w, relu = case
    rng = np.random.default_rng(31 + cin + h)
    x = torch.from_numpy(np.maximum(rng.standard_normal((n, cin, h, w)), 0).astype(np.float32) * 1.5)
    wt = torch.from_numpy((rng.standard_normal((cin, cout, 4, 4)) * np.sqrt(2.0 / (cin * 4))).astype(np.float32))
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    sh = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1)
    d = torch.float64
    want = F.conv_transpose2d(x.to(d), wt.to(d), stride=2, padding=1) * sc.to(d)[None, :, None, None] + sh.to(d)[None, :, None, None]
    if relu:
        want = torch.relu(want)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    got = ops.fused_conv_p2(nhwc(x), wt.to(dev), sc.to(dev), sh.to(dev), relu=bool(relu), transposed=True)
    kept = ops.fused_conv_p2.last.kept_amax().cpu()
    got = got.permute(0, 3, 1, 2).cpu()
    assert got.shape == want.shape
    np.testing.assert_allclose(got.numpy(), want.float().numpy(), rtol=1e-4, atol=3e-5)
    assert torch.allclose(kept, got.abs().amax(dim=(1, 2, 3)), rtol=2.0**-21, atol=0)
    if cin % 32 == 0:
        y = ops.fused_conv(nhwc(x), wt.to(dev), sc.to(dev), sh.to(dev), stride=2, pad=1, relu=bool(relu), algo=ops.ALGO_MFMA, kind=ops.OP_DECONV).permute(0, 3, 1, 2).cpu()
        rms = lambda v: (v.double() - want).pow(2).mean().sqrt().item()
        assert rms(got) <= 1.25 * rms(y) + 1e-8, (rms(got), rms(y))


@pytest.mark.parametrize("case", [(2, 256, 512, 64, 48), (3, 512, 1024, 32, 24), (2, 64, 128, 17, 24), (1, 1024, 2048, 16, 12)], ids=lambda c: "n%d_c%d-%d_%dx%d" % c)
def test_p2_conv1x1_stride2_vs_float64(dev, case):
    """A 1 x 1 stride-2 conv (pose_resnet.py's downsample branches) over P2 planes: the stride-1 kernel over every second input pixel."""
    from multi_view_active_learning_amd import ops

    n, cin, cout, h, w = case
    rng = np.random.default_rng(5 + cin)
    x = torch.from_numpy(rng.standard_normal((n, cin, h, w)).astype(np.float32))
    wt = torch.from_numpy((rng.standard_normal((cout, cin, 1, 1)) * np.sqrt(2.0 / cin)).astype(np.float32))
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    sh = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1)
    d = torch.float64
    want = F.conv2d(x.to(d), wt.to(d), stride=2) * sc.to(d)[None, :, None, None] + sh.to(d)[None, :, None, None]
    got = ops.fused_conv_p2(x.permute(0, 2, 3, 1).contiguous().to(dev), wt.to(dev), sc.to(dev), sh.to(dev), stride=2).permute(0, 3, 1, 2).cpu()
    assert got.shape == want.shape
    np.testing.assert_allclose(got.numpy(), want.float().numpy(), rtol=1e-4, atol=3e-5)


def _report(name, obj):
    """Counts the review wants reproducible: written under gpurun_out/ (merged back from the GPU box), copied into profiles/rNN/."""
    import json

    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, name), "w") as f:
        json.dump(obj, f, indent=1)


def test_p2_argmax_census_vs_exact_fp32(dev, monkeypatch):
    """BASELINE config C2 decode parity over 256 frames x 4 views: the P2 plan against the exact-fp32 MFMA plan
    (MVAL_CONV=fp32).  Counts the (view, joint) maps whose arg-max differs and the top-2 margins involved: no map whose
    margin exceeds twice the heat-map tolerance may flip."""
    from multi_view_active_learning_amd import synth

    c = dict(arch="hrnet_w32", seed=0, n=128, h=256, w=256, j=19)
    m, _ = _load(c, dev)
    frames, v = 256, 4
    flips, flips_above, maps, worst_err, min_margin_agree = 0, 0, 0, 0.0, float("inf")
    with torch.no_grad():
        for b in range(frames * v // 128):
            x = torch.from_numpy(synth.images(500 + b, 32, v, 256, 256)).reshape(128, 3, 256, 256).to(dev)
            monkeypatch.setenv("MVAL_CONV", "p2")
            y = m(x)
            monkeypatch.setenv("MVAL_CONV", "fp32")
            z = m(x)
            tol = 2e-4 * float(z.abs().max())
            worst_err = max(worst_err, float((y - z).abs().max()))
            fy, fz = y.reshape(128, 19, -1), z.reshape(128, 19, -1)
            top2 = torch.topk(fz, 2, dim=-1).values
            margin = top2[..., 0] - top2[..., 1]
            differ = fy.argmax(-1) != fz.argmax(-1)
            flips += int(differ.sum())
            flips_above += int((differ & (margin > 2 * tol)).sum())
            maps += differ.numel()
            if (~differ).any():
                min_margin_agree = min(min_margin_agree, float(margin[~differ].min()))
            assert worst_err <= tol, (worst_err, tol)
    print(f"\narg-max census, P2 vs exact-fp32 plans: {maps} maps, {flips} flips ({flips_above} above margin 2*tol), "
          f"max |heat-map difference| {worst_err:.2e}, smallest top-2 margin among agreeing maps {min_margin_agree:.2e}")
    _report("census_w32_p2_vs_fp32.json", dict(maps=maps, flips=flips, flips_above_margin=flips_above, max_abs_heatmap_difference=worst_err,
                                               smallest_margin_among_agreeing=min_margin_agree, frames=frames, views=v))
    assert maps == 19456 and flips_above == 0
    assert flips == 0  # (pinned: DESIGN 7.0a's "0 flips" is this count, not only the margin-exempt one)


def test_p2_nan_propagates_like_the_other_plans(dev, monkeypatch):
    """A NaN in one image reaches that image's heat-maps in the P2 plan as it does in torch and in the h2 / fp32 plans (the P2
    epilogues clamped with fmaxf, which returns the OTHER operand for a NaN: a diverged model looked finite);
    the other images of the batch are untouched."""
    c = cases.model_cases()["w32"]
    m, _ = _load(c, dev)
    x = torch.from_numpy(cases.model_input(c)).to(dev)
    assert x.shape[0] >= 2
    clean = {}
    for mode in ("p2", "h2"):
        monkeypatch.setenv("MVAL_CONV", mode)
        with torch.no_grad():
            clean[mode] = m(x).clone()
    xn = x.clone()
    xn[0, 1, 100, 80] = float("nan")
    for mode in ("p2", "h2"):
        monkeypatch.setenv("MVAL_CONV", mode)
        with torch.no_grad():
            y = m(xn)
        assert torch.isnan(y[0]).any(), mode
        assert torch.equal(y[1:], clean[mode][1:]), mode  # per-image scales: the NaN stays in its image


def test_p2_bound_slack_check_and_h2_fallback(dev, monkeypatch):
    """P2 scales come from a-priori bounds that compound inside fused operators; parameters with a wide
    per-channel spread can put the bound 2^12+ above the activations, where the fp16 pair loses precision.  The plan measures
    bound / actual maximum on the first forward after every parameter change (p2_slack) and run_network hands over to the
    h2 plan (exact per-image scales) above 2^15 (engine.P2_MAX_SLACK_LOG2)."""
    import warnings

    from multi_view_active_learning_amd import engine

    monkeypatch.setenv("MVAL_CONV", "p2")
    c = cases.model_cases()["w32"]
    m, sd = _load(c, dev)
    x = torch.from_numpy(cases.model_input(c)).to(dev)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")  # synthetic variance-preserving weights: no fallback
        y0 = m(x)
    plan = engine._plan_for(m, x)
    assert plan.p2 and plan.p2_slack is not None and 8.0 < plan.p2_slack < 13.5, plan.p2_slack  # (the fused Bottlenecks' chained bounds)
    # trained-like spread: one BatchNorm channel with a huge gain whose output the next conv ignores -- the activations stay
    # as they were, the bound A * max|x| + B of every operator downstream of that channel explodes
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["stage2.0.branches.0.0.bn1.weight"][0] *= 2.0**18
    sd2["stage2.0.branches.0.0.bn1.bias"][0] = 2.0**17
    sd2["stage2.0.branches.0.0.conv2.weight"][:, 0] = 0.0
    m.load_state_dict(sd2, strict=True)
    monkeypatch.setenv("MVAL_CONV", "h2")
    with torch.no_grad():
        want = m(x).clone()
    monkeypatch.setenv("MVAL_CONV", "p2")
    with torch.no_grad(), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = m(x)
        again = m(x)  # (second call: straight to the h2 plan, no second warning)
    assert any("h2 kernels" in str(i.message) for i in w) and sum("h2 kernels" in str(i.message) for i in w) == 1
    assert torch.equal(got, want) and torch.equal(again, want)
    # MVAL_P2=force keeps the P2 plan (measurement): finite, but it may be less accurate than the tolerance
    # new parameters: P2 again
    m.load_state_dict(sd, strict=True)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")
        y1 = m(x)
    assert torch.equal(y1, y0)
