"""Layer-graph IR for the heat-map networks.

The reference expresses HRNet / PoseResNet as nested ``nn.Module`` classes whose
``forward`` issues ~900 tiny framework ops per image batch.  Here a
network is a flat, topologically ordered list of *fused* operators over symbolic
activation tensors; the same list drives

* the parameter tree (every op names the reference ``state_dict`` prefixes it reads,
  so checkpoints stay interchangeable: SURVEY Appendix B.4),
* the HIP inference engine (one launch per op: conv + BN + residual(s) + ReLU
  [+ nearest-upsample / NCHW store] fused in the epilogue), and
* the training executor (same ops, train-mode BatchNorm, autograd).

Activation layout inside the graph is NHWC fp32; the graph input is the caller's
NCHW image batch and the graph output is NCHW heat-maps, exactly the reference's
tensor contract (pose_estimators/hrnet.py:468-501, pose_resnet.py:139-153).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional


@dataclass
class Act:
    """Symbolic activation: ``channels`` at 1/``down`` of the input resolution."""

    id: int
    channels: int
    down: int
    layout: str = "nhwc"  # "nchw" only for the graph input / output


@dataclass
class Op:
    kind: str  # conv | deconv | maxpool
    src: int
    dst: int
    cin: int
    cout: int
    k: int = 1
    stride: int = 1
    pad: int = 0
    conv: Optional[str] = None  # state_dict prefix of the (de)conv weight
    bn: Optional[str] = None  # state_dict prefix of the BatchNorm that follows
    bias: bool = False
    relu: bool = False
    res1: Optional[int] = None  # out = act(((bn(conv) + res1) + res2)), left to right
    res2: Optional[int] = None
    up: int = 0  # log2 of the nearest-neighbour upsample fused into the store
    # scheduling hints: ops of one phase that sit on different lanes are independent (HRNet
    # branches / fuse outputs) and may run concurrently on separate HIP streams; phases join
    phase: int = 0
    lane: int = 0


@dataclass
class Graph:
    acts: List[Act] = field(default_factory=list)
    ops: List[Op] = field(default_factory=list)
    input: int = 0
    output: int = 0
    cur_phase: int = 0
    cur_lane: int = 0

    def new_phase(self, lane: int = 0):
        self.cur_phase += 1
        self.cur_lane = lane

    def act(self, channels: int, down: int, layout: str = "nhwc") -> int:
        self.acts.append(Act(len(self.acts), channels, down, layout))
        return len(self.acts) - 1

    def conv(self, src, cout, k, stride, conv, bn=None, relu=False, res1=None, res2=None, up=0, bias=False, layout="nhwc"):
        a = self.acts[src]
        down = a.down * stride
        if up:
            down //= 1 << up
        dst = self.act(cout, down, layout)
        self.ops.append(
            Op("conv", src, dst, a.channels, cout, k, stride, k // 2, conv, bn, bias, relu, res1, res2, up,
               self.cur_phase, self.cur_lane)
        )
        return dst

    def param_shapes(self):
        """Ordered {state_dict key: shape} in the reference's registration order is
        not needed for loading (keys match by name); order here is execution order."""
        shapes = {}
        for op in self.ops:
            if op.kind == "conv":
                shapes[op.conv + ".weight"] = (op.cout, op.cin, op.k, op.k)
                if op.bias:
                    shapes[op.conv + ".bias"] = (op.cout,)
            elif op.kind == "deconv":
                shapes[op.conv + ".weight"] = (op.cin, op.cout, op.k, op.k)
            if op.bn:
                for leaf in ("weight", "bias", "running_mean", "running_var"):
                    shapes[f"{op.bn}.{leaf}"] = (op.cout,)
                shapes[f"{op.bn}.num_batches_tracked"] = ()
        return shapes


# --------------------------------------------------------------------------
# HRNet  (reference pose_estimators/hrnet.py)
# --------------------------------------------------------------------------
def _basic_block(g: Graph, x: int, prefix: str) -> int:
    """hrnet.py:36-52: relu(bn2(conv2(relu(bn1(conv1 x)))) + x)."""
    c = g.acts[x].channels
    y = g.conv(x, c, 3, 1, prefix + ".conv1", prefix + ".bn1", relu=True)
    return g.conv(y, c, 3, 1, prefix + ".conv2", prefix + ".bn2", relu=True, res1=x)


def _bottleneck(g: Graph, x: int, prefix: str, planes: int, stride: int = 1, downsample: bool = False) -> int:
    """hrnet.py:75-95 / pose_resnet.py:211-231 (stride lives on the 3x3)."""
    y = g.conv(x, planes, 1, 1, prefix + ".conv1", prefix + ".bn1", relu=True)
    y = g.conv(y, planes, 3, stride, prefix + ".conv2", prefix + ".bn2", relu=True)
    skip = x
    if downsample:
        skip = g.conv(x, planes * 4, 1, stride, prefix + ".downsample.0", prefix + ".downsample.1")
    return g.conv(y, planes * 4, 1, 1, prefix + ".conv3", prefix + ".bn3", relu=True, res1=skip)


def _hr_module(g: Graph, xs: List[int], prefix: str, blocks: int, n_out: int) -> List[int]:
    """hrnet.py:269-287.  The fuse sum  y_i = relu(sum_j f_ij(x_j))  is evaluated in
    the reference's left-to-right order (fp32 addition is not associative) by chaining
    the partial sum through the ``res1``/``res2`` epilogue inputs of the conv that
    produces each term:  out = act(((bn(conv) + res1) + res2)).  The identity term
    (j == i) is folded into the epilogue of the conv term just before it, so no
    stand-alone add/ReLU kernel exists."""
    nb = len(xs)
    xs = list(xs)
    g.new_phase()
    for b in range(nb):
        g.cur_lane = b  # branches are independent until the fuse
        for k in range(blocks):
            xs[b] = _basic_block(g, xs[b], f"{prefix}.branches.{b}.{k}")
    if nb == 1:
        return xs
    outs = []
    g.new_phase()
    for i in range(n_out):
        g.cur_lane = i  # each fused output is its own chain
        ci = g.acts[xs[i]].channels
        acc = None  # activation holding the partial sum so far
        j = 0
        while j < nb:
            if j == i:  # only reachable for i == 0:  y = x[0]
                acc = xs[i]
                j += 1
                continue
            fold_identity = j + 1 == i  # "y = y + x[i]" comes right after this term
            r1, r2 = acc, None
            if fold_identity:
                if r1 is None:
                    r1 = xs[i]  # y = f(x_j) + x_i  (two-operand fp add commutes)
                else:
                    r2 = xs[i]  # y = (y + f(x_j)) + x_i
            nxt = j + 2 if fold_identity else j + 1
            last = nxt >= nb
            if j > i:  # 1x1 conv + BN at the low resolution, nearest-upsampled on store
                q = f"{prefix}.fuse_layers.{i}.{j}"
                acc = g.conv(xs[j], ci, 1, 1, q + ".0", q + ".1", relu=last, res1=r1, res2=r2, up=j - i)
            else:  # chain of (i - j) stride-2 3x3 convs; only the last one joins the sum
                t = xs[j]
                cj = g.acts[xs[j]].channels
                for k in range(i - j):
                    q = f"{prefix}.fuse_layers.{i}.{j}.{k}"
                    if k != i - j - 1:
                        t = g.conv(t, cj, 3, 2, q + ".0", q + ".1", relu=True)
                    else:
                        acc = g.conv(t, ci, 3, 2, q + ".0", q + ".1", relu=last, res1=r1, res2=r2)
            j = nxt
        outs.append(acc)
    return outs


def build_hrnet(num_joints: int, hrnet_cfg) -> Graph:
    """Graph of PoseHighResolutionNet (hrnet.py:293-350, 468-501)."""
    g = Graph()
    g.input = g.act(3, 1, "nchw")
    x = g.conv(g.input, 64, 3, 2, "conv1", "bn1", relu=True)
    x = g.conv(x, 64, 3, 2, "conv2", "bn2", relu=True)
    for k in range(4):
        x = _bottleneck(g, x, f"layer1.{k}", 64, 1, downsample=(k == 0))
    ys = [x]
    stages = (hrnet_cfg.STAGE2, hrnet_cfg.STAGE3, hrnet_cfg.STAGE4)
    for s, st in enumerate(stages):
        if st.BLOCK != "BASIC" or st.FUSE_METHOD != "SUM":
            raise NotImplementedError("only BASIC blocks with SUM fusion (the reference's shipped config)")
        chans = list(st.NUM_CHANNELS)
        nb = st.NUM_BRANCHES
        t = f"transition{s + 1}"
        xs = []
        g.new_phase()
        for i in range(nb):  # hrnet.py:370-413
            g.cur_lane = i
            if i < len(ys):
                if g.acts[ys[i]].channels != chans[i]:
                    xs.append(g.conv(ys[i], chans[i], 3, 1, f"{t}.{i}.0", f"{t}.{i}.1", relu=True))
                else:
                    xs.append(ys[i])
            else:
                z = ys[-1]
                steps = i + 1 - len(ys)
                for j in range(steps):
                    co = chans[i] if j == steps - 1 else g.acts[ys[-1]].channels
                    z = g.conv(z, co, 3, 2, f"{t}.{i}.{j}.0", f"{t}.{i}.{j}.1", relu=True)
                xs.append(z)
        nblk = st.NUM_BLOCKS
        if len(set(nblk)) != 1:
            raise NotImplementedError("per-branch block counts must be equal")
        for m in range(st.NUM_MODULES):
            last = (s == len(stages) - 1) and (m == st.NUM_MODULES - 1)
            xs = _hr_module(g, xs, f"stage{s + 2}.{m}", nblk[0], 1 if last else nb)
        ys = xs
    k = hrnet_cfg.FINAL_CONV_KERNEL
    g.new_phase()
    g.output = g.conv(ys[0], num_joints, k, 1, "final_layer", None, bias=True, layout="nchw")
    return g


# --------------------------------------------------------------------------
# PoseResNet  (reference pose_estimators/pose_resnet.py)
# --------------------------------------------------------------------------
_RESNET_LAYERS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


def build_pose_resnet(num_joints: int, num_layers: int = 50) -> Graph:
    """Graph of PoseResNet (pose_resnet.py:17-67,139-153).  Depths 18/34 are
    unusable in the reference too (its BasicBlock has no ``expansion``;
    SURVEY 2 row 2) and are rejected here."""
    if num_layers not in _RESNET_LAYERS:
        raise NotImplementedError("PoseResNet depth must be one of 50/101/152")
    g = Graph()
    g.input = g.act(3, 1, "nchw")
    x = g.act(64, 2)
    g.ops.append(Op("conv", g.input, x, 3, 64, 7, 2, 3, "conv1", "bn1", relu=True))
    p = g.act(64, 4)
    g.ops.append(Op("maxpool", x, p, 64, 64, 3, 2, 1))
    x = p
    for li, n in enumerate(_RESNET_LAYERS[num_layers]):
        planes = 64 << li
        for k in range(n):
            stride = 2 if (li > 0 and k == 0) else 1
            x = _bottleneck(g, x, f"layer{li + 1}.{k}", planes, stride, downsample=(k == 0))
    for d in range(3):
        a = g.acts[x]
        y = g.act(256, a.down // 2)
        g.ops.append(
            Op("deconv", x, y, a.channels, 256, 4, 2, 1, f"deconv_layers.{3 * d}", f"deconv_layers.{3 * d + 1}", relu=True)
        )
        x = y
    g.output = g.conv(x, num_joints, 1, 1, "final_layer", None, bias=True, layout="nchw")
    return g
