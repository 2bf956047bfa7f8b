"""Parameter tree: ``nn.Parameter``/buffer holders laid out so that ``state_dict()``
produces exactly the reference's key names and shapes, built from
the layer graph instead of from nested layer classes.  The holders have no ``forward``:
compute is done by the HIP engine reading these tensors."""
from __future__ import annotations

import math

import torch
from torch import nn


class ConvWeights(nn.Module):
    """weight (Cout, Cin, k, k) [+ bias (Cout,)] -- nn.Conv2d's parameter layout."""

    def __init__(self, cout, cin, k, bias):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None

    def extra_repr(self):
        return "x".join(map(str, self.weight.shape)) + (", bias" if self.bias is not None else "")


class DeconvWeights(nn.Module):
    """weight (Cin, Cout, k, k) -- nn.ConvTranspose2d's parameter layout, no bias
    (pose_resnet.py:30 ``deconv_with_bias = False``)."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cin, cout, k, k))

    def extra_repr(self):
        return "x".join(map(str, self.weight.shape))


class BatchNormStats(nn.Module):
    """weight, bias, running_mean, running_var, num_batches_tracked -- nn.BatchNorm2d's
    state; momentum 0.1, eps 1e-5 everywhere in the reference."""

    momentum = 0.1
    eps = 1e-5

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _Scope(nn.Module):
    """Pure namespace node ('stage2', '0', 'branches', ...)."""


def _attach(root: nn.Module, path: str, leaf: nn.Module):
    parts = path.split(".")
    node = root
    for p in parts[:-1]:
        child = node._modules.get(p)
        if child is None:
            child = _Scope()
            node.add_module(p, child)
        node = child
    node.add_module(parts[-1], leaf)


def attach_parameters(root: nn.Module, graph) -> dict:
    """Create holders for every op of ``graph`` under ``root``; returns
    {prefix: holder module} for the engine."""
    holders = {}
    for op in graph.ops:
        if op.kind == "conv":
            h = ConvWeights(op.cout, op.cin, op.k, op.bias)
        elif op.kind == "deconv":
            h = DeconvWeights(op.cin, op.cout, op.k)
        else:
            continue
        _attach(root, op.conv, h)
        holders[op.conv] = h
        if op.bn:
            b = BatchNormStats(op.cout)
            _attach(root, op.bn, b)
            holders[op.bn] = b
    # ops are listed in execution order (the projection shortcut runs before conv3); the reference registers a
    # block's ``downsample`` last (hrnet.py:58-73, pose_resnet.py:195-209), and ``parameters()`` order is what an
    # optimizer state_dict is indexed by -- keep it identical so checkpoints carry over in both directions
    for scope in root.modules():
        if isinstance(scope, _Scope) and "downsample" in scope._modules and "conv3" in scope._modules:
            scope._modules["downsample"] = scope._modules.pop("downsample")
    return holders


def init_normal_(holders: dict, std: float = 0.001, only_prefix: tuple = ()):
    """The reference's default init: N(0, std) conv/deconv weights, zero bias, BN 1/0
    (hrnet.py:355-368; pose_resnet.py:48-67 restricts it to deconv + final layers)."""
    for name, h in holders.items():
        if only_prefix and not name.startswith(only_prefix):
            continue
        if isinstance(h, (ConvWeights, DeconvWeights)):
            nn.init.normal_(h.weight, std=std)
            if getattr(h, "bias", None) is not None:
                nn.init.constant_(h.bias, 0)
        elif isinstance(h, BatchNormStats):
            nn.init.constant_(h.weight, 1)
            nn.init.constant_(h.bias, 0)


def init_torch_default_(holders: dict, skip_prefix: tuple = ()):
    """nn.Conv2d's default reset_parameters (kaiming_uniform, a=sqrt(5)) for the
    PoseResNet backbone, which the reference leaves at torch defaults."""
    for name, h in holders.items():
        if skip_prefix and name.startswith(skip_prefix):
            continue
        if isinstance(h, ConvWeights):
            nn.init.kaiming_uniform_(h.weight, a=math.sqrt(5))
