"""Adam with the update of every parameter in ONE launch.

The reference builds ``torch.optim.Adam([{"params": pose_estimator.parameters(), "lr": LR}])``
(``/root/reference/strategy.py:405-407``), steps it once per batch (``:479``), drives its learning rate with a ``StepLR``
(``:408-410``, ``:511``) and stores ``optimizer.state_dict()`` in its checkpoints (``:703``).  :class:`Adam` below is a
subclass of ``torch.optim.Adam`` with the same constructor, ``param_groups`` and state layout (per parameter ``step``,
``exp_avg``, ``exp_avg_sq``), so schedulers, ``state_dict()`` / ``load_state_dict()`` and ``zero_grad()`` are torch's own;
only ``step()`` differs: one ``mval_adam_step`` launch (``csrc/optim.hip``) over a table of (parameter, gradient, exp_avg,
exp_avg_sq) pointers instead of torch's five passes of multi-tensor launches over ~300 small tensors (1.8 ms of a 70 ms
HRNet-W32 training step; 0.2 ms here).  The arithmetic is ``torch/optim/adam.py``'s ``_single_tensor_adam`` in float32.

Configurations the kernel does not implement (amsgrad, maximize, capturable, differentiable, tensor learning rates,
decoupled weight decay, sparse / non-float32 / non-contiguous / CPU tensors) take ``torch.optim.Adam.step`` unchanged.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

_JOB = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8")])  # = mval_adam_job (include/mval_hip.h)


def _torch_step(opt):
    # torch.optim.Adam.step without the profiling / hook wrapper Optimizer.__init__ may have put around it (this class's own step
    # carries that wrapper already: the hooks must not run twice)
    fn = torch.optim.Adam.step
    return getattr(fn, "__wrapped__", fn)(opt)


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)
        self._mval_cache = {}

    # ---- which configurations the kernel covers --------------------------------------------------------------------
    @staticmethod
    def _group_native(group, ps):
        if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable"):
            return False
        if group.get("decoupled_weight_decay") or group.get("fused") or group.get("foreach") is False:
            # (foreach=False / fused=True are explicit requests for torch's own implementations)
            return False
        if torch.is_tensor(group["lr"]) or any(torch.is_tensor(b) for b in group["betas"]):
            return False
        dev = ps[0].device
        if dev.type != "cuda":
            return False
        for p in ps:
            g = p.grad
            if (p.device != dev or p.dtype != torch.float32 or not p.is_contiguous() or g.is_sparse or g.dtype != torch.float32
                    or g.device != dev or not g.is_contiguous()):
                return False
        return True

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._mval_cache = {}

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._mval_cache = {}

    # ---- state in torch's layout, the moments of a group's parameters as views of two flat buffers -----------------------
    def _init_missing_state(self, ps):
        new = [p for p in ps if len(self.state[p]) == 0]
        if not new:
            return
        offs, total = [], 0
        for p in new:
            offs.append(total)
            total += (p.numel() + 3) & ~3  # (16-byte aligned views: the kernel's vector path)
        m = torch.zeros(total, dtype=torch.float32, device=new[0].device)
        v = torch.zeros(total, dtype=torch.float32, device=new[0].device)
        for p, o in zip(new, offs):
            st = self.state[p]
            st["step"] = torch.tensor(0.0, dtype=torch.float32)  # (torch's default: a float32 scalar on the host)
            st["exp_avg"] = m[o : o + p.numel()].view_as(p)
            st["exp_avg_sq"] = v[o : o + p.numel()].view_as(p)

    def _table(self, gi, ps):
        """(jobs, first_block, n_jobs, total_blocks, step count) of group gi: rebuilt when a pointer moved."""
        states = [self.state[p] for p in ps]
        key = (tuple(p.data_ptr() for p in ps) + tuple(p.grad.data_ptr() for p in ps) + tuple(s["exp_avg"].data_ptr() for s in states)
               + tuple(s["exp_avg_sq"].data_ptr() for s in states))
        c = self._mval_cache.get(gi)
        # (the cached step count is re-validated against the state every step: a caller that reset or edited state[p]["step"], or
        # swapped a moment tensor, without load_state_dict gets a rebuilt table, not a stale pointer / bias correction -- ADVICE round 4;
        # the steps are host scalars, so reading two of them costs no synchronisation)
        if (c is not None and c["key"] == key and all(a is b for a, b in zip(c["steps"], (s["step"] for s in states)))
                and float(states[0]["step"]) == c["t"] and float(states[-1]["step"]) == c["t"]):
            return c
        for s in states:
            if (s["exp_avg"].dtype != torch.float32 or not s["exp_avg"].is_contiguous() or not s["exp_avg_sq"].is_contiguous()
                    or s["exp_avg"].device != ps[0].device or s["exp_avg_sq"].device != ps[0].device):
                return None
        steps = {float(s["step"]) for s in states}
        if len(steps) != 1:
            return None  # (parameters that skipped steps: torch's implementation keeps their bias corrections apart)
        tab = np.zeros(len(ps), _JOB)
        tab["p"] = [p.data_ptr() for p in ps]
        tab["g"] = [p.grad.data_ptr() for p in ps]
        tab["m"] = [s["exp_avg"].data_ptr() for s in states]
        tab["v"] = [s["exp_avg_sq"].data_ptr() for s in states]
        tab["n"] = [p.numel() for p in ps]
        be = int(_lib.lib().mval_adam_block_elems())
        blocks = (tab["n"] + be - 1) // be
        first = np.concatenate([[0], np.cumsum(blocks)[:-1]]).astype(np.int32)
        dev = ps[0].device
        c = {
            "key": key,
            # (pinned staging + asynchronous copies: a pageable upload would drain the stream each time a gradient buffer moves)
            "jobs": torch.from_numpy(tab.view(np.uint8).copy()).pin_memory().to(dev, non_blocking=True),
            "first": torch.from_numpy(first).pin_memory().to(dev, non_blocking=True),
            "n": len(ps),
            "blocks": int(blocks.sum()),
            "t": int(steps.pop()),
            "steps": [s["step"] for s in states],
        }
        self._mval_cache[gi] = c
        self.table_builds = getattr(self, "table_builds", 0) + 1
        return c

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        work = []
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if ps:
                work.append((gi, group, ps))
        if not work:
            return loss
        if not all(self._group_native(group, ps) for _, group, ps in work):
            _torch_step(self)
            self._mval_cache = {}
            return loss
        tables = []
        for gi, group, ps in work:
            self._init_missing_state(ps)
            c = self._table(gi, ps)
            if c is None:
                _torch_step(self)
                self._mval_cache = {}
                return loss
            tables.append(c)
        lib = _lib.lib()
        for (gi, group, ps), c in zip(work, tables):
            with torch.cuda.device(ps[0].device):  # (the launch goes to the CURRENT device's stream: make that the parameters' device)
                torch._foreach_add_(c["steps"], 1)
                c["t"] += 1
                t = c["t"]
                beta1, beta2 = group["betas"]
                bias_correction1 = 1 - beta1**t
                bias_correction2 = 1 - beta2**t
                step_size = group["lr"] / bias_correction1
                _lib._check(
                    lib.mval_adam_step(C.c_void_p(c["jobs"].data_ptr()), C.c_void_p(c["first"].data_ptr()), C.c_int(c["n"]), C.c_int(c["blocks"]),
                                       C.c_float(1 - beta1), C.c_float(beta2), C.c_float(1 - beta2), C.c_float(group["eps"]),
                                       C.c_float(group["weight_decay"]), C.c_float(step_size), C.c_float(bias_correction2**0.5), _lib._stream()),
                    "mval_adam_step")
                # the kernel wrote the parameters and moments through raw pointers: tell autograd (saved-tensor checks) and everything keyed on
                # a parameter's version -- the inference / training plans re-pack their weights when it changes -- that they were modified
                torch.autograd.graph.increment_version(ps)
                torch.autograd.graph.increment_version([t for st in (self.state[p] for p in ps) for t in (st["exp_avg"], st["exp_avg_sq"])])
        return loss
