"""Fused Adam over one flat parameter buffer per optimizer.

The reference builds two ``torch.optim.Adam`` objects (<arch>/main.py:34-44: G = encoder+decoder,
lr 1e-4; D = discriminator, lr/2; betas (0.5, 0.999), eps 1e-8, no weight decay).  Here every
parameter of an optimizer becomes a view into ONE contiguous f32 buffer and so does its gradient:
the update is a single bandwidth-bound HIP kernel (28 B/param) and the data-parallel gradient
exchange is an all-reduce over contiguous slices of that buffer with no packing copies.
"""
from __future__ import annotations

from typing import Iterable, List

import torch

from . import lib as L
from . import ops


class FlatAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdam got an empty parameter list")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam needs parameters on the GPU (move the model first); no CPU fallback")
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.step_count = 0
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self.numel = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + n].view(p.shape)
            p.grad = self.flat_g[o:o + n].view(p.shape)
        # packed-weight caches (ops.PackedWeights) of these parameters are keyed on this cell: bumped by step()
        self._epoch = ops.EpochCell(ops.PARAM_EPOCH[0] + 1)
        for p in self.params:
            p._vg_epoch = self._epoch
        ops.PARAM_EPOCH[0] += 1

    # torch.optim surface used by the training loop
    def zero_grad(self, set_to_none: bool = False):
        self.flat_g.zero_()
        base_p, base_g = self.flat_p.data_ptr(), self.flat_g.data_ptr()
        for p, o in zip(self.params, self.offsets):
            if p.data_ptr() != base_p + 4 * o:
                # someone rebound p.data (model.to(), .float(), a manual assignment): the forward would read that
                # storage while step() updates flat_p.  Take the new values and alias again.
                if p.device != self.flat_p.device or p.dtype != torch.float32:
                    raise RuntimeError("FlatAdam: a parameter left the optimizer's device/dtype; rebuild the optimizers")
                self.flat_p[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = self.flat_p[o:o + p.numel()].view(p.shape)
                self._epoch[0] += 1
            if p.grad is None or p.grad.data_ptr() != base_g + 4 * o:   # re-attach if someone set .grad = None
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)

    def step(self, grad_scale: float = 1.0):
        self.step_count += 1
        L.check(L.load().vg_adam_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(),
                                      self.exp_avg_sq.data_ptr(), self.numel, self.lr, self.betas[0], self.betas[1],
                                      self.eps, self.step_count, grad_scale, L.stream()), "adam_step")
        self._epoch[0] += 1
        ops.repack_all(self._epoch)         # every packed weight image of these parameters, one launch

    @property
    def param_groups(self):
        return [{"params": self.params, "lr": self.lr, "betas": self.betas, "eps": self.eps}]

    def set_lr(self, lr: float):
        self.lr = float(lr)

    def state_dict(self):
        """torch.optim.Adam-compatible layout ({'state': {i: {step, exp_avg, exp_avg_sq}}, 'param_groups': [...]})."""
        state = {}
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        return {"state": state, "param_groups": [{"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0,
                                                  "amsgrad": False, "params": list(range(len(self.params)))}]}

    def load_state_dict(self, sd):
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            st = sd["state"].get(i)
            if st is None:
                continue
            n = p.numel()
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = int(st["step"])
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = float(g["lr"]), tuple(g["betas"]), float(g["eps"])


class StepLR:
    """torch.optim.lr_scheduler.StepLR(step_size=30, gamma=0.5) as used by <arch>/main.py:46-48."""

    def __init__(self, optimizer: FlatAdam, step_size: int, gamma: float = 0.1):
        self.opt, self.step_size, self.gamma = optimizer, step_size, gamma
        self.base_lr, self.last_epoch = optimizer.lr, 0

    def step(self):
        self.last_epoch += 1
        self.opt.set_lr(self.base_lr * self.gamma ** (self.last_epoch // self.step_size))

    def state_dict(self):
        return {"step_size": self.step_size, "gamma": self.gamma, "base_lrs": [self.base_lr], "last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.last_epoch = sd["last_epoch"]
        self.base_lr = sd["base_lrs"][0]
        self.opt.set_lr(self.base_lr * self.gamma ** (self.last_epoch // self.step_size))
