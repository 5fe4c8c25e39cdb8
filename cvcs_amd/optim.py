"""Fused optimisers + scheduler behind the reference's `load_optimizer` names (source/scripts/utils.py:208-221).

One launch updates the network's whole flat f32 parameter buffer (cvcs_sgd_step / cvcs_adam_step); the same flat
gradient buffer is what data-parallel training all-reduces, so `grad_scale = 1/world_size` folds the averaging
into the update.
"""
from __future__ import annotations

import torch

from . import ops


class PolynomialLR:
    """closed form of torch.optim.lr_scheduler.PolynomialLR (stepped once per epoch, S/train.py:132-133)."""

    def __init__(self, optimizer, total_iters=5, power=1.0):
        self.optimizer, self.total_iters, self.power = optimizer, total_iters, power
        self.base_lr = optimizer.lr
        self.last_epoch = 0

    def _apply(self):
        e = min(self.last_epoch, self.total_iters)
        self.optimizer.lr = self.base_lr * (1.0 - e / self.total_iters) ** self.power

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [self.optimizer.lr]

    def state_dict(self):
        return {"total_iters": self.total_iters, "power": self.power, "base_lr": self.base_lr,
                "last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.total_iters, self.power = sd["total_iters"], sd["power"]
        self.base_lr, self.last_epoch = sd["base_lr"], sd["last_epoch"]
        self._apply()


class _FusedOptimizer:
    def __init__(self, net, lr):
        self.net, self.lr = net, lr
        self.grad_scale = 1.0        # set to 1/world_size by the data-parallel wrapper
        self.pre_step = None         # hook: wait for the gradient all-reduce
        self.steps = 0

    @property
    def param_groups(self):
        return [{"lr": self.lr, "params": list(self.net.parameters())}]

    def zero_grad(self, set_to_none: bool = True):
        """The HIP backward overwrites every gradient (S/train.py:124 zeroes them first anyway): nothing to clear."""

    def _flat(self):
        p, g = self.net.flat_parameters()
        if self.pre_step is not None:
            self.pre_step()
        return p, g


class FusedSGD(_FusedOptimizer):
    """torch.optim.SGD(lr, momentum, weight_decay) semantics (S/utils.py:211,214)."""

    def __init__(self, net, lr, momentum=0.9, weight_decay=1e-5):
        super().__init__(net, lr)
        self.momentum, self.weight_decay = momentum, weight_decay
        self.buf = None

    def step(self):
        p, g = self._flat()
        if self.buf is None or self.buf.numel() != p.numel() or self.buf.device != p.device:
            self.buf = torch.zeros_like(p)
            self.steps = 0
        ops.sgd_step(p, g, self.buf, self.lr, self.momentum, self.weight_decay, self.grad_scale, self.steps == 0)
        self.steps += 1

    def state_dict(self):
        return {"kind": "SGD", "lr": self.lr, "momentum": self.momentum, "weight_decay": self.weight_decay,
                "steps": self.steps, "momentum_buffer": None if self.buf is None else self.buf.detach().cpu()}

    def load_state_dict(self, sd):
        self.lr, self.momentum, self.weight_decay, self.steps = sd["lr"], sd["momentum"], sd["weight_decay"], sd["steps"]
        if sd["momentum_buffer"] is not None:
            p, _ = self.net.flat_parameters()
            self.buf = sd["momentum_buffer"].to(p.device)


class FusedAdam(_FusedOptimizer):
    """torch.optim.Adam(lr) defaults: betas (0.9, 0.999), eps 1e-8, no weight decay (S/utils.py:217)."""

    def __init__(self, net, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(net, lr)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.m = self.v = None

    def step(self):
        p, g = self._flat()
        if self.m is None or self.m.numel() != p.numel() or self.m.device != p.device:
            self.m, self.v = torch.zeros_like(p), torch.zeros_like(p)
            self.steps = 0
        self.steps += 1
        ops.adam_step(p, g, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                      self.grad_scale, self.steps)

    def state_dict(self):
        return {"kind": "Adam", "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay,
                "steps": self.steps, "exp_avg": None if self.m is None else self.m.detach().cpu(),
                "exp_avg_sq": None if self.v is None else self.v.detach().cpu()}

    def load_state_dict(self, sd):
        self.lr, self.betas, self.eps, self.weight_decay, self.steps = sd["lr"], tuple(sd["betas"]), sd["eps"], \
            sd["weight_decay"], sd["steps"]
        if sd["exp_avg"] is not None:
            p, _ = self.net.flat_parameters()
            self.m, self.v = sd["exp_avg"].to(p.device), sd["exp_avg_sq"].to(p.device)
