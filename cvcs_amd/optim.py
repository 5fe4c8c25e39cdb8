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
        """the keys torch.optim.lr_scheduler.PolynomialLR saves (its load_state_dict updates __dict__: a reference run can resume from
        this) plus `base_lr`"""
        return {"total_iters": self.total_iters, "power": self.power, "base_lr": self.base_lr, "base_lrs": [self.base_lr],
                "last_epoch": self.last_epoch, "_step_count": self.last_epoch + 1, "_last_lr": [self.optimizer.lr],
                "_get_lr_called_within_step": False, "verbose": False}

    def load_state_dict(self, sd):
        """accepts this class's dictionary or torch's (S/train.py resume: `scheduler.load_state_dict(checkpoint[...])`)"""
        self.total_iters, self.power = sd["total_iters"], sd["power"]
        self.base_lr = sd["base_lr"] if "base_lr" in sd else sd["base_lrs"][0]
        self.last_epoch = sd["last_epoch"]
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

    # ---- torch.optim state_dict format: {"state": {i: {...}}, "param_groups": [{..., "params": [0..n-1]}]}, parameter i = the i-th of
    # net.parameters() (what the reference hands to torch.optim, S/utils.py:208-221).  The flat state buffers mirror the flat parameter
    # buffer, so parameter i's slice sits at the offset of its data inside the flat parameters.
    def _slices(self):
        flat, _ = self.net.flat_parameters()
        out = []
        for prm in self.net.parameters():
            off = (prm.data_ptr() - flat.data_ptr()) // 4
            out.append((off, prm.numel(), tuple(prm.shape)))
        return flat, out

    def _export(self, buf):
        _, sl = self._slices()
        return [buf[off:off + n].view(shape).detach().cpu().clone() for off, n, shape in sl]

    def _import(self, tensors, name):
        flat, sl = self._slices()
        buf = torch.zeros_like(flat)
        assert len(tensors) == len(sl), f"optimizer state holds {len(tensors)} '{name}' tensors, the network has {len(sl)} parameters"
        for t, (off, n, shape) in zip(tensors, sl):
            assert tuple(t.shape) == shape, (name, tuple(t.shape), shape)
            buf[off:off + n].copy_(t.reshape(-1).to(flat.device, torch.float32))
        return buf


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
        """torch.optim.SGD's format (a reference run can resume from it, and this class loads the reference's)"""
        n = sum(1 for _ in self.net.parameters())
        state = {} if self.buf is None else {i: {"momentum_buffer": t} for i, t in enumerate(self._export(self.buf))}
        group = {"lr": self.lr, "momentum": self.momentum, "dampening": 0, "weight_decay": self.weight_decay, "nesterov": False,
                 "maximize": False, "foreach": None, "differentiable": False, "fused": None, "params": list(range(n))}
        return {"state": state, "param_groups": [group], "steps": self.steps}

    def load_state_dict(self, sd):
        if "param_groups" in sd:      # torch format (the reference's checkpoints, and this class's own)
            g = sd["param_groups"][0]
            self.lr, self.momentum, self.weight_decay = g["lr"], g["momentum"], g["weight_decay"]
            st = sd["state"]
            if len(st):
                self.buf = self._import([st[i]["momentum_buffer"] for i in range(len(st))], "momentum_buffer")
                self.steps = sd.get("steps", 1)      # (only "first step or not" matters to the SGD update)
            else:
                self.buf, self.steps = None, 0
            return
        self.lr, self.momentum, self.weight_decay, self.steps = sd["lr"], sd["momentum"], sd["weight_decay"], sd["steps"]   # round-1 format
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
        """torch.optim.Adam's format"""
        n = sum(1 for _ in self.net.parameters())
        state = {}
        if self.m is not None:
            for i, (m, v) in enumerate(zip(self._export(self.m), self._export(self.v))):
                state[i] = {"step": torch.tensor(float(self.steps)), "exp_avg": m, "exp_avg_sq": v}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(n))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        if "param_groups" in sd:
            g = sd["param_groups"][0]
            self.lr, self.betas, self.eps, self.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
            st = sd["state"]
            if len(st):
                self.m = self._import([st[i]["exp_avg"] for i in range(len(st))], "exp_avg")
                self.v = self._import([st[i]["exp_avg_sq"] for i in range(len(st))], "exp_avg_sq")
                self.steps = int(float(st[0]["step"]))
            else:
                self.m = self.v = None
                self.steps = 0
            return
        self.lr, self.betas, self.eps, self.weight_decay, self.steps = sd["lr"], tuple(sd["betas"]), sd["eps"], \
            sd["weight_decay"], sd["steps"]                                                                          # round-1 format
        if sd["exp_avg"] is not None:
            p, _ = self.net.flat_parameters()
            self.m, self.v = sd["exp_avg"].to(p.device), sd["exp_avg_sq"].to(p.device)
