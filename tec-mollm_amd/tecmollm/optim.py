"""Optimizer side of the training step on FLAT device buffers (train.py:92-109, :358-366).

`FlatAdamW` owns four flat fp32 buffers -- parameters, gradients, exp_avg, exp_avg_sq -- and re-points
every trainable `p.data` / `p.grad` at views into them, so that
  * autograd accumulates straight into the buffer the RCCL all-reduce sends,
  * `clip_grad_norm_(1.0)` + `AdamW.step()` + `zero_grad()` are two kernel launches (`tecm_adamw_clip_step`)
    over 3 081 996 values instead of a multi-tensor pass per operation, and never sync the host.
`CosineWarmRestarts` is the closed form of torch's CosineAnnealingWarmRestarts stepped once per update
(train.py:366, :108) -- a host-side float, nothing on the device.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Iterable, List

import torch

from . import _lib
from ._lib import TecmAdamW, TecmError, check, lib, stream_ptr


class CosineWarmRestarts:
    """lr_scheduler.CosineAnnealingWarmRestarts(T_0, T_mult, eta_min) with `.step()` called without an epoch."""

    def __init__(self, base_lr: float, T_0: int = 10, T_mult: int = 2, eta_min: float = 1e-7):
        if T_0 <= 0 or T_mult < 1:
            raise ValueError("T_0 must be positive and T_mult >= 1")
        self.base_lr, self.T_0, self.T_mult, self.eta_min = float(base_lr), int(T_0), int(T_mult), float(eta_min)
        self.T_i, self.T_cur, self.last_epoch = self.T_0, 0, 0

    @property
    def lr(self) -> float:
        return self.eta_min + (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * self.T_cur / self.T_i)) / 2

    def step(self) -> float:
        self.last_epoch += 1
        self.T_cur += 1
        if self.T_cur >= self.T_i:
            self.T_cur -= self.T_i
            self.T_i *= self.T_mult
        return self.lr

    def state_dict(self) -> Dict[str, float]:
        return {"T_i": self.T_i, "T_cur": self.T_cur, "last_epoch": self.last_epoch, "base_lr": self.base_lr}

    def load_state_dict(self, s: Dict[str, float]) -> None:
        self.T_i, self.T_cur, self.last_epoch = int(s["T_i"]), int(s["T_cur"]), int(s["last_epoch"])
        self.base_lr = float(s.get("base_lr", self.base_lr))


class FlatAdamW:
    """AdamW(lr, betas, eps, weight_decay) with fused global-norm clipping over flat buffers (HIP only)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, grad_tail: int = 0):
        """grad_tail: extra fp32 slots behind the gradients in the same allocation (`flat_grad_ext`), so that a few
        control words ride in the data-parallel all-reduce of the gradient (tecmollm/train.py: rank-divergence check)."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdamW needs at least one trainable parameter")
        dev = self.params[0].device
        for p in self.params:
            _lib.require_gpu_tensor(p, "parameter")
            if p.device != dev:
                raise TecmError("all parameters must live on one device")
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.sizes = [p.numel() for p in self.params]
        n = sum(self.sizes)
        self.n = n
        self.flat_param = torch.empty(n, device=dev, dtype=torch.float32)
        self.flat_grad_ext = torch.zeros(n + int(grad_tail), device=dev, dtype=torch.float32)
        self.flat_grad = self.flat_grad_ext[:n]
        self.exp_avg = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=torch.float32)
        self._partials = torch.empty(_lib.TECM_NORM_BLOCKS, device=dev, dtype=torch.float64)
        self.total_norm = torch.zeros(1, device=dev, dtype=torch.float32)
        o = 0
        self.grad_views: List[torch.Tensor] = []
        with torch.no_grad():
            for p, k in zip(self.params, self.sizes):
                self.flat_param[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[o:o + k].view(p.shape)
                self.grad_views.append(self.flat_grad[o:o + k].view(p.shape))
                p.grad = self.grad_views[-1]
                o += k
        self.step_count = 0

    def detach_grads(self) -> None:
        """Set every p.grad to None so that the next backward hands its gradient tensors over (autograd's
        AccumulateGrad then stores them instead of launching one `p.grad += g` kernel per parameter);
        `absorb_grads` adds them into the flat buffer with one multi-tensor launch."""
        for p in self.params:
            p.grad = None

    def absorb_grads(self) -> None:
        """flat_grad += every p.grad that does not already live in the flat buffer; those are released (set to None)."""
        dst, src = [], []
        for p, v in zip(self.params, self.grad_views):
            g = p.grad
            if g is None or g.data_ptr() == v.data_ptr():
                continue
            if g.shape != v.shape or g.dtype != torch.float32 or g.device != v.device:
                raise TecmError("a parameter's .grad has the wrong shape / dtype / device")
            dst.append(v)
            src.append(g)
            p.grad = None
        if dst:
            torch._foreach_add_(dst, src)

    def step(self, lr: float = None, max_norm: float = 1.0, grad_scale: float = 1.0, zero_grad: bool = True) -> torch.Tensor:
        """clip_grad_norm_(max_norm) on grad*grad_scale, AdamW update, optional zero_grad.  Returns the
        (device, un-synced) total gradient norm before clipping."""
        self.absorb_grads()                           # gradients autograd stored outside the flat buffer
        self.step_count += 1
        a = TecmAdamW(n=self.n, param=self.flat_param.data_ptr(), grad=self.flat_grad.data_ptr(),
                      exp_avg=self.exp_avg.data_ptr(), exp_avg_sq=self.exp_avg_sq.data_ptr(),
                      partials=self._partials.data_ptr(), total_norm_out=self.total_norm.data_ptr(),
                      lr=self.lr if lr is None else float(lr), beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                      weight_decay=self.weight_decay, max_norm=float(max_norm) if max_norm else 0.0,
                      grad_scale=float(grad_scale), step=self.step_count, zero_grad=1 if zero_grad else 0)
        check(lib().tecm_adamw_clip_step(C.byref(a), stream_ptr()), "tecm_adamw_clip_step")
        return self.total_norm

    def zero_grad(self) -> None:
        self.flat_grad.zero_()

    # torch.optim.AdamW-compatible state (per-parameter tensors), so optimizer checkpoints interchange
    def state_dict(self) -> dict:
        state, o = {}, 0
        for i, (p, k) in enumerate(zip(self.params, self.sizes)):
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[o:o + k].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + k].view(p.shape).clone()}
            o += k
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: dict) -> None:
        group = sd["param_groups"][0]
        if len(group["params"]) != len(self.params):
            raise ValueError("optimizer state has a different number of parameters")
        self.lr, self.eps, self.weight_decay = float(group["lr"]), float(group["eps"]), float(group["weight_decay"])
        self.betas = (float(group["betas"][0]), float(group["betas"][1]))
        o, steps = 0, set()
        with torch.no_grad():
            for i, (p, k) in enumerate(zip(self.params, self.sizes)):
                st = sd["state"].get(i)
                if st is not None:
                    self.exp_avg[o:o + k].copy_(st["exp_avg"].reshape(-1))
                    self.exp_avg_sq[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
                    steps.add(int(st["step"]))
                o += k
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ; FlatAdamW keeps a single count")
        self.step_count = steps.pop() if steps else 0
