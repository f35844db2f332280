"""Adam with the semantics of the reference's optimiser (Adam.py:27-52), which differ from
torch.optim.Adam on complex parameters: the second moment is built from g * conj(g) (the squared
complex modulus, one real number per complex entry), weight decay is the coupled L2 form
(g += wd * p).  Implemented with multi-tensor (_foreach) ops over real views so a step is a handful
of launches instead of a Python loop of complex sqrt/addcdiv per parameter.  Parameters on a HIP device are
updated by the one-pass multi-tensor K10 kernel (csrc/adam.hip: 24 tensors per launch, one native call per step bucket)."""
from __future__ import annotations

import math

import torch
from torch.optim.optimizer import Optimizer


class ComplexAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        """capturable: keep the step count of the device parameters ON THE DEVICE (one counter per parameter group, advanced by the
        update itself), so that step() can be recorded in a HIP graph (harness.GraphedStep) - the bias corrections of the reference
        (Adam.py:27-52: from state['step']) are then evaluated on the device in double.  All device parameters of a group must take
        part in every step (they share the counter); state[p]['step'] is that counter tensor."""
        if lr < 0 or eps < 0 or weight_decay < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._plans = {}            # per param group: pointer tables of the device tensors (not part of state_dict)
        self.capturable = bool(capturable)
        self._dev_counters = {}     # id(group) -> (int32 step counter, float32[2] scalars) on the group's device

    def _device_step(self, group, params, step, lr, beta1, beta2, eps, wd):
        """K10 over all device tensors of the group in one native call; the pointer tables are rebuilt only when a
        parameter or gradient buffer moved (FlatGradients keeps them fixed)."""
        from .. import _native
        # one plan per set of buffers (the pointer tuple is the key): step buckets of one group never evict each other, and a
        # plan is rebuilt only when a parameter, gradient or moment buffer moved (e.g. load_state_dict replaces the moments)
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr())
                    for p in params)
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) >= 4:
                self._plans.clear()
            plan = _native.AdamPlan([p.data for p in params], [p.grad for p in params],
                                    [self.state[p]["exp_avg"] for p in params], [self.state[p]["exp_avg_sq"] for p in params])
            self._plans[key] = plan
        if step is None:            # capturable: the group's device counter
            ctr, scal = self._dev_counters[id(group)]
            plan.step_dev(ctr, scal, lr, beta1, beta2, eps, wd)
        else:
            plan.step(step, lr, beta1, beta2, eps, wd)
        # the kernel writes the parameters through raw pointers: tell autograd they changed (version counters guard saved tensors
        # and key the half-precision weight copies of the mixed-precision layers)
        torch.autograd.graph.increment_version(params)

    @staticmethod
    def _real(t):
        return torch.view_as_real(t) if t.is_complex() else t

    def _init_param_state(self, p):
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(self._real(p))
            # one real second-moment entry per (possibly complex) parameter entry
            st["exp_avg_sq"] = torch.zeros(p.shape, dtype=st["exp_avg"].dtype, device=p.device)
        return st

    def init_state(self):
        """Allocate every parameter's moments (and, when capturable, the device step counters) NOW instead of at the first step():
        an allocation + zero fill recorded inside a HIP graph would be replayed with it."""
        for group in self.param_groups:
            for p in group["params"]:
                if not p.requires_grad:
                    continue
                st = self._init_param_state(p)
                if self.capturable and p.is_cuda and p.dtype in (torch.float32, torch.complex64):
                    if id(group) not in self._dev_counters:
                        self._dev_counters[id(group)] = (torch.zeros(1, dtype=torch.int32, device=p.device),
                                                         torch.zeros(2, dtype=torch.float32, device=p.device))
                    if not torch.is_tensor(st["step"]):
                        st["step"] = self._dev_counters[id(group)][0]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            lr, eps, wd = group["lr"], group["eps"], group["weight_decay"]
            # parameters are bucketed by their own step count (the reference keeps a per-parameter step and bias correction,
            # Adam.py:27-52): a parameter whose grad was None on some steps simply lands in another bucket
            host, devb = {}, {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self._init_param_state(p)
                on_dev = p.is_cuda and p.dtype in (torch.float32, torch.complex64) and p.is_contiguous() and p.grad.is_contiguous()
                if self.capturable and on_dev:
                    if id(group) not in self._dev_counters:
                        self._dev_counters[id(group)] = (torch.zeros(1, dtype=torch.int32, device=p.device),
                                                         torch.zeros(2, dtype=torch.float32, device=p.device))
                    st["step"] = self._dev_counters[id(group)][0]
                    devb.setdefault(None, []).append(p)
                    continue
                st["step"] += 1
                if on_dev:
                    devb.setdefault(st["step"], []).append(p)
                else:
                    host.setdefault(st["step"], []).append(p)
            for t, dev_p in devb.items():
                self._device_step(group, dev_p, t, lr, beta1, beta2, eps, wd)
            for t, plist in host.items():
                ps = [self._real(p) for p in plist]
                gs = [self._real(p.grad) for p in plist]
                ms = [self.state[p]["exp_avg"] for p in plist]
                vs = [self.state[p]["exp_avg_sq"] for p in plist]
                cplx = [p.is_complex() for p in plist]
                bc1 = 1 - beta1 ** t
                bc2 = 1 - beta2 ** t
                if wd != 0:
                    gs = torch._foreach_add(gs, ps, alpha=wd)
                torch._foreach_mul_(ms, beta1)
                torch._foreach_add_(ms, gs, alpha=1 - beta1)
                sq = torch._foreach_mul(gs, gs)
                sq = [s.sum(-1) if c else s for s, c in zip(sq, cplx)]       # |g|^2 for complex entries
                torch._foreach_mul_(vs, beta2)
                torch._foreach_add_(vs, sq, alpha=1 - beta2)
                denom = torch._foreach_sqrt(vs)
                torch._foreach_div_(denom, math.sqrt(bc2))
                torch._foreach_add_(denom, eps)
                denom = [d.unsqueeze(-1) if c else d for d, c in zip(denom, cplx)]
                torch._foreach_addcdiv_(ps, ms, denom, value=-lr / bc1)
        return loss
