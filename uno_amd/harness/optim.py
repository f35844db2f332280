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
        part in every step (they share the counter); state[p]['step'] is that counter tensor.  lr, eps and weight_decay are read by
        the update from three doubles ON THE DEVICE as well (sync_hyper() refreshes them from the param group whenever they changed):
        a scheduler that edits group['lr'] (reference ns_train_2d.py:37,113: StepLR) takes effect on the next replay of a captured
        step, exactly as on the eager path."""
        if lr < 0 or eps < 0 or weight_decay < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._plans = {}            # per param group: pointer tables of the device tensors (not part of state_dict)
        self.capturable = bool(capturable)
        # per group INDEX: (int32 step counter, float32[4] scratch, float64[3] (lr, eps, weight_decay), the three as last uploaded)
        self._dev_counters = {}

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
        if step is None:            # capturable: the group's device counter and device hyper-parameters
            ctr, scal, hyper, _ = self._dev_counters[self._group_index(group)]
            plan.step_dev(ctr, scal, lr, beta1, beta2, eps, wd, hyper=hyper)
        else:
            plan.step(step, lr, beta1, beta2, eps, wd)
        # the kernel writes the parameters through raw pointers: tell autograd they changed (version counters guard saved tensors
        # and key the half-precision weight copies of the mixed-precision layers)
        torch.autograd.graph.increment_version(params)

    def _group_index(self, group):
        for i, g in enumerate(self.param_groups):
            if g is group:
                return i
        raise RuntimeError("ComplexAdam: unknown parameter group")

    def _counter(self, group, device):
        """The group's device-side step counter (+ scratch and hyper-parameters), created on first use.  A count that already exists
        in the group's state - an int from non-capturable steps, or the tensor load_state_dict() put there - is carried over:
        the bias corrections continue from it (reference Adam.py resumes from state['step'])."""
        gi = self._group_index(group)
        entry = self._dev_counters.get(gi)
        if entry is None:
            start = 0
            for p in group["params"]:
                st = self.state.get(p)
                if st and "step" in st:
                    start = max(start, int(st["step"]))            # (a one-off host read when the counter is created)
            ctr = torch.full((1,), start, dtype=torch.int32, device=device)
            now = (float(group["lr"]), float(group["eps"]), float(group["weight_decay"]))
            entry = [ctr, torch.zeros(4, dtype=torch.float32, device=device), torch.tensor(now, dtype=torch.float64).to(device), now]
            self._dev_counters[gi] = entry
        return entry

    def sync_hyper(self):
        """Upload (lr, eps, weight_decay) of every capturable group to the device if they changed since the last upload.  step() calls
        it when no stream capture is running; harness.GraphedStep calls it before every replay."""
        for gi, entry in self._dev_counters.items():
            g = self.param_groups[gi]
            now = (float(g["lr"]), float(g["eps"]), float(g["weight_decay"]))
            if entry[3] != now:
                entry[2].copy_(torch.tensor(now, dtype=torch.float64))
                entry[3] = now

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # the loaded state replaced the moment tensors and the step entries: plans and device counters are rebuilt from it on next use
        self._plans.clear()
        self._dev_counters.clear()
        if not self.capturable:
            # a capturable checkpoint keeps ONE shared int32 device tensor as the step of every parameter, and torch hands `step` entries
            # over un-cloned: `st["step"] += 1` per parameter would then advance that shared tensor once per PARAMETER per step (and an
            # in-memory state_dict would alias the source optimiser's live counter).  Host integers here.
            for st in self.state.values():
                if isinstance(st.get("step"), torch.Tensor):
                    st["step"] = int(st["step"])

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
                    st["step"] = self._counter(group, p.device)[0]
        self.sync_hyper()

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
                    st["step"] = self._counter(group, p.device)[0]
                    devb.setdefault(None, []).append(p)
                    continue
                st["step"] += 1
                if on_dev:
                    devb.setdefault(st["step"], []).append(p)
                else:
                    host.setdefault(st["step"], []).append(p)
            if None in devb and not torch.cuda.is_current_stream_capturing():
                self.sync_hyper()
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
