"""Adam / AdamW for a net whose parameters are views of ONE flat buffer: one HIP launch per step.

The reference's Adam phase is `torch.optim.Adam(self.dnn.parameters())` / `AdamW(..., weight_decay=1e-2)`
(7kdv/PI-7kdv3soliton.py:138,154-157; 7kdv/sw73-adamW.py:170): ten parameter tensors, ~15 foreach launches per
step.  `FusedAdam` has the same constructor arguments, the same update rule (torch's default, non-amsgrad
Adam; `decoupled=True` is AdamW) and the usual `zero_grad()` / `step()` protocol; the step count lives on the
device, so a captured HIP graph replays it as is.  There is no CPU fallback: it needs libpinnjet.so and a GPU.
"""
import ctypes

import torch

from ._native import NativeError, _check, load_library


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False,
                 flat_params=None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=decoupled)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("FusedAdam takes one parameter group")
        self._params = self.param_groups[0]["params"]
        if flat_params is None or not flat_params.is_cuda:
            raise NativeError("FusedAdam needs the flat float32 CUDA buffer the parameters are views of; "
                              "there is no CPU fallback (use torch.optim.Adam)")
        off = 0
        for p in self._params:
            if p.data_ptr() != flat_params.data_ptr() + 4 * off or p.dtype != torch.float32:
                raise ValueError("flat_params is not the buffer the parameters are views of")
            off += p.numel()
        if off != flat_params.numel():
            raise ValueError("flat_params has %d elements, the parameters %d" % (flat_params.numel(), off))
        self.lib = load_library()
        self._x = flat_params
        dev = flat_params.device
        n = flat_params.numel()
        self._g = torch.zeros(n, dtype=torch.float32, device=dev)
        # The moments (flat, unlike the parameters they belong to) and the float64 device-side counters live
        # OUTSIDE `self.state`: torch.optim.Optimizer.load_state_dict casts every floating state tensor to its
        # parameter's dtype, which would hand the kernel a float32 buffer where it reads doubles.
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.device_state = torch.zeros(4, dtype=torch.float64, device=dev)   # step, beta1^t, beta2^t, ticket
        self._attach_grads()

    def state_dict(self):
        """torch's layout plus `fused`: the flat moments and the float64 counters, dtypes preserved."""
        sd = super().state_dict()
        sd["fused"] = {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                       "device_state": self.device_state.clone()}
        return sd

    def load_state_dict(self, state_dict):
        sd = dict(state_dict)
        fused = sd.pop("fused", None)
        super().load_state_dict(sd)
        self._params = self.param_groups[0]["params"]
        if fused is not None:
            for name in ("exp_avg", "exp_avg_sq", "device_state"):
                dst, src = getattr(self, name), fused[name]
                if src.shape != dst.shape or src.dtype != dst.dtype:
                    raise ValueError("FusedAdam.load_state_dict: %s is %s %s, expected %s %s"
                                     % (name, tuple(src.shape), src.dtype, tuple(dst.shape), dst.dtype))
                dst.copy_(src)

    def _intact(self):
        """every `.grad` is still the view this optimizer attached (identity checks: the per-step host path)"""
        views = getattr(self, "_views", None)
        return views is not None and all(p.grad is v for p, v in zip(self._params, views))

    def _attach_grads(self):
        if self._intact():
            return
        off = 0
        for p in self._params:
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self._g.data_ptr() + 4 * off:
                p.grad = self._g[off:off + n].view(p.shape)
            off += n
        self._views = [p.grad for p in self._params]

    def zero_grad(self, set_to_none=True):
        """Zeroes the flat gradient; `.grad` stay views into it (one memset instead of ten)."""
        self._g.zero_()
        self._attach_grads()

    def flat_grad(self):
        return self._g

    def _gather(self):
        if self._intact():
            return
        off = 0
        for p in self._params:
            n = p.numel()
            seg = self._g[off:off + n]
            if p.grad is None:
                seg.zero_()
            elif p.grad.data_ptr() != seg.data_ptr():
                seg.copy_(p.grad.reshape(-1))
            off += n

    @property
    def step_count(self):
        return int(self.device_state[0].item())

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._gather()
        grp = self.param_groups[0]
        dev = self._x.device
        with torch.cuda.device(dev):
            _check(self.lib.pinn_adam_step(self._x.data_ptr(), self._g.data_ptr(), self.exp_avg.data_ptr(),
                                           self.exp_avg_sq.data_ptr(), self.device_state.data_ptr(), self._x.numel(),
                                           float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]),
                                           float(grp["eps"]), float(grp["weight_decay"]), 1 if grp["decoupled"] else 0,
                                           ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                   "pinn_adam_step")
        return loss
