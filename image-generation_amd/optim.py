"""Fused Adam over one flat parameter buffer (``dvg_adam_step``).

Counterpart of the two ``torch.optim.Adam(..., weight_decay=...)`` instances the reference
builds at /root/reference/src/model_wrapper.py:208-217 (coupled L2 weight decay, betas
(0.9, 0.999), eps 1e-8).  All parameters of the optimizer are re-homed as views into one
contiguous float32 buffer, so a step is a single kernel launch and -- on several GPUs -- the
gradient exchange is a single all-reduce of ``flat_grad``.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch

from . import _lib


class FlatAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, grad_buffer: Optional[torch.Tensor] = None):
        self.params: List[torch.nn.Parameter] = [p for p in params]
        if not self.params:
            raise ValueError("FlatAdam got an empty parameter list")
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("FlatAdam needs float32 parameters on one device")
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.empty(self.numel, dtype=torch.float32, device=dev)
        # grad_buffer: a caller-provided view (ModelWrapper packs both optimizers' gradients into ONE buffer so that a
        # data-parallel step is one all-reduce)
        if grad_buffer is not None and (grad_buffer.numel() != self.numel or grad_buffer.dtype != torch.float32
                                        or grad_buffer.device != dev or not grad_buffer.is_contiguous()):
            raise ValueError("FlatAdam: grad_buffer must be a contiguous float32 view of numel elements on the parameters' device")
        self.flat_grad = grad_buffer if grad_buffer is not None else torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.offsets = []
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.flat[off: off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[off: off + n].view(p.shape)  # parameters become views of the flat buffer
                self.offsets.append(off)
                off += n
        self.step_count = 0
        self.dyn_slot = 0  # which adam_* slot of dvg_step_state_t this optimizer reads under graph replay

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None

    def grad_view(self, param: torch.nn.Parameter) -> torch.Tensor:
        """The slice of ``flat_grad`` that holds ``param``'s gradient, shaped like it."""
        for p, off in zip(self.params, self.offsets):
            if p is param:
                return self.flat_grad[off: off + p.numel()].view(p.shape)
        raise KeyError("parameter does not belong to this optimizer")

    def gather_grads(self) -> torch.Tensor:
        """Pack the per-tensor gradients into ``flat_grad`` (missing gradients count as zero).  Gradients that already
        live in their slice of the buffer (the modules' backward wrote them there: modules._grad_targets) cost nothing."""
        base, esz = self.flat_grad.data_ptr(), self.flat_grad.element_size()
        if all(p.grad is not None and p.grad.data_ptr() == base + off * esz and p.grad.is_contiguous()
               for p, off in zip(self.params, self.offsets)):
            return self.flat_grad
        views = []
        for p, off in zip(self.params, self.offsets):
            views.append(p.grad.reshape(-1) if p.grad is not None else torch.zeros(p.numel(), device=self.flat.device))
        torch.cat(views, out=self.flat_grad)
        return self.flat_grad

    def step(self, grad_scale: float = 1.0, gathered: bool = False):
        if not self.flat.is_cuda:
            raise _lib.DvgError("FlatAdam.step needs CUDA tensors; there is no CPU fallback")
        if not gathered:
            self.gather_grads()
        g = self.param_groups[0]
        self.step_count += 1
        with torch.cuda.device(self.flat.device):
            _lib.check(
                _lib.lib().dvg_adam_step(self.flat.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                         self.exp_avg_sq.data_ptr(), self.numel, float(g["lr"]), float(g["betas"][0]),
                                         float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                                         self.step_count, float(grad_scale), _lib.DYN, int(self.dyn_slot),
                                         _lib.stream_ptr(self.flat.device)),
                "dvg_adam_step",
            )

    def hyper(self, step_count=None):
        """(lr / (1 - beta1^t), sqrt(1 - beta2^t)) for step t, rounded exactly as dvg_adam_step derives them from its
        float32 by-value arguments (so a graph replay reading them from device memory is bit-identical to eager)."""
        import numpy as np

        g = self.param_groups[0]
        t = self.step_count if step_count is None else step_count
        lr32, b1, b2 = float(np.float32(g["lr"])), float(np.float32(g["betas"][0])), float(np.float32(g["betas"][1]))
        step_size = np.float32(lr32 / (1.0 - b1 ** t))
        bc2_sqrt = np.float32((1.0 - b2 ** t) ** 0.5)
        return float(step_size), float(bc2_sqrt)

    def state_dict(self):
        return dict(step=self.step_count, exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone(),
                    lr=self.param_groups[0]["lr"])

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups[0]["lr"] = sd["lr"]
