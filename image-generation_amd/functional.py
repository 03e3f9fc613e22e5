"""torch.autograd glue over the C ABI: each Function calls one fused HIP entry
point for forward (+ saved quantities) and one for backward.  torch only owns
the memory and the stream.  No CPU fallback: CPU tensors raise.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from ._lib import MmdCfg, check, lib, require_cuda, stream_ptr

GUMBEL_TAU = 1.0 / 7.0


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ----------------------------------------------------------------------------- MMD


class _MMD(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, n_kernels, factor, bandwidth, squared, reduce_mean, biased):
        L = lib()
        xd = require_cuda(x.detach().float().contiguous(), "x")
        yd = require_cuda(y.detach().float().contiguous(), "y")
        if xd.dim() != 2 or yd.dim() != 2 or xd.shape[1] != yd.shape[1]:
            raise ValueError("maximum_mean_discrepancy_loss expects x (nx, d) and y (ny, d)")
        nx, d = xd.shape
        ny = yd.shape[0]
        cfg = MmdCfg(int(n_kernels), float(factor), float(bandwidth if bandwidth is not None else -1.0),
                     int(bool(squared)), int(bool(reduce_mean)), int(bool(biased)))
        need_grad = x.requires_grad
        loss = torch.empty((), dtype=torch.float32, device=xd.device)
        grad = torch.empty_like(xd) if need_grad else None
        nbytes = L.dvg_mmd_workspace_bytes(nx, ny, d)
        if nbytes == 0:
            raise _lib.DvgError(f"MMD: unsupported shape nx={nx} ny={ny} d={d} (d must be a multiple of 32)")
        ws = _ws(nbytes, xd.device)
        with torch.cuda.device(xd.device):
            check(L.dvg_mmd_fwd_bwd(xd.data_ptr(), nx, yd.data_ptr(), ny, d, ctypes.byref(cfg), loss.data_ptr(),
                                    _lib.ptr(grad), ws.data_ptr(), ws.numel(), stream_ptr(xd.device)), "dvg_mmd_fwd_bwd")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g if grad is not None else None, None, None, None, None, None, None, None)


def mmd_loss(x, y, n_kernels=7, factor=2.0, bandwidth=None, squared=False, reduce_mean=False, biased=False):
    return _MMD.apply(x, y, n_kernels, factor, bandwidth, squared, reduce_mean, biased)


def mmd_loss_and_grad(x, y, n_kernels=7, factor=2.0, bandwidth=None, squared=False, reduce_mean=False, biased=False):
    """(loss, d loss / d x) from the one fused call, outside autograd: for callers that seed the backward pass
    themselves (``torch.autograd.backward(x, grad)``) instead of paying a ones_like + multiply per loss term."""
    L = lib()
    xd = require_cuda(x.detach().float().contiguous(), "x")
    yd = require_cuda(y.detach().float().contiguous(), "y")
    if xd.dim() != 2 or yd.dim() != 2 or xd.shape[1] != yd.shape[1]:
        raise ValueError("maximum_mean_discrepancy_loss expects x (nx, d) and y (ny, d)")
    nx, d = xd.shape
    ny = yd.shape[0]
    cfg = MmdCfg(int(n_kernels), float(factor), float(bandwidth if bandwidth is not None else -1.0),
                 int(bool(squared)), int(bool(reduce_mean)), int(bool(biased)))
    loss = torch.empty((), dtype=torch.float32, device=xd.device)
    grad = torch.empty_like(xd)
    nbytes = L.dvg_mmd_workspace_bytes(nx, ny, d)
    if nbytes == 0:
        raise _lib.DvgError(f"MMD: unsupported shape nx={nx} ny={ny} d={d} (d must be a multiple of 32)")
    ws = _ws(nbytes, xd.device)
    with torch.cuda.device(xd.device):
        check(L.dvg_mmd_fwd_bwd(xd.data_ptr(), nx, yd.data_ptr(), ny, d, ctypes.byref(cfg), loss.data_ptr(),
                                grad.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr(xd.device)), "dvg_mmd_fwd_bwd")
    return loss, grad


# ----------------------------------------------------------------------------- latent -> discrete


class _Gumbel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, n_samples, tau, gumbels, seed, offset):
        L = lib()
        lg = require_cuda(logits.detach().float().contiguous(), "logits")
        B, n = lg.shape
        R = int(n_samples)
        spins = torch.empty((B, R, n), dtype=torch.float32, device=lg.device)
        dspin = torch.empty_like(spins)
        gb = None
        if gumbels is not None:
            gb = require_cuda(gumbels.detach().float().contiguous(), "gumbels")
            if tuple(gb.shape) != (B, R, n, 2):
                raise ValueError(f"gumbels must have shape {(B, R, n, 2)}, got {tuple(gb.shape)}")
        with torch.cuda.device(lg.device):
            check(L.dvg_gumbel_fwd(lg.data_ptr(), B, n, R, float(tau), _lib.ptr(gb), int(seed) & (2**64 - 1),
                                   int(offset) & (2**64 - 1), spins.data_ptr(), dspin.data_ptr(), _lib.DYN,
                                   stream_ptr(lg.device)),
                  "dvg_gumbel_fwd")
        ctx.save_for_backward(dspin)
        ctx.shape = (B, n, R)
        return spins

    @staticmethod
    def backward(ctx, gs):
        (dspin,) = ctx.saved_tensors
        B, n, R = ctx.shape
        gs = gs.contiguous()
        gl = torch.empty((B, n), dtype=torch.float32, device=gs.device)
        with torch.cuda.device(gs.device):
            check(lib().dvg_gumbel_bwd(gs.data_ptr(), dspin.data_ptr(), B, n, R, gl.data_ptr(), stream_ptr(gs.device)),
                  "dvg_gumbel_bwd")
        return gl, None, None, None, None, None


def gumbel_latent_to_discrete(logits, n_samples, gumbels=None, tau=GUMBEL_TAU, seed=0, offset=0):
    return _Gumbel.apply(logits, n_samples, tau, gumbels, seed, offset)


def gumbel_forward_raw(logits, n_samples, gumbels=None, tau=GUMBEL_TAU, seed=0, offset=0):
    """``(spins (B,R,n), dspin (B,R,n))`` outside autograd, for callers that run the backward themselves
    (:func:`gumbel_backward`): the training step feeds it the SUM of two spin gradients in one kernel."""
    L = lib()
    lg = require_cuda(logits.detach().float().contiguous(), "logits")
    B, n = lg.shape
    R = int(n_samples)
    spins = torch.empty((B, R, n), dtype=torch.float32, device=lg.device)
    dspin = torch.empty_like(spins)
    gb = None
    if gumbels is not None:
        gb = require_cuda(gumbels.detach().float().contiguous(), "gumbels")
        if tuple(gb.shape) != (B, R, n, 2):
            raise ValueError(f"gumbels must have shape {(B, R, n, 2)}, got {tuple(gb.shape)}")
    with torch.cuda.device(lg.device):
        check(L.dvg_gumbel_fwd(lg.data_ptr(), B, n, R, float(tau), _lib.ptr(gb), int(seed) & (2**64 - 1),
                               int(offset) & (2**64 - 1), spins.data_ptr(), dspin.data_ptr(), _lib.DYN,
                               stream_ptr(lg.device)), "dvg_gumbel_fwd")
    return spins, dspin


def gumbel_backward(dspin, grad_spins, grad_spins2=None):
    """``d loss / d logits (B,n) = sum_r (grad_spins + grad_spins2) * dspin``: one kernel, no separate add pass."""
    B, R, n = dspin.shape
    g1 = require_cuda(grad_spins.detach().float().contiguous(), "grad_spins")
    g2 = None if grad_spins2 is None else require_cuda(grad_spins2.detach().float().contiguous(), "grad_spins2")
    if g1.numel() != dspin.numel() or (g2 is not None and g2.numel() != dspin.numel()):
        raise ValueError("gumbel_backward: gradients must have the spins' shape")
    gl = torch.empty((B, n), dtype=torch.float32, device=dspin.device)
    with torch.cuda.device(dspin.device):
        check(lib().dvg_gumbel_bwd2(g1.data_ptr(), _lib.ptr(g2), dspin.data_ptr(), B, n, R, gl.data_ptr(),
                                    stream_ptr(dspin.device)), "dvg_gumbel_bwd2")
    return gl


def scalar_add(a, b):
    """``a + b`` of two device scalars by the library (the step's ``mse + mmd``)."""
    a = require_cuda(a.detach().float().reshape(()), "a")
    b = require_cuda(b.detach().float().reshape(()), "b")
    out = torch.empty((), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        check(lib().dvg_scalar_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), stream_ptr(a.device)), "dvg_scalar_add")
    return out


class _Heaviside(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits):
        lg = require_cuda(logits.detach().float().contiguous(), "logits")
        out = torch.empty_like(lg)
        with torch.cuda.device(lg.device):
            check(lib().dvg_heaviside_fwd(lg.data_ptr(), lg.numel(), out.data_ptr(), stream_ptr(lg.device)),
                  "dvg_heaviside_fwd")
        return out.unsqueeze(1)

    @staticmethod
    def backward(ctx, g):
        return g.squeeze(1)  # straight-through identity (/root/reference/src/utils/common.py:173)


def heaviside_latent_to_discrete(logits, n_samples=1):
    return _Heaviside.apply(logits)


# ----------------------------------------------------------------------------- GRBM energy


class _Energy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, linear, quadratic, graph):
        L = lib()
        xd = require_cuda(x.detach().float().contiguous(), "x")
        lin = require_cuda(linear.detach(), "linear")
        quad = require_cuda(quadratic.detach(), "quadratic")
        n = xd.shape[-1]
        rows = xd.numel() // n
        out = torch.empty(xd.shape[:-1], dtype=torch.float32, device=xd.device)
        with torch.cuda.device(xd.device):
            check(L.dvg_grbm_energy(graph.ptr, xd.data_ptr(), rows, lin.data_ptr(), quad.data_ptr(), out.data_ptr(),
                                    stream_ptr(xd.device)), "dvg_grbm_energy")
        ctx.graph = graph
        ctx.save_for_backward(xd)
        ctx.needs = (x.requires_grad, linear.requires_grad or quadratic.requires_grad)
        ctx.meta = (linear.shape, quadratic.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        (xd,) = ctx.saved_tensors
        if ctx.needs[0]:
            raise NotImplementedError("gradient of the GRBM energy wrt its input spins is not on the hot path "
                                      "(the reference detaches them: /root/reference/src/model_wrapper.py:333)")
        L = lib()
        n = xd.shape[-1]
        rows = xd.numel() // n
        gw = g.contiguous().float().reshape(-1)
        gl = torch.empty(ctx.meta[0], dtype=torch.float32, device=xd.device)
        gq = torch.empty(ctx.meta[1], dtype=torch.float32, device=xd.device)
        ws = _ws(L.dvg_grbm_suffstats_workspace_bytes(ctx.graph.ptr), xd.device)
        with torch.cuda.device(xd.device):
            check(L.dvg_grbm_suffstats(ctx.graph.ptr, xd.data_ptr(), rows, gw.data_ptr(), 1.0, gl.data_ptr(),
                                       gq.data_ptr(), 0, ws.data_ptr(), ws.numel(), stream_ptr(xd.device)),
                  "dvg_grbm_suffstats")
        return None, gl, gq, None


def grbm_energy(x, linear, quadratic, graph):
    return _Energy.apply(x, linear, quadratic, graph)


# ----------------------------------------------------------------------------- MSE


class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, recon, images):
        L = lib()
        rd = require_cuda(recon.detach().float().contiguous(), "reconstructed")
        im = require_cuda(images.detach().float().contiguous(), "images")
        B, R = rd.shape[0], rd.shape[1]
        if rd.numel() != B * R * 1024 or im.numel() != B * 1024:
            raise ValueError("mse_loss expects reconstructed (B,R,1,32,32) and images (B,1,32,32)")
        loss = torch.empty((), dtype=torch.float32, device=rd.device)
        grad = torch.empty_like(rd) if recon.requires_grad else None
        ws = _ws(L.dvg_mse_workspace_bytes(), rd.device)
        with torch.cuda.device(rd.device):
            check(L.dvg_mse_fwd_bwd(rd.data_ptr(), im.data_ptr(), B, R, 1.0, loss.data_ptr(), _lib.ptr(grad),
                                    ws.data_ptr(), ws.numel(), stream_ptr(rd.device)), "dvg_mse_fwd_bwd")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g if grad is not None else None), None


def replicated_mse_loss(reconstructed, images):
    """mse_loss(reconstructed, images.unsqueeze(1).repeat(1, R, 1, 1, 1)) without building the repeat
    (/root/reference/src/model_wrapper.py:302-305)."""
    return _MSE.apply(reconstructed, images)


def replicated_mse_loss_and_grad(reconstructed, images):
    """(loss, d loss / d reconstructed) outside autograd (see :func:`mmd_loss_and_grad`)."""
    L = lib()
    rd = require_cuda(reconstructed.detach().float().contiguous(), "reconstructed")
    im = require_cuda(images.detach().float().contiguous(), "images")
    B, R = rd.shape[0], rd.shape[1]
    if rd.numel() != B * R * 1024 or im.numel() != B * 1024:
        raise ValueError("mse_loss expects reconstructed (B,R,1,32,32) and images (B,1,32,32)")
    loss = torch.empty((), dtype=torch.float32, device=rd.device)
    grad = torch.empty_like(rd)
    ws = _ws(L.dvg_mse_workspace_bytes(), rd.device)
    with torch.cuda.device(rd.device):
        check(L.dvg_mse_fwd_bwd(rd.data_ptr(), im.data_ptr(), B, R, 1.0, loss.data_ptr(), grad.data_ptr(),
                                ws.data_ptr(), ws.numel(), stream_ptr(rd.device)), "dvg_mse_fwd_bwd")
    return loss, grad
