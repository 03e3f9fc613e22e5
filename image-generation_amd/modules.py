"""Encoder / Decoder with the reference's module structure and checkpoint keys,
running on the fused HIP path.

Mirrors /root/reference/src/encoder.py:18-49 and /root/reference/src/decoder.py:18-62:
same constructor, same attribute names (``conv``, ``projection``,
``increase_latent_dim``, ``make_2x2_images``, ``merge_batch_dim_and_replica_dim``,
``convtrans``), same ``nn.Sequential`` indices, hence the same ``state_dict`` keys
and the same seeded default initialisation.  ``forward`` hands the whole network
to one C-ABI call (``dvg_encoder_fwd`` / ``dvg_decoder_fwd``) and backward to one
more; the sub-modules stay stock torch modules so the UI code that pokes them
(/root/reference/src/utils/callback_helpers.py:119-141) keeps working.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import DecoderGrads, DecoderParams, EncoderGrads, EncoderParams, check, lib, stream_ptr


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _check_param(t: torch.Tensor, name: str):
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise _lib.DvgError(f"{name}: parameters must be contiguous float32 CUDA tensors (got {t.dtype} on {t.device})")


# ------------------------------------------------------------------------------------ encoder


def _grad_targets(module, params):
    """Where the backward kernels write the parameter gradients.  Normally fresh tensors handed back to autograd.  When
    the module has a gradient sink (``ModelWrapper`` points it at the views of the optimizer's flat gradient buffer) and
    no parameter holds a gradient yet, the kernels write straight into those views: the flat buffer is then already
    packed when the backward pass ends -- no per-tensor allocations, no concatenation launch in front of Adam / the
    all-reduce."""
    if writes_grads_direct(module, len(params)):
        return module._grad_sink, True
    return [torch.empty_like(p) for p in params], False


def writes_grads_direct(module, n_params=None) -> bool:
    """THE predicate for "this module's backward writes its parameter gradients into the gradient sink and sets
    ``.grad`` itself" -- used by :func:`_grad_targets` and by callers that bypass autograd's accumulation for the
    module's parameters (``ModelWrapper`` takes the decoder's spin gradient through ``torch.autograd.grad``, which
    DISCARDS whatever the backward returns for the parameters: correct only on this path)."""
    sink = getattr(module, "_grad_sink", None)
    trainable = module._trainable()
    n_params = len(trainable) if n_params is None else n_params
    return sink is not None and len(sink) == n_params and all(p.grad is None for p in trainable)


def _grad_returns(module, grads, direct):
    if not direct:
        return grads
    for p, g in zip(module._trainable(), grads):
        p.grad = g  # (a leaf's .grad may be set directly; autograd is handed None for these inputs)
    return [None] * len(grads)


class _EncoderFn(torch.autograd.Function):
    """args: images, module, then the 18 trainable tensors in registration order."""

    @staticmethod
    def forward(ctx, images, module, *params):
        L = lib()
        x = _lib.require_cuda(images.detach().float().contiguous(), "images")
        if x.dim() != 4 or tuple(x.shape[1:]) != (1, 32, 32):
            raise ValueError(f"Encoder expects (B,1,32,32) images, got {tuple(x.shape)}")
        B, n = x.shape[0], module.n_latents
        st = module._native_struct(params)
        logits = torch.empty((B, n), dtype=torch.float32, device=x.device)
        ws = _ws(L.dvg_encoder_workspace_bytes(B, n), x.device)
        training = bool(module.training)
        with torch.cuda.device(x.device):
            check(L.dvg_encoder_fwd(ctypes.byref(st), n, x.data_ptr(), B, int(training), logits.data_ptr(), ws.data_ptr(),
                                    ws.numel(), stream_ptr(x.device)), "dvg_encoder_fwd")
        ctx.module, ctx.training, ctx.B = module, training, B
        ctx.save_for_backward(x, ws, *params)
        return logits

    @staticmethod
    def backward(ctx, grad_logits):
        if not ctx.training:
            raise _lib.DvgError("Encoder backward is only implemented for training mode (batch statistics)")
        L = lib()
        x, ws, *params = ctx.saved_tensors
        module = ctx.module
        st = module._native_struct(params)
        grads, direct = _grad_targets(module, params)
        gs = EncoderGrads()
        for l in range(4):
            gs.conv_w[l], gs.conv_b[l] = grads[4 * l].data_ptr(), grads[4 * l + 1].data_ptr()
            gs.bn_g[l], gs.bn_b[l] = grads[4 * l + 2].data_ptr(), grads[4 * l + 3].data_ptr()
        gs.proj_w, gs.proj_b = grads[16].data_ptr(), grads[17].data_ptr()
        gl = grad_logits.contiguous().float()
        with torch.cuda.device(x.device):
            check(L.dvg_encoder_bwd(ctypes.byref(st), module.n_latents, x.data_ptr(), ctx.B, gl.data_ptr(),
                                    ctypes.byref(gs), ws.data_ptr(), ws.numel(), stream_ptr(x.device)), "dvg_encoder_bwd")
        return (None, None, *_grad_returns(module, grads, direct))


class Encoder(torch.nn.Module):
    """An encoder network that maps image data to latent spin-string logits."""

    def __init__(self, n_latents: int):
        super().__init__()
        if n_latents % 32 != 0 or n_latents < 32:
            raise ValueError("n_latents must be a positive multiple of 32 for the MFMA path")
        self.n_latents = n_latents
        channels = [1, 32, 64, 128, n_latents]
        layers = []
        for i in range(len(channels) - 1):
            layers.append(torch.nn.Conv2d(channels[i], channels[i + 1], kernel_size=3, stride=1, padding=1))
            layers.append(torch.nn.BatchNorm2d(channels[i + 1]))
            layers.append(torch.nn.MaxPool2d(kernel_size=2, stride=2))
            layers.append(torch.nn.LeakyReLU())
        layers = layers[:-1]
        self.conv = torch.nn.Sequential(*layers)
        self.flatten_last_two_dims = torch.nn.Flatten(start_dim=-2, end_dim=-1)
        self.projection = torch.nn.Linear(2 * 2, 1)
        self.flatten = torch.nn.Flatten()

    def _trainable(self) -> List[torch.Tensor]:
        out = []
        for l in range(4):
            conv, bn = self.conv[4 * l], self.conv[4 * l + 1]
            out += [conv.weight, conv.bias, bn.weight, bn.bias]
        return out + [self.projection.weight, self.projection.bias]

    def _native_struct(self, params: Sequence[torch.Tensor]) -> EncoderParams:
        st = EncoderParams()
        for l in range(4):
            bn = self.conv[4 * l + 1]
            for t, nm in zip(params[4 * l: 4 * l + 4], ("conv.weight", "conv.bias", "bn.weight", "bn.bias")):
                _check_param(t, f"encoder layer {l} {nm}")
            st.conv_w[l], st.conv_b[l] = params[4 * l].data_ptr(), params[4 * l + 1].data_ptr()
            st.bn_g[l], st.bn_b[l] = params[4 * l + 2].data_ptr(), params[4 * l + 3].data_ptr()
            st.bn_rm[l], st.bn_rv[l] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            st.bn_nbt[l] = bn.num_batches_tracked.data_ptr()
        st.proj_w, st.proj_b = params[16].data_ptr(), params[17].data_ptr()
        return st

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _EncoderFn.apply(x, self, *self._trainable())


# ------------------------------------------------------------------------------------ decoder


def _decoder_call_setup(spins, module, masks, seed, offset, params):
    """What both decoder forward functions do before their library call: checks, the native parameter struct, the
    workspace (a prepared prologue's when it was made for exactly this call), the injected dropout masks."""
    L = lib()
    x = _lib.require_cuda(spins.detach().float().contiguous(), "spins")
    if x.dim() != 3 or x.shape[-1] != module.n_latents:
        raise ValueError(f"Decoder expects (B,R,{module.n_latents}) spins, got {tuple(x.shape)}")
    B, R, n = x.shape
    N = B * R
    st = module._native_struct(params)
    training = bool(module.training)
    # a prologue enqueued ahead of time (Decoder.prepare) is used when it was made for exactly this call
    prep, module._prepared = module._prepared, None
    prepared = (prep is not None and masks is None and prep["key"] == (N, n, training, int(seed), int(offset), x.device)
                and all(a.data_ptr() == b.data_ptr() for a, b in zip(prep["params"], params)))
    ws = prep["ws"] if prepared else _ws(L.dvg_decoder_workspace_bytes(N, n), x.device)
    mask_arr = (ctypes.c_void_p * 4)()
    keep = []
    if training and masks is not None:
        for l, (m, c) in enumerate(zip(masks, (128, 64, 32, 1))):
            m = _lib.require_cuda(m.detach().float().contiguous(), f"dropout mask {l}")
            if m.numel() != N * c:
                raise ValueError(f"dropout mask {l} must have {N}x{c} elements")
            keep.append(m)
            mask_arr[l] = m.data_ptr()
    return x, (B, R, n), st, training, prep, prepared, ws, mask_arr, keep


def _decoder_grad_struct(grads):
    gs = DecoderGrads()
    gs.lin_w, gs.lin_b = grads[0].data_ptr(), grads[1].data_ptr()
    for l in range(4):
        gs.conv_w[l], gs.conv_b[l] = grads[2 + 4 * l].data_ptr(), grads[3 + 4 * l].data_ptr()
        gs.bn_g[l], gs.bn_b[l] = grads[4 + 4 * l].data_ptr(), grads[5 + 4 * l].data_ptr()
    gs.conv_w[4], gs.conv_b[4] = grads[18].data_ptr(), grads[19].data_ptr()
    return gs


class _DecoderFn(torch.autograd.Function):
    """args: spins (B,R,n), module, masks (list or None), seed, offset, then the 28 trainable tensors."""

    @staticmethod
    def forward(ctx, spins, module, masks, seed, offset, *params):
        L = lib()
        x, (B, R, n), st, training, prep, prepared, ws, mask_arr, keep = _decoder_call_setup(spins, module, masks, seed, offset, params)
        N = B * R
        out = torch.empty((B, R, 1, 32, 32), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            if prep is not None and not prepared:
                # made for another call (shape, mode or parameters changed in between): wait it out and do without
                torch.cuda.current_stream(x.device).wait_stream(prep["stream"])
            check(L.dvg_decoder_fwd_ex(ctypes.byref(st), n, x.data_ptr(), N, int(training),
                                       mask_arr if keep else None, int(seed) & (2**64 - 1), int(offset) & (2**64 - 1),
                                       out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.DYN, int(prepared),
                                       stream_ptr(x.device)), "dvg_decoder_fwd_ex")
        ctx.module, ctx.training, ctx.shape = module, training, (B, R, n)
        ctx.need_input_grad = spins.requires_grad
        ctx.save_for_backward(x, ws, *params)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if not ctx.training:
            raise _lib.DvgError("Decoder backward is only implemented for training mode (batch statistics)")
        L = lib()
        x, ws, *params = ctx.saved_tensors
        module = ctx.module
        B, R, n = ctx.shape
        st = module._native_struct(params)
        grads, direct = _grad_targets(module, params)
        gs = _decoder_grad_struct(grads)
        go = grad_out.contiguous().float()
        gx = torch.empty_like(x) if ctx.need_input_grad else None
        # (deferral only when the gradients land in the optimizer's buffer: handed back as fresh tensors, autograd's
        # AccumulateGrad may CLONE them on the caller's stream -- it does when anything else holds a reference, and
        # _deferred_keep below does -- i.e. read them before the library's side stream has written them)
        defer = bool(getattr(module, "_defer_join", False)) and direct
        with torch.cuda.device(x.device):
            # defer_join (set by ModelWrapper around its own step, where the encoder's backward always follows on the same
            # stream): the tail of the weight-gradient chain overlaps the head of the encoder's data-gradient chain
            check(L.dvg_decoder_bwd_ex(ctypes.byref(st), n, x.data_ptr(), B * R, go.data_ptr(), ctypes.byref(gs),
                                       _lib.ptr(gx), ws.data_ptr(), ws.numel(), int(defer),
                                       stream_ptr(x.device)), "dvg_decoder_bwd_ex")
        if defer:
            # The library's side stream is still reading ws / go / x (and writing the slabs in ws and the gradients) when
            # this call returns, and torch's caching allocator knows nothing of that raw stream: autograd would release
            # ws the moment this node is done, and a main-stream allocation made before the join could be carved out of
            # it.  The owner of the deferral (ModelWrapper) drops these references AFTER dvg_stream_join_side.
            module._deferred_keep = (ws, go, x, gx, tuple(grads))
        return (gx, None, None, None, None, *_grad_returns(module, grads, direct))


class _DecoderMseFn(torch.autograd.Function):
    """The decoder with the reconstruction loss fused behind it (``dvg_decoder_fwd_mse_ex`` / ``dvg_decoder_bwd_mse_ex``:
    include/dvg.h): returns ``mse_loss(decoder(spins), images replicated over R)`` without ever writing the
    reconstruction or its gradient.  args: spins (B,R,n), images (B,1,32,32), module, masks, seed, offset, then the 28
    trainable tensors.  The gradient the backward call seeds is d loss / d reconstruction itself, fixed at forward time:
    differentiate the returned loss DIRECTLY (unit upstream gradient) -- a loss that is scaled or combined before it is
    differentiated has to go through ``Decoder.forward`` and ``replicated_mse_loss``."""

    @staticmethod
    def forward(ctx, spins, images, module, masks, seed, offset, *params):
        L = lib()
        if not module.training:
            raise _lib.DvgError("Decoder.forward_mse is a training-mode call (batch statistics, backward sums)")
        x, (B, R, n), st, training, prep, prepared, ws, mask_arr, keep = _decoder_call_setup(spins, module, masks, seed, offset, params)
        im = _lib.require_cuda(images.detach().float().contiguous(), "images")
        if im.numel() != B * 1024:
            raise ValueError(f"forward_mse expects images ({B},1,32,32), got {tuple(images.shape)}")
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            if prep is not None and not prepared:
                torch.cuda.current_stream(x.device).wait_stream(prep["stream"])
            check(L.dvg_decoder_fwd_mse_ex(ctypes.byref(st), n, x.data_ptr(), B * R, mask_arr if keep else None,
                                           int(seed) & (2**64 - 1), int(offset) & (2**64 - 1), im.data_ptr(), R, 1.0,
                                           loss.data_ptr(), ws.data_ptr(), ws.numel(), _lib.DYN, int(prepared),
                                           stream_ptr(x.device)), "dvg_decoder_fwd_mse_ex")
        ctx.module, ctx.shape = module, (B, R, n)
        ctx.need_input_grad = spins.requires_grad
        ctx.save_for_backward(x, im, ws, *params)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        L = lib()
        x, im, ws, *params = ctx.saved_tensors
        module = ctx.module
        B, R, n = ctx.shape
        # (the seed gradient is fixed at forward time -- a loss that was scaled or combined before it was
        # differentiated would get silently wrong gradients.  Checked on the device, without a host wait, on every eager
        # call; a captured step has run eagerly three times before its capture.)
        if grad_loss is not None and not torch.cuda.is_current_stream_capturing():
            torch._assert_async((grad_loss == 1).all(), "Decoder.forward_mse: differentiate the returned loss directly (unit upstream gradient)")
        st = module._native_struct(params)
        grads, direct = _grad_targets(module, params)
        gs = _decoder_grad_struct(grads)
        gx = torch.empty_like(x) if ctx.need_input_grad else None
        defer = bool(getattr(module, "_defer_join", False)) and direct
        with torch.cuda.device(x.device):
            check(L.dvg_decoder_bwd_mse_ex(ctypes.byref(st), n, x.data_ptr(), B * R, im.data_ptr(), R, 1.0,
                                           ctypes.byref(gs), _lib.ptr(gx), ws.data_ptr(), ws.numel(), int(defer),
                                           stream_ptr(x.device)), "dvg_decoder_bwd_mse_ex")
        if defer:
            module._deferred_keep = (ws, im, x, gx, tuple(grads))  # (see _DecoderFn.backward)
        return (gx, None, None, None, None, None, *_grad_returns(module, grads, direct))


class Decoder(torch.nn.Module):
    """A decoder network that maps latent variables to images."""

    def __init__(self, n_latents: int):
        super().__init__()
        if n_latents % 32 != 0 or n_latents < 32:
            raise ValueError("n_latents must be a positive multiple of 32 for the MFMA path")
        self.n_latents = n_latents
        channels = [n_latents, 128, 64, 32, 1]
        layers = []
        self.increase_latent_dim = torch.nn.Linear(n_latents, n_latents * 2 * 2)
        self.make_2x2_images = torch.nn.Unflatten(-1, (n_latents, 2, 2))
        self.merge_batch_dim_and_replica_dim = torch.nn.Flatten(start_dim=0, end_dim=1)
        for i in range(len(channels) - 1):
            layers.append(torch.nn.ConvTranspose2d(channels[i], channels[i + 1], kernel_size=3, stride=1, padding=1))
            layers.append(torch.nn.BatchNorm2d(channels[i + 1]))
            layers.append(torch.nn.Dropout2d(0.2))
            layers.append(torch.nn.Upsample(scale_factor=2))
            layers.append(torch.nn.LeakyReLU())
        layers.append(torch.nn.ConvTranspose2d(channels[-1], channels[-1], kernel_size=3, stride=1, padding=1))
        self.convtrans = torch.nn.Sequential(*layers)
        # parity hook: Dropout2d keep-masks [(N,128),(N,64),(N,32),(N,1)] consumed by the next forward
        self._injected_masks: Optional[List[torch.Tensor]] = None
        self.dropout_seed = 0
        self._dropout_calls = 0
        self._defer_join = False     # see _DecoderFn.backward; set and cleared by ModelWrapper around its own step
        self._deferred_keep = None   # tensors the library's side stream may still touch until the deferred join
        self._prepared = None        # see prepare(): the prologue of the next forward, already enqueued

    def inject_dropout_masks(self, masks: Optional[List[torch.Tensor]]):
        """Use these keep-masks for the next training forward instead of the device RNG."""
        self._injected_masks = masks

    def _trainable(self) -> List[torch.Tensor]:
        out = [self.increase_latent_dim.weight, self.increase_latent_dim.bias]
        for l in range(4):
            conv, bn = self.convtrans[5 * l], self.convtrans[5 * l + 1]
            out += [conv.weight, conv.bias, bn.weight, bn.bias]
        last = self.convtrans[20]
        return out + [last.weight, last.bias]

    def _native_struct(self, params: Sequence[torch.Tensor]) -> DecoderParams:
        for k, t in enumerate(params):
            _check_param(t, f"decoder parameter {k}")
        st = DecoderParams()
        st.lin_w, st.lin_b = params[0].data_ptr(), params[1].data_ptr()
        for l in range(4):
            bn = self.convtrans[5 * l + 1]
            st.conv_w[l], st.conv_b[l] = params[2 + 4 * l].data_ptr(), params[3 + 4 * l].data_ptr()
            st.bn_g[l], st.bn_b[l] = params[4 + 4 * l].data_ptr(), params[5 + 4 * l].data_ptr()
            st.bn_rm[l], st.bn_rv[l] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            st.bn_nbt[l] = bn.num_batches_tracked.data_ptr()
        st.conv_w[4], st.conv_b[4] = params[18].data_ptr(), params[19].data_ptr()
        return st

    def prepare(self, N: int, stream: "torch.cuda.Stream") -> None:
        """Enqueues, on ``stream``, the part of the next ``forward`` over ``N = B * R`` latent rows that needs the
        parameters and the dropout stream only (weight packs, composed weights, keep-masks: ``dvg_decoder_prepare``).
        ``stream`` must already be ordered behind the last parameter update and behind whatever used the memory before
        (``stream.wait_stream(main)`` where the step starts); the forward call joins it.  Without a matching forward the
        preparation is dropped."""
        if self._injected_masks is not None:
            return
        params = self._trainable()
        dev = params[0].device
        if dev.type != "cuda":
            return
        L = lib()
        n, training = self.n_latents, bool(self.training)
        st = self._native_struct(params)
        ws = _ws(L.dvg_decoder_workspace_bytes(N, n), dev)  # (from the CURRENT stream's pool: that is where it lives on)
        seed, offset = int(self.dropout_seed), int(self._dropout_calls)
        with torch.cuda.device(dev):
            check(L.dvg_decoder_prepare(ctypes.byref(st), n, N, int(training), seed & (2**64 - 1), offset & (2**64 - 1),
                                        ws.data_ptr(), ws.numel(), _lib.DYN, stream.cuda_stream), "dvg_decoder_prepare")
        self._prepared = {"key": (N, n, training, seed, offset, dev), "params": tuple(params), "ws": ws, "st": st,
                          "stream": stream}

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        masks, self._injected_masks = self._injected_masks, None
        offset = self._dropout_calls
        if self.training:
            self._dropout_calls += 1
        return _DecoderFn.apply(x, self, masks, self.dropout_seed, offset, *self._trainable())

    def forward_mse(self, x: torch.Tensor, images: torch.Tensor) -> torch.Tensor:
        """``mse_loss(self(x), images.unsqueeze(1).repeat(1, R, ...))`` of a training step without the reconstruction
        ever being written (``_DecoderMseFn``): the same loss to rounding, the same gradients bit for bit, a fifth of the
        memory traffic of the network's tail.  Training mode; differentiate the returned loss directly."""
        masks, self._injected_masks = self._injected_masks, None
        offset = self._dropout_calls
        self._dropout_calls += 1
        return _DecoderMseFn.apply(x, images, self, masks, self.dropout_seed, offset, *self._trainable())
