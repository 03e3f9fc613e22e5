"""Development access to the two MFMA GEMM kernels (include/dvg_dev.h) + layout helpers.
Used by kernel-level tests and micro-benchmarks; not part of the product surface."""
import ctypes

import numpy as np
import torch

from . import _lib

_SIGS = {
    "dvg_dev_conv_igemm": (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 4 +
                           [ctypes.c_int64] + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_void_p]),
    "dvg_dev_conv_splitk_floats": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "dvg_dev_conv_stats_blocks": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int]),
    "dvg_dev_conv_wino": (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 4 +
                          [ctypes.c_int64] + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "dvg_dev_conv_wino4": (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 4 +
                           [ctypes.c_int64] + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "dvg_dev_conv_wino4_shape": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "dvg_dev_wino4_wgrad_slab_floats": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "dvg_dev_conv_wino4_wgrad": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int64] + [ctypes.c_int] * 4 +
                                 [ctypes.c_void_p]),
    "dvg_dev_conv_wino_ok": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "dvg_dev_conv_wino_stats_blocks": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int]),
    "dvg_dev_wgrad_slab_floats": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "dvg_dev_conv_wgrad": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int64] + [ctypes.c_int] * 5 +
                           [ctypes.c_void_p]),
    "dvg_dev_encoder_layout": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]),
    "dvg_dev_wino_wgrad_slab_floats": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "dvg_dev_conv_wino_wgrad": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int64] + [ctypes.c_int] * 5 +
                                [ctypes.c_void_p]),
}
_bound = False


def lib():
    global _bound
    L = _lib.lib()
    if not _bound:
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _bound = True
    return L


def morton_perm(side: int) -> torch.Tensor:
    """perm[m] = row-major index (y*side + x) of the pixel stored at Morton position m."""
    def compact(v):
        v &= 0x55555555
        v = (v | (v >> 1)) & 0x33333333
        v = (v | (v >> 2)) & 0x0F0F0F0F
        v = (v | (v >> 4)) & 0x00FF00FF
        return v
    m = np.arange(side * side)
    return torch.from_numpy((compact(m >> 1) * side + compact(m)).astype(np.int64))


def nchw_to_morton(x: torch.Tensor) -> torch.Tensor:
    """(N,C,H,W) -> (N*H*W, C) NHWC with Morton-ordered pixels."""
    N, C, H, W = x.shape
    perm = morton_perm(H).to(x.device)
    return x.permute(0, 2, 3, 1).reshape(N, H * W, C)[:, perm, :].reshape(N * H * W, C).contiguous()


def morton_to_nchw(t: torch.Tensor, N: int, C: int, side: int) -> torch.Tensor:
    perm = morton_perm(side).to(t.device)
    out = torch.empty((N, side * side, C), dtype=t.dtype, device=t.device)
    out[:, perm, :] = t.reshape(N, side * side, C)
    return out.reshape(N, side, side, C).permute(0, 3, 1, 2).contiguous()


def conv_igemm(x_m, w, mode, M, Cin, Cout, L, ntaps=9, ups=0, poolsum=0, bias=None, stats=False, wp=None, repack=True,
               splitk=True):
    Lb = lib()
    dev = x_m.device
    wp = wp if wp is not None else torch.empty((ntaps * Cin * Cout * 3 + 1) // 2, device=dev)  # (up to three bf16 planes)
    out = torch.empty(((M // 4) if poolsum else M, Cout), device=dev)
    st = torch.empty((Lb.dvg_dev_conv_stats_blocks(M, Cout), Cout, 2), device=dev) if stats else None
    nsk = Lb.dvg_dev_conv_splitk_floats(M, Cin, Cout, ntaps, poolsum) if splitk else 0
    sk = torch.empty(nsk, device=dev) if nsk else None
    _lib.check(Lb.dvg_dev_conv_igemm(x_m.data_ptr(), w.data_ptr(), mode, wp.data_ptr(), _lib.ptr(bias), out.data_ptr(),
                                     _lib.ptr(st), M, Cin, Cout, L, ntaps, ups, poolsum, int(repack), _lib.ptr(sk),
                                     _lib.stream_ptr(dev)))
    return (out, st) if stats else out


def conv_wino_ok(M, Cin, Cout, L):
    return bool(lib().dvg_dev_conv_wino_ok(M, Cin, Cout, L))


def conv_wino(x_m, w, mode, M, Cin, Cout, L, bias=None, stats=False):
    """The stride-1 3x3 layer of conv_igemm (modes 0 / 1) in the Winograd F(2x2,3x3) form (csrc/conv_wino.hip)."""
    Lb = lib()
    dev = x_m.device
    u = torch.empty(16 * Cin * Cout, device=dev)
    out = torch.empty((M, Cout), device=dev)
    st = torch.empty((Lb.dvg_dev_conv_wino_stats_blocks(M, Cout), Cout, 2), device=dev) if stats else None
    _lib.check(Lb.dvg_dev_conv_wino(x_m.data_ptr(), w.data_ptr(), mode, u.data_ptr(), _lib.ptr(bias), out.data_ptr(),
                                    _lib.ptr(st), M, Cin, Cout, L, _lib.stream_ptr(dev)))
    return (out, st) if stats else out


def conv_wino4_shape(M, Cin, Cout, L):
    return bool(lib().dvg_dev_conv_wino4_shape(M, Cin, Cout, L))


def conv_wino4(x_m, w, mode, M, Cin, Cout, L, bias=None, stats=False, cus=0, um=0):
    """The same layer in the Winograd F(4x4,3x3) form (csrc/conv_wino4.hip).  ``um`` = 1: the Upsample(x2) + 3x3 forward
    (``x_m`` = the source map, M / L of the output grid; 25 of the 36 positions); ``um`` = 2: its data gradient (``x_m`` = the
    fine-grid gradient, the result = the source map's gradient, M / 4 rows)."""
    Lb = lib()
    dev = x_m.device
    u = torch.empty(36 * Cin * Cout, device=dev)
    out = torch.empty((M // 4 if um == 2 else M, Cout), device=dev)
    st = torch.empty((M // 1024, Cout, 2), device=dev) if stats else None
    _lib.check(Lb.dvg_dev_conv_wino4(x_m.data_ptr(), w.data_ptr(), mode, u.data_ptr(), _lib.ptr(bias), out.data_ptr(),
                                     _lib.ptr(st), M, Cin, Cout, L, int(cus), int(um), _lib.stream_ptr(dev)))
    return (out, st) if stats else out


def conv_wgrad(x_m, dy, mode, w_shape, M, Cin, Cout, L, ntaps=9, ups=0):
    Lb = lib()
    dev = x_m.device
    slabs = torch.empty(Lb.dvg_dev_wgrad_slab_floats(M, Cin, Cout, ntaps), device=dev)
    gw = torch.empty(w_shape, device=dev)
    _lib.check(Lb.dvg_dev_conv_wgrad(x_m.data_ptr(), dy.data_ptr(), slabs.data_ptr(), gw.data_ptr(), mode, M, Cin, Cout, L,
                                     ntaps, ups, _lib.stream_ptr(dev)))
    return gw


def conv_wino_wgrad(x_m, dy, mode, w_shape, M, Cin, Cout, L, ups=0, cus=0):
    """The 3x3 weight gradient in the Winograd form (csrc/conv_wino_wgrad.hip), or None when the shape does not qualify.
    ``cus``: CUs the persistent grid is sized for (0: the budget the training step gives the launch)."""
    Lb = lib()
    dev = x_m.device
    nf = Lb.dvg_dev_wino_wgrad_slab_floats(M, Cin, Cout, L)
    if nf == 0:
        return None
    slabs = torch.empty(nf, device=dev)
    gw = torch.empty(w_shape, device=dev)
    _lib.check(Lb.dvg_dev_conv_wino_wgrad(x_m.data_ptr(), dy.data_ptr(), slabs.data_ptr(), gw.data_ptr(), mode, M, Cin, Cout, L,
                                          int(ups), int(cus), _lib.stream_ptr(dev)))
    return gw


def encoder_saved(ws: torch.Tensor, B: int, n: int):
    """Views into an encoder workspace after a forward call: per layer the pre-BatchNorm convolution output as NCHW, the
    batch mean and the batch inverse standard deviation (diagnostics).  The offsets are the TRAINING plan's (an evaluation-
    mode forward may plan its packs differently: do not use this after one).  Layer 0's ``Y`` is ``None`` under the
    default ``enc_l0_fused`` != 0: that output is recomputed by every pass that needs it and never written."""
    off = (ctypes.c_size_t * 16)()
    _lib.check(lib().dvg_dev_encoder_layout(B, n, off), "dvg_dev_encoder_layout")
    f = ws.view(torch.float32)
    ch = [32, 64, 128, n]
    out = []
    for l in range(4):
        side, C = 32 >> l, ch[l]
        y = f[off[l]: off[l] + B * side * side * C]
        stored = l > 0 or _lib.get_option("enc_l0_fused") == 0
        out.append(dict(Y=morton_to_nchw(y, B, C, side) if stored else None, mean=f[off[8 + l]: off[8 + l] + C], invstd=f[off[12 + l]: off[12 + l] + C]))
    return out


def conv_wino4_wgrad(x_m, dy, mode, w_shape, M, Cin, Cout, L, cus=0):
    """The 3x3 weight gradient in the Winograd F(4x4,3x3) form (csrc/conv_wino4_wgrad.hip), or None when the shape does not
    qualify.  ``cus``: CUs the grid is sized for (0: the budget the training step gives the launch)."""
    Lb = lib()
    dev = x_m.device
    nf = Lb.dvg_dev_wino4_wgrad_slab_floats(M, Cin, Cout, L)
    if nf == 0:
        return None
    slabs = torch.empty(nf, device=dev)
    gw = torch.empty(w_shape, device=dev)
    _lib.check(Lb.dvg_dev_conv_wino4_wgrad(x_m.data_ptr(), dy.data_ptr(), slabs.data_ptr(), gw.data_ptr(), mode, M, Cin, Cout, L,
                                           int(cus), _lib.stream_ptr(dev)))
    return gw
