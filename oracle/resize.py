"""CPU restatement of the reference's image transform ``Resize((S, S)) -> ToTensor() -> round``
(/root/reference/src/model_wrapper.py:70-77).  Test infrastructure (see oracle/__init__.py).

torchvision's ``Resize`` on the PIL images MNIST yields is ``PIL.Image.resize(size, BILINEAR)``; PIL and
torchvision are third-party dependencies of the reference (``requirements.txt``), torchvision is absent from this
image, Pillow 12.2 is present.  This file restates Pillow's published algorithm (``src/libImaging/Resample.c``:
``precompute_coeffs``, ``normalize_coeffs_8bpc``, ``ImagingResampleHorizontal_8bpc`` / ``Vertical``: two separable
passes with an 8-bit intermediate image and 22-bit fixed-point coefficients) in numpy; it is pinned against Pillow
itself by tests/test_oracle_resize.py (live, when Pillow imports) and by tests/golden/resize_pil.npz (Pillow's own
outputs on committed inputs).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bilinear_coefs(in_size: int, out_size: int):
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, dtype=np.int64)
    xcnt = np.zeros(out_size, dtype=np.int64)
    k = np.zeros((out_size, ksize), dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        n = hi - lo
        w = [max(0.0, 1.0 - abs((x + lo - center + 0.5) * ss)) for x in range(n)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(n):
            v = w[x] / ww if ww != 0.0 else w[x]
            k[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        xmin[xx], xcnt[xx] = lo, n
    return xmin, xcnt, k


def _pass(img: np.ndarray, xmin, xcnt, k) -> np.ndarray:
    """Resamples the LAST axis of a (..., in) uint8 array."""
    out = np.empty(img.shape[:-1] + (len(xmin),), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(len(xmin)):
        acc = np.full(img.shape[:-1], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for t in range(int(xcnt[xx])):
            acc += src[..., xmin[xx] + t] * k[xx, t]
        out[..., xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bilinear_u8(images: np.ndarray, out_size: int) -> np.ndarray:
    """(N, H, W) uint8 -> (N, out, out) uint8: horizontal pass, then vertical pass, as Pillow orders them."""
    images = np.asarray(images, dtype=np.uint8)
    n, h, w = images.shape
    hx = _pass(images, *bilinear_coefs(w, out_size))                      # (N, H, out)
    vy = _pass(np.swapaxes(hx, 1, 2), *bilinear_coefs(h, out_size))       # (N, out, out_y) on the transposed image
    return np.ascontiguousarray(np.swapaxes(vy, 1, 2))


def resize_binarise(images: np.ndarray, out_size: int) -> np.ndarray:
    """The whole transform: (N, H, W) uint8 -> (N, 1, out, out) float32 in {0, 1} (``round(v / 255)`` in float32)."""
    r = resize_bilinear_u8(images, out_size)
    return np.round(r.astype(np.float32) / np.float32(255.0))[:, None].astype(np.float32)
