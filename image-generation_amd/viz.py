"""Host-side picture helpers of the generation entry points (``ModelWrapper.generate_output`` /
``generate_reconstucted_samples``): the image grid the reference builds with ``torchvision.utils.make_grid``
(/root/reference/src/model_wrapper.py:387, :467-474; torchvision is not in this image) and its ``sharpen`` rule
(:382-385, :476-479).  Not on the training path.
"""
from __future__ import annotations

import math

import torch


def _thresholds():
    """The reference keeps the two sharpening thresholds in its UI configuration (/root/reference/demo_configs.py:62-63);
    inside the reference's tree that module is importable and wins, elsewhere its shipped values apply."""
    try:
        from demo_configs import LOWER_THRESHOLD as lo, UPPER_THRESHOLD as up  # type: ignore

        return float(lo), float(up)
    except Exception:
        return 0.4, 0.6


LOWER_THRESHOLD, UPPER_THRESHOLD = _thresholds()


def sharpen(images: torch.Tensor, lower: float = LOWER_THRESHOLD, upper: float = UPPER_THRESHOLD) -> torch.Tensor:
    """``over = H(x - upper)``, ``under = H(x - lower)`` with ``H(0) = 0``; ``(over + |over - 1| x) under``: pixels above
    ``upper`` become 1, pixels not above ``lower`` become 0, the rest stay."""
    over = (images > upper).to(images.dtype)
    under = (images > lower).to(images.dtype)
    return (over + (1 - over) * images) * under


@torch.no_grad()  # (as torchvision's: the reference hands it tensors that still require grad, model_wrapper.py:467)
def make_grid(images: torch.Tensor, nrow: int = 8, padding: int = 2, pad_value: float = 0.0) -> torch.Tensor:
    """(N, C, H, W) -> (3, rows * (H + padding) + padding, cols * (W + padding) + padding): ``nrow`` images per row,
    single-channel images replicated to three channels, cells separated (and framed) by ``padding`` pixels of
    ``pad_value`` -- the layout of ``torchvision.utils.make_grid`` with its default arguments otherwise."""
    if images.dim() != 4:
        raise ValueError("make_grid expects (N, C, H, W)")
    if images.shape[1] == 1:
        images = images.expand(-1, 3, -1, -1)
    n, c, h, w = images.shape
    if n == 1:
        return images[0].clone()
    cols = min(int(nrow), n)
    rows = int(math.ceil(n / cols))
    ch, cw = h + padding, w + padding
    grid = images.new_full((c, ch * rows + padding, cw * cols + padding), pad_value)
    k = 0
    for r in range(rows):
        for q in range(cols):
            if k >= n:
                break
            grid[:, r * ch + padding: r * ch + padding + h, q * cw + padding: q * cw + padding + w] = images[k]
            k += 1
    return grid
