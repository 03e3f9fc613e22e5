"""Input pipeline of the training path.

The reference trains on MNIST resized to 32x32 and rounded to {0,1}
(/root/reference/src/model_wrapper.py:70-103: ``Resize((32,32))``, ``ToTensor()``, ``round``;
``DataLoader(shuffle=True, drop_last=True)``).  This module provides the same batches from
(a) MNIST IDX files when they are present locally (no network here): the raw uint8 images go to the
device once and ``dvg_resize_binarise`` applies the reference's transform there, bit for bit what
torchvision's Resize does to the PIL images (Pillow's two-pass fixed-point BILINEAR; oracle/resize.py
is pinned against Pillow); or (b) a synthetic stand-in of the same shape and ink fraction.  The data
set stays resident in HBM and every shuffled mini-batch is one ``dvg_gather_rows`` launch.
"""
from __future__ import annotations

import gzip
import os
import struct
from typing import Optional

import numpy as np
import torch


class TensorBatches:
    """Minimal DataLoader stand-in: shuffled, ``drop_last`` batches of a device-resident tensor.

    Data parallelism: every rank draws the SAME permutation per epoch (same seed) and takes every ``world_size``-th
    entry starting at ``rank`` -- disjoint shards, ``(n // world_size) // batch_size`` batches per rank and epoch."""

    def __init__(self, images: torch.Tensor, labels: torch.Tensor, batch_size: int, shuffle: bool = True, seed: int = 0,
                 rank: int = 0, world_size: int = 1):
        self.images, self.labels, self.batch_size, self.shuffle = images, labels, int(batch_size), shuffle
        self.rank, self.world_size = int(rank), max(1, int(world_size))
        self._gen = torch.Generator().manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)

    def __len__(self):
        return (self.images.shape[0] // self.world_size) // self.batch_size

    def _shard_order(self, gen):
        n = self.images.shape[0]
        order = torch.randperm(n, generator=gen) if self.shuffle else torch.arange(n)
        order = order[self.rank::self.world_size][: (n // self.world_size)]
        return order.to(self.images.device)

    def __iter__(self):
        order = self._shard_order(self._gen)
        for k in range(len(self)):
            idx = order[k * self.batch_size: (k + 1) * self.batch_size]
            yield gather_rows(self.images, idx), self.labels[idx]

    def preview_batch(self):
        """The first batch the NEXT ``iter()`` of this rank will yield, drawn from a CLONE of the permutation generator:
        the training iterator's stream is not advanced.  (The reference previews with ``next(iter(dataloader))``,
        /root/reference/src/model_wrapper.py:459; here the epoch-end preview runs on rank 0 only, and a permutation
        consumed by one rank alone would leave the ranks sharding DIFFERENT permutations from the next epoch on.)"""
        gen = torch.Generator()
        gen.set_state(self._gen.get_state())
        idx = self._shard_order(gen)[: self.batch_size]
        return gather_rows(self.images, idx), self.labels[idx]


def preview_batch(dataloader):
    """First batch of ``dataloader`` for the epoch-end reconstruction figure, without side effects on a
    :class:`TensorBatches` (any other iterable: ``next(iter(...))`` as the reference does)."""
    if isinstance(dataloader, TensorBatches):
        return dataloader.preview_batch()
    return next(iter(dataloader))


def gather_rows(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """``table[idx]`` along dim 0.  Device-resident float32 tables take the library's gather kernel (one launch, 16-byte
    copies); anything else (CPU-side construction in tests) plain indexing."""
    if not (table.is_cuda and table.dtype == torch.float32 and table.is_contiguous()):
        return table[idx]
    from . import _lib

    idx = idx.to(device=table.device, dtype=torch.int64).contiguous()
    out = torch.empty((idx.numel(),) + tuple(table.shape[1:]), dtype=torch.float32, device=table.device)
    row = int(table[0].numel()) if table.shape[0] else 0
    if idx.numel() and row:
        with torch.cuda.device(table.device):
            _lib.check(_lib.lib().dvg_gather_rows(table.data_ptr(), table.shape[0], row, idx.data_ptr(), idx.numel(),
                                                  out.data_ptr(), None, _lib.stream_ptr(table.device)), "dvg_gather_rows")
    return out


def resize_binarise(raw: torch.Tensor, image_size: int) -> torch.Tensor:
    """(N, H, H) uint8 on the device -> (N, 1, S, S) float32 in {0, 1}: the reference's Resize -> ToTensor -> round."""
    from . import _lib

    if not raw.is_cuda or raw.dtype != torch.uint8 or raw.dim() != 3 or raw.shape[1] != raw.shape[2]:
        raise _lib.DvgError("resize_binarise needs a (N, H, H) uint8 CUDA tensor; there is no CPU fallback")
    raw = raw.contiguous()
    out = torch.empty((raw.shape[0], 1, image_size, image_size), dtype=torch.float32, device=raw.device)
    with torch.cuda.device(raw.device):
        _lib.check(_lib.lib().dvg_resize_binarise(raw.data_ptr(), raw.shape[0], raw.shape[1], image_size, out.data_ptr(),
                                                  _lib.stream_ptr(raw.device)), "dvg_resize_binarise")
    return out


def synthetic_images(count: int, seed: int, ink: float = 0.13, device="cpu") -> torch.Tensor:
    """(count,1,32,32) float32 in {0,1}: i.i.d. Bernoulli(ink), MNIST's ink fraction."""
    g = torch.Generator().manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
    return (torch.rand((count, 1, 32, 32), generator=g) < ink).to(torch.float32).to(device)


def _read_idx_images(path: str) -> np.ndarray:
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        magic, n, h, w = struct.unpack(">IIII", f.read(16))
        if magic != 2051:
            raise ValueError(f"{path}: not an IDX image file")
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(n, h, w)


def load_mnist_raw(root: str = "data") -> Optional[torch.Tensor]:
    """MNIST train images as stored: (60000, 28, 28) uint8, or None when the IDX files are not on disk."""
    for name in ("train-images-idx3-ubyte", "train-images-idx3-ubyte.gz"):
        for sub in ("MNIST/raw", ""):
            path = os.path.join(root, sub, name)
            if os.path.exists(path):
                return torch.from_numpy(_read_idx_images(path).copy())
    return None


def load_mnist(root: str = "data", image_size: int = 32, device="cuda") -> Optional[torch.Tensor]:
    """MNIST train images -> (60000, 1, S, S) float32 {0, 1} on ``device`` (the transform runs there), or None."""
    raw = load_mnist_raw(root)
    return None if raw is None else resize_binarise(raw.to(device), image_size)


def random_subset_indices(n: int, k: int, seed: Optional[int] = None) -> torch.Tensor:
    """``DATASET_SIZE`` as the reference applies it (/root/reference/src/model_wrapper.py:96-100):
    ``torch.utils.data.random_split(dataset, [k, n - k])[0]`` -- the first ``k`` entries of ONE ``randperm(n)``, a random
    subset, not the first ``k`` images.  ``seed=None`` draws from torch's global generator exactly as the reference does
    (the same call, so the same subset after the same seeding); data-parallel runs pass the run's seed so that every
    rank holds the same subset whatever its generator has consumed."""
    if not 0 < k <= n:
        raise ValueError(f"DATASET_SIZE={k} must be in 1..{n}")
    gen = None if seed is None else torch.Generator().manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
    subset = torch.utils.data.random_split(range(n), [k, n - k], generator=gen) if gen is not None else \
        torch.utils.data.random_split(range(n), [k, n - k])
    return torch.tensor(subset[0].indices, dtype=torch.int64)


def get_dataloader(image_size: int, batch_size: int, dataset_size: Optional[int] = None, seed: int = 0,
                   device=None, root: str = "data", rank: int = 0, world_size: int = 1) -> TensorBatches:
    device = device or ("cuda" if torch.cuda.is_available() else "cpu")
    images = load_mnist(root, image_size, device) if torch.device(device).type == "cuda" else None
    if images is None:
        images = synthetic_images(60000, seed)
    if dataset_size:
        images = images[random_subset_indices(images.shape[0], int(dataset_size), seed if world_size > 1 else None)]
    labels = torch.zeros(images.shape[0], dtype=torch.int64)
    return TensorBatches(images.to(device), labels.to(device), batch_size, shuffle=True, seed=seed, rank=rank,
                         world_size=world_size)
