"""Input pipeline of the training path.

The reference trains on MNIST resized to 32x32 and rounded to {0,1}
(/root/reference/src/model_wrapper.py:70-103: ``Resize((32,32))``, ``ToTensor()``, ``round``;
``DataLoader(shuffle=True, drop_last=True)``).  This module provides the same batches from
(a) MNIST IDX files when they are present locally (no network here), resized on the host with
the same antialiased bilinear rule torchvision applies, or (b) a synthetic stand-in of the same
shape and ink fraction, resident on the device.
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

    def __iter__(self):
        n = self.images.shape[0]
        order = torch.randperm(n, generator=self._gen) if self.shuffle else torch.arange(n)
        order = order[self.rank::self.world_size][: (n // self.world_size)]
        order = order.to(self.images.device)
        for k in range(len(self)):
            idx = order[k * self.batch_size: (k + 1) * self.batch_size]
            yield self.images[idx], self.labels[idx]


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


def load_mnist(root: str = "data", image_size: int = 32) -> Optional[torch.Tensor]:
    """MNIST train images -> (60000,1,S,S) float {0,1}, or None when the files are not on disk."""
    for name in ("train-images-idx3-ubyte", "train-images-idx3-ubyte.gz"):
        for sub in ("MNIST/raw", ""):
            path = os.path.join(root, sub, name)
            if os.path.exists(path):
                raw = torch.from_numpy(_read_idx_images(path).copy()).unsqueeze(1)  # uint8, as PIL would hold it
                # torchvision Resize on a PIL image = antialiased bilinear on uint8, then ToTensor (/255), then round
                up = torch.nn.functional.interpolate(raw.float(), size=(image_size, image_size), mode="bilinear",
                                                     antialias=True, align_corners=False)
                return torch.round(up.round().clamp(0, 255) / 255.0)
    return None


def get_dataloader(image_size: int, batch_size: int, dataset_size: Optional[int] = None, seed: int = 0,
                   device=None, root: str = "data", rank: int = 0, world_size: int = 1) -> TensorBatches:
    device = device or ("cuda" if torch.cuda.is_available() else "cpu")
    images = load_mnist(root, image_size)
    if images is None:
        images = synthetic_images(dataset_size or 60000, seed)
    if dataset_size:
        images = images[:dataset_size]
    labels = torch.zeros(images.shape[0], dtype=torch.int64)
    return TensorBatches(images.to(device), labels.to(device), batch_size, shuffle=True, seed=seed, rank=rank,
                         world_size=world_size)
