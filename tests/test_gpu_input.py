"""Device-side input pipeline: ``dvg_resize_binarise`` against the oracle restatement of the
reference's Resize -> ToTensor -> round transform (itself pinned against Pillow: tests/test_oracle_resize.py), bit for
bit; ``dvg_gather_rows`` = ``table[idx]``; the IDX-file path of ``data.get_dataloader``."""
import gzip
import os
import struct

import numpy as np
import pytest
import torch

from image_generation_amd import _lib, data
from oracle import resize

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,a,b", [(64, 28, 32), (1, 28, 32), (5000, 28, 32), (9, 16, 32), (7, 28, 20), (3, 32, 32), (4, 64, 24)])
def test_resize_binarise_is_bit_exact(golden_dir, n, a, b):
    if (n, a, b) == (64, 28, 32):  # the committed Pillow fixture
        fx = np.load(os.path.join(golden_dir, "resize_pil.npz"))
        src, want = fx["src"], (fx["pil32"] >= 128).astype(np.float32)[:, None]
    else:
        src = np.random.default_rng(n + a).integers(0, 256, (n, a, a), dtype=np.uint8)
        want = resize.resize_binarise(src, b)
    got = data.resize_binarise(torch.from_numpy(src).cuda(), b)
    assert got.shape == (n, 1, b, b) and got.dtype == torch.float32
    assert np.array_equal(got.cpu().numpy(), want)


def test_resize_rejects_what_it_cannot_serve():
    with pytest.raises(_lib.DvgError):
        data.resize_binarise(torch.zeros(2, 28, 28, dtype=torch.uint8), 32)  # CPU tensor: no fallback
    with pytest.raises(_lib.DvgError):
        data.resize_binarise(torch.zeros(2, 28, 28, dtype=torch.uint8).cuda(), 128)  # side > 64


def test_gather_rows_and_batches():
    g = torch.Generator().manual_seed(0)
    table = torch.rand(1000, 1, 32, 32, generator=g).cuda()
    idx = torch.randint(0, 1000, (300,), generator=g)
    assert torch.equal(data.gather_rows(table, idx), table[idx.cuda()])
    odd = torch.rand(37, 7, generator=g).cuda()  # rows that are no multiple of 16 bytes
    assert torch.equal(data.gather_rows(odd, torch.arange(36, -1, -1)), odd.flip(0))
    # an out-of-range index is flagged, not dereferenced
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = torch.empty(2, 7, device="cuda")
    bad = torch.tensor([3, 99], dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib().dvg_gather_rows(odd.data_ptr(), 37, 7, bad.data_ptr(), 2, out.data_ptr(), flag.data_ptr(),
                                          _lib.stream_ptr(odd.device)), "dvg_gather_rows")
    assert int(flag.item()) == 1 and torch.equal(out[0], odd[3])
    # TensorBatches: shuffled, drop_last, every image at most once per epoch
    labels = torch.zeros(1000, dtype=torch.int64, device="cuda")
    tb = data.TensorBatches(table, labels, 64, seed=5)
    seen = torch.cat([b for b, _ in tb])
    assert seen.shape == (960, 1, 32, 32)
    keys = {tuple(r.flatten()[:4].tolist()) for r in seen.cpu()}
    assert len(keys) == 960


def test_idx_files_go_through_the_device_transform(tmp_path):
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 256, (200, 28, 28), dtype=np.uint8)
    os.makedirs(tmp_path / "MNIST" / "raw")
    with gzip.open(tmp_path / "MNIST" / "raw" / "train-images-idx3-ubyte.gz", "wb") as f:
        f.write(struct.pack(">IIII", 2051, 200, 28, 28) + raw.tobytes())
    # DATASET_SIZE is the reference's random_split subset (/root/reference/src/model_wrapper.py:96-100): the first 128
    # entries of one randperm(200) from torch's global generator
    torch.manual_seed(77)
    want_idx = torch.utils.data.random_split(range(200), [128, 72])[0].indices
    torch.manual_seed(77)
    dl = data.get_dataloader(32, 16, dataset_size=128, seed=1, device="cuda", root=str(tmp_path))
    assert len(dl) == 8 and dl.images.is_cuda and dl.images.shape == (128, 1, 32, 32)
    assert sorted(want_idx) != list(range(128))
    assert np.array_equal(dl.images.cpu().numpy(), resize.resize_binarise(raw[want_idx], 32))
