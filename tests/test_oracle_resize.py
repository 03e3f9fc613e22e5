"""oracle/resize.py (the reference's Resize -> ToTensor -> round transform, /root/reference/src/model_wrapper.py:70-77)
against Pillow: the committed fixture of Pillow's own outputs, and Pillow itself where it imports."""
import os

import numpy as np
import pytest

from oracle import resize


def test_restatement_equals_pillow_fixture(golden_dir):
    fx = np.load(os.path.join(golden_dir, "resize_pil.npz"))
    assert np.array_equal(resize.resize_bilinear_u8(fx["src"], 32), fx["pil32"])
    b = resize.resize_binarise(fx["src"], 32)
    assert b.shape == (fx["src"].shape[0], 1, 32, 32) and b.dtype == np.float32 and set(np.unique(b)) <= {0.0, 1.0}
    assert np.array_equal(b[:, 0] == 1.0, fx["pil32"] >= 128)


@pytest.mark.parametrize("a,b", [(28, 32), (28, 20), (16, 32), (32, 32), (28, 14), (5, 7)])
def test_restatement_equals_live_pillow(a, b):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(a * 100 + b)
    imgs = rng.integers(0, 256, (6, a, a), dtype=np.uint8)
    want = np.stack([np.asarray(Image.fromarray(x, mode="L").resize((b, b), Image.BILINEAR)) for x in imgs])
    assert np.array_equal(resize.resize_bilinear_u8(imgs, b), want)
