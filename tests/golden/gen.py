"""Deterministic parameter / input generators shared by make_golden.py and the tests.

Data generators only (numpy RNG); no reference code.
"""
import numpy as np


def _shapes_encoder(n):
    ch = [1, 32, 64, 128, n]
    out = []
    for l in range(4):
        ci = 4 * l
        out += [
            (f"conv.{ci}.weight", (ch[l + 1], ch[l], 3, 3), "w"),
            (f"conv.{ci}.bias", (ch[l + 1],), "b"),
            (f"conv.{ci + 1}.weight", (ch[l + 1],), "g"),
            (f"conv.{ci + 1}.bias", (ch[l + 1],), "b"),
            (f"conv.{ci + 1}.running_mean", (ch[l + 1],), "b"),
            (f"conv.{ci + 1}.running_var", (ch[l + 1],), "v"),
            (f"conv.{ci + 1}.num_batches_tracked", (), "i"),
        ]
    out += [("projection.weight", (1, 4), "w"), ("projection.bias", (1,), "b")]
    return out


def _shapes_decoder(n):
    ch = [n, 128, 64, 32, 1]
    out = [("increase_latent_dim.weight", (4 * n, n), "w"), ("increase_latent_dim.bias", (4 * n,), "b")]
    for l in range(4):
        ci = 5 * l
        out += [
            (f"convtrans.{ci}.weight", (ch[l], ch[l + 1], 3, 3), "wt"),
            (f"convtrans.{ci}.bias", (ch[l + 1],), "b"),
            (f"convtrans.{ci + 1}.weight", (ch[l + 1],), "g"),
            (f"convtrans.{ci + 1}.bias", (ch[l + 1],), "b"),
            (f"convtrans.{ci + 1}.running_mean", (ch[l + 1],), "b"),
            (f"convtrans.{ci + 1}.running_var", (ch[l + 1],), "v"),
            (f"convtrans.{ci + 1}.num_batches_tracked", (), "i"),
        ]
    out += [("convtrans.20.weight", (1, 1, 3, 3), "wt"), ("convtrans.20.bias", (1,), "b")]
    return out


def make_params(n, which, seed):
    """dict name -> np.ndarray (float32 / int64) for the encoder or decoder."""
    rng = np.random.default_rng(seed)
    shapes = _shapes_encoder(n) if which == "encoder" else _shapes_decoder(n)
    p = {}
    for name, shape, kind in shapes:
        if kind == "w":
            fan_in = int(np.prod(shape[1:]))
            p[name] = (rng.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
        elif kind == "wt":
            fan_in = int(shape[0] * shape[2] * shape[3])
            p[name] = (rng.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
        elif kind == "b":
            p[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif kind == "g":
            p[name] = (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif kind == "v":
            p[name] = (1.0 + 0.1 * np.abs(rng.standard_normal(shape))).astype(np.float32)
        else:
            p[name] = np.asarray(3, dtype=np.int64)
    return p


def make_images(B, seed, p=0.13):
    rng = np.random.default_rng(seed)
    return (rng.random((B, 1, 32, 32)) < p).astype(np.float32)


def make_spins(B, R, n, seed):
    rng = np.random.default_rng(seed)
    return np.where(rng.random((B, R, n)) < 0.5, -1.0, 1.0).astype(np.float32)


def make_masks(N, seed, keep=0.8):
    rng = np.random.default_rng(seed)
    return [(rng.random((N, c)) < keep).astype(np.float32) for c in (128, 64, 32, 1)]


def subsample(a, stride=97):
    return np.ascontiguousarray(a.reshape(-1)[::stride])


def figure_image(fig):
    """The picture inside a ``px.imshow`` figure as a uint8 (H, W, 3) array: plotly stores RGB pictures as a PNG data URI
    (``source``), small ones as an array (``z``)."""
    import base64
    import io

    tr = fig.data[0] if hasattr(fig, "data") else fig["data"][0]
    get = (lambda k: getattr(tr, k, None)) if hasattr(tr, "type") else (lambda k: tr.get(k))
    if get("z") is not None:
        z = np.asarray(get("z"), dtype=np.float64)
        return np.clip(np.rint(z * (255.0 if z.max() <= 1.0 else 1.0)), 0, 255).astype(np.uint8)
    from PIL import Image

    raw = base64.b64decode(get("source").split(",", 1)[1])
    return np.asarray(Image.open(io.BytesIO(raw)).convert("RGB"), dtype=np.uint8)
