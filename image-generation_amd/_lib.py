"""ctypes binding of the C ABI in include/dvg.h (libdvg.so).

This is the stub a maintainer of the reference would add:
plain pointers and sizes, torch only supplies device memory and the stream.
There is no fallback: if the library is missing or a call fails, an exception
is raised.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int8, c_int32, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# DVG_LIBRARY: development override (A/B runs of two builds in one process launch); the product path is in-tree
LIB_PATH = os.environ.get("DVG_LIBRARY") or os.path.join(_HERE, "libdvg.so")

_lib = None


class DvgError(RuntimeError):
    pass


class MmdCfg(ctypes.Structure):
    _fields_ = [
        ("n_kernels", c_int32),
        ("factor", c_float),
        ("bandwidth", c_float),
        ("squared", c_int32),
        ("reduce_mean", c_int32),
        ("biased", c_int32),
    ]


class EncoderParams(ctypes.Structure):
    _fields_ = [
        ("conv_w", c_void_p * 4),
        ("conv_b", c_void_p * 4),
        ("bn_g", c_void_p * 4),
        ("bn_b", c_void_p * 4),
        ("bn_rm", c_void_p * 4),
        ("bn_rv", c_void_p * 4),
        ("bn_nbt", c_void_p * 4),
        ("proj_w", c_void_p),
        ("proj_b", c_void_p),
    ]


class EncoderGrads(ctypes.Structure):
    _fields_ = [
        ("conv_w", c_void_p * 4),
        ("conv_b", c_void_p * 4),
        ("bn_g", c_void_p * 4),
        ("bn_b", c_void_p * 4),
        ("proj_w", c_void_p),
        ("proj_b", c_void_p),
    ]


class DecoderParams(ctypes.Structure):
    _fields_ = [
        ("lin_w", c_void_p),
        ("lin_b", c_void_p),
        ("conv_w", c_void_p * 5),
        ("conv_b", c_void_p * 5),
        ("bn_g", c_void_p * 4),
        ("bn_b", c_void_p * 4),
        ("bn_rm", c_void_p * 4),
        ("bn_rv", c_void_p * 4),
        ("bn_nbt", c_void_p * 4),
    ]


class DecoderGrads(ctypes.Structure):
    _fields_ = [
        ("lin_w", c_void_p),
        ("lin_b", c_void_p),
        ("conv_w", c_void_p * 5),
        ("conv_b", c_void_p * 5),
        ("bn_g", c_void_p * 4),
        ("bn_b", c_void_p * 4),
    ]


# name -> (restype, argtypes); must list every symbol include/dvg.h declares
_I32P = POINTER(c_int32)
SIGNATURES = {
    "dvg_version": (c_int, []),
    "dvg_source_hash": (c_char_p, []),
    "dvg_last_error": (c_char_p, []),
    "dvg_graph_create": (c_int, [c_int, c_int, _I32P, _I32P, _I32P, _I32P, c_int, _I32P, _I32P, _I32P, POINTER(c_void_p)]),
    "dvg_graph_destroy": (c_int, [c_void_p]),
    "dvg_gibbs_sample": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_float, c_float, c_void_p, c_int,
         c_uint32, c_uint64, c_uint32, c_int, c_int, c_void_p, c_void_p, c_void_p],
    ),
    "dvg_gibbs_launch_info": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "dvg_grbm_energy": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dvg_grbm_suffstats_workspace_bytes": (c_size_t, [c_void_p]),
    "dvg_grbm_suffstats": (
        c_int,
        [c_void_p, c_void_p, c_int64, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p],
    ),
    "dvg_gumbel_fwd": (
        c_int,
        [c_void_p, c_int64, c_int, c_int, c_float, c_void_p, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "dvg_gumbel_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "dvg_gumbel_bwd2": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "dvg_scalar_add": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "dvg_heaviside_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "dvg_resize_binarise": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "dvg_gather_rows": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "dvg_mmd_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "dvg_mmd_spin_flops": (c_int, [c_int64, c_int64, c_int, POINTER(c_double), POINTER(c_double), POINTER(c_int)]),
    "dvg_mmd_fwd_bwd": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_int64, c_int, POINTER(MmdCfg), c_void_p, c_void_p, c_void_p, c_size_t, c_void_p],
    ),
    "dvg_encoder_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "dvg_encoder_fwd": (
        c_int,
        [POINTER(EncoderParams), c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_size_t, c_void_p],
    ),
    "dvg_encoder_bwd": (
        c_int,
        [POINTER(EncoderParams), c_int, c_void_p, c_int64, c_void_p, POINTER(EncoderGrads), c_void_p, c_size_t, c_void_p],
    ),
    "dvg_decoder_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "dvg_decoder_fwd": (
        c_int,
        [POINTER(DecoderParams), c_int, c_void_p, c_int64, c_int, POINTER(c_void_p), c_uint64, c_uint64, c_void_p,
         c_void_p, c_size_t, c_void_p, c_void_p],
    ),
    "dvg_decoder_prepare": (
        c_int,
        [POINTER(DecoderParams), c_int, c_int64, c_int, c_uint64, c_uint64, c_void_p, c_size_t, c_void_p, c_void_p],
    ),
    "dvg_decoder_fwd_ex": (
        c_int,
        [POINTER(DecoderParams), c_int, c_void_p, c_int64, c_int, POINTER(c_void_p), c_uint64, c_uint64, c_void_p,
         c_void_p, c_size_t, c_void_p, c_int, c_void_p],
    ),
    "dvg_decoder_bwd": (
        c_int,
        [POINTER(DecoderParams), c_int, c_void_p, c_int64, c_void_p, POINTER(DecoderGrads), c_void_p, c_void_p,
         c_size_t, c_void_p],
    ),
    "dvg_decoder_bwd_ex": (
        c_int,
        [POINTER(DecoderParams), c_int, c_void_p, c_int64, c_void_p, POINTER(DecoderGrads), c_void_p, c_void_p,
         c_size_t, c_int, c_void_p],
    ),
    "dvg_stream_join_side": (c_int, [c_void_p]),
    "dvg_decoder_fwd_mse_ex": (
        c_int,
        [POINTER(DecoderParams), c_int, c_void_p, c_int64, POINTER(c_void_p), c_uint64, c_uint64, c_void_p, c_int, c_float,
         c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_void_p],
    ),
    "dvg_decoder_bwd_mse_ex": (
        c_int,
        [POINTER(DecoderParams), c_int, c_void_p, c_int64, c_void_p, c_int, c_float, POINTER(DecoderGrads), c_void_p,
         c_void_p, c_size_t, c_int, c_void_p],
    ),
    "dvg_mse_workspace_bytes": (c_size_t, []),
    "dvg_mse_fwd_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dvg_adam_step": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float, c_int64,
         c_float, c_void_p, c_int, c_void_p],
    ),
    "dvg_stream_anchor": (c_int, [c_void_p]),
    "dvg_set_conv_precision": (c_int, [c_int]),
    "dvg_get_conv_precision": (c_int, []),
    "dvg_option_count": (c_int, []),
    "dvg_option_name": (c_char_p, [c_int]),
    "dvg_option_doc": (c_char_p, [c_int]),
    "dvg_set_option": (c_int, [c_char_p, c_int64]),
    "dvg_get_option": (c_int, [c_char_p, POINTER(c_int64)]),
    "dvg_reset_options": (c_int, []),
    "dvg_prof_enable": (c_int, [c_uint64]),
    "dvg_prof_reset": (c_int, []),
    "dvg_prof_num_kernels": (c_int, []),
    "dvg_prof_kernel_name": (c_char_p, [c_int]),
    "dvg_prof_query": (c_int, [c_int, POINTER(c_double), POINTER(c_int64)]),
    "dvg_prof_query_work": (c_int, [c_int, POINTER(c_double)]),
    "dvg_prof_query_share": (c_int, [c_int, POINTER(c_double)]),
}


# include/dvg_dev.h: the A/B references and test knobs (not part of the drop-in boundary; SIGNATURES covers exactly dvg.h)
DEV_OPTION_SIGNATURES = {
    "dvg_dev_option_count": (c_int, []),
    "dvg_dev_option_name": (c_char_p, [c_int]),
    "dvg_dev_option_doc": (c_char_p, [c_int]),
    "dvg_dev_set_option": (c_int, [c_char_p, c_int64]),
    "dvg_dev_get_option": (c_int, [c_char_p, POINTER(c_int64)]),
}


def build(verbose: bool = False) -> str:
    """Compile libdvg.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    cmd = ["make", "-C", _HERE, "-j8"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """The loaded library.  Raises if it has not been built: there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DvgError(
                f"{LIB_PATH} is missing: build it with `make -C {_HERE}` (or __graft_entry__.build()). "
                "The HIP library is required; there is no CPU fallback."
            )
        # torch bundles its own libamdhip64; import it FIRST so libdvg.so binds to the same HIP runtime
        # (two runtimes in one process do not share devices, streams or allocations)
        import torch  # noqa: F401

        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in list(SIGNATURES.items()) + list(DEV_OPTION_SIGNATURES.items()):
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


# Device address of the active dvg_step_state_t, or None.  Set by model_wrapper while a training step is being
# captured into / replayed from a hipGraph: the wrappers then hand it to the kernels that take per-step scalars.
DYN = None


class StepState:
    """Host mirror + device copy of ``dvg_step_state_t`` (include/dvg.h)."""

    FORMAT = "<IIQQ2f2f"  # sweep0, reserved, gumbel_offset, dropout_offset, adam_step_size[2], adam_bc2_sqrt[2]
    RING = 16

    def __init__(self, device):
        import struct

        import torch

        self._struct = struct.Struct(self.FORMAT)
        # The host may run several replays ahead of the device: each write stages through its own pinned slot, and a
        # slot is rewritten only after the async copy that read it has executed (event per slot).
        self._ring = [torch.zeros(self._struct.size, dtype=torch.uint8).pin_memory() for _ in range(self.RING)]
        self._events = [None] * self.RING
        self._n = 0
        self.dev = torch.zeros(self._struct.size, dtype=torch.uint8, device=device)

    @property
    def ptr(self) -> int:
        return self.dev.data_ptr()

    def write(self, sweep0, gumbel_offset, dropout_offset, step_size, bc2_sqrt):
        raw = self._struct.pack(int(sweep0) & 0xFFFFFFFF, 0, int(gumbel_offset), int(dropout_offset), float(step_size[0]),
                                float(step_size[1]), float(bc2_sqrt[0]), float(bc2_sqrt[1]))
        import torch

        k = self._n % self.RING
        self._n += 1
        if self._events[k] is not None:
            self._events[k].synchronize()
        self._ring[k].numpy()[:] = memoryview(raw)
        self.dev.copy_(self._ring[k], non_blocking=True)
        if self._events[k] is None:
            self._events[k] = torch.cuda.Event()
        self._events[k].record()


def set_conv_precision(mode: str) -> None:
    """Operands of the forward / data-gradient convolution GEMMs, process-wide (include/dvg.h, dvg_set_conv_precision):
    ``"f32"`` (default: float32 operands on the f32 MFMA), ``"f32x3"`` (float32 operands as three bf16 pieces, six piece
    products on the bf16 MFMA, float32 accumulation: float32-class results) or ``"bf16"`` (bf16 inputs, f32 accumulate;
    weight gradients stay f32).  A forward pass and its backward pass must run in the same mode."""
    modes = {"f32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "f32x3": 2, "split3": 2}
    if mode not in modes:
        raise ValueError(f"conv precision must be one of {sorted(modes)}, got {mode!r}")
    check(lib().dvg_set_conv_precision(modes[mode]), "dvg_set_conv_precision")


def _dev_names() -> set:
    L = lib()
    return {L.dvg_dev_option_name(i).decode() for i in range(L.dvg_dev_option_count())}


def set_option(name: str, value: int) -> None:
    """A kernel-form option of the library: one of the product's switches (include/dvg.h, dvg_set_option; ``options()``
    lists them) or one of the A/B references / test knobs behind include/dvg_dev.h (``dev_options()``)."""
    if name in _dev_names():
        check(lib().dvg_dev_set_option(name.encode(), int(value)), f"dvg_dev_set_option({name})")
    else:
        check(lib().dvg_set_option(name.encode(), int(value)), f"dvg_set_option({name})")


def get_option(name: str) -> int:
    v = c_int64()
    if name in _dev_names():
        check(lib().dvg_dev_get_option(name.encode(), ctypes.byref(v)), f"dvg_dev_get_option({name})")
    else:
        check(lib().dvg_get_option(name.encode(), ctypes.byref(v)), f"dvg_get_option({name})")
    return int(v.value)


def options() -> dict:
    """name -> (value, doc) of every kernel-form option of the boundary (include/dvg.h)."""
    L = lib()
    out = {}
    for i in range(L.dvg_option_count()):
        name = L.dvg_option_name(i).decode()
        out[name] = (get_option(name), L.dvg_option_doc(i).decode())
    return out


def dev_options() -> dict:
    """name -> (value, doc) of the A/B references and test knobs (include/dvg_dev.h)."""
    L = lib()
    out = {}
    for i in range(L.dvg_dev_option_count()):
        name = L.dvg_dev_option_name(i).decode()
        out[name] = (get_option(name), L.dvg_dev_option_doc(i).decode())
    return out


class option_scope:
    """``with option_scope(dec_lc0=0, dec_d22=0): ...`` sets options for the block and restores the previous values."""

    def __init__(self, **values):
        self.values, self.saved = values, {}

    def __enter__(self):
        for k, v in self.values.items():
            self.saved[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            set_option(k, v)
        return False


def get_conv_precision() -> str:
    return {0: "f32", 1: "bf16", 2: "f32x3"}[lib().dvg_get_conv_precision()]


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().dvg_last_error().decode("utf-8", "replace")
        raise DvgError(f"{what or 'libdvg call'} failed with code {rc}: {msg}")


def ptr(t) -> int:
    """Device (or host) address of a torch tensor, None -> NULL."""
    return None if t is None else t.data_ptr()


def stream_ptr(device=None) -> int:
    import torch

    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(t, name: str):
    if not t.is_cuda:
        raise DvgError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise DvgError(f"{name} must be contiguous")
    return t
