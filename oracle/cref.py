"""ctypes binding of oracle/gibbs_ref.c (test infrastructure; see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libdvgoracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.dvgo_spec_exp.restype = ctypes.c_float
        _LIB.dvgo_spec_exp.argtypes = [ctypes.c_float]
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def philox(ctr, key):
    ctr = np.asarray(ctr, dtype=np.uint32)
    key = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().dvgo_philox(_p(ctr, ctypes.c_uint32), _p(key, ctypes.c_uint32), _p(out, ctypes.c_uint32))
    return out


def spec_exp(z):
    return np.float32(lib().dvgo_spec_exp(ctypes.c_float(float(z))))


def init_state(chain_ids, n, seed, sweep0=0):
    chain_ids = np.ascontiguousarray(chain_ids, dtype=np.uint32)
    st = np.empty((len(chain_ids), n), dtype=np.int8)
    lib().dvgo_init_state(
        ctypes.c_int(len(chain_ids)), ctypes.c_int(n), _p(st, ctypes.c_int8), _p(chain_ids, ctypes.c_uint32),
        ctypes.c_uint64(seed), ctypes.c_uint32(int(sweep0) & 0xFFFFFFFF),
    )
    return st


def gibbs_sweeps(state, chain_ids, hs, Js, beta, order, class_ptr, adj_ptr, adj_idx, adj_eid, seed, sweep0, nsweeps):
    state = np.ascontiguousarray(state, dtype=np.int8)
    C, n = state.shape
    chain_ids = np.ascontiguousarray(chain_ids, dtype=np.uint32)
    hs = np.ascontiguousarray(hs, dtype=np.float32)
    Js = np.ascontiguousarray(Js, dtype=np.float32)
    order = np.ascontiguousarray(order, dtype=np.int32)
    class_ptr = np.ascontiguousarray(class_ptr, dtype=np.int32)
    adj_ptr = np.ascontiguousarray(adj_ptr, dtype=np.int32)
    adj_idx = np.ascontiguousarray(adj_idx, dtype=np.int32)
    adj_eid = np.ascontiguousarray(adj_eid, dtype=np.int32)
    i32 = ctypes.c_int32
    rc = lib().dvgo_gibbs(
        ctypes.c_int(C), ctypes.c_int(n), _p(state, ctypes.c_int8), _p(chain_ids, ctypes.c_uint32),
        _p(hs, ctypes.c_float), _p(Js, ctypes.c_float), ctypes.c_float(beta),
        _p(order, i32), _p(class_ptr, i32), ctypes.c_int(len(class_ptr) - 1),
        _p(adj_ptr, i32), _p(adj_idx, i32), _p(adj_eid, i32),
        ctypes.c_uint64(seed), ctypes.c_uint32(sweep0), ctypes.c_int(nsweeps),
    )
    assert rc == 0
    return state
