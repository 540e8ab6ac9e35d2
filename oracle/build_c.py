"""Builds the C part of the oracle (TEST INFRASTRUCTURE ONLY): oracle/_lib/liboracle_c.so from oracle/*.c with gcc.

    python -m oracle.build_c

`__graft_entry__.build()` calls this; the .so is git-ignored and travels to the GPU box with the gpurun snapshot."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_lib", "liboracle_c.so")
SRCS = [os.path.join(HERE, "device_order_sgemm.c")]


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) > max(os.path.getmtime(s) for s in SRCS):
        return LIB
    # -march=x86-64-v3 (AVX2 + FMA), not -march=native: the .so is built in one container and run on the GPU box's host
    cmd = ["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-fno-math-errno", "-ffp-contract=off", "-shared", "-fPIC", *SRCS, "-o", LIB, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"oracle C build failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[oracle.build_c] {LIB} (rebuilt)")
    return LIB


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build(verbose=False))
        f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        _lib.nk_oracle_sgemm_device_order.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, f32p, ctypes.c_long, f32p, ctypes.c_long,
                                                      f32p, ctypes.c_long, ctypes.c_int, ctypes.c_int]
        _lib.nk_oracle_sgemm_device_order.restype = None
    return _lib


def sgemm_device_order(a: np.ndarray, b: np.ndarray, kc: int = 0, pair_second_first: bool = False) -> np.ndarray:
    """C = A . B (A: M x K, B: K x N, f32) summed in the order `sgemm_kernel` sums it: one fmaf chain per output over k in
    the MFMA feeding order, folded every `kc` values of k (0: a single chain).  See device_order_sgemm.c."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    assert a.ndim == 2 and b.ndim == 2 and a.shape[1] == b.shape[0]
    c = np.empty((a.shape[0], b.shape[1]), np.float32)
    if c.size:
        _load().nk_oracle_sgemm_device_order(a.shape[0], b.shape[1], a.shape[1], a, a.shape[1], b, b.shape[1], c, b.shape[1],
                                             int(kc), int(bool(pair_second_first)))
    return c


if __name__ == "__main__":
    build(force=True)
