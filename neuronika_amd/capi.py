"""ctypes binding of the C ABI in include/neuronika_hip.h (libneuronika_hip.so).

This is the Python twin of the Rust binding a maintainer would add: a
`Device` handle (reference template: neuronika-variable/src/cuda/device.rs:11-58), a
`HipArray` owning one device buffer (cuda/cuarray.rs:10-19) and one function per C entry
point.  There is NO CPU fallback: if the HIP library is missing or a call fails, this module
raises — a GPU test can never silently pass on another code path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NEURONIKA_HIP_LIB") or os.path.join(_HERE, "lib", "libneuronika_hip.so")  # override: A/B builds (benchmarks/ab_build.py)


class NeuronikaHipError(RuntimeError):
    """A C-ABI call returned non-zero.  The reference convention is panic (`.unwrap()`,
    cuda/device.rs:36-45; `assert!`, utils.rs:438-496); here it is an exception."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"[nk status {code}] {msg}")
        self.code = code


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m neuronika_amd.build` "
            "(the HIP backend has no CPU fallback)")
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


lib = _load()

c_f32p = C.POINTER(C.c_float)
c_intp = C.POINTER(C.c_int)
VP = C.c_void_p

# name -> argtypes (restype is int unless listed in _RESTYPES)
_SIGS = {
    "nk_device_count": [c_intp],
    "nk_device_create": [C.c_int, C.POINTER(VP)],
    "nk_device_destroy": [VP],
    "nk_device_sync": [VP],
    "nk_device_index": [VP],
    "nk_stream_compute": [VP],
    "nk_stream_comm": [VP],
    "nk_last_error": [],
    "nk_version": [],
    "nk_alloc_zeroed": [VP, C.c_size_t, C.POINTER(VP)],
    "nk_free": [VP, VP],
    "nk_upload": [VP, VP, VP, C.c_size_t],
    "nk_graph_begin": [VP],
    "nk_graph_end": [VP, C.POINTER(C.c_void_p)],
    "nk_graph_launch": [VP],
    "nk_graph_destroy": [VP],
    "nk_host_alloc": [C.c_size_t, C.POINTER(C.c_void_p)],
    "nk_host_free": [VP],
    "nk_upload_async": [VP, VP, VP, C.c_size_t],
    "nk_download": [VP, VP, VP, C.c_size_t],
    "nk_fill": [VP, VP, C.c_size_t, C.c_float],
    "nk_copy": [VP, VP, VP, C.c_size_t],
    "nk_event_create": [VP, C.POINTER(VP)],
    "nk_event_destroy": [VP],
    "nk_event_record": [VP, C.c_int],
    "nk_event_sync": [VP],
    "nk_event_elapsed_ms": [VP, VP, C.POINTER(C.c_float)],
    "nk_stream_wait_event": [VP, C.c_int, VP],
    "nk_profile_begin": [VP],
    "nk_profile_pause": [VP, C.c_int],
    "nk_profile_end": [VP, C.c_int, c_intp, C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "nk_sgemm": [VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, VP, C.c_int, VP, C.c_int, C.c_float, VP, C.c_int],
    "nk_sgemm_batched": [VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                         VP, C.c_int, C.c_longlong, C.c_longlong, VP, C.c_int, C.c_longlong, C.c_longlong,
                         C.c_float, VP, C.c_int, C.c_longlong, C.c_longlong, C.c_int, C.c_int],
    "nk_sgemm_pair": [VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP, C.c_int, VP, C.c_int, C.c_float, VP, C.c_int,
                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP, C.c_int, VP, C.c_int, C.c_float, VP, C.c_int],
    "nk_sgemm_pair_batched": [VP, C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP, C.c_int, C.c_longlong, C.c_longlong, VP, C.c_int, C.c_longlong,
                              C.c_longlong, C.c_float, VP, C.c_int, C.c_longlong, C.c_longlong,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP, C.c_int, C.c_longlong, C.c_longlong, VP, C.c_int, C.c_longlong,
                              C.c_longlong, C.c_float, VP, C.c_int, C.c_longlong, C.c_longlong],
    "nk_mm_bwd": [VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int],
    "nk_mm_t_bwd": [VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int],
    "nk_mm_fwd": [VP, VP, VP, VP, C.c_int, C.c_int, C.c_int],
    "nk_mm_bwd_left": [VP, VP, VP, VP, C.c_int, C.c_int, C.c_int],
    "nk_mm_bwd_right": [VP, VP, VP, VP, C.c_int, C.c_int, C.c_int],
    "nk_mm_t_fwd": [VP, VP, VP, VP, C.c_int, C.c_int, C.c_int],
    "nk_mm_t_bwd_left": [VP, VP, VP, VP, C.c_int, C.c_int, C.c_int],
    "nk_mm_t_bwd_right": [VP, VP, VP, VP, C.c_int, C.c_int, C.c_int],
    "nk_linear_fwd": [VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int],
    "nk_linear_relu_fwd": [VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int],
    "nk_linear_bwd_input_relu": [VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int],
    "nk_relu_mask_inplace": [VP, VP, VP, C.c_size_t],
    "nk_conv_fwd": [VP, C.c_int, VP, c_intp, VP, c_intp, VP, c_intp, c_intp, C.c_int],
    "nk_conv_bias_fwd": [VP, C.c_int, VP, c_intp, VP, c_intp, VP, VP, c_intp, c_intp, C.c_int],
    "nk_conv_bwd_input_assign": [VP, C.c_int, VP, c_intp, VP, VP, c_intp, c_intp, c_intp, C.c_int],
    "nk_conv_bwd_kernel_assign": [VP, C.c_int, VP, c_intp, VP, VP, c_intp, c_intp, c_intp, C.c_int],
    "nk_conv_bwd_input": [VP, C.c_int, VP, c_intp, VP, VP, c_intp, c_intp, c_intp, C.c_int],
    "nk_conv_bwd_input_padded": [VP, C.c_int, VP, c_intp, c_intp, VP, VP, c_intp, c_intp, c_intp, C.c_int],
    "nk_conv_bwd_input_padded_assign": [VP, C.c_int, VP, c_intp, c_intp, VP, VP, c_intp, c_intp, c_intp, C.c_int],
    "nk_conv_bwd_kernel": [VP, C.c_int, VP, c_intp, VP, VP, c_intp, c_intp, c_intp, C.c_int],
    "nk_conv_bwd_kernel_bias": [VP, C.c_int, VP, VP, c_intp, VP, VP, c_intp, c_intp, c_intp, C.c_int, C.c_int, C.c_int],
    "nk_conv_padding_folds": [VP, C.c_int, c_intp, c_intp, c_intp, c_intp, c_intp, C.c_int, C.POINTER(C.c_int)],
    "nk_conv_bias_fwd_padded": [VP, C.c_int, VP, c_intp, c_intp, VP, c_intp, VP, VP, c_intp, c_intp, C.c_int],
    "nk_conv_bwd_kernel_bias_padded": [VP, C.c_int, VP, VP, c_intp, VP, VP, c_intp, c_intp, c_intp, c_intp, C.c_int, C.c_int, C.c_int],
    "nk_pad_const_fwd": [VP, C.c_int, VP, c_intp, VP, c_intp, C.c_float],
    "nk_pad_reflective_fwd": [VP, C.c_int, VP, c_intp, VP, c_intp],
    "nk_pad_replicative_fwd": [VP, C.c_int, VP, c_intp, VP, c_intp],
    "nk_pad_bwd": [VP, C.c_int, VP, c_intp, VP, c_intp],
    "nk_binary_fwd": [VP, C.c_int, VP, c_intp, C.c_int, VP, c_intp, C.c_int, VP, c_intp, C.c_int],
    "nk_binary_bwd_left": [VP, C.c_int, VP, c_intp, C.c_int, VP, c_intp, C.c_int, VP, c_intp, C.c_int],
    "nk_binary_bwd_right": [VP, C.c_int, VP, c_intp, C.c_int, VP, c_intp, C.c_int, VP, c_intp, C.c_int, VP],
    "nk_unbroadcast_add": [VP, VP, c_intp, C.c_int, VP, c_intp, C.c_int],
    "nk_unary_fwd": [VP, C.c_int, VP, VP, C.c_size_t, C.c_int],
    "nk_unary_bwd": [VP, C.c_int, VP, VP, VP, C.c_size_t, C.c_int],
    "nk_relu_fwd": [VP, VP, VP, C.c_size_t],
    "nk_relu_bwd": [VP, VP, VP, VP, C.c_size_t],
    "nk_sum_fwd": [VP, VP, C.c_size_t, VP],
    "nk_sum_bwd": [VP, VP, C.c_size_t, VP],
    "nk_mean_fwd": [VP, VP, C.c_size_t, VP],
    "nk_mean_bwd": [VP, VP, C.c_size_t, VP],
    "nk_mse_fwd": [VP, VP, VP, C.c_size_t, C.c_int, VP],
    "nk_mse_bwd": [VP, VP, VP, VP, VP, C.c_size_t, C.c_int],
    "nk_binary_bwd_left_assign": [VP, C.c_int, VP, c_intp, C.c_int, VP, c_intp, C.c_int, VP, c_intp, C.c_int],
    "nk_binary_bwd_right_assign": [VP, C.c_int, VP, c_intp, C.c_int, VP, c_intp, C.c_int, VP, c_intp, C.c_int, VP],
    "nk_unbroadcast_assign": [VP, VP, c_intp, C.c_int, VP, c_intp, C.c_int],
    "nk_unary_bwd_assign": [VP, C.c_int, VP, VP, VP, C.c_size_t, C.c_int],
    "nk_softmax_bwd_assign": [VP, VP, VP, VP, c_intp, C.c_int, C.c_int],
    "nk_log_softmax_bwd_assign": [VP, VP, VP, VP, c_intp, C.c_int, C.c_int],
    "nk_dropout_bwd_assign": [VP, VP, VP, VP, C.c_size_t, C.c_double, C.c_int],
    "nk_concat_bwd_part_assign": [VP, VP, VP, c_intp, C.c_int, C.c_int, C.c_int, C.c_int],
    "nk_transpose_bwd_assign": [VP, VP, VP, c_intp, C.c_int],
    "nk_sum_bwd_assign": [VP, VP, C.c_size_t, VP],
    "nk_mean_bwd_assign": [VP, VP, C.c_size_t, VP],
    "nk_relu_bwd_assign": [VP, VP, VP, VP, C.c_size_t],
    "nk_mse_bwd_assign": [VP, VP, VP, VP, VP, C.c_size_t, C.c_int],
    "nk_pad_bwd_assign": [VP, C.c_int, VP, c_intp, VP, c_intp],
    "nk_split_heads_bwd_assign": [VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int],
    "nk_merge_heads_bwd_assign": [VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int],
    "nk_scale_softmax_dropout_bwd_from_scores": [VP, VP, VP, VP, VP, C.c_longlong, C.c_int, C.c_float, C.c_double, C.c_int, C.c_uint64, C.c_uint64, C.c_int],
    "nk_scale_softmax_dropout_bwd_assign": [VP, VP, VP, VP, VP, C.c_longlong, C.c_int, C.c_float, C.c_double, C.c_int, C.c_uint64, C.c_uint64],
    "nk_loss_fwd": [VP, C.c_int, VP, VP, c_intp, C.c_int, C.c_int, VP],
    "nk_loss_bwd": [VP, C.c_int, VP, VP, VP, VP, c_intp, C.c_int, C.c_int],
    "nk_nll_fwd": [VP, VP, VP, c_intp, C.c_int, C.c_int, VP],
    "nk_nll_bwd": [VP, VP, VP, VP, c_intp, C.c_int, C.c_int],
    "nk_mv_fwd": [VP, VP, VP, VP, C.c_int, C.c_int],
    "nk_mv_bwd_left": [VP, VP, VP, VP, C.c_int, C.c_int],
    "nk_mv_bwd_right": [VP, VP, VP, VP, C.c_int, C.c_int],
    "nk_vm_fwd": [VP, VP, VP, VP, C.c_int, C.c_int],
    "nk_vm_bwd_left": [VP, VP, VP, VP, C.c_int, C.c_int],
    "nk_vm_bwd_right": [VP, VP, VP, VP, C.c_int, C.c_int],
    "nk_vv_fwd": [VP, VP, VP, C.c_size_t, VP],
    "nk_vv_bwd": [VP, VP, VP, VP, C.c_size_t],
    "nk_softmax_fwd": [VP, VP, VP, c_intp, C.c_int, C.c_int],
    "nk_softmax_bwd": [VP, VP, VP, VP, c_intp, C.c_int, C.c_int],
    "nk_log_softmax_fwd": [VP, VP, VP, c_intp, C.c_int, C.c_int],
    "nk_log_softmax_bwd": [VP, VP, VP, VP, c_intp, C.c_int, C.c_int],
    "nk_attention_supported": [C.c_int, C.c_int, C.c_double, C.c_int],
    "nk_attention_fwd": [VP, VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_double, C.c_int, C.c_uint64, C.c_uint64],
    "nk_attention_bwd": [VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_double,
                         C.c_int, C.c_int, C.c_int, C.c_int],
    "nk_attention_qkv_fwd": [VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_double, C.c_int, C.c_uint64, C.c_uint64],
    "nk_attention_qkv_bwd": [VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_double, C.c_int, C.c_int],
    "nk_scale_softmax_dropout_fwd": [VP, VP, VP, VP, VP, C.c_longlong, C.c_int, C.c_float, C.c_double, C.c_int, C.c_uint64, C.c_uint64],
    "nk_scale_softmax_dropout_bwd": [VP, VP, VP, VP, VP, C.c_longlong, C.c_int, C.c_float, C.c_double, C.c_int, C.c_uint64, C.c_uint64],
    "nk_dropout_fwd": [VP, VP, VP, VP, C.c_size_t, C.c_double, C.c_int, C.c_uint64, C.c_uint64],
    "nk_dropout_bwd": [VP, VP, VP, VP, C.c_size_t, C.c_double, C.c_int],
    "nk_chunk_fwd": [VP, VP, c_intp, VP, c_intp, C.c_int, C.c_int],
    "nk_chunk_bwd": [VP, VP, c_intp, VP, c_intp, C.c_int, C.c_int],
    "nk_concat_fwd_part": [VP, VP, VP, c_intp, C.c_int, C.c_int, C.c_int, C.c_int],
    "nk_concat_bwd_part": [VP, VP, VP, c_intp, C.c_int, C.c_int, C.c_int, C.c_int],
    "nk_transpose_fwd": [VP, VP, VP, c_intp, C.c_int],
    "nk_transpose_bwd": [VP, VP, VP, c_intp, C.c_int],
    "nk_split_heads_fwd": [VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int],
    "nk_split_heads_bwd": [VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int],
    "nk_merge_heads_fwd": [VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int],
    "nk_merge_heads_bwd": [VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int],
    "nk_sgd_step": [VP, VP, VP, VP, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float],
    "nk_sgd_step_multi": [VP, C.c_int, VP, VP, VP, VP, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float],
    "nk_adam_step": [VP, VP, VP, VP, VP, VP, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float],
    "nk_adagrad_step": [VP, VP, VP, VP, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float],
    "nk_rmsprop_step": [VP, VP, VP, VP, VP, VP, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float],
    "nk_comm_unique_id": [C.c_char_p],
    "nk_comm_init_rank": [VP, C.c_int, C.c_int, C.c_char_p, C.POINTER(VP)],
    "nk_comm_init_replicas": [VP, C.c_int, C.c_int, C.c_double, C.POINTER(VP)],
    "nk_comm_init_all": [C.c_int, C.POINTER(VP), C.POINTER(VP)],
    "nk_dev_tune": [VP, C.c_int, C.POINTER(C.c_int), C.c_int],
    "nk_device_set_busy_slots": [VP, C.c_int],
    "nk_conv_winograd_launches": [VP, C.POINTER(C.c_uint64)],
    "nk_comm_destroy": [VP],
    "nk_allreduce_sum_async": [VP, VP, C.c_size_t, VP],
    "nk_allreduce_sum_group_async": [VP, C.POINTER(VP), C.POINTER(C.c_size_t), C.c_int, VP],
    "nk_comm_join": [VP],
    "nk_comm_rank": [VP],
    "nk_comm_size": [VP],
}
_RESTYPES = {"nk_last_error": C.c_char_p, "nk_version": C.c_char_p, "nk_stream_compute": VP, "nk_stream_comm": VP}
EXPORTED = tuple(_SIGS)

for _name, _args in _SIGS.items():
    try:
        _fn = getattr(lib, _name)  # AttributeError here = header/library mismatch: fail loudly ...
    except AttributeError:
        if not os.environ.get("NEURONIKA_HIP_LIB"):
            raise
        continue                   # ... except for an A/B variant library built from an OLDER revision (tools/build_rev_variant.sh)
    _fn.argtypes = _args
    _fn.restype = _RESTYPES.get(_name, C.c_int)

ADD, SUB, MUL, DIV = 0, 1, 2, 3
OPS = {"add": ADD, "sub": SUB, "mul": MUL, "div": DIV}
REDUCTION = {"sum": 0, "mean": 1}
COMM_ID_BYTES = 128


def check(rc: int) -> None:
    if rc != 0:
        raise NeuronikaHipError(rc, lib.nk_last_error().decode())


def device_count() -> int:
    n = C.c_int(0)
    check(lib.nk_device_count(C.byref(n)))
    return n.value


def ints(v: Sequence[int]):
    return (C.c_int * max(1, len(v)))(*[int(i) for i in v])


class Event:
    def __init__(self, dev: "Device"):
        self.dev = dev
        h = VP()
        check(lib.nk_event_create(dev.h, C.byref(h)))
        self.h = h

    def record(self, comm_stream: bool = False):
        check(lib.nk_event_record(self.h, 1 if comm_stream else 0))
        return self

    def sync(self):
        check(lib.nk_event_sync(self.h))

    def elapsed_ms(self, stop: "Event") -> float:
        ms = C.c_float(0)
        check(lib.nk_event_elapsed_ms(self.h, stop.h, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            lib.nk_event_destroy(self.h)
        except Exception:
            pass


TUNE_GEMM_FORCE, TUNE_GEMM_KPAIR, TUNE_ATTENTION_OCC, TUNE_GEMM_PAIR, TUNE_CONV_NARROW, TUNE_CONV_WINOGRAD, TUNE_GEMM_CHAIN, TUNE_CONV_S2DX = 0, 1, 2, 3, 4, 5, 6, 7   # include/neuronika_hip.h: nk_dev_tune knobs


class Device:
    """`Device::new(idx)` (cuda/device.rs:34-58): one GPU, its compute + communication streams."""

    def __init__(self, idx: int = 0, handle: int | None = None):
        """`handle`: wrap an existing nk_device* (e.g. `tape.Device.raw()`) without owning it."""
        if handle is not None:
            self.h, self.idx, self._own = VP(handle), lib.nk_device_index(VP(handle)), False
        else:
            h = VP()
            check(lib.nk_device_create(idx, C.byref(h)))
            self.h, self.idx, self._own = h, idx, True
        # The sweep scripts (benchmarks/ab_*.py, tools/sessions/*.sh) choose a schedule per process through environment
        # variables; it is THIS harness that reads them and calls nk_dev_tune - the library itself reads none.
        for var, knob in (("NK_GEMM_FORCE", TUNE_GEMM_FORCE), ("NK_GEMM_KPAIR", TUNE_GEMM_KPAIR), ("NK_ATTN_OCC", TUNE_ATTENTION_OCC),
                          ("NK_GEMM_PAIR", TUNE_GEMM_PAIR), ("NK_CONV_NARROW", TUNE_CONV_NARROW), ("NK_CONV_WINOGRAD", TUNE_CONV_WINOGRAD),
                          ("NK_GEMM_CHAIN", TUNE_GEMM_CHAIN), ("NK_CONV_S2DX", TUNE_CONV_S2DX)):
            if os.environ.get(var):
                self.tune(knob, os.environ[var])

    def tune(self, knob: int, values=None):
        """nk_dev_tune: `values` = None (back to the rules), an int, a sequence of ints or "a,b,c"."""
        if values is None:
            vals = []
        elif isinstance(values, str):
            vals = [int(v) for v in values.split(",") if v.strip()]
        elif isinstance(values, (int, np.integer)):
            vals = [int(values)]
        else:
            vals = [int(v) for v in values]
        check(lib.nk_dev_tune(self.h, knob, (C.c_int * max(1, len(vals)))(*vals), len(vals)))

    def gemm_force(self, values=None):
        self.tune(TUNE_GEMM_FORCE, values)

    def gemm_kpair(self, mode=None):
        self.tune(TUNE_GEMM_KPAIR, mode)

    def gemm_pair(self, mode=None):
        self.tune(TUNE_GEMM_PAIR, mode)

    def gemm_chain(self, length=None):
        """NK_TUNE_GEMM_CHAIN: None / -1 the rule (chains of at most 2048), 0 one chain whatever K, L > 0 chains of at most L"""
        self.tune(TUNE_GEMM_CHAIN, length)

    def conv_s2dx(self, mode=None):
        """NK_TUNE_CONV_S2DX: None / -1 rule, 0 the 3x3 stride-2 input gradient never takes the fused-phase kernel, 1 whenever the shape allows,
        2 / 3 the same with narrow / wide blocks forced"""
        self.tune(TUNE_CONV_S2DX, mode)

    def conv_narrow(self, cost=None):
        self.tune(TUNE_CONV_NARROW, cost)

    def conv_winograd_launches(self) -> int:
        """nk_conv_winograd_launches: convolution launches on this handle that took the Winograd kernels so far."""
        n = C.c_uint64(0)
        check(lib.nk_conv_winograd_launches(self.h, C.byref(n)))
        return int(n.value)

    def conv_winograd(self, mode=None, stagger=None, shape=None, dw=None):
        """NK_TUNE_CONV_WINOGRAD: mode -1 rule / 0 never / 1 whenever the shape allows; stagger unit in shader clocks (-1 rule);
        block shape -1 rule / 0 narrow / 1 wide; kernel gradient (F(3x3, 2x2)) -1 rule / 0 never / 1 whenever the shape allows."""
        if stagger is None and shape is None and dw is None:
            self.tune(TUNE_CONV_WINOGRAD, mode)
        else:
            self.tune(TUNE_CONV_WINOGRAD, [-1 if v is None else v for v in (mode, stagger, shape, dw)])

    def busy_slots(self, n: int = 0):
        """nk_device_set_busy_slots: `n` resident-block slots are held by work on another stream (an exchange in flight)."""
        check(lib.nk_device_set_busy_slots(self.h, int(n)))

    def sync(self):
        check(lib.nk_device_sync(self.h))

    def event(self) -> Event:
        return Event(self)

    # ---- arrays
    def zeros(self, shape) -> "HipArray":
        return HipArray(self, shape)

    def array(self, a) -> "HipArray":
        a = np.ascontiguousarray(a, dtype=np.float32)
        out = HipArray(self, a.shape)
        out.upload(a)
        return out

    def full(self, shape, value: float) -> "HipArray":
        out = HipArray(self, shape)
        out.fill(value)
        return out

    def close(self):
        if self.h and self._own:
            lib.nk_device_destroy(self.h)
        self.h = None

    def profile_begin(self):
        check(lib.nk_profile_begin(self.h))

    def profile_pause(self, paused: bool):
        check(lib.nk_profile_pause(self.h, int(bool(paused))))

    def profile_end(self, kernel_class: int = 0):
        """-> (launches, total_ms, total_flop) of one kernel class since profile_begin."""
        n, ms, fl = C.c_int(0), C.c_double(0), C.c_double(0)
        check(lib.nk_profile_end(self.h, kernel_class, C.byref(n), C.byref(ms), C.byref(fl)))
        return n.value, ms.value, fl.value


class HipArray:
    """Device twin of `ndarray::Array<f32, D>` — `CuArray<f32, D>` in the reference's template
    (cuda/cuarray.rs:10-19): owns one zero-initialised device buffer + its shape; frees on drop."""

    def __init__(self, dev: Device, shape):
        if isinstance(shape, (int, np.integer)):
            shape = (int(shape),)
        self.dev = dev
        self.shape = tuple(int(s) for s in shape)
        self.size = int(np.prod(self.shape, dtype=np.int64)) if len(self.shape) else 1
        p = VP()
        check(lib.nk_alloc_zeroed(dev.h, self.size, C.byref(p)))
        self.p = p

    @property
    def ndim(self):
        return len(self.shape)

    def upload(self, a: np.ndarray):
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.size == self.size, (a.shape, self.shape)
        check(lib.nk_upload(self.dev.h, self.p, a.ctypes.data_as(VP), self.size))
        return self

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=np.float32)
        check(lib.nk_download(self.dev.h, out.ctypes.data_as(VP), self.p, self.size))
        return out

    def item(self) -> float:
        return float(self.numpy().reshape(-1)[0])

    def fill(self, v: float):
        check(lib.nk_fill(self.dev.h, self.p, self.size, float(v)))
        return self

    def shape_c(self):
        return ints(self.shape)

    def view_offset(self, first: int) -> "HipArray":
        """Non-owning alias that starts `first` floats into this buffer (column blocks / row blocks of one allocation for the raw
        entry points that take a pointer and a leading dimension); keeps `self` alive."""
        v = HipArray.__new__(HipArray)
        v.dev, v.shape, v.size, v._parent = self.dev, (self.size - first,), self.size - first, self
        v.p = VP(self.p.value + 4 * int(first))
        return v

    def __del__(self):
        try:
            if getattr(self, "_parent", None) is not None:
                return
            if self.p and self.dev.h:
                lib.nk_free(self.dev.h, self.p)
        except Exception:
            pass


KERNEL_SGEMM, KERNEL_CONV, KERNEL_ATTENTION = 0, 1, 2


# ------------------------------------------------------------------------------------------------
# thin per-entry-point wrappers (argument order = C ABI order)
# ------------------------------------------------------------------------------------------------

def sgemm(dev, ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, Cm, ldc):
    check(lib.nk_sgemm(dev.h, int(ta), int(tb), M, N, K, alpha, A.p, lda, B.p, ldb, beta, Cm.p, ldc))


def sgemm_batched(dev, ta, tb, M, N, K, alpha, A, lda, sAo, sAi, B, ldb, sBo, sBi, beta, Cm, ldc, sCo, sCi, bo, bi):
    check(lib.nk_sgemm_batched(dev.h, int(ta), int(tb), M, N, K, alpha, A.p, lda, sAo, sAi, B.p, ldb, sBo, sBi,
                               beta, Cm.p, ldc, sCo, sCi, bo, bi))


def mm_fwd(dev, A, B, out):
    n, m = A.shape; o = B.shape[1]
    check(lib.nk_mm_fwd(dev.h, A.p, B.p, out.p, n, m, o))


def mm_bwd_left(dev, dA, G, B):
    n, o = G.shape; m = B.shape[0]
    check(lib.nk_mm_bwd_left(dev.h, dA.p, G.p, B.p, n, m, o))


def mm_bwd_right(dev, dB, A, G):
    n, m = A.shape; o = G.shape[1]
    check(lib.nk_mm_bwd_right(dev.h, dB.p, A.p, G.p, n, m, o))


def mm_bwd(dev, dA, dB, G, A, B, assign_a=False, assign_b=False):
    """MatrixMatrixMulBackward::backward: both products, one launch when nk_sgemm_pair's rule says so"""
    n, m = A.shape; o = B.shape[1]
    check(lib.nk_mm_bwd(dev.h, dA.p, dB.p, G.p, A.p, B.p, n, m, o, int(assign_a), int(assign_b)))


def mm_t_bwd(dev, dA, dB, G, A, B, assign_a=False, assign_b=False):
    n, m = A.shape; o = B.shape[0]
    check(lib.nk_mm_t_bwd(dev.h, dA.p, dB.p, G.p, A.p, B.p, n, m, o, int(assign_a), int(assign_b)))


def sgemm_pair(dev, ta0, tb0, M0, N0, K0, A0, lda0, B0, ldb0, beta0, C0, ldc0, ta1, tb1, M1, N1, K1, A1, lda1, B1, ldb1, beta1, C1, ldc1):
    check(lib.nk_sgemm_pair(dev.h, int(ta0), int(tb0), M0, N0, K0, A0.p, lda0, B0.p, ldb0, float(beta0), C0.p, ldc0,
                            int(ta1), int(tb1), M1, N1, K1, A1.p, lda1, B1.p, ldb1, float(beta1), C1.p, ldc1))


def sgemm_pair_batched(dev, batch_outer, batch_inner, p0, p1):
    """p = (ta, tb, M, N, K, A, lda, sAo, sAi, B, ldb, sBo, sBi, beta, C, ldc, sCo, sCi)"""
    def flat(q):
        ta, tb, M, N, K, A, lda, sAo, sAi, B, ldb, sBo, sBi, beta, Cm, ldc, sCo, sCi = q
        return [int(ta), int(tb), M, N, K, A.p, lda, sAo, sAi, B.p, ldb, sBo, sBi, float(beta), Cm.p, ldc, sCo, sCi]
    check(lib.nk_sgemm_pair_batched(dev.h, batch_outer, batch_inner, *flat(p0), *flat(p1)))


def mm_t_fwd(dev, A, B, out):
    n, m = A.shape; o = B.shape[0]
    check(lib.nk_mm_t_fwd(dev.h, A.p, B.p, out.p, n, m, o))


def mm_t_bwd_left(dev, dA, G, B):
    n, o = G.shape; m = B.shape[1]
    check(lib.nk_mm_t_bwd_left(dev.h, dA.p, G.p, B.p, n, m, o))


def mm_t_bwd_right(dev, dB, G, A):
    n, o = G.shape; m = A.shape[1]
    check(lib.nk_mm_t_bwd_right(dev.h, dB.p, G.p, A.p, n, m, o))


def conv_fwd(dev, x, w, y, stride, dilation, groups=1, bias=None):
    nd = x.ndim - 2
    if bias is not None:
        check(lib.nk_conv_bias_fwd(dev.h, nd, x.p, x.shape_c(), w.p, w.shape_c(), bias.p, y.p, ints(stride), ints(dilation), groups))
    else:
        check(lib.nk_conv_fwd(dev.h, nd, x.p, x.shape_c(), w.p, w.shape_c(), y.p, ints(stride), ints(dilation), groups))


def conv_bwd_input(dev, dx, g, w, stride, dilation, groups=1, assign=False, padding=None):
    """`padding`: dx is the gradient of the UNPADDED input of a zero Pad node that fed the convolution."""
    nd = dx.ndim - 2
    if padding is not None:
        fn = lib.nk_conv_bwd_input_padded_assign if assign else lib.nk_conv_bwd_input_padded
        check(fn(dev.h, nd, dx.p, dx.shape_c(), ints(padding), g.p, w.p, w.shape_c(), ints(stride), ints(dilation), groups))
        return
    check((lib.nk_conv_bwd_input_assign if assign else lib.nk_conv_bwd_input)(dev.h, nd, dx.p, dx.shape_c(), g.p, w.p, w.shape_c(), ints(stride), ints(dilation), groups))


def conv_bwd_kernel(dev, dw, g, x, stride, dilation, groups=1, assign=False):
    nd = x.ndim - 2
    check((lib.nk_conv_bwd_kernel_assign if assign else lib.nk_conv_bwd_kernel)(dev.h, nd, dw.p, dw.shape_c(), g.p, x.p, x.shape_c(), ints(stride), ints(dilation), groups))


def conv_bwd_kernel_bias(dev, dw, db, g, x, stride, dilation, groups=1, assign=(False, False)):
    """dw (+)= ConvolutionBackwardKernel and db (+)= the (Cout,1,..) bias gradient (sum of g over samples and positions), one pass."""
    nd = x.ndim - 2
    check(lib.nk_conv_bwd_kernel_bias(dev.h, nd, dw.p, db.p, dw.shape_c(), g.p, x.p, x.shape_c(), ints(stride), ints(dilation), groups,
                                      int(assign[0]), int(assign[1])))


def conv_padding_folds(dev, x_shape, padding, w_shape, stride, dilation, groups=1) -> bool:
    """nk_conv_padding_folds: would the Conv module's forward and kernel gradient both run with the Zero padding folded in?"""
    nd = len(x_shape) - 2
    out = C.c_int(0)
    check(lib.nk_conv_padding_folds(dev.h, nd, ints(x_shape), ints(padding), ints(w_shape), ints(stride), ints(dilation), groups, C.byref(out)))
    return bool(out.value)


def conv_fwd_padded(dev, x, w, y, padding, stride, dilation, groups=1, bias=None):
    """y = conv(zero_pad(x, padding), w) (+ bias): `x` is the UNPADDED input (nk_conv_bias_fwd_padded)."""
    nd = x.ndim - 2
    check(lib.nk_conv_bias_fwd_padded(dev.h, nd, x.p, x.shape_c(), ints(padding), w.p, w.shape_c(), bias.p if bias is not None else None, y.p,
                                      ints(stride), ints(dilation), groups))


def conv_bwd_kernel_padded(dev, dw, g, x, padding, stride, dilation, groups=1, db=None, assign=(False, False)):
    """dw (+)= kernel gradient against zero_pad(x, padding), db (+)= bias gradient: `x` is the UNPADDED input."""
    nd = x.ndim - 2
    check(lib.nk_conv_bwd_kernel_bias_padded(dev.h, nd, dw.p, db.p if db is not None else None, dw.shape_c(), g.p, x.p, x.shape_c(), ints(padding),
                                             ints(stride), ints(dilation), groups, int(assign[0]), int(assign[1])))


def linear_fwd(dev, X, W, bias, Y):
    check(lib.nk_linear_fwd(dev.h, X.p, W.p, bias.p, Y.p, X.shape[0], X.shape[1], W.shape[0]))


def linear_relu_fwd(dev, X, W, bias, Y):
    check(lib.nk_linear_relu_fwd(dev.h, X.p, W.p, bias.p, Y.p, X.shape[0], X.shape[1], W.shape[0]))


def linear_bwd_input_relu(dev, dZ, G, W, X, assign=False):
    """dZ (+)= (X > 0) * (G . W): G (n,o), W (o,m), X and dZ (n,m)."""
    check(lib.nk_linear_bwd_input_relu(dev.h, dZ.p, G.p, W.p, X.p, X.shape[0], X.shape[1], W.shape[0], int(assign)))


def relu_mask_inplace(dev, g, y):
    check(lib.nk_relu_mask_inplace(dev.h, g.p, y.p, y.size))


def pad_const_fwd(dev, x, y, padding, value=0.0):
    check(lib.nk_pad_const_fwd(dev.h, x.ndim - 2, x.p, x.shape_c(), y.p, ints(padding), float(value)))


def pad_mode_fwd(dev, x, y, padding, mode):
    fn = {"reflective": lib.nk_pad_reflective_fwd, "replicative": lib.nk_pad_replicative_fwd}[mode]
    check(fn(dev.h, x.ndim - 2, x.p, x.shape_c(), y.p, ints(padding)))


def pad_bwd(dev, dx, g, padding, assign=False):
    check((lib.nk_pad_bwd_assign if assign else lib.nk_pad_bwd)(dev.h, dx.ndim - 2, dx.p, dx.shape_c(), g.p, ints(padding)))


def binary_fwd(dev, op, out, l, r):
    check(lib.nk_binary_fwd(dev.h, OPS[op], out.p, out.shape_c(), out.ndim, l.p, l.shape_c(), l.ndim, r.p, r.shape_c(), r.ndim))


def binary_bwd_left(dev, op, d_left, g, r=None, assign=False):
    rp, rs, rn = (r.p, r.shape_c(), r.ndim) if r is not None else (None, ints([]), 0)
    check((lib.nk_binary_bwd_left_assign if assign else lib.nk_binary_bwd_left)(dev.h, OPS[op], d_left.p, d_left.shape_c(), d_left.ndim, g.p, g.shape_c(), g.ndim, rp, rs, rn))


def binary_bwd_right(dev, op, d_right, g, l=None, r=None, assign=False):
    lp, ls, ln = (l.p, l.shape_c(), l.ndim) if l is not None else (None, ints([]), 0)
    check((lib.nk_binary_bwd_right_assign if assign else lib.nk_binary_bwd_right)(dev.h, OPS[op], d_right.p, d_right.shape_c(), d_right.ndim, g.p, g.shape_c(), g.ndim,
                                  lp, ls, ln, r.p if r is not None else None))


def unbroadcast_add(dev, dst, src, assign=False):
    check((lib.nk_unbroadcast_assign if assign else lib.nk_unbroadcast_add)(dev.h, dst.p, dst.shape_c(), dst.ndim, src.p, src.shape_c(), src.ndim))


UNARY = {"neg": 0, "exp": 1, "ln": 2, "sqrt": 3, "sigmoid": 4, "tanh": 5, "softplus": 6, "leaky_relu": 7, "pow": 8}


def unary_fwd(dev, op, x, y, iparam=0):
    check(lib.nk_unary_fwd(dev.h, UNARY[op], x.p, y.p, x.size, iparam))


def unary_bwd(dev, op, dx, g, ref=None, iparam=0, assign=False):
    check((lib.nk_unary_bwd_assign if assign else lib.nk_unary_bwd)(dev.h, UNARY[op], dx.p, g.p, ref.p if ref is not None else None, dx.size, iparam))


def relu_fwd(dev, x, y):
    check(lib.nk_relu_fwd(dev.h, x.p, y.p, x.size))


def relu_bwd(dev, dx, g, x, assign=False):
    check((lib.nk_relu_bwd_assign if assign else lib.nk_relu_bwd)(dev.h, dx.p, g.p, x.p, x.size))


def sum_fwd(dev, x, out):
    check(lib.nk_sum_fwd(dev.h, x.p, x.size, out.p))


def sum_bwd(dev, dx, g, assign=False):
    check((lib.nk_sum_bwd_assign if assign else lib.nk_sum_bwd)(dev.h, dx.p, dx.size, g.p))


def mean_fwd(dev, x, out):
    check(lib.nk_mean_fwd(dev.h, x.p, x.size, out.p))


def mean_bwd(dev, dx, g, assign=False):
    check((lib.nk_mean_bwd_assign if assign else lib.nk_mean_bwd)(dev.h, dx.p, dx.size, g.p))


def mse_fwd(dev, x, t, out, reduction="mean"):
    check(lib.nk_mse_fwd(dev.h, x.p, t.p, x.size, REDUCTION[reduction], out.p))


def mse_bwd(dev, dx, g, x, t, reduction="mean", assign=False):
    check((lib.nk_mse_bwd_assign if assign else lib.nk_mse_bwd)(dev.h, dx.p, g.p, x.p, t.p, x.size, REDUCTION[reduction]))


LOSS = {"mae": 0, "bce": 1, "bce_with_logits": 2, "kldiv": 3}


def loss_fwd(dev, loss, x, t, out, reduction="mean"):
    check(lib.nk_loss_fwd(dev.h, LOSS[loss], x.p, t.p, x.shape_c(), x.ndim, REDUCTION[reduction], out.p))


def loss_bwd(dev, loss, dx, g, x, t, reduction="mean"):
    check(lib.nk_loss_bwd(dev.h, LOSS[loss], dx.p, g.p, x.p if x is not None else None, t.p, dx.shape_c(), dx.ndim,
                          REDUCTION[reduction]))


def nll_fwd(dev, x, t, out, reduction="mean"):
    check(lib.nk_nll_fwd(dev.h, x.p, t.p, x.shape_c(), x.ndim, REDUCTION[reduction], out.p))


def nll_bwd(dev, dx, g, t, reduction="mean"):
    check(lib.nk_nll_bwd(dev.h, dx.p, g.p, t.p, dx.shape_c(), dx.ndim, REDUCTION[reduction]))


def mv_fwd(dev, A, x, y):
    check(lib.nk_mv_fwd(dev.h, A.p, x.p, y.p, A.shape[0], A.shape[1]))


def mv_bwd_left(dev, dA, g, x):
    check(lib.nk_mv_bwd_left(dev.h, dA.p, g.p, x.p, dA.shape[0], dA.shape[1]))


def mv_bwd_right(dev, dx, A, g):
    check(lib.nk_mv_bwd_right(dev.h, dx.p, A.p, g.p, A.shape[0], A.shape[1]))


def vm_fwd(dev, v, B, y):
    check(lib.nk_vm_fwd(dev.h, v.p, B.p, y.p, B.shape[0], B.shape[1]))


def vm_bwd_left(dev, dv, B, g):
    check(lib.nk_vm_bwd_left(dev.h, dv.p, B.p, g.p, B.shape[0], B.shape[1]))


def vm_bwd_right(dev, dB, v, g):
    check(lib.nk_vm_bwd_right(dev.h, dB.p, v.p, g.p, dB.shape[0], dB.shape[1]))


def vv_fwd(dev, l, r, out):
    check(lib.nk_vv_fwd(dev.h, l.p, r.p, l.size, out.p))


def vv_bwd(dev, d_operand, other, g):
    check(lib.nk_vv_bwd(dev.h, d_operand.p, other.p, g.p, d_operand.size))


def softmax_fwd(dev, x, y, axis):
    check(lib.nk_softmax_fwd(dev.h, x.p, y.p, x.shape_c(), x.ndim, axis))


def softmax_bwd(dev, dx, g, y, axis, assign=False):
    check((lib.nk_softmax_bwd_assign if assign else lib.nk_softmax_bwd)(dev.h, dx.p, g.p, y.p, y.shape_c(), y.ndim, axis))


def log_softmax_fwd(dev, x, y, axis):
    check(lib.nk_log_softmax_fwd(dev.h, x.p, y.p, x.shape_c(), x.ndim, axis))


def log_softmax_bwd(dev, dx, g, y, axis, assign=False):
    check((lib.nk_log_softmax_bwd_assign if assign else lib.nk_log_softmax_bwd)(dev.h, dx.p, g.p, y.p, y.shape_c(), y.ndim, axis))


def dropout_fwd(dev, x, y, noise, p, train=True, seed=0, offset=0):
    check(lib.nk_dropout_fwd(dev.h, x.p, y.p, noise.p if noise is not None else None, x.size, float(p), int(train), seed, offset))


def dropout_bwd(dev, dx, g, noise, p, train=True, assign=False):
    check((lib.nk_dropout_bwd_assign if assign else lib.nk_dropout_bwd)(dev.h, dx.p, g.p, noise.p if noise is not None else None, dx.size, float(p), int(train)))


def scale_softmax_dropout_fwd(dev, scores, probs, out, noise, scale, p, train=True, seed=0, offset=0):
    L = scores.shape[-1]
    check(lib.nk_scale_softmax_dropout_fwd(dev.h, scores.p, probs.p if probs is not None else None, out.p, noise.p if noise is not None else None,
                                           scores.size // L, L, scale, float(p), int(train), seed, offset))


def scale_softmax_dropout_bwd(dev, d_scores, g_out, probs, noise, scale, p, train=True, seed=0, offset=0, assign=False):
    L = probs.shape[-1]
    check((lib.nk_scale_softmax_dropout_bwd_assign if assign else lib.nk_scale_softmax_dropout_bwd)(dev.h, d_scores.p, g_out.p, probs.p, noise.p if noise is not None else None,
                                           probs.size // L, L, scale, float(p), int(train), seed, offset))


def scale_softmax_dropout_bwd_from_scores(dev, d_scores, g_out, scores, noise, scale, p, train=True, seed=0, offset=0, assign=False):
    L = scores.shape[-1]
    check(lib.nk_scale_softmax_dropout_bwd_from_scores(dev.h, d_scores.p, g_out.p, scores.p, noise.p if noise is not None else None,
                                                       scores.size // L, L, scale, float(p), int(train), seed, offset, int(assign)))


def attention_supported(S, dh, p, train=True):
    return bool(lib.nk_attention_supported(S, dh, float(p), int(train)))


def attention_padded(S):
    """Row count / row stride of the fused attention core's scratch tensors (include/neuronika_hip.h)."""
    return (S + 31) // 32 * 32


def attention_fwd(dev, Q, K, V, scores, stats, mask_bits, out, B, S, H, dh, scale, p, train=True, seed=0, offset=0):
    """Fused attention core: scores (B*H,SP,SP), stats (B*H,SP,2), mask_bits (B*H,SP,SP/32 words held in an f32 array; None
    when dropout is inactive) and out (B*S,H*dh) are written, SP = S rounded up to a multiple of 32 (`attention_padded`).
    scores = stats = None: inference (out only)."""
    check(lib.nk_attention_fwd(dev.h, Q.p, K.p, V.p, scores.p if scores is not None else None, stats.p if stats is not None else None,
                               mask_bits.p if mask_bits is not None else None, out.p,
                               B, S, H, dh, scale, float(p), int(train), seed, offset))


def attention_bwd(dev, dQ, dK, dV, dS, dropped, dO, out, scores, stats, mask_bits, Q, K, V, B, S, H, dh, scale, p, train=True,
                  assign=(False, False, False)):
    """dS and dropped (B*H,SP,SP elements, scratch) are written; dQ / dK / dV (+)= the three input gradients per (sample, head)."""
    check(lib.nk_attention_bwd(dev.h, dQ.p, dK.p, dV.p, dS.p, dropped.p, dO.p, out.p, scores.p, stats.p,
                               mask_bits.p if mask_bits is not None else None, Q.p, K.p, V.p, B, S, H, dh, scale, float(p),
                               int(train), int(assign[0]), int(assign[1]), int(assign[2])))


def attention_qkv_fwd(dev, QKV, scores, stats, mask_bits, out, B, S, H, dh, scale, p, train=True, seed=0, offset=0):
    """attention_fwd with Q, K, V as the three column blocks of ONE (B*S, 3*H*dh) array."""
    check(lib.nk_attention_qkv_fwd(dev.h, QKV.p, scores.p if scores is not None else None, stats.p if stats is not None else None,
                                   mask_bits.p if mask_bits is not None else None, out.p, B, S, H, dh, scale, float(p), int(train), seed, offset))


def attention_qkv_bwd(dev, dQKV, dS, dropped, dO, out, scores, stats, mask_bits, QKV, B, S, H, dh, scale, p, train=True, assign=False):
    check(lib.nk_attention_qkv_bwd(dev.h, dQKV.p, dS.p, dropped.p, dO.p, out.p, scores.p, stats.p,
                                   mask_bits.p if mask_bits is not None else None, QKV.p, B, S, H, dh, scale, float(p), int(train), int(assign)))


def chunk_fwd(dev, x, y, chunk_no):
    check(lib.nk_chunk_fwd(dev.h, x.p, x.shape_c(), y.p, y.shape_c(), x.ndim, chunk_no))


def chunk_bwd(dev, dx, g, chunk_no):
    check(lib.nk_chunk_bwd(dev.h, dx.p, dx.shape_c(), g.p, g.shape_c(), dx.ndim, chunk_no))


def concat_fwd(dev, operands, out, axis):
    off = 0
    for o in operands:
        check(lib.nk_concat_fwd_part(dev.h, o.p, out.p, out.shape_c(), out.ndim, axis, off, o.shape[axis]))
        off += o.shape[axis]


def concat_bwd(dev, d_operands, g, axis, assign=False):
    off = 0
    for d in d_operands:
        check((lib.nk_concat_bwd_part_assign if assign else lib.nk_concat_bwd_part)(dev.h, d.p, g.p, g.shape_c(), g.ndim, axis, off, d.shape[axis]))
        off += d.shape[axis]


def transpose_fwd(dev, x, y):
    check(lib.nk_transpose_fwd(dev.h, x.p, y.p, x.shape_c(), x.ndim))


def transpose_bwd(dev, dx, g, assign=False):
    check((lib.nk_transpose_bwd_assign if assign else lib.nk_transpose_bwd)(dev.h, dx.p, g.p, dx.shape_c(), dx.ndim))


def split_heads_fwd(dev, x, y, B, S, H, dh):
    check(lib.nk_split_heads_fwd(dev.h, x.p, y.p, B, S, H, dh))


def split_heads_bwd(dev, dx, g, B, S, H, dh, assign=False):
    check((lib.nk_split_heads_bwd_assign if assign else lib.nk_split_heads_bwd)(dev.h, dx.p, g.p, B, S, H, dh))


def merge_heads_fwd(dev, x, y, B, S, H, dh):
    check(lib.nk_merge_heads_fwd(dev.h, x.p, y.p, B, S, H, dh))


def merge_heads_bwd(dev, dx, g, B, S, H, dh, assign=False):
    check((lib.nk_merge_heads_bwd_assign if assign else lib.nk_merge_heads_bwd)(dev.h, dx.p, g.p, B, S, H, dh))


def _p(a):
    return a.p if a is not None else None


def sgd_step(dev, w, grad, velocity=None, lr=0.01, momentum=0.0, dampening=0.0, nesterov=False, l1=0.0, l2=0.0):
    check(lib.nk_sgd_step(dev.h, w.p, grad.p, _p(velocity), w.size, lr, momentum, dampening, int(nesterov), l1, l2))


def sgd_step_multi(dev, ws, grads, velocities=None, lr=0.01, momentum=0.0, dampening=0.0, nesterov=False, l1=0.0, l2=0.0):
    n = len(ws)
    PA = C.c_void_p * n
    vel = PA(*[(_p(v) if v is not None else None) for v in velocities]) if velocities is not None else None
    check(lib.nk_sgd_step_multi(dev.h, n, PA(*[w.p for w in ws]), PA(*[g.p for g in grads]), vel, (C.c_size_t * n)(*[w.size for w in ws]),
                                lr, momentum, dampening, int(nesterov), l1, l2))


def adam_step(dev, w, grad, exp_avg, exp_avg_sq, max_exp_avg_sq=None, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
              step=1, l1=0.0, l2=0.0):
    check(lib.nk_adam_step(dev.h, w.p, grad.p, exp_avg.p, exp_avg_sq.p, _p(max_exp_avg_sq), w.size, lr, beta1, beta2,
                           eps, step, l1, l2))


def adagrad_step(dev, w, grad, grad_sq, lr=1e-2, lr_decay=0.0, eps=1e-10, step=1, l1=0.0, l2=0.0):
    check(lib.nk_adagrad_step(dev.h, w.p, grad.p, grad_sq.p, w.size, lr, lr_decay, eps, step, l1, l2))


def rmsprop_step(dev, w, grad, square_avg, grad_avg=None, buffer=None, lr=1e-2, alpha=0.99, eps=1e-8, momentum=0.0,
                 l1=0.0, l2=0.0):
    check(lib.nk_rmsprop_step(dev.h, w.p, grad.p, square_avg.p, _p(grad_avg), _p(buffer), w.size, lr, alpha, eps,
                              momentum, l1, l2))


class Comm:
    """One rank of the RCCL communicator (net-new; no reference counterpart)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        check(lib.nk_comm_unique_id(buf))
        return buf.raw

    def __init__(self, dev: Device, nranks: int, rank: int, uid: bytes | None, channels: int = 0, gbps: float = 0.0):
        """uid = None: a replica communicator (nk_comm_init_replicas) of `nranks` virtual ranks."""
        h = VP()
        if uid is None:
            check(lib.nk_comm_init_replicas(dev.h, nranks, int(channels), float(gbps), C.byref(h)))
        else:
            assert len(uid) == COMM_ID_BYTES
            check(lib.nk_comm_init_rank(dev.h, nranks, rank, uid, C.byref(h)))
        self.h, self.dev, self.rank, self.size = h, dev, rank, nranks

    @classmethod
    def init_all(cls, devs):
        """nk_comm_init_all: one communicator per device handle of this process (thread-per-GPU use)."""
        n = len(devs)
        out = (VP * n)()
        check(lib.nk_comm_init_all(n, (VP * n)(*[d.h for d in devs]), out))
        comms = []
        for i, d in enumerate(devs):
            c = cls.__new__(cls)
            c.h, c.dev, c.rank, c.size = VP(out[i]), d, i, n
            comms.append(c)
        return comms

    def allreduce_sum_async(self, buf: HipArray, after: Event | None = None, n: int | None = None):
        check(lib.nk_allreduce_sum_async(self.h, buf.p, buf.size if n is None else n, after.h if after else None))

    def allreduce_sum_group_async(self, bufs, after: Event | None = None):
        ptrs = (VP * len(bufs))(*[b.p for b in bufs])
        cnts = (C.c_size_t * len(bufs))(*[b.size for b in bufs])
        check(lib.nk_allreduce_sum_group_async(self.h, ptrs, cnts, len(bufs), after.h if after else None))

    def join(self):
        check(lib.nk_comm_join(self.h))

    def close(self):
        if self.h:
            lib.nk_comm_destroy(self.h)
            self.h = None
