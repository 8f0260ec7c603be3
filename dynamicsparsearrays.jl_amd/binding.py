"""ctypes binding of the C ABI declared in include/dsa.h.

`Binding(path)` binds `libdsa_hip.so`: every entry point of the header, named
``dsa_*`` (see :func:`product`).  `PREFIX` and `SIGNATURES` are class attributes
so that a test harness can bind another library that exports the same call shapes
(the checker under ``oracle/`` does, in ``oracle/oracle_binding.py``); nothing in
this package does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

I64 = C.c_int64
I32 = C.c_int32
F64 = C.c_double
P_I64 = C.POINTER(C.c_int64)
P_F64 = C.POINTER(C.c_double)
P_U8 = C.POINTER(C.c_uint8)
P_I32 = C.POINTER(C.c_int32)
VP = C.c_void_p

INFO_COUNT = 17
INFO = dict(capacity=0, segment_capacity=1, nb_segments=2, nb_elements=3, height=4,
            nb_partitions=5, table_len=6, stat_window_slots=7, stat_rebalances=8,
            stat_extends=9, stat_shrinks=10, stat_par_rounds=11, stat_par_ops=12, stat_seq_ops=13,
            stat_spmv_nomemset=14, hbm_bytes=15, stat_grid_rebalances=16)

OK, EARG, EBOUNDS, EDELETED, EFULL, EMODE, EASSERT, EHIP, ECAP, EKEY, ERCCL = range(11)
STATUS_NAMES = ["OK", "EARG", "EBOUNDS", "EDELETED", "EFULL", "EMODE", "EASSERT", "EHIP", "ECAP", "EKEY", "ERCCL"]


class DsaError(RuntimeError):
    """Base class; `.code` is the C status.  Subclasses mirror the Julia exception types."""

    def __init__(self, code, msg):
        super().__init__(f"{STATUS_NAMES[code] if 0 <= code < len(STATUS_NAMES) else code}: {msg}")
        self.code = code


class DsaArgumentError(DsaError, ValueError):      # Julia ArgumentError
    pass


class DsaBoundsError(DsaError, IndexError):        # Julia BoundsError
    pass


class DsaErrorException(DsaError):                 # Julia ErrorException / AssertionError
    pass


def _exc_for(code, msg):
    if code in (EARG, EKEY):
        return DsaArgumentError(code, msg)
    if code == EBOUNDS:
        return DsaBoundsError(code, msg)
    return DsaErrorException(code, msg)


# name -> argtypes (restype is always int32 unless listed in _SPECIAL)
_SIGS = {
    "vec_create": [P_I64, P_F64, I64, I32, I64, C.POINTER(VP)],
    "vec_create_empty": [C.POINTER(VP)],
    "vec_destroy": [VP],
    "vec_get": [VP, I64, P_F64],
    "vec_get_batch": [VP, P_I64, I64, P_F64],
    "vec_set": [VP, I64, F64],
    "vec_set_batch": [VP, P_I64, P_F64, I64],
    "vec_nnz": [VP, P_I64],
    "vec_len": [VP, P_I64],
    "vec_shrink_size": [VP],
    "vec_nonzeros": [VP, P_I64, P_F64, I64, P_I64],
    "vec_equal": [VP, VP, P_I32],
    "vec_axpby": [VP, F64, VP, F64, P_I64, P_F64, I64, P_I64],
    "vec_info": [VP, P_I64],
    "vec_export_layout": [VP, P_I64, P_F64, P_U8, I64],
    "vec_rebalance_root": [VP],
    "vec_import_layout": [P_I64, P_F64, P_U8, I64, I64, I64, C.POINTER(VP)],
    "pcsc_import_layout": [P_I64, P_F64, P_U8, I64, I64, P_I64, I64, C.POINTER(VP)],
    "pcsc_create": [P_I64, I64, P_I64, P_F64, I32, C.POINTER(VP)],
    "pcsc_create_empty": [C.POINTER(VP)],
    "pcsc_destroy": [VP],
    "pcsc_get": [VP, I64, I64, P_F64],
    "pcsc_set": [VP, F64, I64, I64],
    "pcsc_deletepartition": [VP, I64],
    "pcsc_nnz": [VP, P_I64],
    "pcsc_nbpartitions": [VP, P_I64],
    "pcsc_info": [VP, P_I64],
    "pcsc_export_layout": [VP, P_I64, P_F64, P_U8, I64, P_I64, I64],
    "mat_create_from_coo": [P_I64, P_I64, P_F64, I64, I64, I64, C.POINTER(VP)],
    "mat_create_empty": [I32, C.POINTER(VP)],
    "mat_destroy": [VP],
    "mat_set": [VP, F64, I64, I64],
    "mat_set_batch": [VP, P_I64, P_I64, P_F64, I64],
    "mat_get": [VP, I64, I64, P_F64],
    "mat_get_batch": [VP, P_I64, P_I64, I64, P_F64],
    "mat_addrow": [VP, I64, P_I64, P_F64, I64],
    "mat_closefillmode": [VP],
    "mat_deletecolumn": [VP, I64],
    "mat_deleterow": [VP, I64],
    "mat_col_view": [VP, I64, P_I64, P_F64, I64, P_I64],
    "mat_row_view": [VP, I64, P_I64, P_F64, I64, P_I64],
    "mat_col_slice": [VP, I64, C.POINTER(VP)],
    "mat_row_slice": [VP, I64, C.POINTER(VP)],
    "mat_nnz": [VP, P_I64],
    "mat_size": [VP, P_I64, P_I64],
    "mat_nbpartitions": [VP, I32, P_I64],
    "mat_info": [VP, I32, P_I64],
    "mat_export_layout": [VP, I32, P_I64, P_F64, P_U8, I64, P_I64, P_I64, P_U8, I64],
    "mat_rebalance_root": [VP, I32],
    "mat_spmv_dense": [VP, I32, P_F64, I64, P_F64, I64],
    "mat_spmv_sparse": [VP, I32, P_I64, P_F64, I64, P_I64, P_F64, I64, P_I64],
}
# device-resident operands, streams, shards, invariant checker, parity hooks
_DEVICE_SIGS = {
    "device_count": [P_I32],
    "set_device": [I32],
    "mat_spmv_dense_dev": [VP, I32, I32, VP, I64, VP, I64],
    "mat_col_view_dev": [VP, I64, VP, VP, I64, P_I64],
    "mat_spmv_sparse_begin": [VP, I32, P_I64, P_F64, I64, P_I64],
    "mat_spmv_sparse_fetch": [VP, P_I64, P_F64, I64, P_I64],
    "mat_spmv_sparse_dev": [VP, I32, VP, VP, I64, VP, VP, I64, VP],
    "mat_row_view_dev": [VP, I64, VP, VP, I64, P_I64],
    "shard_range": [I64, I32, I32, P_I64, P_I64],
    "shard_create_from_coo": [P_I64, P_I64, P_F64, I64, I64, I64, I32, I32, C.POINTER(VP)],
    "shard_spmv_dev": [VP, VP, I64, VP, I64],
    "comm_unique_id": [P_U8],
    "comm_init": [I32, I32, P_U8, C.POINTER(VP)],
    "comm_destroy": [VP],
    "comm_info": [VP, P_I32, P_I32],
    "shard_allreduce_dev": [VP, VP, I64, VP],
    "shard_spmv_allreduce_dev": [VP, VP, VP, I64, VP, I64],
    "vec_check": [VP, P_I64],
    "mat_check": [VP, I32, P_I64],
    "mat_set_stream": [VP, VP],
    "vec_set_stream": [VP, VP],
    "mat_sync": [VP],
    "vec_sync": [VP],
    "vec_dev_relayout": [VP, I32],
    "vec_set_wait_policy": [VP, I32],
    "mat_set_wait_policy": [VP, I32],
    "pool_idle_bytes": [P_I64],
    "pool_trim": [I64],
    "dev_switches": [C.c_char_p, I64, P_I32],
    # parity hooks: device slot-array primitives on raw slot arrays (include/dsa.h)
    "dbg_raw_find": [P_I64, P_F64, P_U8, I64, I64, I64, I64, I32, I32, P_I64, P_I32, P_I64, P_F64],
    "dbg_raw_insert": [P_I64, P_F64, P_U8, I64, I64, F64, I64, I64, P_I64, I64, I32, I32, P_I64, P_I32],
    "dbg_raw_delete": [P_I64, P_F64, P_U8, I64, I64, I64, I64, I32, I32, P_I64, P_I32],
    "dbg_raw_purge": [P_I64, P_F64, P_U8, I64, I64, I64, P_I64, P_I64],
    "dbg_raw_rebalance": [P_I64, P_F64, P_U8, I64, I64, I64, P_I64, I64, I32],
}


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(P_I64)


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(P_F64)


SIGNATURES = {**_SIGS, **_DEVICE_SIGS}


class Binding:
    PREFIX = "dsa"
    SIGNATURES = SIGNATURES

    def __init__(self, path: str):
        self.path = path
        self.prefix = self.PREFIX
        self.lib = C.CDLL(path)
        for name, argtypes in self.SIGNATURES.items():
            fn = getattr(self.lib, f"{self.prefix}_{name}")   # AttributeError if the symbol is missing
            fn.argtypes = argtypes
            fn.restype = I32
            setattr(self, "_" + name, fn)
        self._errmsg = getattr(self.lib, f"{self.prefix}_last_error_message")
        self._errmsg.restype = C.c_char_p
        self._errmsg.argtypes = []

    @classmethod
    def declared_symbols(cls):
        return list(cls.SIGNATURES) + ["last_error_message"]

    def call(self, name, *args):
        rc = getattr(self, "_" + name)(*args)
        if rc != OK:
            raise _exc_for(rc, self._errmsg().decode("utf-8", "replace"))


_PRODUCT = None


def product_library_path() -> str:
    """csrc/libdsa_hip.so.  A development process (DSA_DEV=1, like every other code-path switch: include/dsa.h, dsa_dev_switches) may name
    another build of the SAME sources with DSA_LIBRARY (the footprint-check build csrc/libdsa_hip_fpcheck.so: tools/fuzz.py, the suite's
    check test) — a path that does not exist fails loudly like the default.  A release process ignores an inherited DSA_LIBRARY."""
    override = os.environ.get("DSA_LIBRARY")
    if override and os.environ.get("DSA_DEV") == "1":
        return os.path.abspath(override)
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libdsa_hip.so")


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (soname libamdhip64.so.7).  Two HIP runtimes
    in one process cannot both open the GPU, so when torch is installed its copy is loaded first
    (without importing torch): libdsa_hip.so's DT_NEEDED `libamdhip64.so.7` then resolves to it, and a
    later `import torch` finds the very same file already mapped."""
    import importlib.util
    import sys
    if os.environ.get("DSA_HIP_RUNTIME", "torch") != "torch":
        return
    if "torch" in sys.modules:
        return          # torch's runtime is already mapped
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def product() -> Binding:
    """The HIP product library.  Fails loudly if it has not been built — there is no fallback."""
    global _PRODUCT
    if _PRODUCT is None:
        path = product_library_path()
        _share_hip_runtime_with_torch()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        _PRODUCT = Binding(path)
    return _PRODUCT
