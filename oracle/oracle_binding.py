"""oracle/oracle_binding.py — ctypes binding of liboracle.so.  TEST INFRASTRUCTURE ONLY (see oracle.hpp): imported by tests/,
tools/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; nothing under dynamicsparsearrays.jl_amd/ knows it exists.

The oracle exports the host-pointer subset of include/dsa.h under the prefix `ora_` (same call shapes), so the host-side mirror
of the reference's interface (dsa_amd.api) can drive it for side-by-side comparisons."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "liboracle.so")

# entry points of include/dsa.h the oracle does NOT export: device pointers, streams, shards, device checker, parity hooks
_DEVICE_ONLY = ("device_count", "set_device", "mat_spmv_dense_dev", "shard_range", "shard_create_from_coo", "shard_spmv_dev",
                "vec_check", "mat_check", "mat_set_stream", "vec_set_stream", "mat_sync", "vec_sync", "vec_dev_relayout",
                "vec_set_wait_policy", "mat_set_wait_policy", "pool_idle_bytes", "pool_trim", "dev_switches",
                "dbg_raw_find", "dbg_raw_insert", "dbg_raw_delete", "dbg_raw_purge", "dbg_raw_rebalance", "shard_allreduce_dev",
                "shard_spmv_allreduce_dev", "mat_col_view_dev", "mat_row_view_dev", "mat_spmv_sparse_begin", "mat_spmv_sparse_fetch",
                "mat_spmv_sparse_dev")


def build():
    srcs = [os.path.join(HERE, f) for f in ("oracle.cpp", "oracle_capi.cpp", "oracle.hpp")]
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"])


def load(dsa):
    """the oracle as a `Binding` of the dsa_amd module `dsa` (built on demand with g++)"""
    build()

    class OracleBinding(dsa.Binding):
        PREFIX = "ora"
        SIGNATURES = {k: v for k, v in dsa.Binding.SIGNATURES.items() if k not in _DEVICE_ONLY and not k.startswith("comm_")}

    return OracleBinding(SO)
