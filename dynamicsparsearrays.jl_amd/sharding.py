"""Column-range sharding of a PCSR matrix across the GPUs of one node.

Rank g owns the contiguous column-key range (g*n/G, (g+1)*n/G] as an INDEPENDENT reference-layout
PCSR of its sub-matrix (own capacity / height / semaphores / col_keys, both orientations restricted to
those columns) plus the matching slice of x.  `y_g = A[:, range_g] * x[range_g]`; the only data-path
collective is one all-reduce (sum) of y — RCCL over xGMI on the GPUs (`backend="nccl"`), gloo in the
CPU tests.  One process per GPU; nothing else crosses ranks (writes go to the owner shard, rebalances
are local).  The reference has no counterpart: it is single-process.
"""
from __future__ import annotations

import numpy as np


def column_range(rank: int, world: int, n_total: int):
    """(col0, ncols): rank owns global columns col0+1 .. col0+ncols (1-based keys)."""
    base, rem = divmod(n_total, world)
    col0 = rank * base + min(rank, rem)
    return col0, base + (1 if rank < rem else 0)


def owner_of_column(col: int, world: int, n_total: int) -> int:
    for r in range(world):
        c0, nc = column_range(r, world, n_total)
        if c0 < col <= c0 + nc:
            return r
    raise IndexError(col)


class ColumnShard:
    """The local shard of a column-range sharded matrix.  `api` is the dsa_amd module, `binding` the
    library it runs on (the HIP product by default)."""

    def __init__(self, api, I, J_global, V, m, n_total, rank, world, binding=None):
        self.api, self.rank, self.world, self.m, self.n_total = api, rank, world, m, n_total
        self.col0, self.ncols = column_range(rank, world, n_total)
        I = np.asarray(I, dtype=np.int64)
        J = np.asarray(J_global, dtype=np.int64)
        V = np.asarray(V, dtype=np.float64)
        b = binding if binding is not None else api.product()
        if b.device_api:
            # the C-ABI shard constructor (include/dsa.h: dsa_shard_create_from_coo) — same split, done inside the library
            import ctypes as C
            h = C.c_void_p()
            b.call("shard_create_from_coo", I.ctypes.data_as(C.POINTER(C.c_int64)), J.ctypes.data_as(C.POINTER(C.c_int64)),
                   V.ctypes.data_as(C.POINTER(C.c_double)), len(I), m, n_total, world, rank, C.byref(h))
            self.A = api.DynamicSparseMatrix(b, h)
        else:
            mine = (J > self.col0) & (J <= self.col0 + self.ncols)
            # local column keys 1..ncols: the shard is the reference layout of its own sub-matrix
            self.A = api.dynamicsparse(I[mine], J[mine] - self.col0, V[mine], m, self.ncols, binding=b)

    def x_slice(self, x_global):
        return np.ascontiguousarray(x_global[self.col0:self.col0 + self.ncols])

    def spmv_partial(self, x_local):
        """partial y (length m) of this shard, host arrays."""
        return self.A.mul(np.asarray(x_local, dtype=np.float64), dense_out=self.m)

    def spmv(self, x_local):
        """y = A x: local SpMV + all-reduce of the partial results (identity when world == 1)."""
        import torch
        import torch.distributed as dist
        y = torch.from_numpy(np.ascontiguousarray(self.spmv_partial(x_local)))
        if self.world > 1:
            dist.all_reduce(y, op=dist.ReduceOp.SUM)
        return y.numpy()

    def set(self, row, col_global, val):
        """A[row, col] = val on the owner shard (other ranks ignore the write)."""
        if self.col0 < col_global <= self.col0 + self.ncols:
            self.A[row, col_global - self.col0] = val
