"""Column-range sharding of a PCSR matrix across the GPUs of one node.

Rank g owns the contiguous column-key range (g*n/G, (g+1)*n/G] as an INDEPENDENT reference-layout
PCSR of its sub-matrix (own capacity / height / semaphores / col_keys, both orientations restricted to
those columns) plus the matching slice of x.  `y_g = A[:, range_g] * x[range_g]`; the only data-path
collective is the sum of the partial y over the ranks — RCCL over xGMI on the GPUs (`backend="nccl"`),
gloo in the CPU tests.  One process per GPU; nothing else crosses ranks (writes go to the owner shard,
rebalances are local).  The reference has no counterpart: it is single-process.

x and y are torch tensors on the shard's device: CUDA tensors next to the HIP library — the partial product
is written straight into y's HBM through `dsa_shard_spmv_dev`, on torch's current stream, nothing bounces
through the host — and the collectives are issued on those tensors.  The three places that touch the library
(`_build`, `_device_of`, `spmv_partial`) are methods, so the world-size-2 gloo test (tests/cpu_shard.py) runs
every other line of this class — ranges, slices, the three schedules, write routing — with CPU tensors.

Three schedules for the sum of the m-entry partial results (config 4: m = 10^7, an 80 MB message):
  "all_reduce"   one RCCL all-reduce (ring / tree chosen by RCCL)
  "rs_ag"        reduce_scatter + all_gather (each rank reduces m/G entries, then the slices are gathered)
  "direct"       all_to_all of the G slices (every rank sends slice j of its partial y straight to rank j: G-1
                 point-to-point transfers on G-1 different xGMI links at once), local sum of the G received
                 slices, all_gather of the reduced slices — the schedule that uses the full mesh instead of a ring
All three leave the complete y on every rank (what the next column-generation step reads).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

SCHEDULES = ("all_reduce", "rs_ag", "direct")


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


class AbiComm:
    """The communicator of include/dsa.h (dsa_comm_*): RCCL bound inside libdsa_hip.so, the path a Julia host uses.  Rank 0 draws the
    128-byte id; `bcast(bytes_or_None) -> bytes` hands it to the other ranks — torch.distributed's object broadcast by default (any
    process group will do: the id is 128 bytes of host memory), MPI.Bcast from Julia."""

    def __init__(self, binding, rank, world, bcast=None, with_rccl=True):
        self.b, self.rank, self.world = binding, rank, world
        idb = None
        if world > 1 or with_rccl:
            buf = (C.c_uint8 * 128)()
            payload = None                      # (ok, bytes): the id, or the text of rank 0's failure — tagged, never told apart by length
            if rank == 0:
                # a failure here (no librccl.so) must reach EVERY rank: the others are about to wait for the id, so rank 0 hands them
                # the error text instead of leaving them in the broadcast (a single rank has nobody to tell: its error keeps its type)
                try:
                    binding.call("comm_unique_id", buf)
                    payload = (True, bytes(buf))
                except Exception as e:              # noqa: BLE001 — re-raised on every rank below
                    if world == 1:
                        raise
                    payload = (False, ("rank 0: %s" % e).encode())
            if world > 1:
                if bcast is None:
                    import torch.distributed as dist
                    box = [payload]
                    dist.broadcast_object_list(box, src=0)
                    payload = box[0]
                else:
                    payload = bcast(payload)
            ok, raw = payload if payload is not None else (False, b"rank 0 sent nothing")
            if not ok or len(raw) != 128:
                raise RuntimeError("no RCCL unique id: %s" % (raw.decode(errors="replace") if not ok else "%d bytes instead of 128" % len(raw)))
            idb = (C.c_uint8 * 128).from_buffer_copy(raw)
        self.h = C.c_void_p()
        binding.call("comm_init", rank, world, idb, C.byref(self.h))      # collective over the ranks

    def close(self):
        if self.h:
            self.b.call("comm_destroy", self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ColumnShard:
    """The local shard of a column-range sharded matrix.  `api` is the dsa_amd module, `binding` the
    library it runs on (the HIP product by default), `device` the torch device of x / y (default: the
    current CUDA device)."""

    def __init__(self, api, I, J_global, V, m, n_total, rank, world, binding=None, device=None, local_columns=False, comm=None):
        self.api, self.rank, self.world, self.m, self.n_total = api, rank, world, m, n_total
        self.comm = comm            # an AbiComm: the "all_reduce" schedule then runs RCCL behind the C ABI instead of torch.distributed
        self.col0, self.ncols = column_range(rank, world, n_total)
        I = np.ascontiguousarray(I, dtype=np.int64)
        J = np.ascontiguousarray(J_global, dtype=np.int64)
        V = np.ascontiguousarray(V, dtype=np.float64)
        self.binding = binding if binding is not None else api.product()
        self.A = self._build(I, J, V, local_columns)
        self.device = self._device_of(device)
        # slices of the "rs_ag" / "direct" schedules: m padded to a multiple of the world size
        self.chunk = -(-m // world)
        self._scratch = {}

    # ---- the library-facing part ---------------------------------------------------------------------------------------
    def _build(self, I, J, V, local_columns):
        b = self.binding
        if local_columns:
            # the caller generated only this rank's columns, already as local keys 1..ncols
            return self.api.dynamicsparse(I, J, V, self.m, self.ncols, binding=b)
        # the C-ABI shard constructor (include/dsa.h: dsa_shard_create_from_coo): the triples of this rank's column range as an
        # independent reference-layout matrix with local column keys 1..ncols
        h = C.c_void_p()
        b.call("shard_create_from_coo", I.ctypes.data_as(C.POINTER(C.c_int64)), J.ctypes.data_as(C.POINTER(C.c_int64)),
               V.ctypes.data_as(C.POINTER(C.c_double)), len(I), self.m, self.n_total, self.world, self.rank, C.byref(h))
        return self.api.DynamicSparseMatrix(b, h)

    def _device_of(self, device):
        import torch
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._bind_stream(dev)
        return dev

    def _bind_stream(self, dev):
        """kernels of this shard are ordered on torch's CURRENT stream of `dev`, like the producer of x and the collective that
        consumes y; re-bound by spmv_partial whenever the caller has switched streams since"""
        import torch
        s = torch.cuda.current_stream(dev).cuda_stream
        if getattr(self, "_stream", None) != s:
            self.binding.call("mat_set_stream", self.A.h, C.c_void_p(s))
            self._stream = s

    # ---- operands ----------------------------------------------------------------------------------------------------
    def x_slice(self, x_global):
        """this rank's slice of a global x (numpy) as a tensor on the shard's device."""
        import torch
        return torch.from_numpy(np.ascontiguousarray(x_global[self.col0:self.col0 + self.ncols], dtype=np.float64)).to(self.device)

    def new_y(self):
        import torch
        return torch.zeros(self.m, dtype=torch.float64, device=self.device)

    def _buf(self, name, n):
        import torch
        t = self._scratch.get(name)
        if t is None or t.numel() != n:
            t = torch.zeros(n, dtype=torch.float64, device=self.device)
            self._scratch[name] = t
        return t

    # ---- local product ----------------------------------------------------------------------------------------------
    def spmv_partial(self, x_local, y=None):
        """y_partial = A[:, range] * x[range] into the tensor y (length m) on the shard's device."""
        if y is None:
            y = self.new_y()
        self._bind_stream(self.device)
        self.binding.call("shard_spmv_dev", self.A.h, C.c_void_p(x_local.data_ptr()), self.ncols, C.c_void_p(y.data_ptr()), self.m)
        return y

    # ---- the collective ---------------------------------------------------------------------------------------------
    def reduce(self, y, schedule="all_reduce", async_op=False):
        """sum of the partial y over the ranks, in place; every rank ends with the complete y.
        async_op (all_reduce only): returns the work handle instead of waiting."""
        if self.comm is not None and schedule == "all_reduce" and not async_op:
            # stream-ordered behind the product on the shard's stream: nothing to wait for on the host
            self._bind_stream(self.device)
            self.binding.call("shard_allreduce_dev", self.comm.h, C.c_void_p(y.data_ptr()), self.m, C.c_void_p(self._stream))
            return None
        if self.world == 1:
            return None
        import torch.distributed as dist
        if schedule == "all_reduce":
            return dist.all_reduce(y, op=dist.ReduceOp.SUM, async_op=async_op) if async_op else dist.all_reduce(y, op=dist.ReduceOp.SUM)
        if async_op:
            raise ValueError("async_op is only offered for the all_reduce schedule")
        G, ch = self.world, self.chunk
        pad = self._buf("pad", G * ch)
        pad[:self.m].copy_(y)
        if G * ch > self.m:
            pad[self.m:].zero_()
        mine = self._buf("mine", ch)
        if schedule == "rs_ag":
            dist.reduce_scatter_tensor(mine, pad, op=dist.ReduceOp.SUM)
        elif schedule == "direct":
            recv = self._buf("recv", G * ch)
            dist.all_to_all_single(recv, pad)                 # slice j of every rank's partial y arrives on rank j
            mine.copy_(recv.view(G, ch).sum(dim=0))
        else:
            raise ValueError("schedule must be one of %s" % (SCHEDULES,))
        dist.all_gather_into_tensor(pad, mine)
        y.copy_(pad[:self.m])
        return None

    def spmv(self, x_local, y=None, schedule="all_reduce"):
        """y = A x: local SpMV + the sum over the ranks (identity when world == 1); returns the tensor y."""
        y = self.spmv_partial(x_local, y)
        self.reduce(y, schedule)
        return y

    # ---- writes -----------------------------------------------------------------------------------------------------
    def set(self, row, col_global, val):
        """A[row, col] = val on the owner shard (other ranks ignore the write)."""
        if self.col0 < col_global <= self.col0 + self.ncols:
            self.A[row, col_global - self.col0] = val
