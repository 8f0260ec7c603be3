"""CPU test of the N > 1 path: world_size-2 gloo, column-range sharding + the three schedules of the sum of y
(all_reduce, reduce_scatter + all_gather, all_to_all + local sum + all_gather).  The product class sharding.ColumnShard runs
with its three library-facing methods pointed at the CPU oracle and CPU tensors (tests/cpu_shard.py: there is no GPU in the build
container); on the GPUs the same ranges, slices and schedules run with CUDA tensors on the HIP library and RCCL (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_problem():
    rng = np.random.default_rng(7)
    m, n, nnz = 301, 401, 4000          # odd n: uneven column ranges; odd m: the sliced schedules pad y
    I = rng.integers(1, m + 1, nnz)
    J = rng.integers(1, n + 1, nnz)
    V = rng.integers(1, 10, nnz).astype(np.float64)
    x = 1.0 + rng.random(n)
    return m, n, I, J, V, x


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import dsa_loader
    dsa = dsa_loader.load()
    from dsa_amd import sharding          # registered by dsa_loader
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding
    import cpu_shard
    ora = oracle_binding.load(dsa)
    CpuColumnShard = cpu_shard.make_cpu_shard_class(sharding)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    m, n, I, J, V, x = _make_problem()
    sh = CpuColumnShard(dsa, I, J, V, m, n, rank, world, binding=ora)
    xs = sh.x_slice(x)                                   # a tensor on the shard's device (CPU next to the oracle)
    ys = [sh.spmv(xs, schedule=s).numpy().copy() for s in sharding.SCHEDULES]
    # the overlapped form bench.py uses: partial product, asynchronous all-reduce, wait
    yb = sh.new_y()
    sh.spmv_partial(xs, yb)
    work = sh.reduce(yb, "all_reduce", async_op=True)
    work.wait()
    ys.append(yb.numpy().copy())
    # a write routed to its owner shard, then another product
    sh.set(5, 400, 3.5)
    sh.set(7, 2, 1.25)
    y2 = sh.spmv(xs).numpy().copy()
    np.save(os.path.join(out_dir, f"y_{rank}.npy"), np.stack(ys + [y2]))
    dist.barrier()
    dist.destroy_process_group()


def test_column_sharded_spmv_world2(dsa, oracle, tmp_path):
    from dsa_amd import sharding
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    m, n, I, J, V, x = _make_problem()
    full = dsa.dynamicsparse(I, J, V, m, n, binding=oracle)
    ref = full.mul(x)
    full[5, 400] = 3.5
    full[7, 2] = 1.25
    ref2 = full.mul(x)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"y_{r}.npy"))
        for k in range(len(sharding.SCHEDULES) + 1):      # all_reduce, rs_ag, direct, async all_reduce
            np.testing.assert_allclose(got[k], ref, rtol=1e-12, atol=0)
        np.testing.assert_allclose(got[-1], ref2, rtol=1e-12, atol=0)
    # the column ranges tile 1..n exactly
    cover = []
    for r in range(world):
        c0, nc = sharding.column_range(r, world, n)
        cover += list(range(c0 + 1, c0 + nc + 1))
    assert cover == list(range(1, n + 1))
    assert sharding.owner_of_column(400, world, n) == 1 and sharding.owner_of_column(2, world, n) == 0


def test_abi_communicator_rank_plumbing_with_a_stub_library_and_a_stub_broadcast(dsa):
    """sharding.AbiComm without a GPU: the id is drawn on rank 0 only, handed to the other ranks through `bcast` (MPI.Bcast from Julia,
    torch.distributed's object broadcast in bench.py), and every rank calls comm_init(rank, world, the SAME 128 bytes); a rank 0 that
    cannot load RCCL hands the error to the others instead of leaving them in the broadcast.  The library is a stub that records
    the calls; the real calls run in tests/test_hip_parity.py::test_abi_communicator_world_1_rccl_smoke on the GPU."""
    import ctypes as C
    from dsa_amd import sharding

    class Stub:
        def __init__(self, fail_id=False):
            self.calls, self.fail_id = [], fail_id

        def call(self, name, *args):
            if name == "comm_unique_id":
                if self.fail_id:
                    raise RuntimeError("ERCCL: librccl.so not found")
                for i in range(128):
                    args[0][i] = (7 * i + 3) % 256
            elif name == "comm_init":
                rank, world, idb, out = args
                self.calls.append(("comm_init", rank, world, bytes(idb) if idb is not None else None))
                C.cast(out, C.POINTER(C.c_void_p))[0] = C.c_void_p(0x1000 + rank)
                return
            self.calls.append((name,))

    wire = {}

    def bcast_from(rank):
        def f(raw):
            if rank == 0:
                wire["id"] = raw
            return wire["id"]
        return f

    world = 4
    stubs = [Stub() for _ in range(world)]
    comms = [sharding.AbiComm(stubs[r], r, world, bcast=bcast_from(r)) for r in range(world)]      # rank 0 first, as a broadcast orders it
    expect = bytes((7 * i + 3) % 256 for i in range(128))
    for r in range(world):
        inits = [c for c in stubs[r].calls if c[0] == "comm_init"]
        assert inits == [("comm_init", r, world, expect)], (r, stubs[r].calls)
        assert (("comm_unique_id",) in stubs[r].calls) == (r == 0)
        assert comms[r].h.value == 0x1000 + r
        comms[r].close()
        assert ("comm_destroy",) in stubs[r].calls and not comms[r].h
    # world 1 without RCCL: no id, a communicator without a collective behind it
    s1 = Stub()
    c1 = sharding.AbiComm(s1, 0, 1, with_rccl=False)
    assert s1.calls == [("comm_init", 0, 1, None)]
    c1.close()
    # rank 0 cannot draw the id: every rank raises, nobody calls comm_init
    wire.clear()
    bad = [Stub(fail_id=(r == 0)) for r in range(2)]
    for r in range(2):
        with pytest.raises(RuntimeError, match="no RCCL unique id"):
            sharding.AbiComm(bad[r], r, 2, bcast=bcast_from(r))
        assert not [c for c in bad[r].calls if c[0] == "comm_init"]
