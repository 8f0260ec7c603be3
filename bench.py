#!/usr/bin/env python3
"""bench.py — SpMV GB/s (% of HBM roofline) on the 10M-nnz PCSR of BASELINE.json, 1 -> N MI355X,
plus inserts/s and rebalance GB/s as extra fields of the same JSON line.

A "step" is one pass of the hot path over one batch of synthetic input: y = A x on the
device-resident PCSR, followed — when N > 1 — by the sum of the partial y over the ranks (RCCL over xGMI).
  N = 1 : config C3 of SURVEY.md §8d (the configuration the metric is quoted on): 1M x 1M Float64, 10 entries per
          column = 10M nnz, dense x.
  N > 1 : config C4 per GPU: 10M rows, 1.25M columns and 12.5M nnz per rank (column-range sharding, weak scaling; at
          N = 8 this is BASELINE config 4 exactly: 10M x 10M, 100M nnz), y = 10M doubles = 80 MB summed over the ranks.
The multi-rank path is dynamicsparsearrays.jl_amd/sharding.py (ColumnShard): the same class the gloo tests run.

Launch: python bench.py --gpus 1            (default)
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix_array(seed, n, start=0):
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def unit12(seed, n, start=0):
    return 1.0 + (splitmix_array(seed, n, start) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def c3_triplets(m, ncols, per, col0, seed_rows, seed_vals):
    """every column gets exactly `per` distinct rows 1 + z % m (re-draw on in-column duplicate)."""
    n = ncols * per
    rows = 1 + (splitmix_array(seed_rows, n, start=col0 * per) % np.uint64(m)).astype(np.int64)
    cols_local = np.repeat(np.arange(ncols, dtype=np.int64), per)
    extra = 0
    while True:
        key = cols_local * np.int64(m + 1) + rows
        order = np.argsort(key, kind="stable")
        ks = key[order]
        dup = np.zeros(n, dtype=bool)
        dup[order[1:]] = ks[1:] == ks[:-1]
        nd = int(dup.sum())
        if nd == 0:
            break
        rows[dup] = 1 + (splitmix_array(seed_rows + 1000 + extra, nd) % np.uint64(m)).astype(np.int64)
        extra += 1
    vals = unit12(seed_vals, n, start=col0 * per)
    return rows, cols_local + 1 + col0, vals


def c3_insert_leg(m, n, n_random=100_000, n_new_cols=10_000, per=10):
    """The writes of the `inserts_on_c3` leg (bench.py) and of its parity test (tests/test_hip_parity.py): n_random uniformly random
    A[i, j] = v on the m x n matrix, then n_new_cols NEW columns n+1 .. n+n_new_cols with `per` distinct ascending rows each."""
    ri = 1 + (splitmix_array(61, n_random) % np.uint64(m)).astype(np.int64)
    rj = 1 + (splitmix_array(62, n_random) % np.uint64(n)).astype(np.int64)
    rv = unit12(63, n_random)
    z = 1 + (splitmix_array(64, n_new_cols * per * 2) % np.uint64(m)).astype(np.int64)
    ai, aj = [], []
    pos = 0
    for c in range(n_new_cols):
        seen = set()
        while len(seen) < per:
            seen.add(int(z[pos])); pos += 1
        ai += sorted(seen); aj += [n + 1 + c] * per
    return (ri, rj, rv), (np.array(ai, dtype=np.int64), np.array(aj, dtype=np.int64), unit12(65, len(ai)))


def cold_launch_us(torch, dev, stream, fn, nrep=8, evict_bytes=1 << 30):
    """Time of ONE launch of fn from cold caches: ONE pair of HIP events around nrep x (a device write of evict_bytes, the launch),
    minus ONE pair around nrep x (the same write) alone, divided by nrep.  The write — 1 GiB: four times the 256 MB Infinity Cache,
    far beyond the 8 x 4 MB L2s — makes the operands of every launch come from HBM; no event packet sits next to a launch (a pair
    around every single launch read ~4 us longer than the kernel trace's duration of the same launch: rounds 3-4)."""
    scr = torch.empty(evict_bytes // 4, dtype=torch.float32, device=dev)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for i in range(2):                      # warm both sequences once
        scr.fill_(float(i)); fn()
    torch.cuda.synchronize()
    e[0].record(stream)
    for i in range(nrep):
        scr.fill_(float(i))
        fn()
    e[1].record(stream)
    e[2].record(stream)
    for i in range(nrep):
        scr.fill_(float(i + 1))
    e[3].record(stream)
    torch.cuda.synchronize()
    del scr
    return float((e[0].elapsed_time(e[1]) - e[2].elapsed_time(e[3])) * 1e3 / nrep)


def banded_floor(us):
    """the L2-hit floor of the banded extra (tools/scripts/gather_floor3.sh: 2^24 slots streamed, 10 M gathers from a 1 MB table)"""
    try:
        with open(os.path.join(ROOT, "profiles", "gather_floor_banded.json")) as f:
            g = json.load(f)
        return {"keyed_single_pass_floor_us": g["keyed_single_pass_us"], "kernel_over_floor": round(us / g["keyed_single_pass_us"], 3), "floor_source": g["source"]}
    except (OSError, KeyError, ValueError):
        return {}


def kernel_source_sha():
    """sha256 over the kernel sources whose traffic the committed PMC summaries describe: a summary made from other
    sources is stale and is not quoted."""
    import hashlib
    h = hashlib.sha256()
    for f in ("spmv.hip", "rebalance.hip", "dsa_dev.h"):
        with open(os.path.join(ROOT, "dynamicsparsearrays.jl_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def committed_pmc():
    """newest profiles/*_pmc_summary.json made from the CURRENT kernel sources (tools/summarize_prof.py records the hash), or None."""
    import glob
    sha = kernel_source_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), reverse=True):
        try:
            with open(f) as fh:
                pm = json.load(fh)
        except Exception:
            continue
        if pm.get("kernel_source_sha") == sha:
            return os.path.relpath(f, ROOT), pm
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=("auto", "c3", "c4"), default="auto",
                    help="c3: 1M rows, 1M columns and 10M nnz per GPU (BASELINE config 3; the N = 1 default).  c4: 10M rows, 1.25M columns "
                         "and 12.5M nnz per GPU, y = 10M doubles (the per-GPU shard of BASELINE config 4: at --gpus 8 the 10M x 10M, "
                         "100M-nnz matrix exactly; the N > 1 default)")
    ap.add_argument("--rows", type=int, default=None)
    ap.add_argument("--cols-per-gpu", type=int, default=None)
    ap.add_argument("--per-col", type=int, default=10)
    ap.add_argument("--schedule", choices=("all_reduce", "rs_ag", "direct"), default="all_reduce",
                    help="how the partial y are summed over the ranks in the timed steps (sharding.py)")
    ap.add_argument("--all-schedules", action="store_true",
                    help="N > 1: also time the two schedules that are NOT used in the timed steps (extra collectives: off by default so that "
                         "an unexpected failure of one of them on some node cannot cost the headline line)")
    ap.add_argument("--collective", choices=("torch", "abi"), default="torch",
                    help="N > 1: who issues the all-reduce of y — torch.distributed (RCCL through PyTorch; default) or the C ABI of the library "
                         "itself (dsa_comm_* / dsa_shard_allreduce_dev: RCCL bound inside libdsa_hip.so, the path a Julia host uses; "
                         "stream-ordered behind the SpMV, all_reduce schedule only)")
    ap.add_argument("--dump-y", default=None,
                    help="rank 0 writes y = A x of every reduction schedule (after the timed steps) to this .npz file: what a test compares with "
                         "the CPU oracle (tests/test_hip_parity.py::test_bench_two_ranks_on_one_gpu_gloo_rehearsal)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the inserts/s and rebalance legs")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    # DSA_BENCH_SAME_GPU=1 + DSA_BENCH_BACKEND=gloo: dev-only way to exercise the multi-rank code path on a 1-GPU box
    if os.environ.get("DSA_BENCH_SAME_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = "none"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DSA_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
        # start-up self-check: every rank is there and they agree on the world — a sum of ones over the group must give WORLD_SIZE on
        # every rank (a rank that joined a different rendezvous, or a collective that silently degenerates, shows up here and not as
        # a wrong throughput); the result is printed in the line as `rccl_ranks`
        chk = torch.ones(1, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(chk)
        ranks_seen = int(chk.item())
        if ranks_seen != world:
            raise SystemExit(f"rank {rank}: the process group answers with {ranks_seen} ranks, WORLD_SIZE says {world}")
    else:
        ranks_seen = 1

    import dsa_loader
    dsa = dsa_loader.load()
    hip = dsa.product()                      # raises if libdsa_hip.so is missing: no fallback
    hip.call("set_device", local_rank)

    from dsa_amd import sharding
    cfg = args.config if args.config != "auto" else ("c3" if world == 1 else "c4")
    m = args.rows if args.rows is not None else (1_000_000 if cfg == "c3" else 10_000_000)
    ncl_per = args.cols_per_gpu if args.cols_per_gpu is not None else (1_000_000 if cfg == "c3" else 1_250_000)
    per = args.per_col
    seeds = (5, 6, 7) if cfg == "c3" else (8, 9, 10)          # rows / values / x
    n_total = world * ncl_per
    col0, ncl = sharding.column_range(rank, world, n_total)     # contiguous column-key range of this rank
    I, J, V = c3_triplets(m, ncl, per, col0, seed_rows=seeds[0], seed_vals=seeds[1])
    # the shard is the reference-layout PCSR of its own sub-matrix: local column keys 1..ncl (only this rank's columns are generated)
    t0 = time.time()
    abi_comm = None
    collective_note = None
    if world > 1 and args.collective == "abi" and args.schedule == "all_reduce":
        # every rank must take the same path: the outcome of the (collective) communicator set-up is agreed on before it is used
        try:
            abi_comm = sharding.AbiComm(hip, rank, world)
            ok_local = 1
        except Exception as e:
            ok_local = 0
            collective_note = "ABI communicator failed on rank %d (%s): torch.distributed all_reduce instead" % (rank, str(e)[:120])
        okt = torch.tensor([ok_local], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()) == 0:
            if abi_comm is not None:
                abi_comm.close()
            abi_comm = None
            collective_note = collective_note or "ABI communicator failed on another rank: torch.distributed all_reduce instead"
    shard = sharding.ColumnShard(dsa, I, np.ascontiguousarray(J - col0), V, m, n_total, rank, world, binding=hip, device=dev,
                                 local_columns=True, comm=abi_comm)      # bulk build of both orientations on the device (incl. the H2D of I, J, V)
    build_s = time.time() - t0
    A = shard.A
    info_row = A.info(dsa.ROWMAJOR)
    cap = info_row["capacity"]
    nnz = len(I)

    stream = torch.cuda.current_stream()
    x = torch.from_numpy(unit12(seeds[2], ncl, start=col0)).to(dev)
    # two output vectors: the all-reduce of step k (RCCL's stream) overlaps the SpMV of step k+1
    overlap = world > 1 and args.schedule == "all_reduce" and abi_comm is None      # (the ABI collective is stream-ordered behind the product)
    ys = [shard.new_y() for _ in range(2 if overlap else 1)]
    pending = [None] * len(ys)

    def step(k):
        b = k % len(ys)
        if pending[b] is not None:
            pending[b].wait()                     # the collective that last used this buffer (two steps ago)
            pending[b] = None
        shard.spmv_partial(x, ys[b])             # dsa_shard_spmv_dev: HIP kernel on torch's stream, y written in HBM
        if world > 1:
            pending[b] = shard.reduce(ys[b], args.schedule, async_op=overlap)

    def drain():
        for b in range(len(ys)):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    def barrier():
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        if world == 1:
            return v
        tt = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    for k in range(args.warmup):
        step(k)
    barrier()
    # ONE pair of HIP events brackets the timed region on the stream the kernel is launched on.  (Rounds 1-4 recorded a pair around every
    # launch inside the region: the two event packets cost ~10 us of device time per step — 132.9 us per step for a 122.9 us kernel — and
    # each interval came out ~3 us longer than the kernel trace's duration of the same launch.)
    ev_region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    import gc
    gc.collect(); gc.disable()          # a collector pass of the host interpreter (~3 ms with torch loaded) is not part of a step
    t1 = time.perf_counter()
    ev_region[0].record(stream)
    for k in range(args.steps):
        step(k)
    ev_region[1].record(stream)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t1)
    gc.enable()
    ms_per_step = elapsed * 1e3 / args.steps
    region_ms = ev_region[0].elapsed_time(ev_region[1]) / args.steps
    # the kernel's average launch duration: at N = 1 the region holds nothing but the K launches, back to back on one stream, so the
    # bracket / K is it (idle gaps between launches, if the host fell behind, are charged to the kernel); at N > 1 the stream also
    # waits for collectives inside the region, so the launches are timed again right behind it, K of them alone between one pair
    if world == 1:
        kern_ms = region_ms
    else:
        ev_k = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev_k[0].record(stream)
        for k in range(args.steps):
            shard.spmv_partial(x, ys[0])
        ev_k[1].record(stream)
        torch.cuda.synchronize()
        kern_ms = ev_k[0].elapsed_time(ev_k[1]) / args.steps
    # cross-check in the old form (a pair of events around single launches, outside the timed region)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 20))]
    for a, b in ev:
        a.record(stream); shard.spmv_partial(x, ys[0]); b.record(stream)
    torch.cuda.synchronize()
    kern_pair_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    # the same launch COLD: the timed steps run back to back on 2 x 203 MB of slot buffers + 8 MB of x, within reach of the 256 MB
    # Infinity Cache (FETCH_SIZE counts MALL hits as fetches); here every launch follows a 1 GiB device write
    cold_us = None
    if world == 1:
        try:
            cold_us = cold_launch_us(torch, dev, stream, lambda: shard.spmv_partial(x, ys[0]))
        except Exception:
            cold_us = None

    # algorithmic bytes of one SpMV launch: 16 B per streamed slot + x read + y written
    bytes_launch = 16 * cap + 8 * ncl + 8 * m
    useful_bytes = 16 * nnz + 8 * ncl + 8 * m
    slot_bytes = 16 if (os.environ.get("DSA_KEYS_WIDE") == "1" and os.environ.get("DSA_DEV") == "1") else 12      # int32 keys in HBM while every key fits Int32 (KeyArr)
    physical_bytes = slot_bytes * cap + cap // 8 + 8 * ncl + 8 * m            # what the kernel has to move at least: slots + bitmap + x + y
    value = world * bytes_launch / 1e9 / (ms_per_step / 1e3)
    achieved = bytes_launch / 1e9 / (kern_ms / 1e3)
    nomemset = A.info(dsa.ROWMAJOR)["stat_spmv_nomemset"] > 0

    workload = ("C3: PCSR %dx%d Float64, %d nnz per GPU, dense-x SpMV y=A*x (gather over the rowmajor twin)" % (m, n_total, nnz)) if cfg == "c3" else \
               ("C4: PCSR %dx%d Float64, %d nnz, column-range sharded over %d GPUs (%d columns and %d nnz per GPU), dense-x SpMV "
                "y=A*x + sum of the %d-entry partial y over the ranks" % (m, n_total, world * nnz, world, ncl, nnz, m))
    out = {
        "metric": "spmv_gbps_10M_nnz_pcsr",
        "value": round(value, 2),
        "unit": "GB/s",
        "n_gpus": world,
        "rccl_ranks": ranks_seen,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": workload, "name": cfg,
                   "capacity_slots": cap, "density": round((nnz + info_row["nb_partitions"]) / cap, 4), "sharding": "column-range x%d" % world,
                   "physical_slot_bytes": slot_bytes,
                   "collective": ("%s of y (%d f64 = %.0f MB) over %s%s" % (args.schedule, m, 8 * m / 1e6, ("RCCL / xGMI" + (" behind the C ABI (dsa_shard_allreduce_dev)" if abi_comm is not None else "")) if backend == "nccl" else backend,
                                  ", on RCCL's stream, overlapped with the next step's SpMV (two y buffers)" if overlap else "")) if world > 1 else "none"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernel": "dsa::k_spmv_gather" + (" (no memset of y: the kernel zero-fills rows without a partition)" if nomemset else " (+ memset of y)"),
                     "algorithmic_bytes": bytes_launch,
                     "kernel_ms": round(kern_ms, 5),
                     "kernel_ms_source": ("one HIP event pair around the %d launches of the timed region / %d" % (args.steps, args.steps)) if world == 1 else
                                         ("one HIP event pair around %d launches alone, right behind the timed region / %d" % (args.steps, args.steps)),
                     "kernel_ms_event_pair_per_launch": round(kern_pair_ms, 5),
                     "physical_bytes": physical_bytes,
                     "physical_gbps": round(physical_bytes / 1e9 / (kern_ms / 1e3), 2),
                     "physical_frac": round(physical_bytes / 1e9 / (kern_ms / 1e3) / HBM_PEAK_GBS, 4),
                     "useful_bytes_no_gaps": useful_bytes,
                     "useful_gbps": round(useful_bytes / 1e9 / (kern_ms / 1e3), 2),
                     "note": "achieved / frac divide SURVEY's ALGORITHMIC bytes (16-byte logical slots) by the measured time; physical_* "
                             "count the 12-byte slots + bitmap the kernel actually streams"},
        "nnz_per_s": round(world * nnz / (ms_per_step / 1e3), 1),
        "build_s": round(build_s, 3),
    }
    if collective_note:
        out["config"]["collective_note"] = collective_note
    if cold_us is not None:
        out["roofline"]["cold_kernel_ms"] = round(cold_us / 1e3, 5)
        out["roofline"]["cold_frac"] = round(bytes_launch / 1e3 / cold_us / HBM_PEAK_GBS, 4)
        out["roofline"]["cold_note"] = "8 launches, each behind a 1 GiB device write (caches and Infinity Cache evicted): one event pair around the 8 (write, launch) pairs minus one around 8 writes alone, / 8"

    # what the access pattern itself costs on this part (tools/gatherbench3.hip through tools/scripts/gather_floor3.sh, committed as
    # profiles/gather_floor.json): the slot stream alone, the 10 M gathers from an 8 MB x alone, both in one kernel, and the minimal
    # key-driven kernel (no rows, no semaphores, no y) — the floor a single-pass kernel on uniformly random columns has,
    # in the record next to the fraction it explains.  (Rounds 3-4 quoted gatherbench2's 11 M gathers from an 8.4 MB table: 10 % more
    # gathers than the product issues — semaphores do not read x.)
    if cfg == "c3" and m == 1_000_000 and ncl == 1_000_000 and per == 10:
        try:
            with open(os.path.join(ROOT, "profiles", "gather_floor.json")) as f:
                gf = json.load(f)
            out["roofline"]["stream_only_us"] = gf["stream_only_us"]
            out["roofline"]["gather_only_floor_us"] = gf["gather_only_us"]
            out["roofline"]["stream_and_gather_one_kernel_us"] = gf["stream_and_gather_one_kernel_us"]
            if "keyed_single_pass_us" in gf:
                out["roofline"]["keyed_single_pass_floor_us"] = gf["keyed_single_pass_us"]
                out["roofline"]["kernel_over_floor"] = round(kern_ms * 1e3 / gf["keyed_single_pass_us"], 3)
            floor_us = gf.get("keyed_single_pass_us", gf["gather_only_us"])
            out["roofline"]["frac_ceiling_single_pass"] = round(bytes_launch / 1e9 / (floor_us / 1e6) / HBM_PEAK_GBS, 4)
            out["roofline"]["floor_source"] = gf["source"]
        except (OSError, KeyError, ValueError):
            out["roofline"]["floor_source"] = "none: profiles/gather_floor.json missing"

    # HBM-side traffic per launch: only from a committed rocprofv3 PMC summary made from the CURRENT kernel sources
    # (tools/scripts/profile_round.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes of the same workload); otherwise null
    pmc_file, pm = committed_pmc()
    if pm is not None and cfg == "c3" and m == 1_000_000 and ncl == 1_000_000 and per == 10:
        out["roofline"]["traffic"] = pm.get("k_spmv_gather_C3", {}).get("corrected_traffic_total")
        out["roofline"]["traffic_source"] = "%s (rocprofv3 --pmc, corrected; kernel sources %s)" % (pmc_file, pm.get("kernel_source_sha"))
    elif cfg == "c3" and m == 1_000_000 and ncl == 1_000_000 and per == 10:
        out["roofline"]["traffic_source"] = "none: no committed PMC summary matches the current kernel sources (%s)" % kernel_source_sha()
    else:
        out["roofline"]["traffic_source"] = "none: the committed PMC passes profile the C3 workload only"

    if world > 1:
        # the local product alone and the schedule of the sum of y used above (all three with --all-schedules), timed back to
        # back without overlap (max over ranks): what the collective costs next to the SpMV it follows
        sched = {}
        reps = max(5, min(args.steps, 20))

        def timed(fn):
            barrier()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return round(max_over_ranks(time.perf_counter() - t) / reps * 1e3, 4)

        sched["local_spmv_ms"] = timed(lambda: shard.spmv_partial(x, ys[0]))
        names = sharding.SCHEDULES if (args.all_schedules or os.environ.get("DSA_BENCH_ALL_SCHEDULES") == "1") else (args.schedule,)
        for name in names:
            try:
                sched[name + "_ms"] = timed(lambda: shard.reduce(ys[0], name))
                sched[name + "_algbw_gbps"] = round(8 * m / 1e9 / (sched[name + "_ms"] / 1e3), 1)
            except Exception as e:
                sched[name + "_ms"] = "failed: %s" % str(e)[:120]
        out["collective_schedules"] = sched
        out["end_to_end_ms"] = {"local_spmv": sched["local_spmv_ms"], "step_overlapped": round(ms_per_step, 5)}
    if args.dump_y:
        ydump = {}
        for name in sharding.SCHEDULES:
            yy = shard.new_y()
            shard.spmv_partial(x, yy)
            shard.reduce(yy, name)
            torch.cuda.synchronize()
            ydump[name] = yy.cpu().numpy()
        if abi_comm is not None:
            yy = shard.new_y()
            shard.spmv_partial(x, yy)
            shard.reduce(yy, "all_reduce")
            torch.cuda.synchronize()
            ydump["abi_all_reduce"] = yy.cpu().numpy()
        if rank == 0:
            np.savez(args.dump_y, **ydump)
    if rank == 0 and world == 1 and not args.no_extras:          # the extra legs and the CPU baseline belong to the N = 1 line
        try:
            out.update(extras(dsa, hip, torch, A, dev))
        except Exception as e:               # an extra leg must never cost the headline line
            out["extras_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
        if pm is not None and "roofline_rebalance" in out:
            out["roofline_rebalance"]["traffic"] = pm.get("k_move_root_2^24", {}).get("corrected_traffic_total") \
                if out["roofline_rebalance"]["window_slots"] == 16777216 else None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(dsa, m, per)
        except Exception as e:
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def extras(dsa, hip, torch, A, dev):
    """rebalance roofline (full 2^24-slot window of the 10M-nnz PCSR) and inserts/s (config C2)."""
    res = {}
    stream = torch.cuda.current_stream()
    # --- full-window pack+spread of the colmajor PCSR array (incl. semaphore scatter)
    cap = A.info(dsa.COLMAJOR)["capacity"]
    for _ in range(3):
        A.rebalance_root(dsa.COLMAJOR)
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        A.rebalance_root(dsa.COLMAJOR)
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    b = 32 * cap
    res["roofline_rebalance"] = {"bound": "hbm", "achieved": round(b / 1e9 / (ms / 1e3), 2), "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": round(b / 1e9 / (ms / 1e3) / HBM_PEAK_GBS, 4),
                                 "window_slots": cap, "algorithmic_bytes": b, "ms": round(ms, 5),
                                 "kernels": "k_move2<false, WIDE=false> (one launch)",
                                 "physical_bytes": 2 * (12 * cap + cap // 8),
                                 "physical_frac": round(2 * (12 * cap + cap // 8) / 1e9 / (ms / 1e3) / HBM_PEAK_GBS, 4)}
    try:
        cus = cold_launch_us(torch, dev, stream, lambda: A.rebalance_root(dsa.COLMAJOR))
        res["roofline_rebalance"]["cold_ms"] = round(cus / 1e3, 5)
        res["roofline_rebalance"]["cold_frac"] = round(b / 1e3 / cus / HBM_PEAK_GBS, 4)
        res["roofline_rebalance"]["cold_physical_frac"] = round(2 * (12 * cap + cap // 8) / 1e3 / cus / HBM_PEAK_GBS, 4)
        res["roofline_rebalance"]["cold_note"] = "8 launches, each behind a 1 GiB device write: one event pair around the 8 (write, launch) pairs minus one around 8 writes alone, / 8"
    except Exception as e:
        res["roofline_rebalance"]["cold_error"] = str(e)[:120]
    # --- the isolated rebalance on full windows of 2^20 / 2^21 / 2^24 slots at densities 0.35 / 0.70:
    #     a vector's PMA built from n = density * capacity keys, root pack + spread timed back to back
    sweep = []

    def timed(fn, nrep):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record(stream)
        for _ in range(nrep):
            fn()
        r1.record(stream)
        torch.cuda.synchronize()
        return r0.elapsed_time(r1) / nrep * 1e3

    for lg in (20, 21, 24):
        for dens in (0.35, 0.70):
            capv = 1 << lg
            n = int(dens * capv) + (2 if dens < 0.5 else 0)          # n / 0.7 just above capv / 2 keeps the capacity rule at capv
            n = min(n, int(0.7 * capv))
            vv = dsa.dynamicsparsevec(np.arange(1, n + 1, dtype=np.int64) * 3, unit12(40 + lg, n), binding=hip)
            if vv.info()["capacity"] != capv:
                continue
            hip.call("vec_set_stream", vv.h, C.c_void_p(stream.cuda_stream))
            nrep = 50 if lg < 24 else 20
            us = timed(vv.rebalance_root, nrep)
            phys = 2 * (12 * capv + capv // 8)                         # int32 key + f64 value + 1 occupancy bit, read + written
            row = {"window_slots": capv, "density": round(n / capv, 3), "source": "uniform (already spread)", "us": round(us, 2),
                   "gbps": round(32 * capv / us / 1e3, 1), "frac": round(32 * capv / us / 1e3 / HBM_PEAK_GBS, 4),
                   "physical_frac": round(phys / us / 1e3 / HBM_PEAK_GBS, 4)}
            # skewed sources (dsa_vec_dev_relayout): time(relayout + rebalance) - time(relayout)
            for mode, label in ((1, "packed_left_us"), (2, "gaps_left_us")):
                def both():
                    hip.call("vec_dev_relayout", vv.h, mode)
                    vv.rebalance_root()
                t_both = timed(both, max(nrep // 2, 5))
                t_one = timed(lambda: hip.call("vec_dev_relayout", vv.h, mode), max(nrep // 2, 5))
                row[label] = round(t_both - t_one, 2)
                vv.rebalance_root()
            if dens > 0.5:
                # _extend! (read W, write 2W: 48 B per source slot) and _shrink! back (read 2W, write W), timed as a pair
                def ext_shr():
                    hip.call("vec_dev_relayout", vv.h, 3)
                    hip.call("vec_dev_relayout", vv.h, 4)
                us_pair = timed(ext_shr, max(nrep // 4, 5))
                row["extend_plus_shrink_us"] = round(us_pair, 2)
                row["extend_plus_shrink_frac"] = round(2 * 48 * capv / us_pair / 1e3 / HBM_PEAK_GBS, 4)
            sweep.append(row)
            del vv
    res["rebalance_sweep"] = sweep
    # the 2^24 window from the two sources side by side (the quoted case is the benign one — cells already spread; a
    # source with every gap at the left is what dsa_vec_dev_relayout(2) builds: only the tiles that hold cells write)
    for row in sweep:
        if row["window_slots"] == 1 << 24 and row["density"] > 0.5 and "gaps_left_us" in row:
            gl = row["gaps_left_us"]
            res["roofline_rebalance"]["skewed_sources_2^24"] = {
                "even_us": row["us"], "gaps_left_us": gl, "packed_left_us": row.get("packed_left_us"),
                "gaps_left_frac": round(32 * (1 << 24) / gl / 1e3 / HBM_PEAK_GBS, 4),
                "gaps_left_physical_frac": round(2 * (12 * (1 << 24) + (1 << 21)) / gl / 1e3 / HBM_PEAK_GBS, 4),
                "note": "a vector's 2^24-slot root window at density 0.70: cells already spread / all cells in the last n slots / in the first n slots"}
    # --- wide keys: ONE key outside Int32 widens the whole slot array to 16-byte slots (KeyArr) — the slot size SURVEY's byte formula
    #     assumes.  The 2^24 rebalance and the C3 product on such structures: algorithmic == physical bytes here (+ the bitmap).
    try:
        capw = 1 << 24
        nw = int(0.7 * capw)
        kw = np.arange(1, nw + 1, dtype=np.int64) * 3
        kw[-1] = np.int64(1) << 40
        vw = dsa.dynamicsparsevec(kw, unit12(44, nw), binding=hip)
        assert vw.info()["capacity"] == capw
        hip.call("vec_set_stream", vw.h, C.c_void_p(stream.cuda_stream))
        usw = timed(vw.rebalance_root, 20)
        physw = 2 * (16 * capw + capw // 8)
        res["wide_keys"] = {"rebalance_2^24": {"us": round(usw, 2), "frac": round(32 * capw / usw / 1e3 / HBM_PEAK_GBS, 4),
                                               "physical_bytes": physw, "physical_frac": round(physw / usw / 1e3 / HBM_PEAK_GBS, 4),
                                               "kernel": "k_move2<false, WIDE=true>"}}
        del vw, kw
    except Exception as e:
        res["wide_keys"] = {"rebalance_2^24": {"error": str(e)[:200]}}
    # --- one C4 shard (BASELINE config 4 per GPU: 10M rows, 1.25M columns, 12.5M nnz, capacity 2^25, y = 10M doubles)
    try:
        m4, n4 = 10_000_000, 1_250_000
        I4, J4, V4 = c3_triplets(m4, n4, 10, 0, seed_rows=8, seed_vals=9)
        A4 = dsa.dynamicsparse(I4, J4, V4, m4, n4, binding=hip)
        hip.call("mat_set_stream", A4.h, C.c_void_p(stream.cuda_stream))
        cap4 = A4.info(dsa.ROWMAJOR)["capacity"]
        x4 = torch.from_numpy(unit12(10, n4)).to(dev)
        y4 = torch.zeros(m4, dtype=torch.float64, device=dev)
        for _ in range(3):
            hip.call("mat_spmv_dense_dev", A4.h, 0, 0, C.c_void_p(x4.data_ptr()), n4, C.c_void_p(y4.data_ptr()), m4)
        torch.cuda.synchronize()
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record(stream)
        for _ in range(20):
            hip.call("mat_spmv_dense_dev", A4.h, 0, 0, C.c_void_p(x4.data_ptr()), n4, C.c_void_p(y4.data_ptr()), m4)
        r1.record(stream)
        torch.cuda.synchronize()
        ms4 = r0.elapsed_time(r1) / 20
        b4 = 16 * cap4 + 8 * n4 + 8 * m4
        res["c4_shard_spmv"] = {"rows": m4, "columns": n4, "nnz": len(I4), "capacity_slots": cap4, "ms": round(ms4, 5),
                                "algorithmic_bytes": b4, "gbps": round(b4 / 1e9 / (ms4 / 1e3), 1),
                                "frac": round(b4 / 1e9 / (ms4 / 1e3) / HBM_PEAK_GBS, 4),
                                "note": "local SpMV of one of the 8 column-range shards of BASELINE config 4 (the 80 MB all-reduce of y is not included)"}
        try:      # the measured cost of this shard's slot stream + x gathers alone (tools/scripts/gather_floor3.sh): no 80 MB y, no row keys
            with open(os.path.join(ROOT, "profiles", "gather_floor_c4shard.json")) as f:
                g4 = json.load(f)
            res["c4_shard_spmv"]["stream_and_gather_floor_us"] = g4["keyed_single_pass_us"]
            res["c4_shard_spmv"]["frac_ceiling_stream_and_gathers_only"] = round(b4 / 1e9 / (g4["keyed_single_pass_us"] / 1e6) / HBM_PEAK_GBS, 4)
            res["c4_shard_spmv"]["floor_source"] = g4["source"]
        except (OSError, KeyError, ValueError):
            pass
        del A4, x4, y4
    except Exception as e:           # an extra must never cost the headline line
        res["c4_shard_spmv"] = {"error": str(e)[:200]}
    # --- the C3 product on 16-byte slots: the same matrix plus ONE cell whose column key lies outside Int32 (the twin orientation the
    #     gather kernel walks keeps column keys: it is widened; the cell itself is beyond x and contributes nothing)
    try:
        m3w = n3w = 1_000_000
        Iw, Jw, Vw = c3_triplets(m3w, n3w, 10, 0, seed_rows=5, seed_vals=6)
        Iw = np.concatenate([Iw, [1]]); Jw = np.concatenate([Jw, [np.int64(1) << 33]]); Vw = np.concatenate([Vw, [1.5]])
        Aw = dsa.dynamicsparse(Iw, Jw, Vw, m3w, n3w, binding=hip)
        hip.call("mat_set_stream", Aw.h, C.c_void_p(stream.cuda_stream))
        capw3 = Aw.info(dsa.ROWMAJOR)["capacity"]
        xw = torch.from_numpy(unit12(7, n3w)).to(dev)
        yw = torch.zeros(m3w, dtype=torch.float64, device=dev)
        fw = lambda: hip.call("mat_spmv_dense_dev", Aw.h, 0, 0, C.c_void_p(xw.data_ptr()), n3w, C.c_void_p(yw.data_ptr()), m3w)
        for _ in range(3):
            fw()
        torch.cuda.synchronize()
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record(stream)
        for _ in range(20):
            fw()
        r1.record(stream)
        torch.cuda.synchronize()
        usw3 = r0.elapsed_time(r1) / 20 * 1e3
        bw3 = 16 * capw3 + 8 * n3w + 8 * m3w
        res.setdefault("wide_keys", {})["spmv_c3"] = {"us": round(usw3, 2), "capacity_slots": capw3, "algorithmic_bytes": bw3,
                                                      "frac": round(bw3 / usw3 / 1e3 / HBM_PEAK_GBS, 4),
                                                      "physical_bytes": bw3 + capw3 // 8,
                                                      "physical_frac": round((bw3 + capw3 // 8) / usw3 / 1e3 / HBM_PEAK_GBS, 4),
                                                      "kernel": "k_spmv_gather<WIDE=true, ...>",
                                                      "note": "config 3 + one cell with a column key of 2^33: 16-byte slots in HBM, algorithmic == physical bytes"}
        del Aw, xw, yw, Iw, Jw, Vw
    except Exception as e:
        res.setdefault("wide_keys", {})["spmv_c3"] = {"error": str(e)[:200]}
    # --- an EXTRA, never the headline: the same kernel on a matrix WITH column locality — config 3's shape (1M x 1M, 10 distinct rows
    #     per column) but rows within +-4096 of the column index (banded).  The x entries a stretch of rows gathers then sit in a few
    #     hundred KB: every gather is an L2 hit, the kernel leaves the fabric-bound regime of the uniformly random matrix
    try:
        mb = nb = 1_000_000
        z = splitmix_array(51, nb * 10)
        colb = np.repeat(np.arange(1, nb + 1, dtype=np.int64), 10)
        rowb = np.clip(colb + (z % np.uint64(8192)).astype(np.int64) - 4096, 1, mb)
        keyb = colb * np.int64(mb + 1) + rowb
        _, firstb = np.unique(keyb, return_index=True)
        Ab = dsa.dynamicsparse(rowb[firstb], colb[firstb], unit12(52, len(firstb)), mb, nb, binding=hip)
        hip.call("mat_set_stream", Ab.h, C.c_void_p(stream.cuda_stream))
        capb = Ab.info(dsa.ROWMAJOR)["capacity"]
        xb = torch.from_numpy(unit12(53, nb)).to(dev)
        yb = torch.zeros(mb, dtype=torch.float64, device=dev)
        usb = timed(lambda: hip.call("mat_spmv_dense_dev", Ab.h, 0, 0, C.c_void_p(xb.data_ptr()), nb, C.c_void_p(yb.data_ptr()), mb), 20)
        bb = 16 * capb + 8 * nb + 8 * mb
        res["spmv_banded_extra"] = {"rows": mb, "columns": nb, "nnz": int(len(firstb)), "band": "+-4096", "capacity_slots": capb, "us": round(usb, 2),
                                    "algorithmic_bytes": bb, "gbps": round(bb / usb / 1e3, 1), "frac": round(bb / usb / 1e3 / HBM_PEAK_GBS, 4),
                                    **banded_floor(usb),
                                    "note": "NOT the headline workload: a banded variant of config 3, reported to show what the kernel does when x has locality"}
        del Ab, xb, yb
    except Exception as e:
        res["spmv_banded_extra"] = {"error": str(e)[:200]}
    # --- sparse x through the host-pointer entry point (the _mul shape Coluna calls, src/operations.jl:107-135): result = touched
    #     rows in ascending order; few stored entries -> x-driven kernel, many -> densify + gather + pattern pass.  Times include
    #     the PCIe copies of x and of the result.
    sp = []
    m3, n3 = A.size()
    for nxs in (100, 10_000, 500_000):
        xi = np.unique(1 + (splitmix_array(50 + nxs, nxs) % np.uint64(n3)).astype(np.int64))
        xv = unit12(51, len(xi))
        A.mul((xi, xv))
        ts = []
        for _ in range(7):                 # median of per-call times: a one-off host pause (a Python GC pass of ~3 ms landed in this loop) is not a product
            t = time.perf_counter()
            yi, yv = A.mul((xi, xv))
            ts.append(time.perf_counter() - t)
        row = {"stored_x_entries": int(len(xi)), "touched_rows": int(len(yi)), "ms": round(float(np.median(ts)) * 1e3, 3),
               "ms_max": round(max(ts) * 1e3, 3)}
        try:      # every operand in HBM (dsa_mat_spmv_sparse_dev: xi / xv in, yi / yv / count out): one event pair around 10 enqueued products
            d_xi = torch.from_numpy(xi).to(dev); d_xv = torch.from_numpy(xv).to(dev)
            d_yi = torch.empty(m3, dtype=torch.int64, device=dev); d_yv = torch.empty(m3, dtype=torch.float64, device=dev)
            d_c = torch.zeros(1, dtype=torch.int64, device=dev)
            f = lambda: A.mul_dev(d_xi.data_ptr(), d_xv.data_ptr(), len(xi), d_yi.data_ptr(), d_yv.data_ptr(), m3, d_c.data_ptr())
            for _ in range(3):
                f()
            q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            q0.record(stream)
            for _ in range(10):
                f()
            q1.record(stream)
            torch.cuda.synchronize()
            row["dev_us"] = round(q0.elapsed_time(q1) * 100, 1)
            assert int(d_c.item()) == len(yi)
            del d_xi, d_xv, d_yi, d_yv, d_c
        except Exception as e:
            row["dev_error"] = str(e)[:120]
        sp.append(row)
    res["spmv_sparse_x"] = sp
    # --- m[:, j] as a new device vector: built device to device; creation per call with the handles kept alive
    try:
        rngs = np.random.default_rng(5)
        ms_, ns_ = 200000, 1000
        sizes = {1: 16, 2: 1000, 3: 16000}
        Is = np.concatenate([rngs.choice(ms_, c, replace=False) + 1 for c in sizes.values()] + [rngs.integers(1, ms_ + 1, 50000)])
        Js = np.concatenate([np.full(c, j) for j, c in sizes.items()] + [rngs.integers(4, ns_ + 1, 50000)])
        As = dsa.dynamicsparse(Is, Js, rngs.random(len(Is)) + 1.0, ms_, ns_, binding=hip)
        sl = {}
        for j, c in sizes.items():
            for _ in range(10):
                As.col_slice(j)
            t = time.perf_counter()
            for _ in range(200):
                v_ = As.col_slice(j)          # (the previous slice is destroyed here: its blocks go back to the library's pool and are found again)
            sl["%d_entries_us" % c] = round((time.perf_counter() - t) / 200 * 1e6, 1)
            del v_
        sl["note"] = "m[:, j] -> a new device vector, create + destroy per call through the Python mirror (ctypes + GC included); no cell crosses PCIe"
        res["col_slice"] = sl
        del As
    except Exception as e:
        res["col_slice"] = {"error": str(e)[:200]}
    # --- C2: 2^20-slot PMA, 100k ascending appends (batch A) and 100k uniform odd keys (batch B)
    n0 = 700000
    keys0 = np.arange(1, n0 + 1, dtype=np.int64) * 2
    # (a throwaway vector takes the one-time costs of the write paths first — code objects, the first graph instantiation, pinned
    #  staging: a long-running host pays them once, and the matrix leg below is warmed the same way)
    w = dsa.dynamicsparsevec(keys0[:20000], unit12(3, 20000), binding=hip)
    w.set_batch(np.arange(40001, 45001, dtype=np.int64), unit12(3, 5000))
    w.set_batch(1 + 2 * (splitmix_array(5, 5000) % np.uint64(20000)).astype(np.int64), unit12(4, 5000))
    del w
    app = np.arange(1400001, 1500001, dtype=np.int64)
    odd = np.unique(1 + 2 * (splitmix_array(4, 120000) % np.uint64(700000)).astype(np.int64))[:100000]
    np.random.default_rng(4).shuffle(odd)
    vals0, valsA, valsB = unit12(3, n0), unit12(3, 100000), unit12(4, len(odd))
    tas, tbs, info = [], [], None
    for _ in range(5):                   # median of >= 5 runs after warm-up, every run on a freshly built PMA
        v = dsa.dynamicsparsevec(keys0, vals0, binding=hip)
        t = time.perf_counter()
        v.set_batch(app, valsA)
        tas.append(time.perf_counter() - t)
        info = v.info()
        t = time.perf_counter()
        v.set_batch(odd, valsB)
        tbs.append(time.perf_counter() - t)
        del v
    ta, tb = float(np.median(tas)), float(np.median(tbs))
    res["inserts_per_s"] = {"batch_A_ascending_appends": round(100000 / ta, 1), "batch_B_uniform": round(len(odd) / tb, 1),
                            "batch_A_ms_runs": [round(x * 1e3, 2) for x in tas], "batch_B_ms_runs": [round(x * 1e3, 2) for x in tbs],
                            "config": "C2: 2^20-slot PMA (700k keys) + 100k batched inserts, whole dsa_vec_set_batch call incl. H2D; median of 5 runs, each on a freshly built PMA, after a warm-up on a throwaway vector",
                            "window_slots_per_insert_A": round(info["stat_window_slots"] / 100000, 1),
                            "extends": info["stat_extends"]}
    # --- random A[i,j] = v updates on an existing 20k x 30k structure (each write = 2 PCSR writes, batch-parallel path)
    mm, nn = 20000, 30000
    ri = 1 + (splitmix_array(31, 600000) % np.uint64(mm)).astype(np.int64)
    ci = 1 + (splitmix_array(32, 600000) % np.uint64(nn)).astype(np.int64)
    ui = 1 + (splitmix_array(34, 200000) % np.uint64(mm)).astype(np.int64)
    uj = 1 + (splitmix_array(35, 200000) % np.uint64(nn)).astype(np.int64)
    uv = np.where(splitmix_array(36, 200000) % np.uint64(4) == 0, 0.0, unit12(37, 200000))
    tms = []
    for _ in range(5):
        M = dsa.dynamicsparse(ri, ci, unit12(33, 600000), mm, nn, binding=hip)
        M.set_batch(ui[:128], uj[:128], uv[:128])
        t = time.perf_counter()
        M.set_batch(ui, uj, uv)
        tms.append(time.perf_counter() - t)
        del M
    res["inserts_per_s"]["matrix_random_updates_per_s"] = round(len(ui) / float(np.median(tms)), 1)
    res["inserts_per_s"]["matrix_random_updates_ms_runs"] = [round(x * 1e3, 2) for x in tms]
    res["inserts_per_s"]["matrix_random_updates_config"] = "200k random A[i,j]=v (25% deletes) on a 20k x 30k matrix with 600k nnz; median of 5 runs on fresh matrices"

    # --- C5 (BASELINE config 5, full size): stream 50k new columns (16 distinct rows each, ascending inside a column, 100k rows)
    #     element by element into an empty matrix — both orientations —, SpMV every 1000 columns.  Parity of this loop vs the
    #     oracle: tests/test_hip_parity.py::test_matrix_from_empty_streaming_columns_c5_scaled
    def median_of(runs, key):
        runs = sorted(runs, key=lambda r: r[key])
        mid = dict(runs[len(runs) // 2])
        mid["runs_" + key] = [r[key] for r in runs]
        mid["protocol"] = "median of %d full runs (each on a fresh empty matrix) after one warm-up run" % len(runs)
        return mid

    c5_streaming(dsa, hip, torch, dev, *C5_FULL, stop_after=5000)            # warm-up: code objects, graphs, pools of this process
    res["c5_streaming"] = median_of([c5_streaming(dsa, hip, torch, dev, *C5_FULL) for _ in range(3)], "write_s")
    # --- the same stream as 800 000 single dsa_mat_set calls ("written element by element"): a C client of include/dsa.h
    #     (tools/percall/percall_bench.c, built by __graft_entry__.build) in a child process — one ccall-shaped call per element through
    #     the library's write-combining queue, the matrix observed after every 1000 columns; median of 3 repeats after a warm-up repeat
    try:
        import subprocess
        exe = os.path.join(ROOT, "tools", "percall", "percall_bench")
        r = subprocess.run([exe, str(C5_FULL[0]), str(C5_FULL[1]), str(C5_FULL[2]), str(C5_FULL[3]), "3"], capture_output=True, text=True, timeout=300)
        res["c5_per_call"] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": (r.stderr or r.stdout)[-300:]}
        res["c5_per_call"]["note"] = ("config 5 as single dsa_mat_set(A, v, i, j) calls from a plain-C client (what a Julia `A[i, j] = v` ccall costs): the "
                                      "library queues each write on the host and applies the queue in order before the next observing call")
    except Exception as e:
        res["c5_per_call"] = {"error": str(e)[:200]}
    # --- an EXTRA: the same stream as column generation does it — after every batch 5 % of its columns are deleted again
    #     (deletecolumn!, src/matrix.jl:95-102: tombstones in the colmajor tables).  A batch of new, larger column ids cannot fail, so the two
    #     orientations keep running side by side (dsa_host.hip: mat_apply_sets); parity: test_column_generation_with_deletions_matches_oracle
    try:
        res["c5_streaming_with_deletions"] = median_of([c5_streaming(dsa, hip, torch, dev, *C5_FULL, delete_every=20) for _ in range(3)], "write_s")
    except Exception as e:
        res["c5_streaming_with_deletions"] = {"error": str(e)[:200]}
    # --- buffered writes: the C3 triples through the fill buffer in ten batches of 1 M, then
    #     closefillmode! (the flush = two bulk builds); and dynamicsparse(I, J, V) from caller memory.  Second of two passes (the
    #     first one pays the pinned staging chunks, which are kept).
    I3, J3, V3 = c3_triplets(1_000_000, 1_000_000, 10, 0, seed_rows=5, seed_vals=6)
    fill = {}
    for _ in range(2):
        F = dsa.dynamicsparse(fill_mode=True, binding=hip)
        t = time.perf_counter()
        for c in range(0, len(I3), 1_000_000):
            F.set_batch(I3[c:c + 1_000_000], J3[c:c + 1_000_000], V3[c:c + 1_000_000])
        t1 = time.perf_counter()
        F.closefillmode()
        t2 = time.perf_counter()
        fill = {"triples": int(len(I3)), "appends_ms": round((t1 - t) * 1e3, 2), "closefillmode_ms": round((t2 - t1) * 1e3, 2), "nnz": int(F.nnz())}
        del F
    t = time.perf_counter()
    F = dsa.dynamicsparse(I3, J3, V3, 1_000_000, 1_000_000, binding=hip)
    fill["dynamicsparse_from_caller_memory_ms"] = round((time.perf_counter() - t) * 1e3, 2)
    # the flush against its stream bound: 24 B per triple in, both orientations' slot arrays out (16 B per slot of capacity), once each
    capF = F.info(dsa.COLMAJOR)["capacity"] + F.info(dsa.ROWMAJOR)["capacity"]
    fill["closefillmode_stream_bound_ms"] = round((24 * len(I3) + 16 * capF) / (HBM_PEAK_GBS * 1e9) * 1e3, 4)
    fill["closefillmode_frac_of_stream_bound"] = round(fill["closefillmode_stream_bound_ms"] / fill["closefillmode_ms"], 4)
    fill["kbuild"] = ("hand-written LSD radix sort of (partition, key) composite << 24 | input index: 5 passes of 8 bits over 8-byte words at this size, "
                      "values gathered by index at the emit (csrc/build.hip)")
    del F
    res["buffered_writes"] = fill
    # --- inserts ON the 10 M-nnz PCSR itself (what BASELINE's metric names): 100 k uniformly random A[i, j] = v, then 10 k NEW columns
    #     of 10 rows appended — every element write updates both 2^24-slot orientations.  Last leg: it changes the headline matrix.
    #     Parity of exactly these writes at full size: tests/test_hip_parity.py::test_inserts_on_the_c3_matrix_match_oracle
    try:
        m3, n3 = A.size()
        (ri, rj, rv), (ai, aj, av) = c3_insert_leg(m3, n3)

        def stats():
            return {o: A.info(o) for o in (dsa.COLMAJOR, dsa.ROWMAJOR)}

        def delta(a, b_):
            return {name: {k: b_[o][k] - a[o][k] for k in ("stat_window_slots", "stat_rebalances", "stat_grid_rebalances", "stat_extends", "nb_elements")}
                    for o, name in ((dsa.COLMAJOR, "colmajor"), (dsa.ROWMAJOR, "rowmajor"))}
        # median of 3 passes, each on a freshly built copy of the C3 matrix (the leg changes the matrix it runs on), after one warm-up
        # pass that takes the first-launch costs of this process (the first append run / count-only model launch on a new hardware
        # queue costs 1.4-2.3 ms on this runtime: round 4's single shot had it inside the 'appended_columns' time)
        passes = []
        Aw = A
        for rep in range(4):
            Aw.set_batch(ri[:256], rj[:256], rv[:256])              # (first batch on this handle: graph capture, op buffers)
            s0 = {o: Aw.info(o) for o in (dsa.COLMAJOR, dsa.ROWMAJOR)}
            t = time.perf_counter(); Aw.set_batch(ri[256:], rj[256:], rv[256:]); t_rand = time.perf_counter() - t
            s1 = {o: Aw.info(o) for o in (dsa.COLMAJOR, dsa.ROWMAJOR)}
            t = time.perf_counter(); Aw.set_batch(ai, aj, av); t_app = time.perf_counter() - t
            s2 = {o: Aw.info(o) for o in (dsa.COLMAJOR, dsa.ROWMAJOR)}
            if rep > 0:
                passes.append((t_rand, t_app, s0, s1, s2))
            if rep < 3:
                if Aw is not A:
                    del Aw
                Aw = dsa.dynamicsparse(I3, J3, V3, 1_000_000, 1_000_000, binding=hip)
        if Aw is not A:
            del Aw
        rand_runs = [p[0] for p in passes]; app_runs = [p[1] for p in passes]
        t_rand = float(np.median(rand_runs)); t_app = float(np.median(app_runs))
        s0, s1, s2 = passes[0][2], passes[0][3], passes[0][4]
        res["inserts_on_c3"] = {
            "matrix": "%d x %d, %d nnz before the leg, capacity %d slots per orientation" % (m3, n3, s0[dsa.COLMAJOR]["nb_elements"] - s0[dsa.COLMAJOR]["nb_partitions"], s0[dsa.COLMAJOR]["capacity"]),
            "random_writes": {"element_writes": int(len(ri) - 256), "ms": round(t_rand * 1e3, 3), "element_writes_per_s": round((len(ri) - 256) / t_rand, 1),
                              "pcsr_inserts_per_s": round(2 * (len(ri) - 256) / t_rand, 1), "stats": delta(s0, s1)},
            "appended_columns": {"columns": 10_000, "element_writes": int(len(ai)), "ms": round(t_app * 1e3, 3), "element_writes_per_s": round(len(ai) / t_app, 1),
                                 "pcsr_inserts_per_s": round(2 * len(ai) / t_app, 1), "stats": delta(s1, s2)},
            "random_writes_ms_runs": [round(x * 1e3, 3) for x in rand_runs], "appended_columns_ms_runs": [round(x * 1e3, 3) for x in app_runs],
            "protocol": "medians of 3 passes, each on a freshly built copy of the C3 matrix, after one warm-up pass",
            "note": "whole dsa_mat_set_batch calls incl. H2D; stat_grid_rebalances = launches of the grid-wide pack/spread kernel (windows above 8192 "
                    "slots); the appended columns are one append run per batch in the colmajor orientation (bitmap replay + one K-permute of the "
                    "array, no per-window launches) and random inserts in the rowmajor twin"}
    except Exception as e:
        res["inserts_on_c3"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    return res


C5_FULL = (100_000, 50_000, 16, 1000)


def c5_columns(m5, ncols5, per5):
    """(I, J, V) of the C5 stream: column j = 1..ncols5 in order, its rows ascending (seeds 11 / 12)."""
    I, J, V = c3_triplets(m5, ncols5, per5, 0, seed_rows=11, seed_vals=12)
    order = np.lexsort((I, J))
    return I[order], J[order], V


def c5_streaming(dsa, hip, torch, dev, m5, ncols5, per5, every, binding=None, stop_after=None, delete_every=None):
    B = dsa.dynamicsparse(fill_mode=False, binding=binding or hip)     # keeps its two own streams: the orientations update concurrently
    I5, J5, V5 = c5_columns(m5, ncols5, per5)
    x5 = unit12(13, ncols5)
    if binding is None:
        xd = torch.from_numpy(x5).to(dev)
        yd = torch.zeros(m5, dtype=torch.float64, device=dev)
    t_w, t_s, nsp, t_first = 0.0, 0.0, 0, None
    t_del, n_del = 0.0, 0
    for c0 in range(0, ncols5, every):
        sl = slice(c0 * per5, (c0 + every) * per5)
        t = time.perf_counter()
        B.set_batch(I5[sl], J5[sl], V5[sl])
        if binding is None:
            hip.call("mat_sync", B.h)      # the batch's device work (table merge, meta prefetch) belongs to the write, not to the product behind it
        t_w += time.perf_counter() - t
        t = time.perf_counter()
        if binding is None:
            hip.call("mat_spmv_dense_dev", B.h, 0, 0, C.c_void_p(xd.data_ptr()), c0 + every, C.c_void_p(yd.data_ptr()), m5)
            torch.cuda.synchronize()
        else:
            B.mul(x5[: c0 + every])
        t_s += time.perf_counter() - t
        nsp += 1
        if delete_every:
            t = time.perf_counter()
            for j in range(c0 + 1, c0 + every + 1, delete_every):
                B.deletecolumn(j); n_del += 1
            t_del += time.perf_counter() - t
        if c0 + every == 10_000:
            t_first = t_w
        if stop_after is not None and c0 + every >= stop_after:
            break
    ncols_done = min(ncols5, (nsp * every))
    nw = ncols_done * per5
    final_spmv = None
    if binding is None and not delete_every and stop_after is None:
        # the product on the FINAL matrix (x = 50 k doubles = 400 KB: L2-resident on every XCD — the regime column generation runs in),
        # 20 launches between one pair of HIP events on the matrix's own stream order (torch's current stream: set for the product)
        st = torch.cuda.current_stream()
        hip.call("mat_set_stream", B.h, C.c_void_p(st.cuda_stream))
        for _ in range(3):
            hip.call("mat_spmv_dense_dev", B.h, 0, 0, C.c_void_p(xd.data_ptr()), ncols_done, C.c_void_p(yd.data_ptr()), m5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20):
            hip.call("mat_spmv_dense_dev", B.h, 0, 0, C.c_void_p(xd.data_ptr()), ncols_done, C.c_void_p(yd.data_ptr()), m5)
        e1.record(st)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        capr = B.info(dsa.ROWMAJOR)["capacity"]
        bb = 16 * capr + 8 * ncols_done + 8 * m5
        final_spmv = {"us": round(us, 2), "capacity_slots": capr, "algorithmic_bytes": bb, "gbps": round(bb / us / 1e3, 1),
                      "frac": round(bb / us / 1e3 / HBM_PEAK_GBS, 4)}
        try:
            with open(os.path.join(ROOT, "profiles", "gather_floor_c5.json")) as f:
                g5 = json.load(f)
            final_spmv["keyed_single_pass_floor_us"] = g5["keyed_single_pass_us"]
            final_spmv["kernel_over_floor"] = round(us / g5["keyed_single_pass_us"], 3)
            final_spmv["floor_source"] = g5["source"]
        except (OSError, KeyError, ValueError):
            pass
    if delete_every:
        return {"columns": ncols_done, "columns_per_s": round(ncols_done / t_w, 1), "write_s": round(t_w, 3), "deleted_columns": n_del,
                "deletecolumn_us_each": round(t_del / max(n_del, 1) * 1e6, 1), "spmv_ms_avg": round(t_s / nsp * 1e3, 4),
                "note": "config 5 with 1 of %d streamed columns deleted again after every batch of %d" % (delete_every, every)}
    return {"columns": ncols_done, "rows": m5, "element_writes": nw, "columns_per_s": round(ncols_done / t_w, 1),
            "element_writes_per_s": round(nw / t_w, 1), "write_s": round(t_w, 3), "first_10k_columns_write_s": None if t_first is None else round(t_first, 3),
            "spmv_ms_avg": round(t_s / nsp * 1e3, 4), "spmv_every_columns": every, "final_spmv": final_spmv,
            "note": "each element write updates both orientations (2 PCSR inserts); new rows arrive in random key order "
                    "(middle inserts of addpartition!, src/pcsr.jl:114-146)"}


def cpu_baseline(dsa, m, per):
    """The CPU oracle (a single-thread C++ restatement of the reference; the Julia reference cannot run
    here) on the SAME C3 input as the GPU leg (all 1M columns, 10M nnz: ~3 s of build + 3 products of ~0.5 s)."""
    import multiprocessing
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_binding
    ora = oracle_binding.load(dsa)
    ncs = 1_000_000 if m == 1_000_000 else 400_000
    I, J, V = c3_triplets(m, ncs, per, 0, seed_rows=5, seed_vals=6)
    t = time.perf_counter()
    B = dsa.dynamicsparse(I, J, V, m, ncs, binding=ora)
    tb = time.perf_counter() - t
    x = unit12(7, ncs)
    cap = B.info(dsa.COLMAJOR)["capacity"]
    reps = 3
    t = time.perf_counter()
    for _ in range(reps):
        B.mul(x)                                  # _mul with the reference's Dict accumulator
    ts = (time.perf_counter() - t) / reps
    bytes_ = 16 * cap + 8 * ncs + 8 * m
    xx, xp = dsa.binding._f64(x)
    yy = np.empty(m)
    lib = ora.lib
    lib.ora_mat_spmv_dense_fastacc.argtypes = [C.c_void_p, C.c_int32, dsa.binding.P_F64, C.c_int64, dsa.binding.P_F64, C.c_int64]
    t = time.perf_counter()
    for _ in range(reps):
        lib.ora_mat_spmv_dense_fastacc(B.h, 0, xp, ncs, yy.ctypes.data_as(dsa.binding.P_F64), m)
    tf = (time.perf_counter() - t) / reps
    # the same C2 insert batches on the oracle (single thread)
    n0 = 700000
    vo = dsa.dynamicsparsevec(np.arange(1, n0 + 1, dtype=np.int64) * 2, unit12(3, n0), binding=ora)
    app = np.arange(1400001, 1500001, dtype=np.int64)
    t = time.perf_counter()
    vo.set_batch(app, unit12(3, 100000))
    ta = time.perf_counter() - t
    odd = np.unique(1 + 2 * (splitmix_array(4, 120000) % np.uint64(700000)).astype(np.int64))[:100000]
    np.random.default_rng(4).shuffle(odd)
    t = time.perf_counter()
    vo.set_batch(odd, unit12(4, len(odd)))
    tb2 = time.perf_counter() - t
    c5 = c5_streaming(dsa, None, None, None, *C5_FULL, binding=ora, stop_after=10_000)
    return {"value": round(bytes_ / 1e9 / ts, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "c5_first_10k_columns": {k: c5[k] for k in ("columns", "element_writes", "write_s", "element_writes_per_s", "spmv_ms_avg")},
            "inserts_per_s": {"batch_A_ascending_appends": round(100000 / ta, 1), "batch_B_uniform": round(len(odd) / tb2, 1)},
            "sample": "%s C3 matrix (%d x %d, %d nnz, colmajor capacity %d): y = A*x with the "
                      "reference's Dict accumulator (src/operations.jl:101), 3 products, 1 thread of %d host cores"
                      % ("the full" if ncs == 1_000_000 else "first %d columns of the" % ncs, m, ncs, len(I), cap, multiprocessing.cpu_count()),
            "nnz_per_s": round(len(I) / ts, 1),
            "dense_accumulator_gbps": round(bytes_ / 1e9 / tf, 4), "dense_accumulator_nnz_per_s": round(len(I) / tf, 1),
            "build_s": round(tb, 2)}


if __name__ == "__main__":
    main()
