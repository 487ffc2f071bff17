#!/usr/bin/env python3
"""bench.py -- rows/sec of the SELECT/WHERE scan-and-filter on the commands_* schema.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one execution of the hot path: the reference's Sample-1 query shape
    WHERE sudo_used = FALSE AND user_name = "student1030"
(the query QPESeq really runs through linearSearchRecords over every row, and the one
BASELINE.md anchors on) over the rank's row-range shard of the seeded synthetic table,
inputs resident in HBM, through the C-ABI (pqps_qstream_scan / pqps_exchange_select).  With N > 1
the step also merges the matching row IDs of all shards on every rank (RCCL), pipelined on a second
stream.  Workload = BASELINE.json configs[1]: 100 M synthetic rows per GPU (weak scaling: N GPUs
scan N x 100 M rows); `--rows-total 1000000000` gives configs[3] (1 B rows sharded over the ranks),
with `--mode count` configs[4] (COUNT(*) + all-reduce).

The table exists TWICE in HBM and consecutive steps alternate between the copies: a step's 300 MB would
otherwise find part of itself in the 256 MiB Infinity Cache left by the step before, and the HBM roofline
would be flattered.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import importlib.util
import json
import os
import pathlib
import sys
import time

# The query stream keeps two queries in flight on two HIP streams; they only overlap if the runtime gives the two
# streams different hardware queues.  ROCm's default pool is 4 queues shared by every stream of the process (torch's
# included), assigned in an order the host cannot see; with 8 the two lanes never collide.  Has to be in the
# environment before the HIP runtime starts (i.e. before torch is imported).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = pathlib.Path(__file__).resolve().parent
PKG = ROOT / "parallel-query-processing-system_amd"
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_MEASURED_COPY_GBPS = 6290.0  # the same guide's measured float4 copy (79 % of the spec): what a streaming kernel has been SEEN to reach

QUERIES = {
    # name: (chain, SQL text as in the reference's sample-queries.txt where it exists)
    "S1": ([("sudo_used", "=", "FALSE"), "AND", ("user_name", "=", "student1030")],
           'sudo_used = FALSE AND user_name = "student1030"'),
    "Q_A": ([("risk_level", ">", "3")], "risk_level > 3"),
    "Q_B": ([("sudo_used", "=", "TRUE"), "AND", ("risk_level", ">", "2")], "sudo_used = TRUE AND risk_level > 2"),
    "Q_C": ([("exit_code", "!=", "0"), "AND", ("user_id", ">=", "1500"), "OR", ("risk_level", "=", "5")],
            "exit_code != 0 AND user_id >= 1500 OR risk_level = 5"),
    "Q_u8": ([("sudo_used", "=", "TRUE")], "sudo_used = TRUE"),
    "Q_u16": ([("user_name", "=", "student1030")], 'user_name = "student1030"'),
    # dense answers: 13.5 % and 43 % of the rows
    "Q_r2": ([("risk_level", ">", "2")], "risk_level > 2"),
    "Q_r1": ([("risk_level", ">", "1")], "risk_level > 1"),
}


def load_pkg():
    spec = importlib.util.spec_from_file_location("pqps_amd", PKG / "__init__.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["pqps_amd"] = mod
    spec.loader.exec_module(mod)
    spec2 = importlib.util.spec_from_file_location("pqps_amd_merge", PKG / "merge.py")
    mg = importlib.util.module_from_spec(spec2)
    spec2.loader.exec_module(mg)
    return mod, mg


def cpu_baseline(pq, chain, sql, seed, log):
    """QPESeq-equivalent CPU rate on this box's host cores, bounded sample (rank 0, N=1).

    kind "reference": the reference's own linearSearchRecords (oracle/_ref, compiled
    from its sources) over 1040-byte AoS records;  else kind "port": the oracle's
    row-at-a-time evaluator over the same columns.  The OpenMP row-parallel port is
    reported next to it (`omp`)."""
    import numpy as np
    sys.path.insert(0, str(ROOT / "tests"))
    import qpelib as q

    import statistics
    ncores = os.cpu_count() or 1
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    n_col = 20_000_000
    host = q.HostSynth(n_col, seed=seed)

    def timed(fn, reps):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            k = fn()
            ts.append(time.perf_counter() - t0)
        return k, statistics.median(ts), min(ts)

    k1, t_port, t_port_best = timed(lambda: len(host.oracle_scan(chain, nthreads=1)), 3)
    nth = min(ncores, 64)
    k2, t_omp, t_omp_best = timed(lambda: len(host.oracle_scan(chain, nthreads=nth)), 5)
    assert k1 == k2
    # flat keys: the driver's record keeps one level of this object
    common = {"cpu_model": cpu_model, "cores_online": ncores,
              "port_serial_value": n_col / t_port, "port_serial_best": n_col / t_port_best,
              "omp_value": n_col / t_omp, "omp_best": n_col / t_omp_best, "omp_cores": nth,
              "port_sample": f"{n_col} synthetic rows (columnar), query {sql!r}, {k1} matches: the oracle's row-at-a-time evaluator, "
                             f"serial (median of 3) and OpenMP row ranges x{nth} (median of 5)"}
    log(f"cpu port ({cpu_model}, {ncores} cores online): serial {n_col / t_port / 1e6:.1f} M rows/s, omp x{nth} {n_col / t_omp / 1e6:.1f} M rows/s")

    ref = q.load_ref()
    if ref is None:
        return dict({"value": n_col / t_port, "best": n_col / t_port_best, "unit": "rows/s", "cores": 1, "kind": "port",
                     "sample": common["port_sample"]}, **common)
    # the real reference over AoS records built from the same synthetic columns
    n_aos = 12_000_000
    rec_dt = np.dtype({"names": ["command_id", "raw_command", "base_command", "shell_type", "exit_code", "timestamp",
                                 "sudo_used", "working_directory", "user_id", "user_name", "host_name", "risk_level"],
                       "formats": ["<u8", "S512", "S100", "S20", "<i4", "S30", "u1", "S200", "<i4", "S50", "S100", "<i4"],
                       "offsets": [0, 8, 520, 620, 640, 644, 674, 675, 876, 880, 930, 1032], "itemsize": 1040})
    recs = np.zeros(n_aos, dtype=rec_dt)
    a = {k: v[:n_aos] for k, v in host.arr.items()}
    recs["command_id"] = a["command_id"]
    recs["exit_code"], recs["user_id"], recs["risk_level"] = a["exit_code"], a["user_id"], a["risk_level"]
    recs["sudo_used"] = a["sudo_used"]
    recs["shell_type"] = np.array(pq.SYNTH_SHELLS, dtype="S20")[a["shell_type"]]
    recs["user_name"] = np.array(pq.SYNTH_USERS_DICT, dtype="S50")[a["user_name"]]
    recs["host_name"] = np.array(pq.SYNTH_HOSTS, dtype="S100")[a["host_name"]]
    recs["base_command"] = np.array(pq.SYNTH_BASES, dtype="S100")[a["base_command"]]
    recs["raw_command"], recs["timestamp"], recs["working_directory"] = b"cmd", b"2025-01-01T00:00:00.000Z", b"/home/u"
    ptrs = (recs.ctypes.data + np.arange(n_aos, dtype=np.uint64) * 1040).astype(np.uint64)
    ref.linearSearchRecords.restype = C.c_void_p
    ref.linearSearchRecords.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    wl = pq.WhereList(chain)
    ts = []
    for _ in range(5):
        cnt = C.c_int()
        t0 = time.perf_counter()
        res = ref.linearSearchRecords(ptrs.ctypes.data, n_aos, C.cast(wl.ptr, C.c_void_p), C.byref(cnt))
        ts.append(time.perf_counter() - t0)
        libc.free(res)
    med, best = statistics.median(ts), min(ts)
    want = int(np.searchsorted(host.oracle_scan(chain), n_aos))
    assert cnt.value == want, (cnt.value, want)
    log(f"cpu reference (QPESeq linearSearchRecords): {n_aos / med / 1e6:.2f} M rows/s on 1 core (best {n_aos / best / 1e6:.2f})")
    return dict({"value": n_aos / med, "best": n_aos / best, "unit": "rows/s", "cores": 1, "kind": "reference",
                 "sample": f"{n_aos} AoS records (1040 B) of the same synthetic table, query {sql!r}, "
                           f"reference linearSearchRecords compiled -O2 (oracle/_ref), median of 5"}, **common)


def numpy_checksum(ids):
    """(sum of ids, sum of ids[i] * (2 i + 1)) mod 2^64 -- what pqps_ids_checksum / hipQueryChecksumHIP compute on the device."""
    import numpy as np
    v = np.asarray(ids, dtype=np.uint64)
    with np.errstate(over="ignore"):
        s0 = int(v.sum(dtype=np.uint64))
        s1 = int((v * (np.arange(len(v), dtype=np.uint64) * np.uint64(2) + np.uint64(1))).sum(dtype=np.uint64))
    return s0, s1


def gathered_checksum(parts):
    """Checksum of the concatenation of lists whose own (count, s0, s1) are `parts`, in order: entry i of a part lands at
    displacement + i, so its s1 grows by 2 * displacement * s0."""
    mask = (1 << 64) - 1
    s0 = s1 = displ = 0
    for k, a, b in parts:
        s0 = (s0 + a) & mask
        s1 = (s1 + b + 2 * displ * a) & mask
        displ += k
    return displ, s0, s1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed region of `steps` queries: `value` is the median, the spread is reported")
    ap.add_argument("--rows", type=int, default=100_000_000, help="rows per GPU (weak scaling)")
    ap.add_argument("--rows-total", type=int, default=0,
                    help="total rows over all ranks (strong scaling, e.g. 1000000000 = configs[3]/[4]); overrides --rows")
    ap.add_argument("--mode", default="ids", choices=["ids", "count"],
                    help="ids: ascending row-ID list (+ all-gatherv merge for N > 1); count: COUNT(*) (+ all-reduce), configs[4]")
    ap.add_argument("--copies", type=int, default=2, help="table copies the steps alternate between (1: the same buffers every step)")
    ap.add_argument("--query", default="S1", choices=sorted(QUERIES))
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary queries / index / projection / 1 B-row legs")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "torch"],
                    help="N>1 exchange step: 'rccl' = the product calls RCCL itself (engine API / shim); "
                         "'torch' = torch.distributed collectives from Python (also the gloo rehearsal path)")
    ap.add_argument("--rccl-library", default=None, help="the librccl.so the product loads (default: the one of this torch build)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="N=1, shim level: one query at a time on one stream (no second query in flight)")
    ap.add_argument("--level", default="engine", choices=["engine", "shim"],
                    help="where `value` is measured, AT EVERY N.  engine (default): through the reference-facing engine API -- "
                         "initializeEngineSynthetic{,Rank}HIP + executeQuery{Select,Count}AsyncHIP tickets issued by a C loop "
                         "(host/engineBench.c), no torch in the timed region; with N > 1 the engines are joined over RCCL "
                         "(hipEngineJoinRanksHIP) and every ticket's answer is the all-gathered list.  shim: pqps_qstream_scan / "
                         "pqps_exchange_select of include/pqps_hip.h driven from Python, one level below.  Both figures are reported "
                         "(config.engine_ms_per_query / config.shim_ms_per_step).")
    ap.add_argument("--engine-threads", type=int, default=1, help="host threads issuing queries in the engine-level leg")
    ap.add_argument("--engine-in-flight", type=int, default=3, help="tickets each thread keeps outstanding in the engine-level leg")
    ap.add_argument("--force-merge", action="store_true",
                    help="N=1 rehearsal: run the N>1 exchange (RCCL communicator of one rank: sizes, payload, merge) at both levels")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    # stdout carries exactly ONE line (the JSON): RCCL prints its version banner to fd 1 when the
    # communicator comes up, so everything incidental is sent to stderr until the result is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    os.environ["PQPS_DEVICE"] = str(dev_index)                       # the engines of this process live on this rank's GPU
    device = torch.device("cuda", dev_index)
    exchange = world > 1 or args.force_merge              # is there an exchange step after the scan?
    if exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
    cdev = device if args.backend == "nccl" else torch.device("cpu")      # where small control tensors live

    def agree(ok):
        """True when EVERY rank says so (an all-reduce every rank reaches, whatever happened on it before)."""
        if not exchange or world == 1:
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_u64(values):
        """every rank's tuple of u64 values, as a list of tuples in rank order"""
        if world == 1:
            return [tuple(int(v) for v in values)]
        mine = torch.tensor([v - (1 << 64) if v >= (1 << 63) else v for v in values], dtype=torch.int64, device=cdev)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        return [tuple(int(v) & ((1 << 64) - 1) for v in p.cpu().tolist()) for p in parts]

    def rank_fence():
        torch.cuda.synchronize()
        if exchange and world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pq, mg = load_pkg()
    ctx = pq.Context(dev_index)
    dev_name, cus, hbm = ctx.info()
    # an explicit (non-default) stream: the shim treats a NULL handle as "the context's own stream"
    compute = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(compute)
    sptr = compute.cuda_stream
    assert sptr != 0

    chain, sql = QUERIES[args.query]
    strong = args.rows_total > 0
    n_global = args.rows_total if strong else args.rows * world
    start, count = mg.shard_rows(n_global, world, rank)
    if start + count > 2**32:
        sys.exit("row IDs are u32: the table must stay below 2^32 rows")
    count_mode = args.mode == "count"
    copies = max(1, args.copies)
    rccl_library = args.rccl_library or mg.default_rccl_library(torch)

    # ---- table: this rank's row range, generated in place on the device, `copies` times -------
    keep = []

    def alloc(nbytes):
        t = torch.empty(nbytes + 64, dtype=torch.uint8, device=device)
        keep.append(t)
        return t.data_ptr()

    extras = [] if (args.no_extras or world > 1) else [k for k in QUERIES if k != args.query]
    needed = {leaf[0] for k in [args.query] + extras for leaf in _leaves(QUERIES[k][0])} & set(pq.COLUMNS)
    t0 = time.perf_counter()
    tables = []
    for c in range(copies):
        cols_c = set(needed)
        if extras and c == 0:
            cols_c |= {"command_id", "user_id", "risk_level"}      # index-mode leg (configs[2]), first copy only
        tables.append(pq.SyntheticTable(ctx, count, seed=args.seed, row0=start, columns=sorted(cols_c), alloc=alloc, stream=sptr))
    torch.cuda.synchronize()
    log(f"{dev_name}, {cus} CUs: generated {count:,} rows x {sorted(needed)} x {copies} copies in {time.perf_counter() - t0:.2f} s")

    bound = [t.bind(chain) for t in tables]                         # (pred, cols, n_cols, bytes per row) per copy
    bytes_per_row = bound[0][3]
    widths = sorted({tables[0].width[leaf[0]] for leaf in _leaves(chain)}, reverse=True)
    L = pq.lib()

    # ---- calibration: one shim-level scan of this rank's shard.  Its count sizes the buffers, its list (checksummed where
    # it lies) is what every other path's answer is compared with.
    cal_ids = torch.empty(max(count, 1), dtype=torch.int32, device=device)
    cal_cnt = torch.zeros(1, dtype=torch.int64, device=device)
    pred, cols, nc, _ = bound[0]
    pq.check(L.pqps_filter_scan(ctx.h, cols, nc, count, start, C.byref(pred), cal_ids.data_ptr(), count,
                                cal_cnt.data_ptr(), sptr), "calibration scan")
    torch.cuda.synchronize()
    kernel_chosen = L.pqps_last_kernel().decode()                    # what the shim launched for this query on this table (this thread's last call)
    local_matches = int(cal_cnt.item())
    local_sums = ctx.ids_checksum(cal_ids.data_ptr(), local_matches, sptr)
    per_rank = gather_u64((local_matches,) + local_sums)             # [(count, s0, s1)] in rank order
    total_matches, want_s0, want_s1 = gathered_checksum(per_rank)   # the all-gathered list's checksum, from the ranks' own
    max_matches = max(p[0] for p in per_rank)
    del cal_ids
    torch.cuda.empty_cache()

    result_notes = []

    # ==== engine level: the reference-facing API, a C loop issuing tickets =====================================================
    def engine_level():
        """-> dict, or raises.  Every rank calls this (it contains collective steps)."""
        B = pq.bench_lib()
        engines = []
        try:
            if exchange:
                for _ in range(copies):
                    engines.append(pq.HipEngine.synthetic_rank(n_global, world, rank, seed=args.seed))
                # bring-up in two halves with an agreement in between: a rank that fails locally must not leave the others
                # inside ncclCommInitRank (merge.open_exchange does the same for the shim-level exchange)
                for e in engines:
                    box = [None, None]
                    if rank == 0:
                        try:
                            box[0] = pq.HipEngine.rccl_id(rccl_library)
                        except Exception as ex:                      # noqa: BLE001
                            box[1] = repr(ex)
                    if world > 1:
                        dist.broadcast_object_list(box, src=0)
                    if box[1] is not None:
                        raise pq.PqpsError("RCCL id: " + box[1])
                    ok = True
                    try:
                        e.join_prepare(rccl_library)
                    except Exception as ex:                          # noqa: BLE001
                        log(f"rank {rank}: {ex!r}")
                        ok = False
                    if not agree(ok):
                        raise pq.PqpsError("exchange set-up failed on some rank")
                    ok = True
                    try:
                        e.join_connect(box[0])
                    except Exception as ex:                          # noqa: BLE001
                        log(f"rank {rank}: {ex!r}")
                        ok = False
                    if not agree(ok):
                        raise pq.PqpsError("communicator bring-up failed on some rank")
            else:
                engines = [pq.HipEngine.synthetic(count, seed=args.seed) for _ in range(copies)]
            arr = (C.POINTER(pq.EngineS) * copies)(*[e.e for e in engines])
            wl = pq.WhereList(chain)
            # ---- untimed: every engine answers once, and the answer is checked BEFORE anything is timed -- the ticket's list,
            # checksummed where it lies, against the checksum of the ranks' own shim-level lists put together in rank order
            ok, why = True, ""
            try:
                for e in engines:
                    tk = e.select_async(chain, count_only=count_mode)
                    if not tk:
                        raise pq.PqpsError("no ticket")
                    k, res = e.await_ticket(tk)
                    if k != total_matches:
                        ok, why = False, f"engine answers {k} rows, the ranks' scans {total_matches}"
                    elif not count_mode:
                        got = e.ticket_checksum(tk)
                        if got != (want_s0, want_s1):
                            ok, why = False, f"the engine's list differs from the ranks' lists in rank order (checksum {got} vs {(want_s0, want_s1)})"
                        if exchange and int(res.shard_count[0]) != local_matches:
                            ok, why = False, f"this rank's part: {int(res.shard_count[0])} vs {local_matches}"
                    e.release_ticket(tk)
            except Exception as ex:                                  # noqa: BLE001
                ok, why = False, repr(ex)
            if not ok:
                print(f"[bench] rank {rank}: engine-level verification failed: {why}", file=sys.stderr, flush=True)
            if not agree(ok):
                raise pq.PqpsError("engine-level verification failed: " + (why or "on another rank"))
            for e in engines:
                L.hipEngineKernelTiming(e.e, 1)
                e.wire_bytes(reset=True)
            # ---- timed: `reps` regions of exactly `steps` queries, each bracketed by a barrier + device synchronise
            regions, issue_s, await_s, last = [], 0.0, 0.0, None
            for r in range(max(1, args.reps)):
                res = pq.BenchResult()
                res.want_checksum = 0 if count_mode else 1
                rank_fence()
                rc = B.hipEngineBench(arr, copies, wl.ptr, 1 if count_mode else 0, args.engine_threads, args.engine_in_flight,
                                      args.warmup if r == 0 else 2, args.steps, C.byref(res))
                rank_fence()
                good = rc == 0 and res.mismatches == 0 and res.matches == total_matches and \
                    (count_mode or (res.have_checksum and (int(res.checksum[0]), int(res.checksum[1])) == (want_s0, want_s1)))
                if not agree(good):
                    raise pq.PqpsError(f"hipEngineBench: rc {rc}, {res.matches} matches (want {total_matches}), {res.mismatches} mismatches, "
                                       f"checksum {'ok' if good else 'differs or missing'}")
                regions.append(max_over_ranks(res.seconds))
                issue_s += res.issue_seconds
                await_s += res.await_seconds
                last = res
            kern_ms, launches = 0.0, 0
            for e in engines:
                ev, tot, k = C.c_double(), C.c_double(), C.c_int()
                if L.hipEngineKernelTime(e.e, C.byref(ev), C.byref(tot), C.byref(k)) == 0:
                    kern_ms += tot.value
                    launches += k.value
            wire = [sum(x) for x in zip(*[e.wire_bytes() for e in engines])]
            eager = [e.eager_queries() for e in engines]
            n_q = args.steps * max(1, args.reps) * args.engine_threads
            n_issued = n_q + (args.warmup + 2 * (max(1, args.reps) - 1)) * args.engine_threads + copies     # timed + warm-up + verification queries
            med = sorted(regions)[len(regions) // 2]
            per_q = med / (args.steps * args.engine_threads)
            out = {"value": n_global / per_q, "unit": "rows/s", "ms_per_query": per_q * 1e3, "queries_per_region": args.steps * args.engine_threads,
                   "regions_s": regions, "threads": args.engine_threads, "tickets_in_flight_per_thread": args.engine_in_flight,
                   "engines_alternated": copies, "matches": int(last.matches),
                   "host_issue_us_per_query": issue_s / n_q * 1e6, "host_await_us_per_query": await_s / n_q * 1e6,
                   "in_stream_kernel_ms": kern_ms / max(launches, 1), "launches_timed": launches,
                   "wire_bytes_in_per_query": wire[0] / n_issued, "u32_bytes_in_per_query": wire[1] / n_issued,
                   "one_collective_queries": sum(x[0] for x in eager), "exchanged_queries": sum(x[1] for x in eager), "eager_ids_per_rank": eager[0][2],
                   "verified": "every engine's answer before the timed regions and the last ticket of every timed region: count and "
                               "device-side checksum of the list == the ranks' shim-level lists in rank order",
                   "api": ("initializeEngineSyntheticRankHIP + hipEngineJoinRanksHIP + " if exchange else "initializeEngineSyntheticHIP + ")
                          + "executeQuery%sAsyncHIP / awaitQueryHIP / releaseQueryHIP (host/engineBench.c)" % ("Count" if count_mode else "Select")}
            log(f"engine level: {per_q * 1e6:.1f} us/query (median of {len(regions)} regions: {min(regions) / args.steps * 1e6:.1f} .. {max(regions) / args.steps * 1e6:.1f}), "
                f"{out['value'] / 1e12:.3f} T rows/s ({args.engine_threads} thread(s) x {args.engine_in_flight} tickets, {copies} engines); "
                f"host {out['host_issue_us_per_query']:.1f} us issuing + {out['host_await_us_per_query']:.1f} us awaiting per query; "
                f"launch in the stream {out['in_stream_kernel_ms'] * 1e3:.1f} us")
            return out
        finally:
            for e in engines:
                try:
                    e.close()
                except Exception:                                    # noqa: BLE001
                    pass

    # ==== shim level: include/pqps_hip.h driven from Python (round 2's headline path; the fallback of the engine level) =========
    def shim_level():
        """-> dict.  Every rank calls this."""
        slot_cap = (count + 4096) // 4096 * 4096 if (exchange and not count_mode) else \
            ((int(max_matches * 1.25) + 4096) // 4096 * 4096 if not count_mode else 4096)
        native = exchange and args.exchange == "rccl" and (args.backend == "nccl" or args.rccl_library is not None)
        qs = None
        if not exchange and not args.no_pipeline:
            qs = C.c_void_p()
            pq.check(L.pqps_qstream_create(ctx.h, RING, C.byref(qs)), "pqps_qstream_create")
        xch = None
        if native:
            # every rank must take the same path: if the shim-driven exchange cannot be set up on any of them
            # (RCCL library not found, allocation failed, communicator refused), all fall back to the
            # torch.distributed collectives -- the ranks agree before and after the communicator is built
            xch = mg.ShardExchange.open(pq, ctx, torch, dist, world, rank, slot_cap, ring=RING, rccl_library=rccl_library, control_device=cdev)
            native = xch is not None
        st = {"native": native, "xch": xch, "mergers": None}

        def make_torch_path():
            st["mergers"] = [mg.IdMerger(torch, dist, world, rank, slot_cap, device, ctx=ctx, pq=pq, shard=(count, start),
                                         host_staged=(args.backend != "nccl"), always_collective=args.force_merge) for _ in range(RING)]
            st["comm"] = torch.cuda.Stream(device=device)
            st["filt_done"] = [torch.cuda.Event() for _ in range(RING)]
            st["merge_done"] = [torch.cuda.Event() for _ in range(RING)]
            st["count_out"] = torch.zeros(2 * RING, dtype=torch.int64, device=device)
            st["count_all"] = torch.zeros(RING, dtype=torch.int64, device=cdev)
        if not native:
            make_torch_path()

        def step(k):
            r = k % RING
            pred, cols, nc, _ = bound[k % copies]                       # consecutive steps read different buffers
            if st["native"]:
                if count_mode:
                    st["xch"].count(cols, nc, count, C.byref(pred), r, sptr)
                else:
                    st["xch"].select(cols, nc, count, start, C.byref(pred), r, sptr)
                return
            m = st["mergers"][r]
            count_out = st["count_out"]
            if qs is not None:
                if count_mode:
                    pq.check(L.pqps_qstream_count(qs, cols, nc, count, C.byref(pred), count_out[2 * r:].data_ptr(), sptr), "pqps_qstream_count")
                else:
                    pq.check(L.pqps_qstream_scan(qs, cols, nc, count, start, C.byref(pred), m.ids_ptr, m.cap, m.count_ptr, sptr),
                             "pqps_qstream_scan")
                return
            merge_done, filt_done, comm = st["merge_done"], st["filt_done"], st["comm"]
            merge_done[r].synchronize()
            if count_mode:
                pq.check(L.pqps_filter_count(ctx.h, cols, nc, count, C.byref(pred), count_out[2 * r:].data_ptr(), sptr), "pqps_filter_count")
            else:
                pq.check(L.pqps_filter_scan(ctx.h, cols, nc, count, start, C.byref(pred), m.ids_ptr, m.cap,
                                            m.count_ptr, sptr), "pqps_filter_scan")
            if not exchange:
                return
            filt_done[r].record(compute)
            with torch.cuda.stream(comm):
                comm.wait_event(filt_done[r])
                if count_mode:                                          # mpi:745 through torch.distributed
                    count_all = st["count_all"]
                    if args.backend == "nccl":
                        count_all[r] = count_out[2 * r]
                        dist.all_reduce(count_all[r:r + 1])
                    else:
                        t = count_out[2 * r:2 * r + 1].cpu()
                        dist.all_reduce(t)
                        count_all[r] = t[0]
                else:
                    # sizes of query k, then the payload of the query three calls back, whose sizes have long reached
                    # the host (same order as the shim-driven exchange: the host never waits for a scan that still runs)
                    m.begin(stream_ptr=comm.cuda_stream)
                    back = 3 if RING >= 5 else (2 if RING == 4 else 1)
                    prev = st["mergers"][(k - back) % RING]
                    if k >= back and prev is not m and prev._pending:
                        prev.finish()
                        merge_done[(k - back) % RING].record(comm)
                    if RING == 1:
                        m.finish()
                merge_done[r].record(comm)

        def drain():
            """Payload phases still held back (the last query's) go out."""
            if qs is not None:
                pq.check(L.pqps_qstream_sync(qs), "pqps_qstream_sync")
            if st["native"]:
                st["xch"].sync()
            elif exchange and not count_mode:
                with torch.cuda.stream(st["comm"]):
                    for j in sorted(range(RING), key=lambda i: st["mergers"][i]._issued):       # oldest first, on every rank alike
                        if st["mergers"][j]._pending:
                            st["mergers"][j].finish()
                            st["merge_done"][j].record(st["comm"])

        def phase(n_steps):
            """n_steps steps + everything they enqueued finished, on every rank -> (all ranks ok, seconds, host seconds enqueuing).
            A rank on which the product's exchange fails (a bounded wait ran out, the communicator was aborted) stops
            issuing and still arrives at the agreement."""
            ok = True
            t0 = time.perf_counter()
            enq = 0.0
            try:
                for k in range(n_steps):
                    step(k)
                enq = time.perf_counter() - t0
                drain()
            except pq.PqpsError as ex:
                print(f"[bench] rank {rank}: {ex}", file=sys.stderr, flush=True)
                ok = False
            rank_fence()
            dt = time.perf_counter() - t0
            return agree(ok), dt, enq

        def fall_back(where):
            """All ranks leave the shim-driven exchange together and go on with the torch.distributed form of the same step."""
            log(f"the shim-driven exchange failed {where}: every rank falls back to torch.distributed (IdMerger)")
            result_notes.append(f"shim-driven exchange failed {where}; finished on torch.distributed")
            try:
                st["xch"].close()                                    # (a dead exchange tears down without waiting for its peers)
            except Exception as ex:                                  # noqa: BLE001
                print(f"[bench] rank {rank}: closing the exchange: {ex!r}", file=sys.stderr, flush=True)
            st["xch"], st["native"] = None, False
            make_torch_path()

        def verify_slot(slot):
            """What the steps left in ring slot `slot`, against the ranks' own lists: on EVERY rank."""
            if count_mode:
                if st["native"]:
                    got_total, got_local = st["xch"].count_result(slot)
                elif exchange:
                    torch.cuda.synchronize()
                    got_total, got_local = int(st["count_all"][slot].item()), int(st["count_out"][2 * slot].item())
                else:
                    torch.cuda.synchronize()
                    got_local = int(st["count_out"][2 * slot].item())
                    got_total = got_local
                return got_local == local_matches and got_total == total_matches, f"counts {got_local}/{got_total} vs {local_matches}/{total_matches}"
            if st["native"]:
                merged, got_local = st["xch"].result(slot)
            else:
                got_local = st["mergers"][slot].local_count()
                merged = st["mergers"][slot].result() if exchange else None
            if got_local != local_matches:
                return False, f"local count {got_local} vs {local_matches}"
            if exchange:
                if len(merged) != total_matches:
                    return False, f"gathered {len(merged)} IDs, the ranks hold {total_matches}"
                if numpy_checksum(merged) != (want_s0, want_s1):
                    return False, "the gathered list is not the ranks' lists in rank order"
                if len(merged) and not bool(np.all(merged[1:] > merged[:-1])):
                    return False, "the gathered list is not strictly ascending"
            else:
                m = st["mergers"][slot]
                if (ctx.ids_checksum(m.ids_ptr, local_matches, sptr) if local_matches else (0, 0)) != local_sums:
                    return False, "the query stream's list differs from the single scan's"
            return True, ""

        # ---- warm-up, then verification BEFORE anything is timed (a wrong answer ends the run with rc != 0 on every rank)
        n_warm = max(args.warmup, RING if exchange else 1)
        ok, _, _ = phase(n_warm)
        if not ok and st["native"]:
            fall_back("during warm-up")
            ok, _, _ = phase(n_warm)
        if not ok:
            sys.exit("bench: the warm-up steps failed")
        good, why = True, ""
        try:
            good, why = verify_slot((n_warm - 1) % RING)
        except pq.PqpsError as ex:
            good, why = False, str(ex)
        if not good:
            print(f"[bench] rank {rank}: verification failed: {why}", file=sys.stderr, flush=True)
        if not agree(good):
            sys.exit("bench: the warm-up answer is wrong: " + (why or "on another rank"))
        if qs is not None:
            L.pqps_qstream_wait_ns(qs, 1)
            pq.check(L.pqps_qstream_set_timing(qs, 1), "pqps_qstream_set_timing")
        if st["native"]:
            L.pqps_exchange_wait_ns(st["xch"].h, 1)
            wb = (C.c_uint64 * 2)()
            L.pqps_exchange_wire_bytes(st["xch"].h, wb, 1)
        elif exchange and not count_mode:
            for m in st["mergers"]:
                m.wire_bytes_in = m.u32_bytes_in = 0
        # ---- timed: `reps` regions of exactly `steps` steps
        regions, enq_s = [], 0.0
        r = 0
        while r < max(1, args.reps):
            ok, dt, enq = phase(args.steps)
            if not ok:
                if st["native"]:
                    fall_back("in a timed region")
                    regions, enq_s, r = [], 0.0, 0                   # every region counts on ONE path
                    continue
                sys.exit("bench: a timed region failed")
            regions.append(max_over_ranks(dt))
            enq_s += enq
            r += 1
        good, why = verify_slot((args.steps - 1) % RING)
        if not agree(good):
            sys.exit("bench: the timed steps left a wrong answer: " + (why or "on another rank"))
        waited = 0.0
        in_stream_ms = None
        wire = eager = None
        if qs is not None:
            waited = L.pqps_qstream_wait_ns(qs, 1) * 1e-9
            ev, tot, k = C.c_double(), C.c_double(), C.c_int()
            pq.check(L.pqps_qstream_kernel_time(qs, C.byref(ev), C.byref(tot), C.byref(k)), "pqps_qstream_kernel_time")
            pq.check(L.pqps_qstream_set_timing(qs, 0), "pqps_qstream_set_timing")
            in_stream_ms = tot.value / max(k.value, 1)
        if st["native"]:
            waited = L.pqps_exchange_wait_ns(st["xch"].h, 1) * 1e-9
            wb = (C.c_uint64 * 2)()
            L.pqps_exchange_wire_bytes(st["xch"].h, wb, 0)
            wire = (int(wb[0]), int(wb[1]))
            eg = (C.c_uint64 * 3)()
            L.pqps_exchange_eager(st["xch"].h, eg, 0)
            eager = (int(eg[0]), int(eg[1]), int(eg[2]))
        elif exchange and not count_mode:
            wire = (sum(m.wire_bytes_in for m in st["mergers"]), sum(m.u32_bytes_in for m in st["mergers"]))
        n_steps_all = args.steps * len(regions)
        med = sorted(regions)[len(regions) // 2]
        log(f"shim level: {med / args.steps * 1e6:.1f} us/step (median of {len(regions)} regions: {min(regions) / args.steps * 1e6:.1f} .. "
            f"{max(regions) / args.steps * 1e6:.1f}); host: {(enq_s - waited) / n_steps_all * 1e6:.1f} us/step in runtime calls + "
            f"{waited / n_steps_all * 1e6:.1f} us/step waiting for a free ring slot")
        # ---- the same steps once more with HIP events on the dispatch of every scan kernel (ID output: the ONE launch of
        # the query -- scan tiles + expanders; COUNT(*): the scan, the 1-workgroup reduction after it in `pipeline`):
        # its average duration feeds `roofline`.  Kept out of the timed regions because in timing mode a query
        # runs whole on one stream (no overlap with its neighbours), which is not how `value` is produced.
        ctx.set_timing(True)
        ok, _, _ = phase(args.steps)
        kern_ms, pipe_ms, launches = ctx.kernel_time()
        ctx.set_timing(False)
        if not ok:
            sys.exit("bench: the per-launch timing steps failed")
        out = {"value": n_global * args.steps / med, "ms_per_step": med / args.steps * 1e3, "regions_s": regions,
               "avg_kernel_ms": kern_ms / max(launches, 1), "avg_pipeline_ms": pipe_ms / max(launches, 1), "launches_timed": launches,
               "in_stream_kernel_ms": in_stream_ms, "native": st["native"], "pipelined": qs is not None or st["native"],
               "wire_bytes_in_per_step": wire[0] / n_steps_all if wire else None, "u32_bytes_in_per_step": wire[1] / n_steps_all if wire else None,
               "one_collective_queries": eager[0] if eager else None, "exchanged_queries": eager[1] if eager else None,
               "eager_ids_per_rank": eager[2] if eager else None,
               "api": ("pqps_exchange_%s" % ("count" if count_mode else "select") if st["native"] else
                       ("pqps_qstream_%s" % ("count" if count_mode else "scan") if qs is not None else
                        "pqps_filter_%s%s" % ("count" if count_mode else "scan", " + torch.distributed (IdMerger)" if exchange else "")))
                      + " (include/pqps_hip.h), issued from Python"}
        if st["xch"] is not None:
            st["xch"].close()
        if qs is not None:
            pq.check(L.pqps_qstream_destroy(qs), "pqps_qstream_destroy")
        return out

    # ---- run the levels.  Engine level first (its engines are gone before the shim level allocates its buffers).
    eng = None
    # (the product drives RCCL itself whenever torch.distributed runs over it -- or over anything else when the host names the library
    #  to load: the rehearsal of N > 1 on a one-GPU box uses gloo as torch's transport and tests/loopback/libloopback_mp.so as "RCCL")
    product_rccl = args.exchange == "rccl" and (args.backend == "nccl" or args.rccl_library is not None)
    want_engine = args.level == "engine" and not args.no_pipeline and (not exchange or product_rccl)
    if want_engine:
        err = None
        try:
            eng = engine_level()
        except Exception as ex:                                      # noqa: BLE001
            err = repr(ex)
            print(f"[bench] rank {rank}: engine-level leg failed: {err}", file=sys.stderr, flush=True)
        # all ranks use the engine-level figure or none does
        if not agree(eng is not None):
            if eng is not None or err is None:
                err = "failed on another rank"
            eng = None
        if eng is None:
            result_notes.append("engine level failed (" + str(err) + "); value is the shim-level figure")
    torch.cuda.empty_cache()
    # the shim level runs at N = 1 always (its one-launch-at-a-time leg feeds `roofline`), at N > 1 when the engine level is not the headline
    shim = None
    if world == 1 or eng is None:
        shim = shim_level()
    level = "engine" if eng is not None else "shim"

    if eng is not None:
        rows_per_s, ms_per_step, regions = eng["value"], eng["ms_per_query"], eng["regions_s"]
        per_region = args.steps * args.engine_threads
    else:
        rows_per_s, ms_per_step, regions = shim["value"], shim["ms_per_step"], shim["regions_s"]
        per_region = args.steps
    # n * sum w(c) + 4 * matches (ID list) or + 8 (COUNT(*))
    alg_bytes = count * bytes_per_row + (8 if count_mode else 4 * local_matches)
    avg_kernel_ms = shim["avg_kernel_ms"] if shim else None
    achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9 if avg_kernel_ms else None
    job_achieved = alg_bytes / (ms_per_step * 1e-3) / 1e9

    traffic, traffic_src = pmc_traffic(args.query, count) if (world == 1 and not count_mode) else (None, None)
    cfg_no = (4 if count_mode else 3) if strong else 1
    wire_in = eng.get("wire_bytes_in_per_query") if eng is not None else (shim or {}).get("wire_bytes_in_per_step")
    u32_in = eng.get("u32_bytes_in_per_query") if eng is not None else (shim or {}).get("u32_bytes_in_per_step")
    via = ("engine API (hipEngineJoinRanksHIP: the product calls RCCL)" if eng is not None else
           ("shim-driven (pqps_exchange_*: the product calls RCCL)" if (shim and shim["native"]) else "torch.distributed (merge.IdMerger)"))
    parallelism = f"row-range shards x{world}"
    if exchange:
        parallelism += (f", one {'RCCL' if args.backend == 'nccl' else args.backend + ' (host-staged rehearsal)'} "
                        + ("all-reduce of the counts" if count_mode else
                           "all-gatherv of the row IDs per query on every rank (sizes all-gather with room for 16 384 IDs per rank: an answer "
                           "that fits arrives in that ONE collective; otherwise exactly-sized send/recv at displacements, a list of 32 768 IDs "
                           "and more in compact form: 2 bytes per ID + 4 per 65 536-row group)")
                        + f", {via}")
        if wire_in is not None and not count_mode:
            parallelism += f"; payload received per query and rank: {wire_in:,.0f} bytes on the wire for {u32_in:,.0f} bytes of u32 IDs"
            src = eng if eng is not None else (shim or {})
            if src.get("exchanged_queries"):
                parallelism += f"; {src['one_collective_queries']} of {src['exchanged_queries']} exchanged queries finished in the one collective"
    if result_notes:
        parallelism += "; NOTE: " + "; ".join(result_notes)
    result = {
        "metric": "rows/sec SELECT-filter on commands_* schema",
        "value": rows_per_s, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "/".join(f"u{8 * w}" for w in widths), "data": "synthetic",
        "config": {"workload": (f"configs[{cfg_no}]: " + ("COUNT(*) of the " if count_mode else "") + "SELECT/WHERE scan-filter on the commands_* schema, "
                                + (f"{n_global:,} synthetic rows sharded over {world} GPU(s)" if strong else f"{args.rows:,} synthetic rows per GPU")),
                   "query": sql, "mode": args.mode, "rows_per_gpu": count if strong else args.rows, "rows_total": n_global,
                   "matches_total": total_matches, "selectivity": total_matches / max(n_global, 1),
                   "bytes_per_row": bytes_per_row, "table_copies_alternated": copies,
                   "level": level,
                   # both levels as flat keys (the driver's record keeps one level of this object)
                   "engine_ms_per_query": eng["ms_per_query"] if eng else None,
                   "engine_value": eng["value"] if eng else None,
                   "shim_ms_per_step": shim["ms_per_step"] if shim else None,
                   "shim_value": shim["value"] if shim else None,
                   "timed_regions": len(regions), "value_is": "median over the timed regions of `steps` queries each (roofline.value_spread_*)",
                   "pipelining": ((f"{args.engine_threads} host thread(s) x {args.engine_in_flight} asynchronous tickets outstanding over {copies} engines "
                                   "(copies of the table, alternated); every engine runs its queries on two lanes = two HIP streams "
                                   "(one for tables of 537 M rows and more): the tail of one launch under the scan tiles of the next")
                                  if eng is not None else
                                  ("two queries in flight, each whole on a stream of its own: the tail of query k's launch (+ its exchange) under the scan tiles of query k+1"
                                   if shim["pipelined"] else "none: the queries back to back on one stream")),
                   "parallelism": parallelism,
                   "verified": "before the timed regions and after them, on every rank: count + checksum of the answer == the ranks' own single-scan lists in rank order",
                   "device": dev_name},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS if achieved else None, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": (kernel_chosen if not count_mode else kernel_chosen.replace("MODE_IDS", "MODE_COUNT"))
                               + (" -- the ONE launch of an ID query: scan tiles (the only readers of the table) + expander workgroups"
                                  if not count_mode else " -- the scan; the one-workgroup reduction of the totals follows it"),
                     "frac_is": "ONE launch at a time at the shim level (HIP events on the dispatch), re-run after the timed regions; job_frac is the timed regions' figure",
                     "avg_kernel_ms": avg_kernel_ms, "avg_pipeline_ms": shim["avg_pipeline_ms"] if shim else None,
                     # the same launches as they ran IN the timed region (several in flight): longer each, shorter per query
                     "in_stream_kernel_ms": eng["in_stream_kernel_ms"] if eng else (shim["in_stream_kernel_ms"] if shim else None),
                     "launches_timed": shim["launches_timed"] if shim else None, "algorithmic_bytes_per_launch": alg_bytes,
                     # the same algorithmic bytes over the TIMED region's time per query -- per GPU
                     "job_achieved": job_achieved, "job_frac": job_achieved / HBM_PEAK_GBPS,
                     # beside the spec: the guide's measured streaming rate on this part (6.29 TB/s, float4 copy) -- `peak` stays the spec
                     "guide_measured_copy": HBM_MEASURED_COPY_GBPS, "frac_of_guide_measured": achieved / HBM_MEASURED_COPY_GBPS if achieved else None,
                     "job_frac_of_guide_measured": job_achieved / HBM_MEASURED_COPY_GBPS,
                     "value_spread_reps": len(regions), "value_spread_min": n_global * per_region / max(regions),
                     "value_spread_median": rows_per_s, "value_spread_max": n_global * per_region / min(regions)},
    }
    if eng is not None:
        result["engine_level"] = eng
    if shim is not None:
        result["shim_level"] = shim

    # side measurements must never cost the headline line
    if rank == 0 and world == 1 and not args.no_extras:
        try:
            result["extra"] = extras_leg(pq, L, ctx, tables, count, start, sptr, torch, device, extras, log, alloc, args.seed)
            ns = result["extra"].get("north_star_1b", {})
            for name in ("S1", "Q_A", "Q_B", "Q_C"):                 # flat keys under `roofline`: the north star's own size
                if f"{name}_ids" in ns:
                    result["roofline"][f"north_star_1b_{name}_frac"] = ns[f"{name}_ids"]["frac_of_8TBps"]
                    result["roofline"][f"north_star_1b_{name}_us"] = ns[f"{name}_ids"]["avg_query_ms"] * 1e3
                if f"{name}_count" in ns:
                    result["roofline"][f"north_star_1b_{name}_count_frac"] = ns[f"{name}_count"]["frac_of_8TBps"]
            for name in ("Q_A", "Q_B", "Q_C"):                       # engine-level figures of the other shapes, beside S1's
                if f"{name}_engine" in result["extra"]:
                    result["config"][f"engine_ms_per_query_{name}"] = result["extra"][f"{name}_engine"]["ms_per_query"]
        except Exception as e:                                   # noqa: BLE001
            log(f"extras leg failed: {e!r}")
            result["extra"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(pq, chain, sql, args.seed, log)
        except Exception as e:                                   # noqa: BLE001
            log(f"cpu baseline failed: {e!r}")
            result["cpu_baseline"] = {"value": None, "unit": "rows/s", "cores": 0, "kind": "port", "sample": "failed: " + repr(e)}
        try:
            result["cpu_baseline"].update(e2e_leg(log))
        except Exception as e:                                   # noqa: BLE001
            log(f"end-to-end CSV leg failed: {e!r}")
            result["cpu_baseline"]["e2e_error"] = repr(e)
    if exchange:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(result), flush=True)


RING = 6                       # result slots in flight (the exchange holds two queries' payload back: needs >= 5)


def e2e_leg(log):
    """the real-CSV path -- full `record` rows, tokenizer -> connectEngine -> engine -- timed at 50 k and 1 M rows:
    the reference's own driver (oracle/_ref/QPESeq_ref, compiled from its sources) and QPEHIP on the same seeded CSV
    (scripts/make_csv.py) and the reference's sample-queries.txt, phases as both drivers print them (QPESeq.c:89-94:
    initialisation, query execution, total).  Flat keys for cpu_baseline."""
    import re
    import shutil
    import subprocess
    import tempfile
    out = {}
    qpeseq = ROOT / "oracle" / "_ref" / "QPESeq_ref"
    qpehip = PKG / "QPEHIP"
    if not qpeseq.exists() or not qpehip.exists():
        return {"e2e_note": "oracle/_ref/QPESeq_ref or QPEHIP not built"}

    def phases(text):
        t = re.sub(r"\x1b\[[0-9;]*m", "", text)

        def g(k):
            return float(re.search(k + r" Time:\s*([0-9.]+)", t).group(1))
        return g("Engine Initialization"), g("Query Execution"), g("Total Execution")

    def norm(text):
        text = text.split("\x1b[36m=======")[0]
        text = re.sub(r"Query Time: [0-9.]+ seconds", "Query Time: X seconds", text)
        return re.sub(r"Execution Time: [0-9.]+", "Execution Time: X", text)

    for rows, tag in ((50_000, "50k"), (1_000_000, "1m")):
        with tempfile.TemporaryDirectory() as td:
            td = pathlib.Path(td)
            subprocess.run([sys.executable, str(ROOT / "scripts" / "make_csv.py"), str(rows), str(td / "data.csv")], check=True, stdout=subprocess.DEVNULL)
            texts = {}
            for name, exe in (("qpeseq", qpeseq), ("qpehip", qpehip)):
                run = td / name
                run.mkdir()
                shutil.copy(td / "data.csv", run / "data.csv")
                shutil.copy(ROOT / "tests" / "golden" / "sample-queries.txt", run / "sample-queries.txt")
                t0 = time.perf_counter()
                r = subprocess.run([str(exe), "data.csv"], cwd=run, capture_output=True, timeout=600, stdin=subprocess.DEVNULL)
                wall = time.perf_counter() - t0
                text = r.stdout.decode("latin-1")
                init_s, query_s, total_s = phases(text)
                out.update({f"e2e_{tag}_{name}_init_s": init_s, f"e2e_{tag}_{name}_query_s": query_s, f"e2e_{tag}_{name}_total_s": total_s,
                            f"e2e_{tag}_{name}_wall_s": wall})
                texts[name] = norm(text)
            out[f"e2e_{tag}_stdout_identical"] = texts["qpeseq"] == texts["qpehip"]
            log(f"end to end, {rows:,}-row CSV, sample-queries.txt: QPESeq {out[f'e2e_{tag}_qpeseq_total_s']:.3f} s (init {out[f'e2e_{tag}_qpeseq_init_s']:.3f} + queries "
                f"{out[f'e2e_{tag}_qpeseq_query_s']:.3f}), QPEHIP {out[f'e2e_{tag}_qpehip_total_s']:.3f} s (init {out[f'e2e_{tag}_qpehip_init_s']:.3f} + queries "
                f"{out[f'e2e_{tag}_qpehip_query_s']:.3f}); stdout identical: {out[f'e2e_{tag}_stdout_identical']}")
    out["e2e_sample"] = ("seeded CSV (scripts/make_csv.py) + the reference's sample-queries.txt through both drivers on this box: the reference's QPESeq "
                         "(oracle/_ref, 1 core) and QPEHIP (1 GPU); init / query / total as the drivers print them, wall = process wall clock")
    return out


def engine_shape_leg(pq, chain, rows, seed, copies, steps=100, warmup=20, in_flight=3):
    """One more query shape through the engine API as `value` is measured (extras: Q_A / Q_B / Q_C beside S1)."""
    B = pq.bench_lib()
    engines = [pq.HipEngine.synthetic(rows, seed=seed) for _ in range(copies)]
    try:
        arr = (C.POINTER(pq.EngineS) * copies)(*[e.e for e in engines])
        wl = pq.WhereList(chain)
        secs = []
        res = pq.BenchResult()
        for r in range(3):
            res = pq.BenchResult()
            if B.hipEngineBench(arr, copies, wl.ptr, 0, 1, in_flight, warmup if r == 0 else 2, steps, C.byref(res)) != 0 or res.mismatches:
                raise pq.PqpsError("hipEngineBench failed")
            secs.append(res.seconds)
        med = sorted(secs)[1]
        return {"ms_per_query": med / steps * 1e3, "rows_per_s": rows * steps / med, "matches": int(res.matches)}
    finally:
        for e in engines:
            e.close()


def pmc_traffic(query, rows):
    """HBM bytes per launch of the query's kernel from the committed rocprofv3 PMC summary of
    this same workload (profiles/rNN_<query>_<rows>_pmc.json, made by scripts/profile_pmc.sh +
    scripts/summarize_profiles.py: separate --pmc passes, FETCH_SIZE x 1024 x 2 for the gfx950
    half-count of wide coalesced reads, WRITE_SIZE x 1024).  None when no profile matches."""
    tag = {100_000_000: "100m", 1_000_000_000: "1b"}.get(rows)
    if tag is None:
        return None, None
    files = sorted((ROOT / "profiles").glob(f"r*_{query.lower().replace('_', '')}_{tag}_pmc.json"))
    if not files:
        return None, None
    d = json.loads(files[-1].read_text())
    k1 = [v for k, v in d.items() if k.startswith("scan_eval_") and "<0," in k]        # the ID-output launch
    if not k1 or "hbm_read_bytes_per_launch" not in k1[0]:
        return None, None
    return k1[0]["hbm_read_bytes_per_launch"] + k1[0].get("hbm_write_bytes_per_launch", 0.0), files[-1].name


def _leaves(chain):
    for x in chain[0::2]:
        if isinstance(x, list):
            yield from _leaves(x)
        else:
            yield x


def time_query(pq, L, ctx, tables, chain, mode, count, start, ids, cnt, sptr, torch, reps=20, stream=None):
    """One query shape, one launch at a time on one stream, HIP events on the dispatch; consecutive launches
    alternate between the table copies.  -> dict (whole-query numbers; COUNT(*): scan + 1-workgroup reduction)."""
    bound = [t.bind(chain) for t in tables]
    bpr = bound[0][3]

    def run(i):
        pred, cols, nc, _ = bound[i % len(bound)]
        if mode == "ids":
            pq.check(L.pqps_filter_scan(ctx.h, cols, nc, count, start, C.byref(pred), ids.data_ptr(), ids.numel(), cnt.data_ptr(), sptr))
        else:
            pq.check(L.pqps_filter_count(ctx.h, cols, nc, count, C.byref(pred), cnt.data_ptr(), sptr))
    for i in range(4):
        run(i)
    torch.cuda.synchronize()
    ctx.set_timing(True)
    for i in range(reps):
        run(i)
    ms, pipe, k = ctx.kernel_time()
    ctx.set_timing(False)
    matches = int(cnt[0].item())
    byts = count * bpr + (4 * matches if mode == "ids" else 8)
    out = {"rows_per_s": count / (pipe / k * 1e-3), "matches": matches, "bytes_per_row": bpr,
           "GBps": byts / (pipe / k * 1e-3) / 1e9, "frac_of_8TBps": byts / (pipe / k * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "scan_kernel_ms": ms / k, "avg_query_ms": pipe / k}
    if mode == "ids" and stream is not None:
        # the same queries as `value` is produced: a stream, two in flight (wall clock over the batch, results left on the device)
        qs, ring = stream
        n_q = max(2 * reps, 20)
        for i in range(len(ring)):
            pred, cols, nc, _ = bound[i % len(bound)]
            pq.check(L.pqps_qstream_scan(qs, cols, nc, count, start, C.byref(pred), ring[i][0].data_ptr(), ring[i][0].numel(), ring[i][1].data_ptr(), sptr))
        pq.check(L.pqps_qstream_sync(qs), "pqps_qstream_sync")
        torch.cuda.synchronize()
        L.pqps_qstream_hint_answer(qs, matches, count)                 # (as the engine does when it has awaited a query: dense answers on one lane)
        t0 = time.perf_counter()
        for i in range(n_q):
            pred, cols, nc, _ = bound[i % len(bound)]
            b = ring[i % len(ring)]
            pq.check(L.pqps_qstream_scan(qs, cols, nc, count, start, C.byref(pred), b[0].data_ptr(), b[0].numel(), b[1].data_ptr(), sptr))
        pq.check(L.pqps_qstream_sync(qs), "pqps_qstream_sync")
        dt = (time.perf_counter() - t0) / n_q
        assert int(ring[(n_q - 1) % len(ring)][1][0].item()) == matches
        out["stream_ms_per_query"] = dt * 1e3
        out["stream_frac_of_8TBps"] = byts / dt / 1e9 / HBM_PEAK_GBPS
    return out


def extras_leg(pq, L, ctx, tables, count, start, sptr, torch, device, names, log, alloc, seed):
    """Untimed-by-the-contract side measurements: the other query shapes of SURVEY 8(d) as ID list and as
    COUNT(*) (whole query, (n * bytes/row + 4 * matches) / time), index probes (configs[2]), device
    projection, the PCIe-inclusive rate, and the north-star size: S1 / Q_A / Q_B at 1 G rows on this GPU."""
    out = {}
    table = tables[0]
    ids = torch.empty(max(count, 1), dtype=torch.int32, device=device)
    cnt = torch.zeros(2, dtype=torch.int64, device=device)
    reps = 20
    # a query stream of its own for the "as a stream" figures: three result buffers (pqps_qstream keeps two queries in flight)
    qs = C.c_void_p()
    pq.check(L.pqps_qstream_create(ctx.h, 3, C.byref(qs)), "pqps_qstream_create")
    ring = [(ids, cnt)] + [(torch.empty(max(count, 1), dtype=torch.int32, device=device), torch.zeros(2, dtype=torch.int64, device=device))
                           for _ in range(2)]
    for name in names:
        chain, sql = QUERIES[name]
        for mode in ("ids", "count"):
            r = time_query(pq, L, ctx, tables, chain, mode, count, start, ids, cnt, sptr, torch, reps, stream=(qs, ring))
            r["query"] = sql
            out[f"{name}_{mode}"] = r
            log(f"{name:>5} {mode:>5}: {r['avg_query_ms'] * 1e3:7.1f} us  {r['rows_per_s'] / 1e9:7.1f} G rows/s  {r['GBps']:5.0f} GB/s "
                f"({100 * r['frac_of_8TBps']:.1f} % of 8 TB/s), {r['matches']:,} matches"
                + (f"; as a stream {r['stream_ms_per_query'] * 1e3:.1f} us per query ({100 * r['stream_frac_of_8TBps']:.1f} %)" if "stream_ms_per_query" in r else ""))
    del ring[1:]
    torch.cuda.empty_cache()
    # ---- configs[2]: index range-probe SELECT (sorted-permutation index = B+-tree leaf order) --------
    # bytes per SURVEY 8(d): slice_len * (4 [perm] + sum of gathered predicate column widths) + 4 * matches
    have = set(table.ptr)
    # (key in the record, indexed column, key kind, label, WHERE, probe window); the first four are the probed comparison itself (the
    # probe's rows are copied), the last two carry a second condition (the gather filter evaluates the WHERE on the probe's rows)
    specs = [("index_command_id", "command_id", 0, "command_id >= n-1e6", [("command_id", ">=", str(start + count - 1_000_000))], start + count - 1_000_000, 2**64 - 1),
             ("index_user_id", "user_id", 1, "user_id = 1001", [("user_id", "=", "1001")], 1001, 1001),
             ("index_risk_level", "risk_level", 1, "risk_level = 5", [("risk_level", "=", "5")], 5, 5),
             ("index_risk_level_gt3", "risk_level", 1, "risk_level > 3 (the reference's Sample 3)", [("risk_level", ">", "3")], 4, 2**31 - 1),
             ("index_user_id_and_sudo", "user_id", 1, "user_id = 1001 AND sudo_used = TRUE", [("user_id", "=", "1001"), "AND", ("sudo_used", "=", "TRUE")], 1001, 1001),
             ("index_risk_level_gt3_and_sudo", "risk_level", 1, "sudo_used = TRUE AND risk_level > 3", [("sudo_used", "=", "TRUE"), "AND", ("risk_level", ">", "3")], 4, 2**31 - 1)]
    rng = torch.zeros(4, dtype=torch.int64, device=device)
    for key, col, kind, label, chain, lo, hi in specs:
        if not {leaf[0] for leaf in _leaves(chain)} <= have:
            continue
        w = table.width[col]
        perm = torch.empty(count, dtype=torch.int32, device=device)
        keys = torch.empty(count * w, dtype=torch.uint8, device=device)
        carr = pq.column_array([(table.ptr[col], w)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pq.check(L.pqps_index_build(ctx.h, carr, count, kind, perm.data_ptr(), keys.data_ptr(), sptr), "pqps_index_build")
        torch.cuda.synchronize()
        build_ms = (time.perf_counter() - t0) * 1e3
        pred, cols, nc, bpr = table.bind(chain)

        def probe():
            cnt.zero_()
            # one call per probe, as the engine issues it (engine/hip/executeEngine-hip.c): the probe, then its rows that pass the WHERE
            # appended -- copied when the WHERE is the probed comparison itself, evaluated by the gather filter otherwise
            pq.check(L.pqps_index_select(ctx.h, cols, nc, carr, perm.data_ptr(), keys.data_ptr(), kind, count, lo & (2**64 - 1), hi & (2**64 - 1),
                                         start, C.byref(pred), rng.data_ptr(), ids.data_ptr(), count, cnt.data_ptr(), sptr), "index select")
        for _ in range(3):
            probe()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            probe()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        slice_len = int(rng[1].item() - rng[0].item())
        matches = int(cnt[0].item())
        byts = slice_len * (4 + bpr) + 4 * matches
        out[key] = {"query": label, "index_build_ms": build_ms, "ms_per_query": ms, "slice_len": slice_len,
                    "matches": matches, "GBps": byts / (ms * 1e-3) / 1e9, "rows_per_s_table": count / (ms * 1e-3),
                    "kernel": pq.lib().pqps_last_kernel().decode()}
        log(f"index {label}: build {build_ms:.1f} ms, probe+filter {ms * 1e3:.0f} us, slice {slice_len:,}, {matches:,} matches")
        del perm, keys
    # ---- projection on the device: the selected columns of the result rows, gathered by ID ----
    if "Q_A" in names and {"command_id", "user_id", "risk_level"} <= have:
        chain, sql = QUERIES["Q_A"]
        pred, cols, nc, bpr = table.bind(chain)
        pq.check(L.pqps_filter_scan(ctx.h, cols, nc, count, start, C.byref(pred), ids.data_ptr(), count, cnt.data_ptr(), sptr))
        torch.cuda.synchronize()
        matches = int(cnt[0].item())
        proj_cols = ["command_id", "user_id", "risk_level"]
        outs = [torch.empty(max(matches, 1) * table.width[c], dtype=torch.uint8, device=device) for c in proj_cols]
        carrs = [pq.column_array([(table.ptr[c], table.width[c])]) for c in proj_cols]

        def project():
            for c, carr, o in zip(proj_cols, carrs, outs):
                pq.check(L.pqps_project_column(ctx.h, carr, ids.data_ptr(), cnt.data_ptr(), matches, start, o.data_ptr(), sptr), "pqps_project_column")
        for _ in range(3):
            project()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            project()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out_bytes = matches * sum(table.width[c] for c in proj_cols)
        out["Q_A_project_3_columns"] = {"query": sql, "columns": proj_cols, "rows": matches, "ms": ms,
                                        "rows_per_s": matches / (ms * 1e-3), "output_GBps": out_bytes / (ms * 1e-3) / 1e9}
        log(f"projection of {proj_cols} for {matches:,} result rows: {ms * 1e3:.0f} us")
        del outs
    # PCIe-inclusive, one blocking query at a time (what a host caller of the shim sees): scan, wait,
    # read the count back, download the IDs into host memory.  Never `value`.
    import numpy as np
    main_q = [k for k in QUERIES if k not in names][0]
    for name in [main_q] + [n for n in names if n == "Q_A"]:
        chain, sql = QUERIES[name]
        pred, cols, nc, bpr = table.bind(chain)
        host_ids = np.empty(max(count, 1), dtype=np.uint32)
        k_reps, m, got = 20, 0, C.c_uint64()
        for i in range(k_reps + 3):
            if i == 3:
                t0 = time.perf_counter()
            pq.check(L.pqps_filter_scan(ctx.h, cols, nc, count, start, C.byref(pred), ids.data_ptr(), count, cnt.data_ptr(), sptr))
            ctx.download(C.byref(got), cnt.data_ptr(), 8, sptr)            # waits for the scan
            m = got.value
            if m:
                ctx.download(host_ids.ctypes.data, ids.data_ptr(), 4 * m, sptr)
        dt = (time.perf_counter() - t0) / k_reps
        out[f"{name}_ids_to_host"] = {"query": sql, "matches": m, "ms_per_query": dt * 1e3, "rows_per_s": count / dt,
                                      "note": "blocking: scan + sync + count readback + ID download (pageable host memory)"}
        log(f"{name:>4} ids -> host: {dt * 1e6:.0f} us per query, {count / dt / 1e9:.1f} G rows/s (PCIe-inclusive)")
    # ---- the other shapes through the ENGINE API, as `value` is measured for the headline query (tickets from a C loop) ----
    del ids
    del ring
    torch.cuda.empty_cache()
    for name in ("Q_A", "Q_B", "Q_C"):
        if name in names:
            try:
                r = engine_shape_leg(pq, QUERIES[name][0], count, seed, 2)
                r["query"] = QUERIES[name][1]
                out[f"{name}_engine"] = r
                log(f"{name:>5} engine level: {r['ms_per_query'] * 1e3:.1f} us per query, {r['rows_per_s'] / 1e12:.3f} T rows/s")
            except Exception as e:                               # noqa: BLE001
                log(f"{name} engine-level leg failed: {e!r}")
    # ---- the north-star size on this one GPU: 1 G rows (3 - 12 GB per query: far beyond the Infinity Cache, one copy) ----
    torch.cuda.empty_cache()
    n1b = 1_000_000_000
    big = pq.SyntheticTable(ctx, n1b, seed=seed, row0=0, columns=["sudo_used", "user_name", "risk_level", "exit_code", "user_id"], alloc=alloc, stream=sptr)
    ids = torch.empty(n1b // 8, dtype=torch.int32, device=device)
    ring = [(ids, cnt)] + [(torch.empty(n1b // 8, dtype=torch.int32, device=device), torch.zeros(2, dtype=torch.int64, device=device))
                           for _ in range(2)]
    ns = {"rows": n1b}
    for name in ("S1", "Q_A", "Q_B", "Q_C", "Q_u8", "Q_u16", "Q_r2", "Q_r1"):
        chain, sql = QUERIES[name]
        if name == "Q_r2":                                                  # dense answers (135 M and 430 M IDs): one buffer of n / 2 entries,
            del ring, ids                                                   # no stream leg (one lane at this size: stream = single)
            torch.cuda.empty_cache()
            ids = torch.empty(n1b // 2, dtype=torch.int32, device=device)
            ring = None
        for mode in ("ids", "count"):
            r = time_query(pq, L, ctx, [big], chain, mode, n1b, 0, ids, cnt, sptr, torch, reps=10,
                           stream=(qs, ring) if ring is not None else None)
            r["query"] = sql
            ns[f"{name}_{mode}"] = r
            log(f"1 G rows {name:>4} {mode:>5}: {r['avg_query_ms'] * 1e3:7.1f} us  {r['rows_per_s'] / 1e12:5.2f} T rows/s  "
                f"({100 * r['frac_of_8TBps']:.1f} % of 8 TB/s whole query)"
                + (f"; as a stream {r['stream_ms_per_query'] * 1e3:.1f} us ({100 * r['stream_frac_of_8TBps']:.1f} %)" if "stream_ms_per_query" in r else ""))
    out["north_star_1b"] = ns
    pq.check(L.pqps_qstream_destroy(qs), "pqps_qstream_destroy")
    return out


if __name__ == "__main__":
    main()
