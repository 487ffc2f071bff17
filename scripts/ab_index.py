#!/usr/bin/env python3
"""Device time of one index-mode query (probe + order-preserving gather filter) through the C-ABI.

    python scripts/ab_index.py --rows 100000000 --column user_id --lo 1001 --hi 1001 [--reps 20]
"""
import argparse
import ctypes as C
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--column", default="user_id")
    ap.add_argument("--lo", type=int, default=1001)
    ap.add_argument("--hi", type=int, default=1001)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--select", action="store_true", help="one pqps_index_select per query (copies when the WHERE is the probed comparison) instead of probe + gather")
    args = ap.parse_args()
    pq, _ = bench.load_pkg()
    L = pq.lib()
    ctx = pq.Context(0)
    n = args.rows
    table = pq.SyntheticTable(ctx, n, seed=0x5EED, columns=[args.column])
    w = table.width[args.column]
    kind = 0 if args.column == "command_id" else 1
    perm, keys = ctx.malloc(4 * n), ctx.malloc(w * n)
    ids, cnt, rng = ctx.malloc(4 * n), ctx.malloc(64), ctx.malloc(64)
    carr = pq.column_array([(table.ptr[args.column], w)])
    pq.check(L.pqps_index_build(ctx.h, carr, n, kind, perm, keys, None), "build")
    ctx.sync()
    top = 2**63 - 1 if kind == 0 else 2**31 - 1
    if args.lo == args.hi:                                       # ONE comparison = the probe's own window: the engine's call copies
        chain = [(args.column, "=", str(args.lo))]
    elif args.hi >= top:
        chain = [(args.column, ">=", str(args.lo))]
    else:
        chain = [(args.column, ">=", str(args.lo)), "AND", (args.column, "<=", str(args.hi))]
    pred, cols, nc, _ = table.bind(chain)
    mask = 2**64 - 1

    def query():
        ctx.memset(cnt, 0, 8)
        if args.select:                                          # what the engine issues per probe (include/pqps_hip.h)
            pq.check(L.pqps_index_select(ctx.h, cols, nc, carr, perm, keys, kind, n, args.lo & mask, args.hi & mask, 0, C.byref(pred), rng,
                                         ids, n, cnt, None), "index select")
            return
        pq.check(L.pqps_index_probe(ctx.h, keys, w, kind, n, args.lo & mask, args.hi & mask, rng, None), "probe")
        pq.check(L.pqps_filter_gather(ctx.h, cols, nc, perm, rng, n, 0, C.byref(pred), ids, n, cnt, None), "gather")

    for _ in range(3):
        query()
    ctx.sync()
    ctx.set_timing(True)
    for _ in range(args.reps):
        query()
    ev, tot, k = ctx.kernel_time()
    ctx.set_timing(False)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        query()
    ctx.sync()
    wall = (time.perf_counter() - t0) / args.reps * 1e6
    m = C.c_uint64()
    ctx.download(C.byref(m), cnt, 8)
    what = f"pqps_index_select ({L.pqps_last_kernel().decode()[:24]})" if args.select else "probe + gather"
    print(f"{args.column} in [{args.lo}, {args.hi}] of {n:,} rows: {m.value:,} matches; " + (f"gather launch {tot / k * 1e3:.1f} us; " if k else "")
          + f"memset + {what} back to back {wall:.1f} us per query", flush=True)


if __name__ == "__main__":
    main()
