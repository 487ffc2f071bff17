#!/usr/bin/env python3
"""Where the +-5 % between processes at 1 G rows comes from: ONE process, ONE build, and the three things a launch touches
placed anew one at a time -- the ID buffer (offsets into one allocation, then fresh allocations), the context's scratch
(slots, count words, sums: a fresh context), the table (a fresh copy).  Device time from the HIP events on the dispatch.

    python scripts/ab_placement.py --rows 1000000000 --queries Q_A,Q_B
"""
import argparse
import ctypes as C
import pathlib
import statistics
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--queries", default="Q_A,Q_B")
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--contexts", type=int, default=4)
    ap.add_argument("--tables", type=int, default=3)
    ap.add_argument("--buffers", type=int, default=4)
    args = ap.parse_args()
    pq, _ = bench.load_pkg()
    L = pq.lib()
    ctx0 = pq.Context(0)
    names = args.queries.split(",")
    needed = {leaf[0] for k in names for leaf in bench._leaves(bench.QUERIES[k][0])}
    n = args.rows
    cap = max(n // 4, 1024)
    slack = 64 << 20
    tables = [pq.SyntheticTable(ctx0, n, seed=0x5EED, columns=sorted(needed))]
    bufs = [ctx0.malloc(4 * cap + slack)]
    cnt = ctx0.malloc(64)
    ctxs = [ctx0]

    def measure(ctx, table, qname, ids_ptr):
        pred, cols, nc, bpr = table.bind(bench.QUERIES[qname][0])
        for _ in range(2):
            pq.check(L.pqps_filter_scan(ctx.h, cols, nc, n, 0, C.byref(pred), ids_ptr, cap, cnt, None), "scan")
        ctx.sync()
        ts = []
        for _ in range(args.reps):
            ctx.set_timing(True)
            for _ in range(3):
                pq.check(L.pqps_filter_scan(ctx.h, cols, nc, n, 0, C.byref(pred), ids_ptr, cap, cnt, None), "scan")
            kern_ms, _, k = ctx.kernel_time()
            ctx.set_timing(False)
            ts.append(kern_ms / k * 1e3)
        return statistics.median(ts), min(ts), max(ts)

    def show(tag, qname, t):
        print(f"{qname:4s} {tag:44s} {t[0]:8.1f} us [{t[1]:.1f}..{t[2]:.1f}]", flush=True)

    for qname in names:
        print(f"--- {qname}, {n:,} rows", flush=True)
        base = int(bufs[0])
        for off in (0, 4096, 65536, 1 << 20, (2 << 20) + 4096, 16 << 20, (32 << 20) + 128):
            show(f"ID buffer + {off:,} bytes", qname, measure(ctx0, tables[0], qname, C.c_void_p(base + off)))
    while len(bufs) < args.buffers:
        bufs.append(ctx0.malloc(4 * cap + slack))
    for qname in names:
        for i, b in enumerate(bufs):
            show(f"ID buffer allocation {i}", qname, measure(ctx0, tables[0], qname, b))
    while len(ctxs) < args.contexts:
        ctxs.append(pq.Context(0))
    for qname in names:
        for i, c in enumerate(ctxs):
            show(f"context {i} (its own scratch)", qname, measure(c, tables[0], qname, bufs[0]))
    while len(tables) < args.tables:
        tables.append(pq.SyntheticTable(ctx0, n, seed=0x5EED, columns=sorted(needed)))
    for qname in names:
        for i, t in enumerate(tables):
            show(f"table copy {i}", qname, measure(ctx0, t, qname, bufs[0]))


if __name__ == "__main__":
    main()
