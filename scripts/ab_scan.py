#!/usr/bin/env python3
"""Device time of single queries through the C-ABI, for A/B runs of the shim's tuning switches.

    VAR=value python scripts/ab_scan.py --rows 100000000 --queries S1,Q_A --reps 30 [--count]

Prints, per query: matches, us per query (HIP events on the dispatch: the launch's own begin / end) and
(n * bytes/row + 4 * matches) / time as a fraction of 8 TB/s.  No torch: start-up is a second.
Environment switches are read once per process by the shim, hence one process per setting."""
import argparse
import ctypes as C
import importlib.util
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402  (QUERIES, load_pkg)

# selectivities between the bench's query shapes (tuning runs of the hand-over thresholds): ~1 %, ~2 %, ~5 % of the rows
bench.QUERIES.update({
    "Q_1pct": ([("risk_level", "=", "5")], "risk_level = 5"),
    "Q_2pct": ([("risk_level", "=", "5"), "OR", ("exit_code", "=", "2")], "risk_level = 5 OR exit_code = 2"),
    "Q_5pct": ([("exit_code", "!=", "0")], "exit_code != 0"),
    "Q_none": ([("risk_level", ">", "9")], "risk_level > 9"),
    "Q_01pct": ([("risk_level", "=", "5"), "AND", ("exit_code", "=", "2")], "risk_level = 5 AND exit_code = 2"),
})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--queries", default="S1,Q_A")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--count", action="store_true", help="also time COUNT(*) of the same predicates")
    ap.add_argument("--tag", default="")
    ap.add_argument("--copies", type=int, default=1, help="table copies the launches alternate between (2: nothing a launch reads is cache-resident)")
    ap.add_argument("--wall", action="store_true", help="also report wall time per query of a back-to-back stream")
    args = ap.parse_args()
    pq, _ = bench.load_pkg()
    L = pq.lib()
    ctx = pq.Context(0)
    names = args.queries.split(",")
    needed = {leaf[0] for k in names for leaf in bench._leaves(bench.QUERIES[k][0])}
    tables = [pq.SyntheticTable(ctx, args.rows, seed=0x5EED, columns=sorted(needed)) for _ in range(max(1, args.copies))]
    n = args.rows
    ids = ctx.malloc(4 * max(n // 2, 1024))
    cnt = ctx.malloc(64)
    out = []
    for name in names:
        chain, _sql = bench.QUERIES[name]
        bound = [t.bind(chain) for t in tables]
        bpr = bound[0][3]
        turn = [0]
        modes = ["ids"] + (["count"] if args.count else [])
        for mode in modes:
            def run():
                pred, cols, nc, _ = bound[turn[0] % len(bound)]
                turn[0] += 1
                if mode == "ids":
                    pq.check(L.pqps_filter_scan(ctx.h, cols, nc, n, 0, C.byref(pred), ids, n // 2, cnt, None))
                else:
                    pq.check(L.pqps_filter_count(ctx.h, cols, nc, n, C.byref(pred), cnt, None))
            for _ in range(3):
                run()
            ctx.sync()
            ctx.set_timing(True)
            for _ in range(args.reps):
                run()
            ev, tot, k = ctx.kernel_time()
            ctx.set_timing(False)
            m = C.c_uint64()
            ctx.download(C.byref(m), cnt, 8)
            us = tot / k * 1e3
            byts = n * bpr + (4 * m.value if mode == "ids" else 8)
            frac = byts / (us * 1e-6) / 8e12
            s = f"{name}/{mode}: {us:7.1f} us (scan kernel {ev / k * 1e3:.1f})  frac {frac:.3f}  ({m.value} matches)"
            if args.wall:
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(args.reps):
                    run()
                ctx.sync()
                s += f"  wall {(time.perf_counter() - t0) / args.reps * 1e6:7.1f} us"
            out.append(s)
    print(f"[{args.tag}] rows={n:,}  " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
