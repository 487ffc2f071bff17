#!/usr/bin/env python3
"""Experiment: a stream of independent ID queries issued alternately on TWO contexts (each its own HIP stream and
scratch), optionally at different stream priorities, against the same queries on one context.

    python scripts/ab_two_streams.py --rows 100000000 --query S1 --reps 200 [--prio -1,0]

Wall time per query over the whole batch (inputs resident, results left on the device)."""
import argparse
import ctypes as C
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--query", default="S1")
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--prio", default="", help="priorities of the two streams, e.g. -1,0 (empty: default streams)")
    ap.add_argument("--copies", type=int, default=2)
    ap.add_argument("--count", action="store_true", help="COUNT(*) instead of ID lists (the one- / N-context legs)")
    ap.add_argument("--ring", type=int, default=6, help="result slots of the exchange")
    ap.add_argument("--exchange", action="store_true", help="also time pqps_exchange_select with a world of one")
    ap.add_argument("--streams", type=int, default=2, help="contexts (streams) the queries go round")
    args = ap.parse_args()
    pq, _ = bench.load_pkg()
    L = pq.lib()
    L.pqps_exchange_wait_ns.restype = C.c_uint64
    prios = args.prio.split(",") if args.prio else [None] * args.streams
    ctxs = []
    for p in prios:
        if p is None:
            os.environ.pop("PQPS_STREAM_PRIORITY", None)
        else:
            os.environ["PQPS_STREAM_PRIORITY"] = p
        ctxs.append(pq.Context(0))
    os.environ.pop("PQPS_STREAM_PRIORITY", None)
    chain, _sql = bench.QUERIES[args.query]
    needed = sorted({leaf[0] for leaf in bench._leaves(chain)})
    tables = [pq.SyntheticTable(ctxs[0], args.rows, seed=0x5EED, columns=needed) for _ in range(args.copies)]
    ctxs[0].sync()
    n = args.rows
    outs = [(c.malloc(4 * max(n // 2, 1024)), c.malloc(64)) for c in ctxs]
    bound = [t.bind(chain) for t in tables]

    def run(k, which):
        pred, cols, nc, _ = bound[k % len(bound)]
        c = ctxs[which]
        ids, cnt = outs[which]
        if args.count:
            pq.check(L.pqps_filter_count(c.h, cols, nc, n, C.byref(pred), cnt, None))
        else:
            pq.check(L.pqps_filter_scan(c.h, cols, nc, n, 0, C.byref(pred), ids, n // 2, cnt, None))

    # the same through the shim's query stream (pqps_qstream_scan: two lanes, completion events on the dispatch packets)
    qs = C.c_void_p()
    pq.check(L.pqps_qstream_create(ctxs[0].h, 4, C.byref(qs)), "pqps_qstream_create")
    ring = [(ctxs[0].malloc(4 * max(n // 2, 1024)), ctxs[0].malloc(64)) for _ in range(4)]
    for rep in range(2):
        t0 = time.perf_counter()
        for k in range(args.reps):
            pred, cols, nc, _ = bound[k % len(bound)]
            ids, cnt = ring[k % 4]
            pq.check(L.pqps_qstream_scan(qs, cols, nc, n, 0, C.byref(pred), ids, n // 2, cnt, None))
        pq.check(L.pqps_qstream_sync(qs))
        dt = (time.perf_counter() - t0) / args.reps * 1e6
    print(f"[{args.query} rows={n:,}] pqps_qstream: {dt:7.1f} us per query", flush=True)
    # ... and through the exchange with a world of one (the N > 1 path: sizes to the host, payload held back one query)
    if args.exchange:
        mg = bench.load_pkg()[1]
        xch = mg.ShardExchange.open(pq, ctxs[0], None, None, 1, 0, n, ring=args.ring)
        for rep in range(2):
            t0 = time.perf_counter()
            for k in range(args.reps):
                pred, cols, nc, _ = bound[k % len(bound)]
                xch.select(cols, nc, n, 0, C.byref(pred), k % args.ring, None)
            xch.sync()
            dt = (time.perf_counter() - t0) / args.reps * 1e6
            waited = L.pqps_exchange_wait_ns(xch.h, 1) * 1e-3 / args.reps
        print(f"[{args.query} rows={n:,}] pqps_exchange (world 1): {dt:7.1f} us per query, of which the host waited {waited:.1f} us "
              f"(for a free slot or for sizes)", flush=True)
        xch.close()
    for mode in ("one", "two"):
        for k in range(10):
            run(k, k % len(ctxs) if mode == "two" else 0)
        for c in ctxs:
            c.sync()
        t0 = time.perf_counter()
        for k in range(args.reps):
            run(k, k % len(ctxs) if mode == "two" else 0)
        for c in ctxs:
            c.sync()
        dt = (time.perf_counter() - t0) / args.reps * 1e6
        ns = 1 if mode == "one" else len(ctxs)
        m = C.c_uint64()
        ctxs[0].download(C.byref(m), outs[0][1], 8)
        print(f"[{args.query} rows={n:,} prio={args.prio or 'default'}] {ns} stream(s): {dt:7.1f} us per query ({m.value} matches)", flush=True)


if __name__ == "__main__":
    main()
