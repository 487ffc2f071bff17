#!/usr/bin/env python3
"""Same-process A/B of several builds of the shim library: ONE table in device memory, every build scans the very same
buffers, launches interleaved (A B C A B C ...), device time from the HIP events on the dispatch.  Takes the physical
placement of the columns -- which moves a 1 G-row figure by +-5 % from process to process -- out of the comparison.

    python scripts/ab_libs.py --rows 1000000000 --queries S1,Q_A,Q_B --libs r02=scripts/_ab_r02/parallel-query-processing-system_amd/libpqps_hip.so,head=parallel-query-processing-system_amd/libpqps_hip.so

The table is generated (and the predicates are bound) by the in-tree library; the other builds only run pqps_filter_scan /
pqps_filter_count on it (pqps_column / pqps_predicate have not changed since round 1)."""
import argparse
import ctypes as C
import pathlib
import statistics
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def bind(path, pq):
    L = C.CDLL(str(path))
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    L.pqps_last_error.restype = C.c_char_p
    L.pqps_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.pqps_ctx_destroy.argtypes = [vp]
    L.pqps_ctx_destroy.restype = None
    L.pqps_ctx_sync.argtypes = [vp, vp]
    L.pqps_ctx_set_timing.argtypes = [vp, C.c_int]
    L.pqps_ctx_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.pqps_filter_scan.argtypes = [vp, C.POINTER(pq.Column), u32, u64, u32, C.POINTER(pq.Predicate), vp, u64, vp, vp]
    L.pqps_filter_count.argtypes = [vp, C.POINTER(pq.Column), u32, u64, C.POINTER(pq.Predicate), vp, vp]
    try:
        L.pqps_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    except AttributeError:                                       # builds before round 4
        pass
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--queries", default="S1,Q_A,Q_B")
    ap.add_argument("--libs", required=True, help="name=path,name=path,...")
    ap.add_argument("--rounds", type=int, default=6, help="rounds of (every build x `reps` launches)")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--count", action="store_true")
    ap.add_argument("--ctxs", type=int, default=1, help="contexts per build (each draws its own scratch placement: name#k; the line also gives the mean over them)")
    ap.add_argument("--copies", type=int, default=1, help="table copies the launches alternate between (2: nothing a launch reads is cache-resident)")
    args = ap.parse_args()
    pq, _ = bench.load_pkg()
    ctx = pq.Context(0)
    names = args.queries.split(",")
    needed = {leaf[0] for k in names for leaf in bench._leaves(bench.QUERIES[k][0])}
    tables = [pq.SyntheticTable(ctx, args.rows, seed=0x5EED, columns=sorted(needed)) for _ in range(max(1, args.copies))]
    n = args.rows
    ids = ctx.malloc(4 * max(n // 2, 1024))
    cnt = ctx.malloc(64)
    builds = []
    for spec in args.libs.split(","):                                # name=path[@option:value[@option:value ...]]
        name, rest = spec.split("=", 1)
        path, *opts = rest.split("@")
        L = bind(ROOT / path, pq)
        for k_ctx in range(max(1, args.ctxs)):
            h = C.c_void_p()
            if L.pqps_ctx_create(0, C.byref(h)) != 0:
                sys.exit(f"{name}: {L.pqps_last_error().decode()}")
            for o in opts:
                k, v = o.split(":")
                if L.pqps_ctx_set_option(h, k.encode(), int(v)) != 0:
                    sys.exit(f"{name}: {L.pqps_last_error().decode()}")
            builds.append((name if args.ctxs <= 1 else f"{name}#{k_ctx}", L, h))
    if args.ctxs > 1:                                                # interleave the contexts of the builds: a#0 b#0 a#1 b#1 ...
        builds.sort(key=lambda b: int(b[0].rsplit("#", 1)[1]))
    print(f"rows={n:,}  builds: {[b[0] for b in builds]}  rounds={args.rounds} x reps={args.reps}", flush=True)
    for qname in names:
        chain, _sql = bench.QUERIES[qname]
        bound = [t.bind(chain) for t in tables]
        bpr = bound[0][3]
        turn = [0]
        for mode in ["ids"] + (["count"] if args.count else []):
            def run(L, h):
                pred, cols, nc, _ = bound[turn[0] % len(bound)]
                turn[0] += 1
                if mode == "ids":
                    rc = L.pqps_filter_scan(h, cols, nc, n, 0, C.byref(pred), ids, n // 2, cnt, None)
                else:
                    rc = L.pqps_filter_count(h, cols, nc, n, C.byref(pred), cnt, None)
                if rc != 0:
                    sys.exit(L.pqps_last_error().decode())
            times = {b[0]: [] for b in builds}
            for name, L, h in builds:                                # warm-up: scratch, list area, code objects
                for _ in range(2):
                    run(L, h)
                L.pqps_ctx_sync(h, None)
            for _ in range(args.rounds):
                for name, L, h in builds:
                    L.pqps_ctx_set_timing(h, 1)
                    for _ in range(args.reps):
                        run(L, h)
                    ev, tot, k = C.c_double(), C.c_double(), C.c_int()
                    L.pqps_ctx_kernel_time(h, C.byref(ev), C.byref(tot), C.byref(k))
                    L.pqps_ctx_set_timing(h, 0)
                    times[name].append(tot.value / k.value * 1e3)
            m = C.c_uint64()
            ctx.download(C.byref(m), cnt, 8)
            byts = n * bpr + (4 * m.value if mode == "ids" else 8)
            line = f"{qname}/{mode} ({m.value:,} matches):"
            for name, _, _ in builds:
                med = statistics.median(times[name])
                line += f"  {name} {med:7.1f} us [{min(times[name]):.1f}..{max(times[name]):.1f}] {byts / (med * 1e-6) / 8e12:.3f}"
            print(line, flush=True)
            if args.ctxs > 1:
                means = {}
                for name, _, _ in builds:
                    means.setdefault(name.rsplit("#", 1)[0], []).append(statistics.median(times[name]))
                print("    mean over contexts: " + "  ".join(f"{k} {statistics.mean(v):7.1f} us {byts / (statistics.mean(v) * 1e-6) / 8e12:.3f}" for k, v in means.items()), flush=True)


if __name__ == "__main__":
    main()
