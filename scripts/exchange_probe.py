#!/usr/bin/env python3
"""What the exchange's OWN kernels cost a query (not RCCL's): a world of N ranks as threads of this process on one GPU, the
product's exchange between them through tests/loopback/libloopback_rccl.so, a stream of SELECTs per rank.  Run it under
rocprofv3 --kernel-trace --stats: wire_pack_kernel, eager_unpack_kernel / wire_expand_many_kernel and the scan launches show
with their durations (the loopback's copies stand where RCCL's kernels would).

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 scripts/exchange_probe.py --world 8 --rows 100000000 --query S1
"""
import argparse
import ctypes as C
import pathlib
import sys
import threading
import time
import traceback

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rows", type=int, default=100_000_000, help="rows of the whole table (sharded over the ranks)")
    ap.add_argument("--query", default="S1")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--ring", type=int, default=6)
    args = ap.parse_args()
    pq, mg = bench.load_pkg()
    L = pq.lib()
    path = str(ROOT / "tests" / "loopback" / "libloopback_rccl.so").encode()
    world, n = args.world, args.rows
    chain = bench.QUERIES[args.query][0]
    needed = sorted({leaf[0] for leaf in bench._leaves(chain)})
    gate = threading.Barrier(world)
    ident = [None]
    report = [None] * world
    failed = []

    def rank_main(rank):
        try:
            ctx = pq.Context(0)
            start, count = mg.shard_rows(n, world, rank)
            dev = pq.SyntheticTable(ctx, count, seed=0x5EED, row0=start, columns=needed)
            if rank == 0:
                buf = C.create_string_buffer(128)
                pq.check(L.pqps_exchange_unique_id(path, buf), "unique id")
                ident[0] = buf.raw
            h = C.c_void_p()
            pq.check(L.pqps_exchange_prepare(ctx.h, path, world, rank, count + 16, args.ring, C.byref(h)), "prepare")
            gate.wait()
            pq.check(L.pqps_exchange_connect(h, C.create_string_buffer(ident[0], 128)), "connect")
            xch = mg.ShardExchange(pq, ctx, h, world, rank, args.ring)
            pred, cols, nc, _ = dev.bind(chain)
            for k in range(6):                                               # warm-up
                xch.select(cols, nc, count, start, C.byref(pred), k % args.ring, None)
            xch.sync()
            gate.wait()
            t0 = time.monotonic()
            for k in range(args.steps):
                xch.select(cols, nc, count, start, C.byref(pred), k % args.ring, None)
            xch.sync()
            dt = time.monotonic() - t0
            arr, local = xch.result((args.steps - 1) % args.ring)
            eg, wb = (C.c_uint64 * 3)(), (C.c_uint64 * 2)()
            L.pqps_exchange_eager(h, eg, 0)
            L.pqps_exchange_wire_bytes(h, wb, 0)
            report[rank] = dict(us_per_query=dt / args.steps * 1e6, ids=len(arr), own=local, eager=list(eg), wire=list(wb))
            gate.wait()
            xch.close()
            dev.free()
            ctx.close()
        except BaseException:
            traceback.print_exc()
            failed.append(rank)
            import os
            os._exit(3)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for r, rep in enumerate(report):
        print(f"rank {r}: {rep}", flush=True)


if __name__ == "__main__":
    main()
