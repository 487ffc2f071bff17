#!/usr/bin/env python3
"""Soak test of the query stream: thousands of queries of mixed shapes back to back through
pqps_qstream_scan, every result compared with the oracle's.  usage: python tests/soak_qstream.py [queries]   (manual; not collected by pytest)"""
import ctypes as C
import pathlib
import random
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))       # the oracle is the checker here, as in the test suite
import qpelib as q  # noqa: E402

pq = q.pq
QUERIES = {
    "S1": [("sudo_used", "=", "FALSE"), "AND", ("user_name", "=", "student1030")],
    "Q_A": [("risk_level", ">", "3")],
    "Q_B": [("sudo_used", "=", "TRUE"), "AND", ("risk_level", ">", "2")],
    "Q_C": [("exit_code", "!=", "0"), "AND", ("user_id", ">=", "1500"), "OR", ("risk_level", "=", "5")],
    "u8": [("sudo_used", "=", "TRUE")],
    "none": [("risk_level", ">", "9")],
    "S7": [("sudo_used", "=", "TRUE"), "OR", [("risk_level", "=", "5"), "AND", ("shell_type", "=", "bash")]],
    "dense": [("risk_level", ">", "1")],                      # 43 %: 16-bit row lists in every step
    "u16": [("user_name", "=", "student1030")],
}


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    n, depth, nbuf = 3_000_017, 4, 12
    L = pq.lib()
    ctx = pq.Context(0)
    host = q.HostSynth(n, seed=99)
    dev = pq.SyntheticTable(ctx, n, seed=99)
    want = {k: host.oracle_scan(c) for k, c in QUERIES.items()}
    bound = {k: dev.bind(c) for k, c in QUERIES.items()}
    qs = C.c_void_p()
    pq.check(L.pqps_qstream_create(ctx.h, depth, C.byref(qs)))
    ids = [ctx.malloc(4 * n) for _ in range(nbuf)]
    cnt = [ctx.malloc(64) for _ in range(nbuf)]
    rng = random.Random(5)
    names = list(QUERIES)
    done, t0 = 0, time.time()
    while done < total:
        batch = [rng.choice(names) for _ in range(nbuf)]
        for j, name in enumerate(batch):
            pred, cols, nc, _ = bound[name]
            pq.check(L.pqps_qstream_scan(qs, cols, nc, n, 0, C.byref(pred), ids[j], n, cnt[j], None), name)
        pq.check(L.pqps_qstream_sync(qs))
        ctx.sync()
        for j, name in enumerate(batch):
            k = C.c_uint64()
            ctx.download(C.byref(k), cnt[j], 8)
            got = np.zeros(max(k.value, 1), dtype=np.uint32)
            if k.value:
                ctx.download(got.ctypes.data, ids[j], 4 * k.value)
            if k.value != len(want[name]) or not np.array_equal(got[:k.value], want[name]):
                sys.exit(f"MISMATCH at query {done + j} ({name}): {k.value} vs {len(want[name])}")
        done += nbuf
        if done % 1200 == 0:
            print(f"{done} queries ok ({time.time() - t0:.0f} s)", flush=True)
    print(f"soak ok: {done} queries, all equal to the oracle")


if __name__ == "__main__":
    main()
