#!/usr/bin/env python3
"""Runs ONE parity sweep of the HIP filter in this process and exits non-zero on the first difference.

The shim reads its tuning switches (PQPS_NT_LOADS, PQPS_EXPAND_LAG, PQPS_SUM_LAG, PQPS_EXPAND_SPIN_LIMIT, ...)
once per process, so tests/test_gpu_variants.py starts this script once per setting.  Test infrastructure:
the oracle (and, for hand-built predicates, the numpy model of the kernel arithmetic) is the checker.

    python tests/variant_driver.py sizes  [n ...]     seeded synthetic tables, the suite's queries, vs the oracle
    python tests/variant_driver.py shapes [n]         every (W0, W1, W2) kernel shape, chain and tree form, vs numpy
"""
import ctypes as C
import itertools
import sys

import numpy as np

import qpelib as q
import kernel_model

pq = q.pq

QUERIES = {
    "S1": [("sudo_used", "=", "FALSE"), "AND", ("user_name", "=", "student1030")],
    "Q_A": [("risk_level", ">", "3")],
    "Q_B": [("sudo_used", "=", "TRUE"), "AND", ("risk_level", ">", "2")],
    "Q_C": [("exit_code", "!=", "0"), "AND", ("user_id", ">=", "1500"), "OR", ("risk_level", "=", "5")],
    "S7": [("sudo_used", "=", "TRUE"), "OR", [("risk_level", "=", "5"), "AND", ("shell_type", "=", "bash")]],
    "dense": [("sudo_used", "=", "FALSE")],
    "u8": [("sudo_used", "=", "TRUE")],                        # one 1-byte column, ~7 %: four steps per wave scan in the expander
    "u8_dict": [("shell_type", "=", "zsh")],
    "mid": [("risk_level", ">", "1")],                          # ~43 %: steps of ~440 matches (staged 64-row path)
    "mid_u8": [("shell_type", "!=", "bash")],
    "all": [],
    "none": [("risk_level", ">", "9")],
    "seven_leaves": [("risk_level", "=", "1"), "OR", ("risk_level", "=", "2"), "AND", ("exit_code", "=", "0"), "OR",
                     [("host_name", "<", "labpc-05"), "AND", ("shell_type", "!=", "zsh"), "AND", ("user_id", "<", "1900")],
                     "OR", ("sudo_used", "=", "1")],
}


class Out:
    def __init__(self, ctx, cap):
        self.ctx, self.cap = ctx, cap
        self.ids, self.count = ctx.malloc(max(cap, 1) * 4), ctx.malloc(64)

    def result(self):
        self.ctx.sync()                      # raises if a launch reported an incomplete result
        k = C.c_uint64()
        self.ctx.download(C.byref(k), self.count, 8)
        a = np.zeros(max(k.value, 1), dtype=np.uint32)
        if k.value:
            self.ctx.download(a.ctypes.data, self.ids, 4 * min(k.value, self.cap))
        return k.value, a[:min(k.value, self.cap)]

    def free(self):
        self.ctx.free(self.ids)
        self.ctx.free(self.count)


def sweep_sizes(ctx, sizes):
    L = pq.lib()
    for n in sizes:
        dev = pq.SyntheticTable(ctx, n, seed=0xC0FFEE)
        host = q.HostSynth(n, seed=0xC0FFEE)
        out = Out(ctx, n + 8)
        for name, chain in QUERIES.items():
            pred, cols, nc, _ = dev.bind(chain)
            for rep in range(2):             # twice: the second run finds the words the first one left behind
                pq.check(L.pqps_filter_scan(ctx.h, cols, nc, n, 0, C.byref(pred), out.ids, out.cap, out.count, None), name)
                k, got = out.result()
                want = host.oracle_scan(chain)
                if k != len(want) or not np.array_equal(got, want):
                    sys.exit(f"DIFF ids: n={n} query={name} rep={rep}: {k} vs {len(want)} matches")
            pq.check(L.pqps_filter_count(ctx.h, cols, nc, n, C.byref(pred), out.count, None), name)
            k, _ = out.result()
            if k != len(want):
                sys.exit(f"DIFF count: n={n} query={name}: {k} vs {len(want)}")
        out.free()
        dev.free()
        print(f"sizes: n={n} ok", flush=True)


SHAPES = [s for s in itertools.product((8, 4, 2, 1, 0), repeat=3)
          if s[0] != 0 and s[0] >= s[1] >= s[2] and not (s[1] == 0 and s[2] != 0)]
DT = {8: np.uint64, 4: np.uint32, 2: np.uint16, 1: np.uint8}


def sweep_shapes(ctx, n):
    """Columns of random small values (so that equality windows hit), one to two leaves per column."""
    L = pq.lib()
    rng = np.random.default_rng(1234)
    pad = (n + 4095) // 4096 * 4096
    out = Out(ctx, n + 8)
    assert len(SHAPES) == 34
    for shape in SHAPES:
        widths = [w for w in shape if w]
        arrays, ptrs = [], []
        for w in widths:
            a = rng.integers(0, 7, n).astype(DT[w])
            a[rng.integers(0, n, n // 50)] = np.iinfo(DT[w]).max       # a few extreme values
            p = ctx.malloc(pad * w)
            ctx.memset(p, 0, pad * w)
            ctx.upload(p, a.ctypes.data, a.nbytes)
            arrays.append(a)
            ptrs.append(p)
        cols = pq.column_array(list(zip(ptrs, widths)))
        for form in ("and", "or", "tree", "one"):
            pred = pq.Predicate()
            leaves = []
            for c in range(len(widths)):
                for _ in range(1 if form == "one" or len(widths) == 3 else 2):
                    lo, span = int(rng.integers(0, 5)), int(rng.integers(0, 3))
                    leaves.append((c, int(rng.integers(0, 2)), lo, span))
                if form == "one":
                    break
            k = len(leaves)
            pred.n_leaves, pred.n_columns = k, len(widths)
            for i, (c, neg, lo, span) in enumerate(leaves):
                pred.leaf[i].column, pred.leaf[i].negate, pred.leaf[i].lo, pred.leaf[i].span = c, neg, lo, span
                pred.on_true[i], pred.on_false[i], pred.order[i] = pq.ACCEPT, pq.REJECT, i
            rows = 1 << k
            if form in ("and", "one"):
                pred.truth = 1 << (rows - 1)                         # all leaves true
            elif form == "or":
                pred.truth = ((1 << rows) - 1) & ~1                  # any leaf true
            else:
                pred.truth = int(rng.integers(1, 1 << min(rows, 62))) | (1 << (rows - 1))
            want = np.nonzero(kernel_model.evaluate(pred, arrays))[0].astype(np.uint32)
            pq.check(L.pqps_filter_scan(ctx.h, cols, len(widths), n, 0, C.byref(pred), out.ids, out.cap, out.count, None), "scan")
            kk, got = out.result()
            if kk != len(want) or not np.array_equal(got, want):
                sys.exit(f"DIFF ids: shape={shape} form={form}: {kk} vs {len(want)} matches")
            pq.check(L.pqps_filter_count(ctx.h, cols, len(widths), n, C.byref(pred), out.count, None), "count")
            kk, _ = out.result()
            if kk != len(want):
                sys.exit(f"DIFF count: shape={shape} form={form}: {kk} vs {len(want)}")
        for p in ptrs:
            ctx.free(p)
    out.free()
    print(f"shapes: {len(SHAPES)} shapes x 4 forms ok at n={n}", flush=True)


def sweep_wrap(ctx, n):
    """Scan, index gather and query-stream queries in turn, many times over: with PQPS_EPOCH_START=65530 every scratch
    (the context's and both lanes') takes its epoch through 65535 and starts over at 1 (the tagged hand-off words are
    zeroed, csrc/pqps_hip.hip:run_filter) while words of the epochs before are still lying around."""
    L = pq.lib()
    dev = pq.SyntheticTable(ctx, n, seed=0xC0FFEE)
    host = q.HostSynth(n, seed=0xC0FFEE)
    out = Out(ctx, n + 8)
    names = ["S1", "Q_A", "dense", "Q_C", "none", "u8"]
    want = {k: host.oracle_scan(QUERIES[k]) for k in names}
    bound = {k: dev.bind(QUERIES[k]) for k in names}
    # an index on risk_level for the gather launches
    perm, keys, rng = ctx.malloc(4 * n), ctx.malloc(4 * n), ctx.malloc(64)
    col = pq.column_array([(dev.ptr["risk_level"], 4)])
    pq.check(L.pqps_index_build(ctx.h, col, n, 1, perm, keys, None), "index build")
    order = q.host_index_order(host.arr["risk_level"])
    full = {k: np.zeros(n, dtype=bool) for k in names}
    for k in names:
        full[k][want[k]] = True
    qs = C.c_void_p()
    pq.check(L.pqps_qstream_create(ctx.h, 3, C.byref(qs)), "qstream")
    ring = [Out(ctx, n + 8) for _ in range(3)]
    issued = []
    for it in range(40):
        k = names[it % len(names)]
        pred, cols, nc, _ = bound[k]
        # plain scan on the context
        pq.check(L.pqps_filter_scan(ctx.h, cols, nc, n, 0, C.byref(pred), out.ids, out.cap, out.count, None), k)
        kk, got = out.result()
        if kk != len(want[k]) or not np.array_equal(got, want[k]):
            sys.exit(f"DIFF scan: it={it} query={k}")
        # index gather: rows with risk_level >= 4 in leaf order, re-filtered by the query
        ctx.memset(out.count, 0, 8)
        pq.check(L.pqps_index_probe(ctx.h, keys, 4, 1, n, 4, 0x7FFFFFFF, rng, None), "probe")
        pq.check(L.pqps_filter_gather(ctx.h, cols, nc, perm, rng, n, 0, C.byref(pred), out.ids, out.cap, out.count, None), "gather")
        kk, got = out.result()
        cand = order[host.arr["risk_level"][order] >= 4]
        exp = cand[full[k][cand]].astype(np.uint32)
        if kk != len(exp) or not np.array_equal(got, exp):
            sys.exit(f"DIFF gather: it={it} query={k}: {kk} vs {len(exp)}")
        # the query stream: two lanes, each with a scratch (and an epoch) of its own
        slot = ring[it % 3]
        if len(issued) == 3:
            k0, s0 = issued.pop(0)
            pq.check(L.pqps_qstream_wait(qs, ring.index(s0)), "wait")
            c = C.c_uint64()
            ctx.download(C.byref(c), s0.count, 8)
            a = np.zeros(max(c.value, 1), dtype=np.uint32)
            if c.value:
                ctx.download(a.ctypes.data, s0.ids, 4 * c.value)
            if c.value != len(want[k0]) or not np.array_equal(a[:c.value], want[k0]):
                sys.exit(f"DIFF qstream: it={it} query={k0}")
        pq.check(L.pqps_qstream_scan_slot(qs, it % 3, cols, nc, n, 0, C.byref(pred), slot.ids, slot.cap, slot.count, None), "qstream scan")
        issued.append((k, slot))
    pq.check(L.pqps_qstream_sync(qs), "qstream sync")
    pq.check(L.pqps_qstream_destroy(qs), "qstream destroy")
    print(f"wrap: 40 rounds of scan + gather + query stream ok at n={n}", flush=True)


def main():
    mode = sys.argv[1]
    ctx = pq.Context(0)
    if mode == "sizes":
        sweep_sizes(ctx, [int(x) for x in sys.argv[2:]] or [1, 4097, 65_537, 300_001, (1 << 21) + 17])
    elif mode == "shapes":
        sweep_shapes(ctx, int(sys.argv[2]) if len(sys.argv) > 2 else 70_001)
    elif mode == "wrap":
        sweep_wrap(ctx, int(sys.argv[2]) if len(sys.argv) > 2 else 300_001)
    else:
        sys.exit("usage: variant_driver.py sizes|shapes ...")
    ctx.close()
    print("OK")


if __name__ == "__main__":
    main()
