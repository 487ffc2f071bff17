"""Parity of the HIP path with the oracle, through the C-ABI, on a real MI355X.

  * shim level  (include/pqps_hip.h): synthetic tables generated on the device,
    the same bits regenerated on the host, oracle scan vs pqps_filter_* --
    bit-exact row-ID sequences; sizes straddle tile / chunk boundaries.
  * engine level (include/executeEngine-hip.h): every committed golden vector
    the REAL reference produced (tests/golden/select_golden.json) -- row IDs and
    the sha256 of every projected cell; index leaf order; known-answer tests of
    the reference's tests/*.c.
  * full size (BASELINE configs[1], 100 M rows): size-independent properties --
    ascending IDs, count == COUNT(*) == popcount(flags), sampled membership
    against the host twin, shard concatenation == whole-table result.
"""
import ctypes as C
import os
import hashlib
import json
import random

import numpy as np
import pytest

import qpelib as q

pq = q.pq
pytestmark = pytest.mark.gpu

SELECT = (json.loads((q.GOLDEN / "select_golden.json").read_text())
          + json.loads((q.GOLDEN / "select_random_golden.json").read_text())     # + seeded random WHERE trees, same reference
          + json.loads((q.GOLDEN / "select_wide_golden.json").read_text()))       # + lists of more than 32 comparisons, reference engine API
INDEX_CONFIGS = {
    "none": [],
    "default": pq.DEFAULT_INDEXES,
    "cmdid": [("command_id", 0)],
    "risk": [("risk_level", 1)],
    "risk_twice": [("risk_level", 1), ("risk_level", 1)],
}

QUERIES = {
    "S1": [("sudo_used", "=", "FALSE"), "AND", ("user_name", "=", "student1030")],
    "Q_A": [("risk_level", ">", "3")],
    "Q_B": [("sudo_used", "=", "TRUE"), "AND", ("risk_level", ">", "2")],
    "Q_C": [("exit_code", "!=", "0"), "AND", ("user_id", ">=", "1500"), "OR", ("risk_level", "=", "5")],
    "S7": [("sudo_used", "=", "TRUE"), "OR", [("risk_level", "=", "5"), "AND", ("shell_type", "=", "bash")]],
    "S8": [("user_id", "=", "1001"), "OR", [("user_name", "=", "student1002"), "AND", ("shell_type", "=", "zsh")]],
    "cid_range": [("command_id", ">=", "1000"), "AND", ("command_id", "<", "70000")],
    "u8": [("sudo_used", "=", "TRUE")],                        # one 1-byte column, ~7 %: four steps per wave scan in the expander
    "u8_dict": [("shell_type", "=", "zsh")],
    "mid": [("risk_level", ">", "1")],                          # ~43 %: steps of ~440 matches (staged 64-row path)
    "mid_u8": [("shell_type", "!=", "bash")],
    "all": [],
    "none": [("risk_level", ">", "9")],
    "neq": [("risk_level", "!=", "1")],
    "seven_leaves": [("risk_level", "=", "1"), "OR", ("risk_level", "=", "2"), "AND", ("exit_code", "=", "0"), "OR",
                     [("host_name", "<", "labpc-05"), "AND", ("shell_type", "!=", "zsh"), "AND", ("user_id", "<", "1900")],
                     "OR", ("sudo_used", "=", "1")],
}


@pytest.fixture(scope="module")
def ctx():
    c = pq.Context(0)
    yield c
    c.close()


class DeviceOut:
    def __init__(self, ctx, cap):
        self.ctx, self.cap = ctx, cap
        self.ids = ctx.malloc(max(cap, 1) * 4)
        self.count = ctx.malloc(64)
        self.flags = None

    def read_count(self):
        v = C.c_uint64()
        self.ctx.download(C.byref(v), self.count, 8)
        return v.value

    def read_ids(self, k):
        a = np.zeros(max(k, 1), dtype=np.uint32)
        if k:
            self.ctx.download(a.ctypes.data, self.ids, 4 * k)
        return a[:k]

    def free(self):
        self.ctx.free(self.ids)
        self.ctx.free(self.count)


def gpu_scan(ctx, table, chain, out, id_base=0):
    pred, cols, nc, _ = table.bind(chain)
    pq.check(pq.lib().pqps_filter_scan(ctx.h, cols, nc, table.n, id_base, C.byref(pred), out.ids, out.cap, out.count, None),
             "pqps_filter_scan")
    ctx.sync()
    k = out.read_count()
    assert k <= out.cap
    return out.read_ids(k)


def gpu_count(ctx, table, chain, out):
    pred, cols, nc, _ = table.bind(chain)
    pq.check(pq.lib().pqps_filter_count(ctx.h, cols, nc, table.n, C.byref(pred), out.count, None), "pqps_filter_count")
    ctx.sync()
    return out.read_count()


@pytest.mark.parametrize("n", [0, 1, 3, 63, 64, 255, 256, 257, 1023, 4095, 4096, 4097, 8191, 12288, 100001, 1 << 20, (1 << 21) + 17])
def test_scan_matches_oracle_across_sizes(ctx, n):
    dev = pq.SyntheticTable(ctx, n, seed=0x5EED)
    host = q.HostSynth(n, seed=0x5EED)
    out = DeviceOut(ctx, n + 8)
    try:
        for name, chain in QUERIES.items():
            got = gpu_scan(ctx, dev, chain, out)
            want = host.oracle_scan(chain)
            assert got.shape == want.shape and np.array_equal(got, want), (name, n)
            assert gpu_count(ctx, dev, chain, out) == len(want), (name, n)
    finally:
        out.free()
        dev.free()


def test_device_generator_equals_host_twin(ctx):
    n = 300_007
    dev = pq.SyntheticTable(ctx, n, seed=77, row0=123_456_789)
    host = q.HostSynth(n, seed=77, row0=123_456_789)
    for name, _f, w in pq.SYNTH_LAYOUT:
        a = np.zeros(n, dtype=host.arr[name].dtype)
        ctx.download(a.ctypes.data, dev.ptr[name], n * w)
        assert np.array_equal(a, host.arr[name]), name
    # distributions of SURVEY App. B hold roughly
    r = host.arr["risk_level"]
    assert abs((r == 1).mean() - 0.568) < 0.01 and abs((r > 3).mean() - 0.044) < 0.005
    assert abs(host.arr["sudo_used"].mean() - 0.068) < 0.01
    dev.free()


def test_id_base_and_shard_concatenation(ctx):
    """Row-range shards (mpi:703-715 partition) filtered separately and concatenated in
    rank order give the whole-table answer -- the all-gatherv merge shape."""
    n, world = 1_000_003, 8
    whole = q.HostSynth(n, seed=5).oracle_scan(QUERIES["Q_B"])
    parts = []
    for r in range(world):
        s, c = C.c_uint64(), C.c_uint64()
        pq.lib().pqps_partition(n, world, r, C.byref(s), C.byref(c))
        dev = pq.SyntheticTable(ctx, c.value, seed=5, row0=s.value)
        out = DeviceOut(ctx, c.value + 8)
        parts.append(gpu_scan(ctx, dev, QUERIES["Q_B"], out, id_base=s.value))
        out.free()
        dev.free()
    assert np.array_equal(np.concatenate(parts), whole)

def test_merge_slots_is_the_allgatherv_layout(ctx):
    """8 shards filtered straight into [count | IDs] slots (what pqps_exchange_select / merge.py hand to
    the all-gather), laid out rank after rank as the collective delivers them, compacted on the device
    by pqps_merge_slots: equals the whole-table answer; a slot that overflowed is reported."""
    n, world, hdr = 1_000_003, 8, pq.SLOT_HEADER_WORDS
    chain = QUERIES["Q_B"]
    whole = q.HostSynth(n, seed=5).oracle_scan(chain)
    shard_max = 0
    bounds = []
    for r in range(world):
        s_, c_ = C.c_uint64(), C.c_uint64()
        pq.lib().pqps_partition(n, world, r, C.byref(s_), C.byref(c_))
        bounds.append((s_.value, c_.value))
    for cap, expect_overflow in ((len(whole) // world * 2 + 64) & ~1, False), (1000, True):
        stride = cap + hdr
        slots = ctx.malloc(world * stride * 4)
        ctx.memset(slots, 0, world * stride * 4)
        merged = ctx.malloc(world * cap * 4)
        totals = ctx.malloc(16)
        for r, (start, count) in enumerate(bounds):
            dev = pq.SyntheticTable(ctx, count, seed=5, row0=start)
            pred, cols, nc, _ = dev.bind(chain)
            base = slots + r * stride * 4
            rc = pq.lib().pqps_filter_scan(ctx.h, cols, nc, count, start, C.byref(pred), base + 4 * hdr, cap, base, None)
            pq.check(rc, "pqps_filter_scan")
            ctx.sync()
            dev.free()
        pq.check(pq.lib().pqps_merge_slots(ctx.h, slots, world, stride, merged, world * cap, totals, None), "pqps_merge_slots")
        ctx.sync()
        t = (C.c_uint64 * 2)()
        ctx.download(t, totals, 16)
        assert t[1] == len(whole)
        if expect_overflow:
            assert t[0] == world * cap < t[1]
        else:
            assert t[0] == len(whole)
            got = np.zeros(len(whole), dtype=np.uint32)
            ctx.download(got.ctypes.data, merged, got.nbytes)
            assert np.array_equal(got, whole)
        for p_ in (slots, merged, totals):
            ctx.free(p_)
    # argument validation
    assert pq.lib().pqps_merge_slots(ctx.h, None, world, 100, None, 0, None, None) == -1
    assert pq.lib().pqps_merge_slots(ctx.h, 8, 0, 100, 8, 0, None, None) == -1
    assert pq.lib().pqps_merge_slots(ctx.h, 8, 2, 4, 8, 0, None, None) == -1       # stride leaves no room for IDs


def test_native_exchange_world_of_one(ctx):
    """pqps_exchange_*: scan + sizes + exactly-sized payload behind one call.  One GPU here, so a world of 1:
    checks the RCCL plumbing (dlopen, two-step bring-up, communicator, the COUNT all-reduce on the exchange
    stream), the held-back payload phase and slot reuse round the ring, growth of the gathered list, and that
    every query's merged list equals the oracle's."""
    import importlib.util
    torch = None        # no torch in this process: the shim's HIP runtime is the system one, and so is the RCCL it loads
    spec = importlib.util.spec_from_file_location("pqps_merge", q.PKG / "merge.py")
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    n = 300_007
    host = q.HostSynth(n, seed=9)
    dev = pq.SyntheticTable(ctx, n, seed=9)
    names = ["Q_B", "S1", "none", "Q_A", "S7", "Q_B", "neq"]
    want = {k: host.oracle_scan(QUERIES[k]) for k in set(names)}
    xch = mg.ShardExchange.open(pq, ctx, torch, None, 1, 0, n, ring=2)
    assert xch is not None
    try:
        pending = []
        for i, name in enumerate(names):
            pred, cols, nc, _ = dev.bind(QUERIES[name])
            slot = i % 2
            if len(pending) == 2:                          # read a slot's result before the ring reuses it
                j, nm = pending.pop(0)
                got, local = xch.result(j)
                assert local == len(want[nm]) and np.array_equal(got, want[nm]), nm
            xch.select(cols, nc, n, 0, C.byref(pred), slot, None)
            pending.append((slot, name))
        for j, nm in pending:
            got, local = xch.result(j)
            assert local == len(want[nm]) and np.array_equal(got, want[nm]), nm
        # COUNT(*) = count kernel + all-reduce (mpi:745), interleaved with an ID query on the other slot
        for nm in ("Q_B", "none", "seven_leaves"):
            pred, cols, nc, _ = dev.bind(QUERIES[nm])
            xch.count(cols, nc, n, C.byref(pred), 0, None)
            pred2, cols2, nc2, _ = dev.bind(QUERIES["S1"])
            xch.select(cols2, nc2, n, 0, C.byref(pred2), 1, None)
            assert xch.count_result(0) == (len(host.oracle_scan(QUERIES[nm])),) * 2, nm
            assert np.array_equal(xch.result(1)[0], want["S1"])
    finally:
        xch.close()
        dev.free()
    # an answer larger than the first allocation of the gathered list (2^20 IDs): the list grows, nothing is cut;
    # a ring of three with results read late, and a ring of one (nothing to hide the sizes' round trip behind)
    n2 = 2_600_003
    host2 = q.HostSynth(n2, seed=10)
    dev = pq.SyntheticTable(ctx, n2, seed=10, columns=["sudo_used", "risk_level", "user_name"])
    dense = [("sudo_used", "=", "FALSE")]
    for ring in (3, 1):
        xch = mg.ShardExchange.open(pq, ctx, torch, None, 1, 0, n2, ring=ring)
        try:
            plan = [dense, QUERIES["Q_A"], dense, QUERIES["S1"]]
            for i, chain in enumerate(plan):
                pred, cols, nc, _ = dev.bind(chain)
                if i >= ring:
                    got, _ = xch.result(i % ring)
                    assert np.array_equal(got, host2.oracle_scan(plan[i - ring]))
                xch.select(cols, nc, n2, 0, C.byref(pred), i % ring, None)
            xch.sync()
            for i in range(max(len(plan) - ring, 0), len(plan)):
                got, local = xch.result(i % ring)
                want2 = host2.oracle_scan(plan[i])
                assert local == len(want2) and np.array_equal(got, want2)
            assert len(host2.oracle_scan(dense)) > (1 << 20)
        finally:
            xch.close()
    dev.free()
    # overflow: a slot smaller than the rank's own answer is reported, not silently truncated
    dev = pq.SyntheticTable(ctx, n, seed=9)
    xch = mg.ShardExchange.open(pq, ctx, torch, None, 1, 0, 1000, ring=1)
    try:
        pred, cols, nc, _ = dev.bind(QUERIES["neq"])
        xch.select(cols, nc, n, 0, C.byref(pred), 0, None)
        with pytest.raises(pq.PqpsError):
            xch.result(0)
    finally:
        xch.close()
        dev.free()


def test_query_stream_keeps_queries_apart(ctx):
    """pqps_qstream_*: two queries in flight, each whole on one of the stream's two lanes.  40 ID queries of 6
    different shapes (sparse, dense, > 6 leaves -> generic kernel, empty) go through a depth-3 stream back to
    back, each with its own output buffer, COUNT(*) queries (pqps_qstream_count) in between; every answer must
    equal the oracle's, and a sync in the middle must not disturb the sequence."""
    L = pq.lib()
    n = 700_001
    host = q.HostSynth(n, seed=17)
    dev = pq.SyntheticTable(ctx, n, seed=17)
    names = ["S1", "Q_A", "seven_leaves", "none", "Q_C", "S7"]
    want = {k: host.oracle_scan(QUERIES[k]) for k in names}
    bound = {k: dev.bind(QUERIES[k]) for k in names}
    qs = C.c_void_p()
    pq.check(L.pqps_qstream_create(ctx.h, 3, C.byref(qs)), "pqps_qstream_create")
    assert L.pqps_qstream_create(ctx.h, 0, C.byref(C.c_void_p())) == -1           # depth >= 1
    outs = [DeviceOut(ctx, n) for _ in range(8)]
    counts = ctx.malloc(8 * 8)
    try:
        order = [names[(3 * i + i // 5) % len(names)] for i in range(40)]
        for base in range(0, 40, 8):                                  # 8 queries in flight per batch of buffers
            for j, name in enumerate(order[base:base + 8]):
                pred, cols, nc, _ = bound[name]
                pq.check(L.pqps_qstream_scan(qs, cols, nc, n, 0, C.byref(pred), outs[j].ids, outs[j].cap, outs[j].count, None),
                         "pqps_qstream_scan")
                if j % 3 != 2:                                        # ... and its COUNT(*) right behind, on the other lane
                    pq.check(L.pqps_qstream_count(qs, cols, nc, n, C.byref(pred), counts + 8 * j, None), "pqps_qstream_count")
            pq.check(L.pqps_qstream_sync(qs), "pqps_qstream_sync")
            ctx.sync()
            got_counts = (C.c_uint64 * 8)()
            ctx.download(got_counts, counts, 64)
            for j, name in enumerate(order[base:base + 8]):
                k = outs[j].read_count()
                assert k == len(want[name]) and np.array_equal(outs[j].read_ids(k), want[name]), (base + j, name)
                if j % 3 != 2:
                    assert got_counts[j] == len(want[name]), (base + j, name)
        # capacity overflow is still reported through the count, nothing is written past the buffer
        pred, cols, nc, _ = bound["Q_A"]
        small = DeviceOut(ctx, 100)
        pq.check(L.pqps_qstream_scan(qs, cols, nc, n, 0, C.byref(pred), small.ids, 100, small.count, None))
        pq.check(L.pqps_qstream_sync(qs))
        assert small.read_count() == len(want["Q_A"]) and np.array_equal(small.read_ids(100), want["Q_A"][:100])
        small.free()
    finally:
        pq.check(L.pqps_qstream_destroy(qs))
        ctx.free(counts)
        for o in outs:
            o.free()
        dev.free()


def test_shim_rejects_malformed_calls(ctx):
    """Every shape the kernels assume is checked on the host first: a bad call returns PQPS_EINVAL
    with a message and launches nothing (a kernel fault can take the whole GPU down)."""
    L, EINVAL = pq.lib(), -1
    dev = pq.SyntheticTable(ctx, 10_000, seed=2)
    out = DeviceOut(ctx, 10_000)
    pred, cols, nc, _ = dev.bind(QUERIES["Q_B"])

    def scan(cols_=cols, nc_=nc, n=10_000, base=0, pred_=pred, ids=out.ids, cap=out.cap, cnt=out.count):
        return L.pqps_filter_scan(ctx.h, cols_, nc_, n, base, C.byref(pred_) if pred_ is not None else None, ids, cap, cnt, None)

    assert scan() == 0
    assert scan(pred_=None) == EINVAL and b"predicate is NULL" in L.pqps_last_error()
    assert scan(cnt=None) == EINVAL
    assert scan(ids=None) == EINVAL
    assert scan(cols_=None) == EINVAL
    assert scan(nc_=nc + 1) == EINVAL                                 # predicate / call disagree on the column count
    assert scan(n=2**32) == EINVAL and scan(n=10_000, base=2**32 - 5_000) == EINVAL      # u32 row IDs
    assert L.pqps_filter_scan(None, cols, nc, 10, 0, C.byref(pred), out.ids, out.cap, out.count, None) == EINVAL

    def mutated(fn):
        p2 = pq.Predicate.from_buffer_copy(pred)
        c2 = pq.column_array([(cols[i].data, cols[i].width) for i in range(nc)])
        fn(p2, c2)
        return scan(cols_=c2, pred_=p2)

    assert mutated(lambda p, c: setattr(c[0], "width", 3)) == EINVAL
    assert mutated(lambda p, c: setattr(c[0], "data", cols[0].data + 4)) == EINVAL          # 16-byte alignment
    assert mutated(lambda p, c: setattr(c[1], "data", None)) == EINVAL
    assert mutated(lambda p, c: setattr(p.leaf[0], "column", 7)) == EINVAL
    assert mutated(lambda p, c: setattr(p.leaf[0], "negate", 2)) == EINVAL
    assert mutated(lambda p, c: setattr(p, "n_leaves", 33)) == EINVAL

    def swap_leaves(p, c):
        p.leaf[0].column, p.leaf[1].column = 1, 0
    assert mutated(swap_leaves) == EINVAL                                               # leaves sorted by column
    # the jump program of a > 6-leaf predicate must only jump forward and name real leaf slots
    pred7, cols7, nc7, _ = dev.bind(QUERIES["seven_leaves"])
    for field, value in (("on_true", 0), ("on_false", 1), ("order", 9), ("on_true", 200)):
        p7 = pq.Predicate.from_buffer_copy(pred7)
        getattr(p7, field)[2] = value
        assert L.pqps_filter_scan(ctx.h, cols7, nc7, 10_000, 0, C.byref(p7), out.ids, out.cap, out.count, None) == EINVAL, field
    # count / flags / gather / index / merge entry points
    assert L.pqps_filter_count(ctx.h, cols, nc, 10_000, C.byref(pred), None, None) == EINVAL
    assert L.pqps_filter_flags(ctx.h, cols, nc, 10_000, C.byref(pred), None, out.count, None) == EINVAL
    assert L.pqps_index_build(ctx.h, None, 10, 0, out.ids, out.ids, None) == EINVAL
    assert L.pqps_compact_rows(ctx.h, cols, nc, 10, None, C.byref(C.c_uint64()), None) == EINVAL
    assert L.pqps_compact_rows(ctx.h, cols, 0, 10, out.ids, C.byref(C.c_uint64()), None) == EINVAL
    assert L.pqps_exchange_select(None, cols, nc, 10, 0, C.byref(pred), 0, None) == EINVAL
    assert L.pqps_qstream_scan(None, cols, nc, 10, 0, C.byref(pred), out.ids, out.cap, out.count, None) == EINVAL
    assert L.pqps_qstream_create(ctx.h, 65, C.byref(C.c_void_p())) == EINVAL and L.pqps_qstream_create(None, 2, C.byref(C.c_void_p())) == EINVAL
    assert L.pqps_qstream_wait(None, 0) == EINVAL and L.pqps_qstream_mark(None, 0) == EINVAL
    assert L.pqps_copy_peer(None, out.ids, ctx.h, out.ids, 4, None) == EINVAL
    assert L.pqps_project_column(ctx.h, cols, None, out.count, 10, 0, out.ids, None) == EINVAL
    assert L.pqps_project_column(ctx.h, cols, out.ids, out.count, 10, 0, None, None) == EINVAL
    assert L.pqps_gather_keys(ctx.h, cols, 1, out.ids, out.count, 10, 0, None, None) == EINVAL
    u8col = pq.column_array([(cols[0].data, 1)])
    assert L.pqps_gather_keys(ctx.h, u8col, 1, out.ids, out.count, 10, 0, out.ids, None) == EINVAL       # signed keys are 4 bytes wide
    assert L.pqps_merge_index_slots(ctx.h, out.ids, None, 2, 100, out.ids, 10, out.count, None) == EINVAL
    assert L.pqps_merge_index_slots(ctx.h, out.ids, out.ids, 2, 101, out.ids, 10, out.count, None) == EINVAL   # odd stride
    assert L.pqps_ctx_reserve(None, 10) == EINVAL
    # a good call still works afterwards
    assert np.array_equal(gpu_scan(ctx, dev, QUERIES["Q_B"], out), q.HostSynth(10_000, seed=2).oracle_scan(QUERIES["Q_B"]))
    out.free()
    dev.free()


def test_compact_rows_equals_numpy_boolean_indexing(ctx):
    """pqps_compact_rows (DELETE on the device): flags from the predicate kernel, every column width
    compacted in place, survivors in order -- numpy `col[~flags]` is the checker."""
    rng = np.random.default_rng(3)
    for n in (1, 4097, 300_001):
        host = {1: rng.integers(0, 256, n, dtype=np.uint8), 2: rng.integers(0, 65536, n, dtype=np.uint16),
                4: rng.integers(0, 2**32, n, dtype=np.uint32), 8: rng.integers(0, 2**63, n, dtype=np.uint64)}
        pad = (n + 4095) // 4096 * 4096
        bufs = {}
        for w, arr in host.items():
            bufs[w] = ctx.malloc(pad * w)
            ctx.memset(bufs[w], 0, pad * w)
            ctx.upload(bufs[w], arr.ctypes.data, arr.nbytes)
        cols = pq.column_array([(bufs[w], w) for w in (8, 1, 4, 2)])
        for flags in (rng.random(n) < 0.3, np.zeros(n, bool), np.ones(n, bool), np.arange(n) % 2 == 0):
            for w, arr in host.items():
                ctx.upload(bufs[w], arr.ctypes.data, arr.nbytes)
            f8 = flags.astype(np.uint8)
            fdev = ctx.malloc(pad)
            ctx.memset(fdev, 0, pad)
            ctx.upload(fdev, f8.ctypes.data, n)
            kept = C.c_uint64()
            pq.check(pq.lib().pqps_compact_rows(ctx.h, cols, 4, n, fdev, C.byref(kept), None), "pqps_compact_rows")
            assert kept.value == int((~flags).sum())
            for w, arr in host.items():
                got = np.zeros(max(kept.value, 1), dtype=arr.dtype)
                if kept.value:
                    ctx.download(got.ctypes.data, bufs[w], kept.value * w)
                assert np.array_equal(got[:kept.value], arr[~flags]), (n, w)
            ctx.free(fdev)
        for b_ in bufs.values():
            ctx.free(b_)


def test_bench_exchange_path_runs_on_one_gpu():
    """bench.py --force-merge: the N > 1 code path (torch.distributed bootstrap over RCCL, shim-driven
    all-gather + merge on the exchange stream, ring of slots) with a world of 1, in its own process
    with bench.py's load order.  stdout must be exactly one JSON line."""
    import json
    import subprocess
    import sys
    # (level, exchange) -> who drives the exchange: the engine API in C (hipEngineJoinRanksHIP: round 4), the shim from
    # Python (pqps_exchange_*), torch.distributed (merge.IdMerger)
    for level, mode, who in (("engine", "rccl", "engine API"), ("shim", "rccl", "shim-driven"), ("engine", "torch", "torch.distributed")):
        p = subprocess.run([sys.executable, str(q.ROOT / "bench.py"), "--rows", "3000000", "--steps", "9", "--warmup", "2", "--reps", "3",
                            "--no-extras", "--no-cpu-baseline", "--force-merge", "--exchange", mode, "--level", level],
                           capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533"))
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, p.stdout[:2000]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 1 and d["steps"] == 9 and d["config"]["matches_total"] > 0
        assert who in d["config"]["parallelism"], d["config"]["parallelism"]
        assert d["config"]["level"] == ("engine" if (level, mode) == ("engine", "rccl") else "shim")
        assert d["roofline"]["value_spread_reps"] == 3 and d["roofline"]["value_spread_min"] <= d["value"] <= d["roofline"]["value_spread_max"]
        assert d["config"]["shim_ms_per_step"] is not None and "NOTE" not in d["config"]["parallelism"]


@pytest.mark.parametrize("ranks,extra", [(2, []), (3, ["--query", "Q_A"]), (2, ["--mode", "count"])])
def test_bench_with_several_ranks_sharing_the_gpu(ranks, extra):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank), with the ranks sharing
    this box's one GPU and gloo as the transport (RCCL refuses two ranks on one device): row-range shards, the sizes-first
    exchange through merge.IdMerger with queries in flight, the COUNT all-reduce, and bench.py's own verification of the
    gathered result on EVERY rank (it asserts).  One JSON line from rank 0."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(q.ROOT / "bench.py"),
                        "--gpus", str(ranks), "--backend", "gloo", "--rows", "2000003", "--steps", "12", "--warmup", "3",
                        "--no-extras", "--no-cpu-baseline", *extra],
                       capture_output=True, text=True, timeout=900, cwd=str(q.ROOT))
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["steps"] == 12 and d["config"]["rows_total"] == 2000003 * ranks
    assert d["config"]["matches_total"] > 0 and d["scaling"] == "weak"


@pytest.mark.parametrize("ranks,extra,level", [(2, [], "engine"), (3, ["--query", "Q_A"], "engine"), (2, ["--mode", "count"], "engine"),
                                               (2, ["--query", "Q_A", "--level", "shim"], "shim")])
def test_bench_engine_level_with_several_ranks_through_the_process_loopback(ranks, extra, level):
    """bench.py exactly as the driver launches it for N > 1 -- torch.distributed.run, one process per rank, `value` at the ENGINE
    level: rank engines (initializeEngineSyntheticRankHIP) joined through hipEngineJoinRanksHIP, tickets issued by the C loop,
    every answer the whole table's -- on this box's one GPU: torch's transport is gloo and the "RCCL" the product loads is
    tests/loopback/libloopback_mp.so (ranks as processes, data through shared memory).  Also once at the shim level
    (pqps_exchange_* from Python).  bench.py verifies count + checksum on every rank before and after timing and exits non-zero
    on a wrong answer; the record must say which path ran."""
    import json
    import socket
    import subprocess
    import sys
    lib = q.ROOT / "tests" / "loopback" / "libloopback_mp.so"
    assert lib.exists(), "build it first: make -C tests/loopback"
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(q.ROOT / "bench.py"),
                        "--gpus", str(ranks), "--backend", "gloo", "--exchange", "rccl", "--rccl-library", str(lib),
                        "--rows", "2000003", "--steps", "12", "--warmup", "3", "--reps", "2", "--no-extras", "--no-cpu-baseline", *extra],
                       capture_output=True, text=True, timeout=900, cwd=str(q.ROOT), env=dict(os.environ, LOOPBACK_MP_OUTBOX_MB="64"))
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["config"]["rows_total"] == 2000003 * ranks and d["config"]["matches_total"] > 0
    assert d["config"]["level"] == level and "NOTE" not in d["config"]["parallelism"], d["config"]["parallelism"]
    assert ("engine API" if level == "engine" else "shim-driven") in d["config"]["parallelism"]
    if "Q_A" in extra:
        assert "bytes on the wire" in d["config"]["parallelism"]               # the compact form's accounting made it into the record
        assert " 0 of " in d["config"]["parallelism"]                          # 44 000 / 29 000 IDs per rank: the two-step path
    elif "count" not in extra:
        # S1 (the default workload): 67 IDs per rank -- every exchanged query finished in the sizes all-gather
        import re
        m = re.search(r"(\d+) of (\d+) exchanged queries finished in the one collective", d["config"]["parallelism"])
        assert m and int(m.group(1)) == int(m.group(2)) > 0, d["config"]["parallelism"]


def test_flags_mode(ctx):
    n = 70_001
    dev = pq.SyntheticTable(ctx, n, seed=9)
    host = q.HostSynth(n, seed=9)
    flags_dev = ctx.malloc(n + 4096)
    cnt = ctx.malloc(64)
    for name in ("Q_A", "S7", "all", "none", "seven_leaves"):
        pred, cols, nc, _ = dev.bind(QUERIES[name])
        pq.check(pq.lib().pqps_filter_flags(ctx.h, cols, nc, n, C.byref(pred), flags_dev, cnt, None))
        ctx.sync()
        f = np.zeros(n, dtype=np.uint8)
        ctx.download(f.ctypes.data, flags_dev, n)
        want = np.zeros(n, dtype=np.uint8)
        want[host.oracle_scan(QUERIES[name])] = 1
        assert np.array_equal(f, want), name
    ctx.free(flags_dev)
    ctx.free(cnt)
    dev.free()


def test_random_trees_on_device(ctx):
    import test_predicate_compile as tpc
    n = 50_000
    dev = pq.SyntheticTable(ctx, n, seed=3)
    host = q.HostSynth(n, seed=3)
    out = DeviceOut(ctx, n + 8)
    rng = random.Random(2024)
    leaves = [l for l in tpc.LEAVES if l[0] not in ("working_directory", "timestamp", "raw_command")]
    done = 0
    for _ in range(150):
        saved, tpc.LEAVES = tpc.LEAVES, leaves
        try:
            chain = tpc.random_chain(rng, depth=3, max_items=5)
        finally:
            tpc.LEAVES = saved
        if tpc.count_leaves(chain) > 40:
            continue
        try:
            got = gpu_scan(ctx, dev, chain, out)
        except pq.PqpsError as e:
            assert "limit" in str(e)
            continue
        assert np.array_equal(got, host.oracle_scan(chain)), chain
        done += 1
    assert done > 100
    out.free()
    dev.free()


def test_capacity_overflow_is_reported_not_written(ctx):
    n = 10_000
    dev = pq.SyntheticTable(ctx, n, seed=1)
    out = DeviceOut(ctx, 16)
    pred, cols, nc, _ = dev.bind([])
    guard = ctx.malloc(4 * 64)
    pq.check(pq.lib().pqps_filter_scan(ctx.h, cols, nc, n, 0, C.byref(pred), out.ids, 16, out.count, None))
    ctx.sync()
    assert out.read_count() == n                       # true count, caller sees it exceeds capacity
    assert list(out.read_ids(16)) == list(range(16))
    ctx.free(guard)
    out.free()
    dev.free()


@pytest.mark.parametrize("cap", [1, 103, 1001, 1003, 4096, 70_001, 200_002])
def test_a_result_buffer_that_is_too_small_is_filled_and_not_overrun(ctx, cap):
    """Dense and sparse steps (row lists of both kinds, bit masks of a 1-byte column): the first `cap` IDs arrive, the
    count is the true one, nothing behind the buffer's end is touched -- whatever lane and store the end falls into."""
    n = 500_001
    dev = pq.SyntheticTable(ctx, n, seed=5)
    host = q.HostSynth(n, seed=5)
    guard = 4096
    buf = ctx.malloc(4 * (cap + guard))
    cnt = ctx.malloc(64)
    sentinel = np.full(cap + guard, 0xDEADBEEF, dtype=np.uint32)
    for chain in ([("risk_level", ">", "1")], [("risk_level", ">", "2")], [("risk_level", ">", "3")], [("sudo_used", "=", "FALSE")],
                  [("user_name", "!=", "student1030")], [("exit_code", "!=", "0"), "AND", ("user_id", ">=", "1500"), "OR", ("risk_level", "=", "5")]):
        want = host.oracle_scan(chain)
        if len(want) <= cap:
            continue
        ctx.upload(buf, sentinel.ctypes.data, sentinel.nbytes)
        pred, cols, nc, _ = dev.bind(chain)
        pq.check(pq.lib().pqps_filter_scan(ctx.h, cols, nc, n, 0, C.byref(pred), buf, cap, cnt, None))
        ctx.sync()
        k = C.c_uint64()
        ctx.download(C.byref(k), cnt, 8)
        got = np.zeros(cap + guard, dtype=np.uint32)
        ctx.download(got.ctypes.data, buf, got.nbytes)
        assert k.value == len(want), chain
        assert np.array_equal(got[:cap], want[:cap]), chain
        assert np.all(got[cap:] == 0xDEADBEEF), chain
    ctx.free(buf)
    ctx.free(cnt)
    dev.free()


# ---- engine level: golden vectors of the real reference -------------------------------
def sha_rows(rows):
    h = hashlib.sha256()
    for r in rows:
        for c in r:
            h.update(c.encode("latin-1"))
            h.update(b"\x1f")
        h.update(b"\x1e")
    return h.hexdigest()


_engines = {}


def engine_for(csv, cfg):
    key = (csv, cfg)
    if key not in _engines:
        _engines[key] = pq.HipEngine(q.GOLDEN / csv, INDEX_CONFIGS[cfg])
    return _engines[key]


@pytest.mark.parametrize("case", SELECT, ids=[f"{c['csv'][:4]}-{c['name']}-{c['indexes']}" for c in SELECT])
def test_engine_select_matches_reference_golden(case):
    eng = engine_for(case["csv"], case["indexes"])
    chain = q.chain_from_jsonable(case["where"])
    ids = eng.select_ids(chain)
    assert len(ids) == case["num_records"]
    if q.case_ids(case) is not None:
        assert ids == q.case_ids(case)
    sql = case["sql"]
    sel = sql[len("SELECT "):sql.index(" FROM ")]
    cols = None if sel.strip() == "*" else [c.strip() for c in sel.split(",")]
    res = eng.select(cols, chain)
    assert res["success"] and res["numRecords"] == case["num_records"]
    assert sha_rows(res["rows"]) == case["rows_sha256"]
    if case.get("columns"):
        assert res["columns"] == case["columns"]
    if case["indexes"] == "none":
        assert eng.count(chain) == case["num_records"]


BOOLPROBE = json.loads((q.GOLDEN / "select_boolprobe_golden.json").read_text())
BOOLPROBE_INDEX_CONFIGS = {
    "default": pq.DEFAULT_INDEXES,
    "bool_only": [("sudo_used", 3)],
    "bool_twice": [("sudo_used", 3), ("risk_level", 1), ("sudo_used", 3)],
}


@pytest.mark.parametrize("cfg", sorted(BOOLPROBE_INDEX_CONFIGS))
def test_engine_boolprobe_matches_qpeomp_golden(cfg):
    """hipEngineProbeBoolIndexes: index mode follows QPEOMP / QPEMPI (BOOL indexes probed too, omp:424-459) -- the
    compiled OpenMP engine's answers, one thread (select_boolprobe_golden.json); switched off again, QPESeq's."""
    eng = pq.HipEngine(q.GOLDEN / "commands_2k.csv", BOOLPROBE_INDEX_CONFIGS[cfg])
    orc = q.OracleTable(q.GOLDEN / "commands_2k.csv", BOOLPROBE_INDEX_CONFIGS[cfg])
    assert eng.probe_bool_indexes(True) == 0
    cases = [c for c in BOOLPROBE if c["indexes"] == cfg]
    assert len(cases) > 20
    for case in cases:
        chain = q.chain_from_jsonable(case["where"])
        ids = eng.select_ids(chain)
        assert ids == q.case_ids(case), case["name"]
        res = eng.select(case["columns"], chain)
        assert res["success"] and res["numRecords"] == case["num_records"] and res["columns"] == case["columns"], case["name"]
        assert sha_rows(res["rows"]) == case["rows_sha256"], case["name"]
    assert eng.probe_bool_indexes(False) == 1
    for case in cases[:12]:
        chain = q.chain_from_jsonable(case["where"])
        assert eng.select_ids(chain) == orc.select_ids(chain)[0], case["name"]
    eng.close()


def test_engine_index_order_matches_reference_btree(ctx):
    gold = json.loads((q.GOLDEN / "index_order_golden.json").read_text())
    for csv, per_attr in gold.items():
        attrs = [("command_id", 0), ("user_id", 1), ("risk_level", 1), ("exit_code", 1), ("sudo_used", 3),
                 ("shell_type", 2), ("user_name", 2)]
        eng = pq.HipEngine(q.GOLDEN / csv, attrs)
        # read the device permutations back through the table handle
        tbl = C.cast(eng.e.contents.record_block, C.POINTER(HipTable)).contents
        for i, (a, _t) in enumerate(attrs):
            perm = np.zeros(max(eng.n, 1), dtype=np.uint32)
            if eng.n:
                ctx.download(perm.ctypes.data, tbl.index[i].perm_dev, 4 * eng.n)
            assert list(perm[:eng.n]) == per_attr[a], (csv, a)
        eng.close()


class HipDictionary(C.Structure):
    _fields_ = [("count", C.c_int), ("values", C.c_void_p), ("storage", C.c_void_p), ("storage_bytes", C.c_size_t)]


class HipIndex(C.Structure):
    _fields_ = [("column", C.c_int), ("key_kind", C.c_int), ("perm_dev", C.c_void_p), ("keys_dev", C.c_void_p)]


class HipTable(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("n_rows", C.c_uint64), ("capacity_rows", C.c_uint64),
                ("col", pq.Column * 12), ("dict", HipDictionary * 12), ("index", C.POINTER(HipIndex)),
                ("ids_dev", C.c_void_p), ("capacity_ids", C.c_uint64), ("count_dev", C.c_void_p),
                ("row_block", C.c_void_p), ("row_capacity", C.c_size_t)]


def test_kat_reference_unit_tests():
    """tests/executeEngine-serial-test.c:29-114 through evaluateWhereClause (HIP, one-row table)."""
    L = pq.lib()
    r = pq.Record()
    r.command_id, r.risk_level, r.user_id, r.sudo_used, r.exit_code = 100, 5, 10, True, 0
    r.user_name, r.raw_command, r.base_command, r.shell_type = b"admin", b"ls -la", b"ls", b"bash"
    r.timestamp, r.working_directory, r.host_name = b"2023-01-01", b"/home/admin", b"localhost"
    for chain, want in [
        ([("risk_level", ">", "3")], True),
        ([[("risk_level", ">", "3"), "AND", ("user_id", "=", "10")]], True),
        ([[("risk_level", ">", "10")], "OR", [("user_id", "=", "10")]], True),
        ([[("risk_level", ">", "10")], "AND", [("user_id", "=", "10")]], False),
        ([("user_name", "=", "admin"), "AND", ("sudo_used", "=", "true")], True),
    ]:
        wl = pq.WhereList(chain)
        assert L.evaluateWhereClause(C.byref(r), wl.ptr) is want


def test_kat_duplicates_and_ranges(tmp_path):
    """tests/duplicate-test.c:37-54 and tests/bplus-serial-test.c:40-43 through the HIP engine."""
    p = tmp_path / "dup.csv"
    p.write_text(
        "command_id,raw_command,base_command,shell_type,exit_code,timestamp,sudo_used,working_directory,user_id,user_name,host_name,risk_level\n"
        "1,cmd1,base,bash,0,ts,0,wd,1001,user,host,1\n2,cmd2,base,bash,0,ts,0,wd,1001,user,host,1\n"
        "3,cmd3,base,bash,0,ts,0,wd,1001,user,host,2\n4,cmd4,base,bash,0,ts,0,wd,1001,user,host,1\n")
    eng = pq.HipEngine(p, [("risk_level", 1)])
    assert eng.n == 4
    assert eng.select_ids([("risk_level", "=", "1")]) == [3, 1, 0]
    assert eng.select_ids([("risk_level", "=", "2")]) == [2]
    eng.close()
    p2 = tmp_path / "r.csv"
    p2.write_text("h\n" + "".join(f"{k},c,b,bash,0,ts,0,wd,1,u,h,1\n" for k in (5, 15, 25, 35, 45)))
    eng = pq.HipEngine(p2, [("command_id", 0)])
    ids = eng.select_ids([("command_id", ">=", "10"), "AND", ("command_id", "<=", "30")])
    assert [eng.record(i).command_id for i in ids] == [15, 25, 15, 25]
    ids = eng.select_ids([("command_id", ">=", "5"), "AND", ("command_id", "<=", "45")])
    assert sorted(eng.record(i).command_id for i in ids) == [5, 5, 15, 15, 25, 25, 35, 35, 45, 45]
    eng.close()


def test_linear_search_records_keeps_input_order():
    t = q.OracleTable(q.GOLDEN / "commands_2k.csv", [])
    L = pq.lib()
    order = list(range(t.n))
    random.Random(7).shuffle(order)
    order = order[:500]
    ptrs = (C.POINTER(pq.Record) * len(order))(*[C.pointer(t.rows[i]) for i in order])
    chain = [("risk_level", ">", "1"), "AND", ("shell_type", "=", "bash")]
    wl = pq.WhereList(chain)
    n = C.c_int()
    res = L.linearSearchRecords(ptrs, len(order), wl.ptr, C.byref(n))
    got = [res[i].contents.command_id for i in range(n.value)]
    want = [t.rows[i].command_id for i in order if q.load_oracle().orc_eval_where(C.byref(t.rows[i]), wl.ptr)]
    assert got == want
    L.free(res)


# ---- full size: BASELINE configs[1] = 100 M rows on one GPU --------------------------------
def test_full_size_properties(ctx):
    n = 100_000_000
    dev = pq.SyntheticTable(ctx, n, seed=0x5EED, columns=["risk_level", "sudo_used", "user_name", "exit_code", "user_id"])
    out = DeviceOut(ctx, n // 4)
    flags_dev = ctx.malloc(n + 4096)
    rng = np.random.default_rng(1)
    try:
        for name in ("S1", "Q_A", "Q_B", "Q_C", "dense_13_percent"):
            chain = QUERIES[name] if name in QUERIES else [("risk_level", ">", "2")]       # (16-bit row lists in every step)
            ids = gpu_scan(ctx, dev, chain, out)
            k = len(ids)
            assert k > 0 and np.all(ids[1:] > ids[:-1]), name              # strictly ascending
            assert gpu_count(ctx, dev, chain, out) == k, name              # COUNT(*) agrees
            pred, cols, nc, _ = dev.bind(chain)
            pq.check(pq.lib().pqps_filter_flags(ctx.h, cols, nc, n, C.byref(pred), flags_dev, out.count, None))
            ctx.sync()
            assert out.read_count() == k
            f = np.zeros(n, dtype=np.uint8)
            ctx.download(f.ctypes.data, flags_dev, n)
            assert int(f.sum()) == k and np.array_equal(np.nonzero(f)[0].astype(np.uint32), ids), name
            # membership of sampled rows against the host twin / oracle, row by row
            sample = np.unique(np.concatenate([rng.integers(0, n, 2000), ids[rng.integers(0, k, 2000)],
                                               [0, 1, 4095, 4096, n - 1, n - 4097]]))
            member = np.isin(sample, ids)
            for row, m in zip(sample, member):
                h = q.HostSynth(1, seed=0x5EED, row0=int(row))
                assert (len(h.oracle_scan(chain)) == 1) == bool(m), (name, row)
            # a prefix of the table is bit-exact against the oracle
            m = 3_000_000
            want = q.HostSynth(m, seed=0x5EED).oracle_scan(chain)
            assert np.array_equal(ids[:len(want)], want) and (len(want) == k or ids[len(want)] >= m)
    finally:
        ctx.free(flags_dev)
        out.free()
        dev.free()


def test_row_ids_above_two_to_the_31(ctx):
    """4 000 000 001 rows on one GPU (the shim's row IDs are u32; the reference's int counts stop at
    2^31-1): the last partial step, the streaming-load kernel variant, one group per wave in K3 and
    every 64-bit offset.  The head (first 2 M rows) and the tail (last 2 M rows, IDs ~ 4e9) of
    the ID list are bit-exact against the host twin, the middle by COUNT(*) and order."""
    n = 4_000_000_001
    seed = 77
    dev = pq.SyntheticTable(ctx, n, seed=seed, columns=["sudo_used", "user_name"])
    chain = QUERIES["S1"]
    out = DeviceOut(ctx, 1 << 20)
    try:
        ids = gpu_scan(ctx, dev, chain, out)
        k = len(ids)
        assert 100_000 < k < out.cap
        assert np.all(ids[1:] > ids[:-1])
        assert int(ids[-1]) > 2**31 and int(ids[-1]) < n
        assert gpu_count(ctx, dev, chain, out) == k
        m = 2_000_000
        head = q.HostSynth(m, seed=seed).oracle_scan(chain)
        assert np.array_equal(ids[:len(head)], head) and ids[len(head)] >= m
        tail = q.HostSynth(m, seed=seed, row0=n - m).oracle_scan(chain, id_base=n - m)
        assert len(tail) > 0 and np.array_equal(ids[k - len(tail):], tail.astype(np.uint32))
        assert ids[k - len(tail) - 1] < n - m
        # a dense answer near the top of the ID space, through the index-free gather of a shard
        start = n - 3_000_000
        shard = pq.SyntheticTable(ctx, 3_000_000, seed=seed, row0=start, columns=["sudo_used", "user_name"])
        out2 = DeviceOut(ctx, 3_000_000)
        got = gpu_scan(ctx, shard, [("sudo_used", "=", "FALSE")], out2, id_base=start)
        want = q.HostSynth(3_000_000, seed=seed, row0=start).oracle_scan([("sudo_used", "=", "FALSE")], id_base=start)
        assert np.array_equal(got, want.astype(np.uint32)) and int(got[-1]) > 2**31
        out2.free()
        shard.free()
    finally:
        out.free()
        dev.free()
