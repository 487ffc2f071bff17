"""BASELINE configs[2]: index range-probe SELECT at synthetic scale, through the C-ABI shim.

pqps_index_build (perm sorted key asc / row desc = leaf order of the reference's B+ tree,
bplus.c:282-358,471-517) + pqps_index_probe (inclusive window) + pqps_filter_gather
(order-preserving re-filter with the complete WHERE, appended probe after probe) against a
numpy restatement of executeQuerySelectSerial's index path (serial:358-474) over the host
twin of the same seeded table.  5 M rows bit-exact, 100 M rows by properties."""
import ctypes as C

import numpy as np
import pytest

import qpelib as q

pq = q.pq
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pq.Context(0)
    yield c
    c.close()


class DeviceIndex:
    def __init__(self, ctx, table, name, signed):
        self.ctx, self.table, self.signed, self.name = ctx, table, signed, name
        self.width = table.width[name]
        n = max(table.n, 1)
        self.perm = ctx.malloc(4 * n)
        self.keys = ctx.malloc(self.width * n)
        col = pq.column_array([(table.ptr[name], self.width)])
        pq.check(pq.lib().pqps_index_build(ctx.h, col, table.n, 1 if signed else 0, self.perm, self.keys, None), "index build")

    def free(self):
        self.ctx.free(self.perm)
        self.ctx.free(self.keys)


def index_select(ctx, table, probes, chain, out_ids, cap, scratch, id_base=0, whole=False):
    """probes: [(DeviceIndex, key_lo, key_hi)] in the order QPESeq would probe them.  `whole`: every probe through
    pqps_index_select (what the engine calls: copies the probe's rows when the WHERE is the probed comparison itself)
    instead of pqps_index_probe + pqps_filter_gather."""
    pred, cols, nc, _ = table.bind(chain)
    count_dev, range_dev = scratch, scratch + 16
    ctx.memset(count_dev, 0, 8)
    L = pq.lib()
    for ix, lo, hi in probes:
        if whole:
            key_col = pq.column_array([(table.ptr[ix.name], ix.width)])
            pq.check(L.pqps_index_select(ctx.h, cols, nc, key_col, ix.perm, ix.keys, 1 if ix.signed else 0, table.n,
                                         lo & 0xFFFFFFFFFFFFFFFF, hi & 0xFFFFFFFFFFFFFFFF, id_base, C.byref(pred), range_dev,
                                         out_ids, cap, count_dev, None), "index select")
            continue
        pq.check(L.pqps_index_probe(ctx.h, ix.keys, ix.width, 1 if ix.signed else 0, table.n,
                                    lo & 0xFFFFFFFFFFFFFFFF, hi & 0xFFFFFFFFFFFFFFFF, range_dev, None), "probe")
        pq.check(L.pqps_filter_gather(ctx.h, cols, nc, ix.perm, range_dev, table.n, id_base, C.byref(pred),
                                      out_ids, cap, count_dev, None), "gather")
    ctx.sync()
    k = C.c_uint64()
    ctx.download(C.byref(k), count_dev, 8)
    ids = np.zeros(max(min(k.value, cap), 1), dtype=np.uint32)
    if k.value:
        ctx.download(ids.ctypes.data, out_ids, 4 * min(k.value, cap))
    return ids[:min(k.value, cap)], k.value


host_index_order, host_index_select = q.host_index_order, q.host_index_select


I32_MIN, I32_MAX = -2**31, 2**31 - 1


def test_index_mode_matches_serial_semantics_5m(ctx):
    n = 5_000_000
    dev = pq.SyntheticTable(ctx, n, seed=21)
    host = q.HostSynth(n, seed=21)
    ix = {"risk_level": DeviceIndex(ctx, dev, "risk_level", True), "user_id": DeviceIndex(ctx, dev, "user_id", True),
          "command_id": DeviceIndex(ctx, dev, "command_id", False), "exit_code": DeviceIndex(ctx, dev, "exit_code", True)}
    perms = {name: host_index_order(host.arr[name]) for name in ix}
    for name, d in ix.items():                                   # leaf order itself
        p = np.zeros(n, dtype=np.uint32)
        ctx.download(p.ctypes.data, d.perm, 4 * n)
        assert np.array_equal(p, perms[name].astype(np.uint32)), name
    # 1- and 2-byte keys (BOOL / dictionary-coded STRING indexes), and a column with a single value
    for name in ("sudo_used", "user_name", "host_name"):
        d = DeviceIndex(ctx, dev, name, False)
        p = np.zeros(n, dtype=np.uint32)
        ctx.download(p.ctypes.data, d.perm, 4 * n)
        assert np.array_equal(p, host_index_order(host.arr[name]).astype(np.uint32)), name
        k = np.zeros(n, dtype=host.arr[name].dtype)
        ctx.download(k.ctypes.data, d.keys, k.nbytes)
        assert np.array_equal(k, np.sort(host.arr[name], kind="stable")), name
        d.free()
    const = pq.SyntheticTable(ctx, 70_001, seed=21, columns=["risk_level"])
    L_ = pq.lib()
    ctx.memset(const.ptr["risk_level"], 0, 4 * 70_001)           # every key equal: no pass runs, rows stay descending
    d = DeviceIndex(ctx, const, "risk_level", True)
    p = np.zeros(70_001, dtype=np.uint32)
    ctx.download(p.ctypes.data, d.perm, 4 * 70_001)
    assert np.array_equal(p, np.arange(70_000, -1, -1, dtype=np.uint32))
    d.free()
    const.free()
    out = ctx.malloc(4 * 3 * n)
    scratch = ctx.malloc(256)
    cases = [
        # (probes in chain order, WHERE)
        ([("risk_level", 4, I32_MAX)], [("risk_level", ">", "3")]),                                        # Sample 3
        ([("risk_level", 5, 5)], [("risk_level", "=", "5")]),                                               # Sample 4
        ([("risk_level", 3, I32_MAX)], [("sudo_used", "=", "TRUE"), "AND", ("risk_level", ">", "2")]),      # Sample 2
        ([("user_id", 1001, 1001)], [("user_id", "=", "1001"), "OR", [("user_name", "=", "student1002"), "AND", ("shell_type", "=", "zsh")]]),
        ([("command_id", n - 1000, 2**64 - 1)], [("command_id", ">=", str(n - 1000))]),
        ([("command_id", 0, 9)], [("command_id", "<", "10")]),
        ([("risk_level", I32_MIN, I32_MAX)], [("risk_level", "!=", "1")]),                                 # != probes everything
        ([("risk_level", 4, I32_MAX), ("exit_code", 130, 130)], [("risk_level", ">=", "4"), "AND", ("exit_code", "=", "130")]),   # duplicates
        ([("risk_level", 5, 5)], [("risk_level", "=", "5"), "OR", ("user_name", "=", "student1030")]),     # OR-loss
        ([("user_id", 2990, I32_MAX), ("user_id", I32_MIN, 1003)], [("user_id", ">=", "2990"), "OR", ("user_id", "<=", "1003")]),
        ([("risk_level", 9, I32_MAX)], [("risk_level", ">", "8")]),                                         # empty range
    ]
    for probes, chain in cases:
        want = host_index_select(host, perms, probes, chain)
        for whole in (False, True):                                # probe + gather filter, and the engine's one call per probe
            got, k = index_select(ctx, dev, [(ix[nm], lo, hi) for nm, lo, hi in probes], chain, out, 3 * n, scratch, whole=whole)
            assert k == len(want) and np.array_equal(got, want), (chain, whole)
    # the copy of a probe's rows (WHERE = the probed comparison): shifted row numbers, a result buffer that is too small (filled,
    # not overrun, the count says what there was), two probes one behind the other
    probes, chain = [("risk_level", 4, I32_MAX)], [("risk_level", ">", "3")]
    want = host_index_select(host, perms, probes, chain)
    got, k = index_select(ctx, dev, [(ix["risk_level"], 4, I32_MAX)], chain, out, 3 * n, scratch, id_base=1000, whole=True)
    assert k == len(want) and np.array_equal(got, want + 1000)
    ctx.memset(out, 0xEE, 4 * 2048)
    got, k = index_select(ctx, dev, [(ix["risk_level"], 4, I32_MAX)], chain, out, 1001, scratch, whole=True)
    assert k == len(want) and np.array_equal(got, want[:1001])
    guard = np.zeros(8, dtype=np.uint32)
    ctx.download(guard.ctypes.data, out + 4 * 1001, guard.nbytes)
    assert np.all(guard == 0xEEEEEEEE)
    assert pq.lib().pqps_last_kernel().decode().startswith("append_range_kernel")
    for d in ix.values():
        d.free()
    ctx.free(out)
    ctx.free(scratch)
    dev.free()


def test_index_probe_100m_properties(ctx):
    n = 100_000_000
    dev = pq.SyntheticTable(ctx, n, seed=0x5EED, columns=["risk_level", "exit_code", "sudo_used"])
    ix = DeviceIndex(ctx, dev, "risk_level", True)
    out = ctx.malloc(4 * (n // 8))
    scratch = ctx.malloc(256)
    chain = [("risk_level", ">", "3"), "AND", ("exit_code", "=", "0")]
    got, k = index_select(ctx, dev, [(ix, 4, I32_MAX)], chain, out, n // 8, scratch)
    # same set as the scan-mode answer ...
    pred, cols, nc, _ = dev.bind(chain)
    cnt = ctx.malloc(64)
    pq.check(pq.lib().pqps_filter_count(ctx.h, cols, nc, n, C.byref(pred), cnt, None))
    ctx.sync()
    c = C.c_uint64()
    ctx.download(C.byref(c), cnt, 8)
    assert k == c.value == len(got) and len(np.unique(got)) == k
    # ... in index order: keys ascending, rows descending inside a key
    risk = np.zeros(n, dtype=np.int32)
    ctx.download(risk.ctypes.data, dev.ptr["risk_level"], 4 * n)
    kk = risk[got]
    assert np.all(kk[1:] >= kk[:-1]) and set(np.unique(kk)) == {4, 5}
    same = kk[1:] == kk[:-1]
    assert np.all(got[1:][same] < got[:-1][same])
    for p in (out, scratch, cnt):
        ctx.free(p)
    ix.free()
    dev.free()


@pytest.mark.parametrize("name,signed,lo,hi,chain", [
    ("risk_level", True, 4, I32_MAX, [("risk_level", ">", "3"), "AND", ("exit_code", "=", "0")]),
    ("user_id", True, 1001, 1003, [("user_id", ">=", "1001"), "AND", ("user_id", "<=", "1003")]),
    ("command_id", False, 1_500_000, 2**64 - 1, [("command_id", ">=", "1500000"), "AND", ("sudo_used", "=", "FALSE")]),
])
def test_index_mode_merge_across_shards(ctx, name, signed, lo, hi, chain):
    """SURVEY 8(e), index mode across shards: 8 row-range shards each answer the probe from their own
    index (leaf order inside the shard), the [count | ids] slots and the key slots are laid out as the
    all-gathers deliver them, and pqps_merge_index_slots must reproduce the WHOLE table's leaf order
    (key asc, row desc) -- i.e. exactly what the single-table index path returns."""
    n, world, seed, hdr = 2_000_003, 8, 31, pq.SLOT_HEADER_WORDS
    L = pq.lib()
    whole = pq.SyntheticTable(ctx, n, seed=seed)
    wix = DeviceIndex(ctx, whole, name, signed)
    out = ctx.malloc(4 * n)
    scratch = ctx.malloc(256)
    want, k_want = index_select(ctx, whole, [(wix, lo, hi)], chain, out, n, scratch)
    assert k_want == len(want) > 1000
    wix.free()
    whole.free()

    cap = (n // world + 1026) & ~1                       # a shard can contribute all of its rows
    stride = cap + hdr
    slots = ctx.malloc(world * stride * 4)
    keys = ctx.malloc(world * cap * 8)
    ctx.memset(slots, 0, world * stride * 4)
    for r in range(world):
        s_, c_ = C.c_uint64(), C.c_uint64()
        L.pqps_partition(n, world, r, C.byref(s_), C.byref(c_))
        start, count = s_.value, c_.value
        shard = pq.SyntheticTable(ctx, count, seed=seed, row0=start)
        six = DeviceIndex(ctx, shard, name, signed)
        slot = slots + r * stride * 4
        got, k = index_select(ctx, shard, [(six, lo, hi)], chain, slot + 4 * hdr, cap, scratch, id_base=start)
        assert k <= cap
        ctx.upload(slot, C.byref(C.c_uint64(k)), 8)
        col = pq.column_array([(shard.ptr[name], shard.width[name])])
        pq.check(L.pqps_gather_keys(ctx.h, col, 1 if signed else 0, slot + 4 * hdr, slot, cap, start, keys + r * cap * 8, None), "gather keys")
        ctx.sync()
        six.free()
        shard.free()
    merged = ctx.malloc(4 * world * cap)
    totals = ctx.malloc(16)
    pq.check(L.pqps_merge_index_slots(ctx.h, slots, keys, world, stride, merged, world * cap, totals, None), "index merge")
    t = (C.c_uint64 * 2)()
    ctx.download(t, totals, 16)
    assert t[0] == t[1] == k_want
    got = np.zeros(k_want, dtype=np.uint32)
    ctx.download(got.ctypes.data, merged, 4 * k_want)
    assert np.array_equal(got, want)
    for p_ in (slots, keys, merged, totals, out, scratch):
        ctx.free(p_)
