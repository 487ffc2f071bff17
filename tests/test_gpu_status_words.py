"""A launch whose bounded waits ran out is reported to whoever awaits THAT launch (round-3 advice): the status words of a
lane context carry the launch's epoch, pqps_qstream_wait looks at its own slot's launches only.  Two slots of a query
stream share a lane (depth 4, two lanes); the test marks one slot's launch as failed the way the kernel would
(pqps_qstream_test_fail_slot writes its epoch into the lane's status words) and awaits the slots in an order in which the
old single sticky word would have blamed the wrong query."""
import ctypes as C

import numpy as np
import pytest

import qpelib as q

pq = q.pq
pytestmark = pytest.mark.gpu


def test_a_failed_launch_is_reported_for_its_own_slot_only():
    L = pq.lib()
    ctx = pq.Context(0)
    n = 700_003
    dev = pq.SyntheticTable(ctx, n, seed=5)
    host = q.HostSynth(n, seed=5)
    chain = [("risk_level", ">", "3")]
    want = host.oracle_scan(chain)
    pred, cols, nc, _ = dev.bind(chain)
    qs = C.c_void_p()
    pq.check(L.pqps_qstream_create(ctx.h, 4, C.byref(qs)), "qstream")
    ids = [ctx.malloc(4 * n) for _ in range(4)]
    cnt = [ctx.malloc(64) for _ in range(4)]

    def issue(slot):
        pq.check(L.pqps_qstream_scan_slot(qs, slot, cols, nc, n, 0, C.byref(pred), ids[slot], n, cnt[slot], None), "issue")

    def answer(slot):
        k = C.c_uint64()
        ctx.download(C.byref(k), cnt[slot], 8)
        out = np.zeros(k.value, dtype=np.uint32)
        ctx.download(out.ctypes.data, ids[slot], 4 * k.value)
        return out

    for slot in range(4):                                                   # slots 0 and 2 run on lane 0, 1 and 3 on lane 1
        issue(slot)
    pq.check(L.pqps_qstream_test_fail_slot(qs, 2), "mark")
    assert L.pqps_qstream_wait(qs, 0) == 0                                   # same lane as slot 2, awaited first: not its failure
    assert L.pqps_qstream_wait(qs, 1) == 0 and L.pqps_qstream_wait(qs, 3) == 0
    rc = L.pqps_qstream_wait(qs, 2)
    assert rc != 0 and b"gave up" in L.pqps_last_error()                     # reported for the slot whose launch it was ...
    assert L.pqps_qstream_wait(qs, 2) == 0                                   # ... once
    for slot in (0, 1, 3):
        assert np.array_equal(answer(slot), want), slot
    # the lane starts over (hand-off words reset before its next ID launch) and goes on answering
    for rep in range(3):
        for slot in range(4):
            issue(slot)
        for slot in (3, 2, 1, 0):
            assert L.pqps_qstream_wait(qs, slot) == 0
            assert np.array_equal(answer(slot), want), (rep, slot)
    # a COUNT in a slot has no ID launch to blame
    pq.check(L.pqps_qstream_count_slot(qs, 1, cols, nc, n, C.byref(pred), cnt[1], None), "count")
    assert L.pqps_qstream_test_fail_slot(qs, 1) != 0
    assert L.pqps_qstream_wait(qs, 1) == 0
    pq.check(L.pqps_qstream_sync(qs), "sync")
    pq.check(L.pqps_qstream_destroy(qs), "destroy")
    for p in ids + cnt:
        ctx.free(p)
    dev.free()
    ctx.close()
