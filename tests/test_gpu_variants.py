"""Kernel variants that the default settings reach only at sizes too large for a bit-exact oracle run, forced at
small sizes through the shim's switches -- each in its own process, because the shim reads them once.

  * streaming (`nt`) loads for every one of the 34 (W0, W1, W2) shapes, chain and tree form
  * expanders placed among the scan tiles (the default from 4096 groups = 268 M rows on) with the smallest lags,
    so that expanders really are early and wait for their tiles, sums and neighbours
  * the recovery pass: every expander gives up at its first look (spin limit 0), no tile does sum duty -- the
    last expander publishes every sum and expands every group on its own
  * bit masks for the fuller steps (what a context without a list area falls back to)
  * the three forms a step's matches are left in (entries in the slot up to 0 / 64 / 128 matches), under each of the above
  * a dense answer at > 600 M rows (default placement among the tiles; row lists, ranked bit masks and the 64-row
    expansion path): head and tail of the ID list bit-exact against the host twin, the middle by count and order
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import qpelib as q

pq = q.pq
pytestmark = pytest.mark.gpu
DRIVER = str(q.ROOT / "tests" / "variant_driver.py")


def run_driver(env, *args, timeout=900):
    p = subprocess.run([sys.executable, DRIVER, *args], capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, **env), cwd=str(q.ROOT / "tests"))
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), (p.stdout[-2000:], p.stderr[-2000:])


@pytest.mark.parametrize("nt", ["0", "1"])
def test_every_shape_with_and_without_streaming_loads(nt):
    run_driver({"PQPS_NT_LOADS": nt}, "shapes", "70001")


def test_streaming_loads_across_sizes():
    run_driver({"PQPS_NT_LOADS": "1"}, "sizes", "0", "1", "1023", "4097", "100001", str((1 << 21) + 17))


@pytest.mark.parametrize("lag,sum_lag", [("0", "1"), ("1", "1"), ("5", "2"), ("40", "20")])
def test_expanders_among_the_tiles_with_small_lags(lag, sum_lag):
    run_driver({"PQPS_EXPAND_LAG": lag, "PQPS_SUM_LAG": sum_lag}, "sizes", "1", "4097", "65537", "300001",
               str((1 << 21) + 17), "9000001")


def test_recovery_pass_expands_everything_on_its_own():
    env = {"PQPS_EXPAND_SPIN_LIMIT": "0", "PQPS_SUM_LAG": "2000000000", "PQPS_EXPAND_LAG": "0"}
    run_driver(env, "sizes", "1", "65537", "300001", str((1 << 21) + 17))
    run_driver(dict(env, PQPS_EXPAND_LAG="100000"), "sizes", "300001", "5000001")      # ... and when all expanders trail the tiles


@pytest.mark.parametrize("list_max", ["0", "64", "128"])
def test_steps_with_and_without_entries_in_their_slots(list_max):
    """A step with at most `list_max` matches (default 104) leaves them as 16-bit entries in its slot (two 128-byte lines
    from 65 entries on) and the expanders copy them; 0 = every step as a bit mask or a 16-bit list, 128 = up to a full
    slot.  Every shape, the sizes, expanders among the tiles, the recovery pass (which copies step by step), and no
    list area (entries or bit masks only)."""
    env = {"PQPS_LIST_MAX": list_max}
    run_driver(env, "shapes", "70001")
    run_driver(env, "sizes", "1", "4097", "300001", str((1 << 21) + 17))
    run_driver(dict(env, PQPS_EXPAND_LAG="5", PQPS_SUM_LAG="2"), "sizes", "65537", "300001", "9000001")
    run_driver(dict(env, PQPS_EXPAND_SPIN_LIMIT="0", PQPS_SUM_LAG="2000000000", PQPS_EXPAND_LAG="0"), "sizes", "65537", "300001")
    run_driver(dict(env, PQPS_LIST16="0"), "sizes", "4097", "300001", str((1 << 21) + 17))


def test_sparse_steps_without_tiny_words():
    """A step with 1 - 3 matches leaves its entries in a tiny word beside its count word (the group's leader has them with the
    look that settles the group); PQPS_TINY_MAX=0: such steps leave entries in their slots like fuller ones.  Every shape,
    the sizes, expanders among the tiles, the recovery pass -- the defaults of every other test in this file run WITH tiny words."""
    env = {"PQPS_TINY_MAX": "0"}
    run_driver(env, "shapes", "70001")
    run_driver(env, "sizes", "1", "4097", "300001", str((1 << 21) + 17))
    run_driver(dict(env, PQPS_EXPAND_LAG="5", PQPS_SUM_LAG="2"), "sizes", "65537", "300001", "9000001")
    run_driver(dict(env, PQPS_EXPAND_SPIN_LIMIT="0", PQPS_SUM_LAG="2000000000", PQPS_EXPAND_LAG="0"), "sizes", "65537", "300001")
    run_driver({"PQPS_TINY_MAX": "1"}, "sizes", "4097", "300001")


def test_fuller_steps_as_bit_masks_when_there_is_no_list_area():
    """Steps with more than 104 matches leave 16-bit row lists in the context's list area (2 bytes per table row); a
    context that cannot get one (or PQPS_LIST16=0) keeps the bit masks and the ranking expansion for them."""
    run_driver({"PQPS_LIST16": "0"}, "shapes", "70001")
    run_driver({"PQPS_LIST16": "0"}, "sizes", "4097", "300001", str((1 << 21) + 17))
    run_driver({"PQPS_LIST16": "0", "PQPS_EXPAND_LAG": "5", "PQPS_SUM_LAG": "2"}, "sizes", "300001", "9000001")


def test_index_probes_evaluated_instead_of_copied():
    """pqps_index_select copies a probe's rows when the WHERE is the probed comparison itself; PQPS_INDEX_COPY=0 sends every
    probe through the gather filter: the reference's SELECT goldens (both index configurations) once more that way."""
    p = subprocess.run([sys.executable, "-m", "pytest", str(q.ROOT / "tests" / "test_gpu_parity.py"), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "select_matches_reference"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, PQPS_INDEX_COPY="0"), cwd=str(q.ROOT))
    assert p.returncode == 0 and " passed" in p.stdout, (p.stdout[-2000:], p.stderr[-2000:])


def test_epoch_wrap_of_the_handoff_words():
    """The hand-off words carry a 16-bit epoch; after 65535 ID queries on a scratch it starts over and the tagged arrays
    are zeroed.  A stream of queries gets there every few seconds; here the epoch starts 5 queries before the wrap."""
    run_driver({"PQPS_EPOCH_START": "65530"}, "wrap", "300001")
    run_driver({"PQPS_EPOCH_START": "65533", "PQPS_EXPAND_LAG": "2", "PQPS_SUM_LAG": "1"}, "wrap", "1100001")      # expanders among the tiles


def test_dense_answer_above_600m_rows():
    n = 640_000_003
    seed = 99
    ctx = pq.Context(0)
    dev = pq.SyntheticTable(ctx, n, seed=seed, columns=["sudo_used", "risk_level"])
    ids_dev, cnt_dev = ctx.malloc(4 * n), ctx.malloc(64)
    flags_dev = ctx.malloc(n + 4096)
    try:
        for chain in ([("sudo_used", "=", "FALSE")], [("risk_level", "<", "3")], [("risk_level", ">", "3")]):
            pred, cols, nc, _ = dev.bind(chain)
            pq.check(pq.lib().pqps_filter_scan(ctx.h, cols, nc, n, 0, C.byref(pred), ids_dev, n, cnt_dev, None))
            ctx.sync()
            k = C.c_uint64()
            ctx.download(C.byref(k), cnt_dev, 8)
            k = k.value
            pq.check(pq.lib().pqps_filter_count(ctx.h, cols, nc, n, C.byref(pred), cnt_dev, None))
            ctx.sync()
            c2 = C.c_uint64()
            ctx.download(C.byref(c2), cnt_dev, 8)
            assert k == c2.value and 0 < k < n
            m = 2_000_000
            head = q.HostSynth(m, seed=seed).oracle_scan(chain)
            tail = q.HostSynth(m, seed=seed, row0=n - m).oracle_scan(chain, id_base=n - m).astype(np.uint32)
            got = np.zeros(len(head) + 1, dtype=np.uint32)
            ctx.download(got.ctypes.data, ids_dev, got.nbytes)
            assert np.array_equal(got[:-1], head) and got[-1] >= m
            got = np.zeros(len(tail) + 1, dtype=np.uint32)
            ctx.download(got.ctypes.data, ids_dev + 4 * (k - len(tail) - 1), got.nbytes)
            assert np.array_equal(got[1:], tail) and got[0] < n - m
            # the middle: strictly ascending over a 64 M-ID window that straddles many groups
            w = min(k, 64_000_000)
            mid = np.zeros(w, dtype=np.uint32)
            ctx.download(mid.ctypes.data, ids_dev + 4 * ((k - w) // 2), mid.nbytes)
            assert np.all(mid[1:] > mid[:-1])
        # DELETE flags of a scan that outgrows the Infinity Cache (streaming-load flag kernel)
        chain = [("sudo_used", "=", "TRUE")]
        pred, cols, nc, _ = dev.bind(chain)
        pq.check(pq.lib().pqps_filter_flags(ctx.h, cols, nc, n, C.byref(pred), flags_dev, cnt_dev, None))
        ctx.sync()
        kf = C.c_uint64()
        ctx.download(C.byref(kf), cnt_dev, 8)
        f = np.zeros(4_000_000, dtype=np.uint8)
        ctx.download(f.ctypes.data, flags_dev + n - len(f), len(f))
        want = np.zeros(len(f), dtype=np.uint8)
        want[q.HostSynth(len(f), seed=seed, row0=n - len(f)).oracle_scan(chain)] = 1
        assert np.array_equal(f, want)
        pq.check(pq.lib().pqps_filter_count(ctx.h, cols, nc, n, C.byref(pred), cnt_dev, None))
        ctx.sync()
        kc = C.c_uint64()
        ctx.download(C.byref(kc), cnt_dev, 8)
        assert kf.value == kc.value
    finally:
        for p in (ids_dev, cnt_dev, flags_dev):
            ctx.free(p)
        dev.free()
        ctx.close()
