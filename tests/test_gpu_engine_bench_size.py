"""Engine-level parity AT BENCH SIZE (round 4): bench.py's `value` comes from initializeEngineSyntheticHIP at 100 M rows
with asynchronous tickets -- three in flight over two engines alternated, as host/engineBench.c issues them.  This test
drives exactly that shape for S1, Q_A, Q_B and Q_C and checks every awaited ticket's list (downloaded from
result.ids_dev) the way test_full_size_properties checks the shim: strictly ascending, == COUNT(*) through the engine, the
first 3 M rows bit-exact against the oracle, sampled membership row by row against the host twin + oracle, and the
device-side checksum (hipQueryChecksumHIP, what bench.py compares) == numpy's over the downloaded list.  Once on one
device and once with the table split over two shards on one card (PQPS_DEVICES=0,0: the one-process merge on the
device); each in its own process (the device list is read when the engine is built).  Finally the C bench loop itself
(hipEngineBench) over the same engines: its last ticket's checksum == the checksum of the list checked above."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import qpelib as q

pq = q.pq
pytestmark = pytest.mark.gpu

WORKER = textwrap.dedent("""
    import ctypes as C, json, os, sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import qpelib as q
    pq = q.pq
    L, B = pq.lib(), pq.bench_lib()
    n, seed = int(os.environ["ROWS"]), 0x5EED
    queries = {k: q.chain_from_jsonable(v) for k, v in json.loads(os.environ["QUERIES"]).items()}
    engines = [pq.HipEngine.synthetic(n, seed=seed) for _ in range(2)]
    assert all(L.hipEngineShards(e.e, None, 0) == int(os.environ["SHARDS"]) for e in engines)
    ctx = pq.Context(0)
    rng = np.random.default_rng(1)

    def numpy_checksum(ids):
        v = np.asarray(ids, dtype=np.uint64)
        with np.errstate(over="ignore"):
            return (int(v.sum(dtype=np.uint64)), int((v * (np.arange(len(v), dtype=np.uint64) * np.uint64(2) + np.uint64(1))).sum(dtype=np.uint64)))

    def check(name, chain, tk, eng):
        k, res = eng.await_ticket(tk)
        assert k > 0, name
        ids = np.zeros(k, dtype=np.uint32)
        # (several shards: the gathered list lies on shard 0's device = device 0 here)
        ctx.download(ids.ctypes.data, res.ids_dev, 4 * k)
        assert np.all(ids[1:] > ids[:-1]), name                                  # strictly ascending
        assert eng.ticket_checksum(tk) == numpy_checksum(ids), name             # the device-side checksum bench.py relies on
        assert sum(int(res.shard_count[s]) for s in range(res.n_shards)) == k, name
        m = 3_000_000                                                           # a prefix of the table bit-exact against the oracle
        want = q.HostSynth(m, seed=seed).oracle_scan(chain)
        assert np.array_equal(ids[:len(want)], want) and (len(want) == k or ids[len(want)] >= m), name
        sample = np.unique(np.concatenate([rng.integers(0, n, 1500), ids[rng.integers(0, k, 1500)], [0, 1, 4095, 4096, n - 1, n - 4097]]))
        member = np.isin(sample, ids)
        for row, mb in zip(sample, member):                                     # membership of sampled rows, row by row
            assert (len(q.HostSynth(1, seed=seed, row0=int(row)).oracle_scan(chain)) == 1) == bool(mb), (name, int(row))
        return k, numpy_checksum(ids)

    out = {}
    # as hipEngineBench issues them: ticket k goes to engine k % 2, three outstanding, the oldest awaited first
    plan = [(name, rep) for name in queries for rep in range(3)]
    ring = []
    for i, (name, rep) in enumerate(plan):
        if len(ring) == 3:
            nm, tk, eng = ring.pop(0)
            k, sums = check(nm, queries[nm], tk, eng)
            assert out.setdefault(nm, [k, list(sums)]) == [k, list(sums)], nm    # every repetition, both engines: the same list
            eng.release_ticket(tk)
        eng = engines[i % 2]
        ring.append((name, eng.select_async(queries[name]), eng))
    while ring:
        nm, tk, eng = ring.pop(0)
        k, sums = check(nm, queries[nm], tk, eng)
        assert out.setdefault(nm, [k, list(sums)]) == [k, list(sums)], nm
        eng.release_ticket(tk)
    for name, chain in queries.items():                                          # COUNT(*) through the engine agrees
        assert engines[0].count(chain) == out[name][0], name
    # the C loop bench.py times: 1 thread x 3 tickets over the 2 engines; its last list's checksum is the one checked above
    arr = (C.POINTER(pq.EngineS) * 2)(*[e.e for e in engines])
    for name, chain in queries.items():
        wl = pq.WhereList(chain)
        br = pq.BenchResult()
        br.want_checksum = 1
        assert B.hipEngineBench(arr, 2, wl.ptr, 0, 1, 3, 5, 30, C.byref(br)) == 0, name
        assert br.mismatches == 0 and br.matches == out[name][0] and br.have_checksum, name
        assert [int(br.checksum[0]), int(br.checksum[1])] == out[name][1], name
    # more tickets than the engines have lanes cannot be outstanding: refused, nothing runs, nothing hangs
    br = pq.BenchResult()
    assert B.hipEngineBench(arr, 2, pq.WhereList(queries["S1"]).ptr, 0, 2, 16, 1, 4, C.byref(br)) == -2
    for e in engines:
        e.close()
    ctx.close()
    with open(os.environ["OUT_FILE"], "w") as f:
        json.dump(out, f)
    print("OK")
""")

QUERIES = {
    "S1": [("sudo_used", "=", "FALSE"), "AND", ("user_name", "=", "student1030")],
    "Q_A": [("risk_level", ">", "3")],
    "Q_B": [("sudo_used", "=", "TRUE"), "AND", ("risk_level", ">", "2")],
    "Q_C": [("exit_code", "!=", "0"), "AND", ("user_id", ">=", "1500"), "OR", ("risk_level", "=", "5")],
}


@pytest.mark.parametrize("devices,shards", [(None, 1), ("0,0", 2)])
def test_engine_tickets_at_bench_size_match_the_oracle(tmp_path, devices, shards):
    rows = 100_000_000
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {str(q.ROOT)!r}\n" + WORKER)
    env = dict(os.environ, ROWS=str(rows), QUERIES=json.dumps({k: q.chain_to_jsonable(v) for k, v in QUERIES.items()}),
               OUT_FILE=str(tmp_path / "out.json"), SHARDS=str(shards), OMP_NUM_THREADS="8")
    env.pop("PQPS_DEVICES", None)
    if devices:
        env["PQPS_DEVICES"] = devices
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=1100)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), (p.stdout[-1500:], p.stderr[-3000:])
    got = json.loads((tmp_path / "out.json").read_text())
    # the counts of the seeded table (the same numbers bench.py reports as matches_total at this size)
    assert got["S1"][0] == 6_691 and got["Q_A"][0] == 4_397_793 and got["Q_B"][0] == 6_684_885 and got["Q_C"][0] == 3_871_923
