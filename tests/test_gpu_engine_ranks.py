"""One process per GPU behind the ENGINE API (round 4): initializeEngineSyntheticRankHIP + hipEngineJoinRanksHIP make the
engine's SELECT / COUNT the whole table's -- the shard's scan + the all-gatherv of the row numbers / the all-reduce of the
counts over RCCL, issued by executeQuery{Select,Count}AsyncHIP from C (the reference: QPEMPI.c:145-155 + engine/mpi/
executeEngine-mpi.c:703-768).  On this pool's one-GPU boxes the ranks are THREADS of one process and the nccl* entry
points come from tests/loopback/libloopback_rccl.so; what runs is the product's engine + exchange code with worlds of 2 and
3: several tickets in flight per rank, results awaited out of order, COUNT(*) in between, the compact wire form, and the C
bench loop (hipEngineBench) on every rank at once.  Every rank's answer must be the oracle's for the WHOLE table."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import qpelib as q


@pytest.fixture(autouse=True, scope="module")
def _compact_lists_of_any_size():
    """The product keeps lists below 32 768 IDs as u32 on the wire; these cases want the compact form at test sizes (the workers inherit it)."""
    old = os.environ.get("PQPS_WIRE_MIN_IDS")
    os.environ["PQPS_WIRE_MIN_IDS"] = "0"
    yield
    if old is None:
        os.environ.pop("PQPS_WIRE_MIN_IDS", None)
    else:
        os.environ["PQPS_WIRE_MIN_IDS"] = old

pq = q.pq
pytestmark = pytest.mark.gpu
LOOPBACK = q.ROOT / "tests" / "loopback" / "libloopback_rccl.so"

WORKER = textwrap.dedent("""
    import ctypes as C, json, os, sys, threading, traceback
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import qpelib as q
    pq = q.pq
    L = pq.lib()
    B = pq.bench_lib()
    world = int(os.environ["WORLD"])
    n = int(os.environ["ROWS"])
    queries = {k: q.chain_from_jsonable(v) for k, v in json.loads(os.environ["QUERIES"]).items()}
    gate = threading.Barrier(world)
    ident = [None]
    out = [dict() for _ in range(world)]

    def rank_main(rank):
        try:
            eng = pq.HipEngine.synthetic_rank(n, world, rank, seed=21)
            assert L.hipEngineLanes(eng.e) >= 6
            if rank == 0:
                ident[0] = pq.HipEngine.rccl_id(LOOPBACK)
            gate.wait()
            eng.join_ranks(LOOPBACK, ident[0])
            ctx = pq.Context(0)
            res = {}
            # three tickets in flight, awaited out of issue order; every rank issues the same queries in the same order
            names = list(queries)
            tickets = [(name, eng.select_async(queries[name])) for name in names[:3]]
            order = [1, 0, 2]
            for i in order:
                name, tk = tickets[i]
                k, r = eng.await_ticket(tk)
                ids = np.zeros(max(k, 1), dtype=np.uint32)
                if k > 0:
                    ctx.download(ids.ctypes.data, r.ids_dev, 4 * k)
                f = os.path.join(os.environ["OUT_DIR"], f"{rank}_{name}.npy")
                np.save(f, ids[:max(k, 0)])
                res[name] = [int(k), f, int(r.shard_count[0]), list(eng.ticket_checksum(tk))]
            for _, tk in tickets:
                eng.release_ticket(tk)
            # COUNT(*) between SELECTs, and the rest of the queries one by one
            for name in names[3:]:
                tc = eng.select_async(queries[name], count_only=True)
                tk = eng.select_async(queries[name])
                kc, _ = eng.await_ticket(tc)
                k, r = eng.await_ticket(tk)
                ids = np.zeros(max(k, 1), dtype=np.uint32)
                if k > 0:
                    ctx.download(ids.ctypes.data, r.ids_dev, 4 * k)
                f = os.path.join(os.environ["OUT_DIR"], f"{rank}_{name}.npy")
                np.save(f, ids[:max(k, 0)])
                res[name] = [int(k), f, int(r.shard_count[0]), list(eng.ticket_checksum(tk)), int(kc)]
                eng.release_ticket(tc)
                eng.release_ticket(tk)
            res["_wire"] = list(eng.wire_bytes())
            # the C bench loop on every rank at once (what bench.py --gpus N times): 1 thread x 3 tickets in flight
            wl = pq.WhereList(queries[names[0]])
            arr = (C.POINTER(pq.EngineS) * 1)(eng.e)
            br = pq.BenchResult()
            br.want_checksum = 1
            gate.wait()
            rc = B.hipEngineBench(arr, 1, wl.ptr, 0, 1, 3, 4, 24, C.byref(br))
            res["_bench"] = [int(rc), int(br.matches), int(br.mismatches), int(br.have_checksum), int(br.checksum[0]), int(br.checksum[1])]
            # an index-mode or multi-pass query is refused, the engine stays usable
            gate.wait()
            eng.leave_ranks()
            local = eng.select_ids(queries[names[0]])                       # back to a local engine: this rank's rows only
            res["_local_after_leave"] = [len(local), int(local[0]) if local else -1]
            out[rank] = res
            gate.wait()
            ctx.close()
            eng.close()
        except BaseException:
            traceback.print_exc()
            sys.stderr.flush()
            os._exit(3)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads: t.start()
    for t in threads: t.join()
    with open(os.environ["OUT_FILE"], "w") as f:
        json.dump(out, f)
    print("OK")
""")

QUERIES = {
    "q_a": [("risk_level", ">", "3")],                                                       # a few per cent: compact on the wire
    "s1": [("sudo_used", "=", "FALSE"), "AND", ("user_name", "=", "student1030")],           # sparse: u32 on the wire
    "r1": [("risk_level", ">", "1")],                                                        # 43 % of the rows
    "last": [("command_id", ">=", "1400000")],                                               # all matches on the last rank(s)
    "none": [("risk_level", ">", "9")],
    "nested": [("sudo_used", "=", "TRUE"), "OR", [("risk_level", "=", "5"), "AND", ("shell_type", "=", "bash")]],
}


def numpy_checksum(ids):
    v = np.asarray(ids, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return [int(v.sum(dtype=np.uint64)), int((v * (np.arange(len(v), dtype=np.uint64) * np.uint64(2) + np.uint64(1))).sum(dtype=np.uint64))]


@pytest.mark.parametrize("world", [2, 3])
def test_rank_engines_answer_for_the_whole_table(tmp_path, world):
    assert LOOPBACK.exists(), "build it first: make -C tests/loopback (python __graft_entry__.py does)"
    rows = 1_500_001
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {str(q.ROOT)!r}\nLOOPBACK = {str(LOOPBACK)!r}\n" + WORKER)
    env = dict(os.environ, WORLD=str(world), ROWS=str(rows), QUERIES=json.dumps({k: q.chain_to_jsonable(v) for k, v in QUERIES.items()}),
               OUT_FILE=str(tmp_path / "out.json"), OUT_DIR=str(tmp_path), OMP_NUM_THREADS="1", PQPS_EXCHANGE_TIMEOUT_S="60")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), (p.stdout[-1500:], p.stderr[-3000:])
    got = json.loads((tmp_path / "out.json").read_text())
    mg = q.pq_merge()
    host = q.HostSynth(rows, seed=21)
    first = list(QUERIES)[0]
    for r in range(world):
        start, count = mg.shard_rows(rows, world, r)
        for name, chain in QUERIES.items():
            want = host.oracle_scan(chain)
            entry = got[r][name]
            mine = int(((want >= start) & (want < start + count)).sum())
            assert entry[0] == len(want) and entry[2] == mine, (name, r, entry[:3], len(want), mine)
            assert np.array_equal(np.load(entry[1]), want), (name, r)
            assert entry[3] == numpy_checksum(want), (name, r)                # the device-side checksum bench.py relies on
            if len(entry) > 4:
                assert entry[4] == len(want), (name, r)                       # COUNT(*): the all-reduced count
        wire, as_u32 = got[r]["_wire"]
        assert 0 < wire < 0.62 * as_u32, (r, wire, as_u32)                     # the dense answers dominate: about half the bytes
        want0 = host.oracle_scan(QUERIES[first])
        assert got[r]["_bench"] == [0, len(want0), 0, 1] + numpy_checksum(want0), (r, got[r]["_bench"])
        local0 = want0[(want0 >= start) & (want0 < start + count)]
        assert got[r]["_local_after_leave"] == [len(local0), int(local0[0]) if len(local0) else -1], r
