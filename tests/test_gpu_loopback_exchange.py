"""The shim-driven exchange (pqps_exchange_prepare / _connect / _select / _count / _result) with a world of TWO and
THREE on a one-GPU box: the ranks are threads of one process and the ten nccl* calls the exchange resolves with
dlsym come from tests/loopback/libloopback_rccl.so (a stand-in with RCCL's stream semantics, see its header) instead
of librccl.so.  What runs is the product's own world > 1 code: capacities all-gather at connect, the per-query sizes
all-gather, displacements, the held-back send / recv group (hold = 2 with a ring of 5 or more), ragged capacities, a
gathered list that has to grow, a rank without rows, a local slot that overflows, COUNT(*) all-reduces in between.
Every rank's gathered list and every count must equal the oracle's answer for the whole table
(engine/mpi/executeEngine-mpi.c:745-765 is the shape being reproduced).

With two or more GPUs the ranks spread over them (peer copies); tests/test_gpu_two_ranks.py is the same through the
real RCCL, one process per GPU, and needs two cards.

Round 4: the payload travels in the compact wire form wherever that is smaller (every case below that has more than a
few IDs per 65 536 rows: Q_A, Q_B, risk_level > 1, the dense and the skewed ones), the bytes received are checked
against the model, and one run repeats the cases with PQPS_EXCHANGE_COMPACT=0.  A STALL injected into the loopback (one
rank's payload group never finishes on the device, the way a stuck RCCL kernel would not) must end in every rank's
bounded wait running out, the communicator aborted, and all ranks finishing the query on a fallback -- within seconds."""
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
    pq, mg = q.pq, q.pq_merge()
    L = pq.lib()
    path = LOOPBACK.encode()
    world = int(os.environ["WORLD"])
    cases = json.loads(os.environ["CASES"])
    ndev = L.pqps_device_count()
    out = [dict() for _ in range(world)]
    gate = threading.Barrier(world)
    ids = {}

    def rank_main(rank):
        try:
            ctx = pq.Context(rank % ndev)
            for name, (n, chain, cap, ring, plan) in cases.items():
                chain = q.chain_from_jsonable(chain)
                start, count = mg.shard_rows(n, world, rank)
                dev = pq.SyntheticTable(ctx, count, seed=21, row0=start)
                if rank == 0:
                    ident = C.create_string_buffer(128)
                    pq.check(L.pqps_exchange_unique_id(path, ident), "unique id")
                    ids[name] = ident.raw
                h = C.c_void_p()
                pq.check(L.pqps_exchange_prepare(ctx.h, path, world, rank, int(cap if cap else count + 16), ring, C.byref(h)), "prepare")
                gate.wait()
                pq.check(L.pqps_exchange_connect(h, C.create_string_buffer(ids[name], 128)), "connect")
                xch = mg.ShardExchange(pq, ctx, h, world, rank, ring)
                pred, cols, nc, _ = dev.bind(chain)
                got = []
                for step in plan:                                            # "s<slot>" select, "c<slot>" count, "r<slot>" result, "k<slot>" count result
                    op, slot = step[0], int(step[1:])
                    if op == "s":
                        xch.select(cols, nc, count, start, C.byref(pred), slot, None)
                    elif op == "c":
                        xch.count(cols, nc, count, C.byref(pred), slot, None)
                    elif op == "r":
                        try:
                            arr, local = xch.result(slot)
                            f = os.path.join(os.environ["OUT_DIR"], f"{rank}_{name}_{len(got)}.npy")
                            np.save(f, arr)
                            got.append([f, local])
                        except pq.PqpsError as e:
                            got.append("error: " + str(e))
                    elif op == "k":
                        total, mine = xch.count_result(slot)
                        got.append(["count", total, mine])
                    elif op == "y":
                        xch.sync()
                eg = (C.c_uint64 * 3)()
                L.pqps_exchange_eager(h, eg, 0)
                got.append(["eager", int(eg[0]), int(eg[1]), int(eg[2])])
                wb = (C.c_uint64 * 2)()
                L.pqps_exchange_wire_bytes(h, wb, 0)
                got.append(["wire", int(wb[0]), int(wb[1])])
                out[rank][name] = got
                gate.wait()                                                  # nobody tears down while a peer is still in a call
                xch.close()
                dev.free()
            ctx.close()
        except BaseException:
            traceback.print_exc()
            sys.stderr.flush()
            os._exit(3)                                                      # the peers would wait for this rank forever

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads: t.start()
    for t in threads: t.join()
    with open(os.environ["OUT_FILE"], "w") as f:
        json.dump(out, f)
    print("OK")
""")

N = 3_000_001
# name: (rows, chain, slot capacity (0: the shard's rows + 16), ring, plan)
CASES = {
    # a ring of 6: the payload of a query goes out three calls later (hold = 2); results out of issue order
    "q_b_ring6": (N, [("sudo_used", "=", "TRUE"), "AND", ("risk_level", ">", "2")], 0, 6,
                  ["s0", "s1", "s2", "s3", "s4", "s5", "r2", "r0", "s0", "s1", "r5", "r0", "r1", "y0"]),
    "first_only": (N, [("command_id", "<=", "1000000")], 0, 2, ["s0", "s1", "r0", "s0", "r1", "r0"]),       # all matches on rank 0
    "last_only": (N, [("command_id", ">", "2900000")], 0, 3, ["s0", "s1", "s2", "r2", "r0", "r1"]),          # ... on the last rank
    "dense_grows": (N, [("sudo_used", "=", "FALSE")], 0, 2, ["s0", "r0", "s1", "r1"]),                      # > 2^20 IDs: the gathered list grows
    "none": (N, [("risk_level", ">", "9")], 0, 1, ["s0", "r0", "s0", "r0"]),                                 # ring of one, nobody has a match: no group at all
    "one_row": (1, [("risk_level", ">=", "0")], 0, 1, ["s0", "r0"]),                                         # the other ranks own no rows
    "counts_between": (N, [("risk_level", ">", "3")], 0, 4, ["s0", "c1", "s2", "k1", "c3", "r0", "r2", "k3"]),
    "overflow": (N, [("risk_level", ">=", "1")], 4096, 2, ["s0", "r0", "s1", "r1"]),                         # every rank's own slot is too small
    "ring5_long": (700_001, [("sudo_used", "=", "TRUE")], 0, 5,
                   ["s0", "s1", "s2", "s3", "s4", "s0", "s1", "s2", "s3", "s4", "r0", "r1", "r2", "r3", "r4"]),
    # the north star's answer shapes over shards of 15 - 23 groups of 65 536 rows: a few per cent (Q_A), 43 % of the rows
    "q_a_compact": (N, [("risk_level", ">", "3")], 0, 6, ["s0", "s1", "s2", "r0", "s3", "r2", "r1", "r3"]),
    "r1_dense": (N, [("risk_level", ">", "1")], 0, 5, ["s0", "s1", "r1", "r0"]),
    "s1_sparse": (N, [("sudo_used", "=", "FALSE"), "AND", ("user_name", "=", "student1030")], 0, 5, ["s0", "s1", "r0", "r1"]),
}


@pytest.mark.parametrize("world,compact,floor", [(2, True, "0"), (3, True, "0"), (2, False, "0"), (2, True, None), (8, True, None),
                                                 (20, True, "0")])      # 19 compact payloads per query: more than one rebuild launch takes (16)
def test_exchange_world_of_several_through_the_loopback(tmp_path, monkeypatch, world, compact, floor):
    # floor None: the product's own rules -- only lists of 32 768 IDs and more travel compact (s1_sparse does not), and an answer of up to
    # 16 384 IDs per rank arrives with the sizes (one collective); floor "0": compact lists of any size, no eager blocks
    if floor is None:
        monkeypatch.delenv("PQPS_WIRE_MIN_IDS")
        monkeypatch.delenv("PQPS_EXCHANGE_EAGER_IDS", raising=False)
    else:
        monkeypatch.setenv("PQPS_EXCHANGE_EAGER_IDS", "0")
    assert LOOPBACK.exists(), "build it first: make -C tests/loopback (python __graft_entry__.py does)"
    picked = CASES if world <= 8 else {k: CASES[k] for k in ("q_a_compact", "dense_grows", "s1_sparse", "none", "first_only")}
    cases = {k: (v[0], q.chain_to_jsonable(v[1]), v[2], v[3], v[4]) for k, v in picked.items()}
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {str(q.ROOT)!r}\nLOOPBACK = {str(LOOPBACK)!r}\n" + WORKER)
    env = dict(os.environ, WORLD=str(world), CASES=json.dumps(cases), OUT_FILE=str(tmp_path / "out.json"), OUT_DIR=str(tmp_path),
               OMP_NUM_THREADS="1", PQPS_EXCHANGE_COMPACT="1" if compact else "0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), (p.stdout[-1500:], p.stderr[-3000:])
    got = json.loads((tmp_path / "out.json").read_text())
    mg = q.pq_merge()
    for name, (rows, chain, cap, ring, plan) in picked.items():
        want = q.HostSynth(rows, seed=21).oracle_scan(chain)
        for r in range(world):
            start, count = mg.shard_rows(rows, world, r)
            mine = int(((want >= start) & (want < start + count)).sum())
            results = got[r][name]
            wire = results.pop()                                            # ["wire", bytes as they travelled, bytes as u32 IDs]
            assert wire[0] == "wire"
            eager = results.pop()                                           # ["eager", SELECTs done in the one collective, SELECTs done, room]
            assert eager[0] == "eager"
            if floor is None:
                caps, counts = [], []
                for p in range(world):
                    ps, pc = mg.shard_rows(rows, world, p)
                    caps.append(((cap if cap else pc + 16) + 1) & ~1)
                    counts.append(int(((want >= ps) & (want < ps + pc)).sum()))
                room = min(16384, min(caps), (1 << 20) // world) & ~1
                room = room if room >= 256 else 0                           # (one_row: the smallest slot holds 16 IDs -- not worth a block)
                assert eager[3] == room, (name, eager, room)
                if room and max(counts) <= room:
                    assert eager[1] == eager[2] > 0, (name, r, eager)       # every answer of this case arrived with its sizes
                else:
                    assert eager[1] == 0 and eager[2] > 0, (name, r, eager)  # none did: they took the two steps
            else:
                assert eager[1] == 0 and eager[3] == 0, (name, eager)
            # the model: every finished SELECT brings in each peer's list -- compact where that is smaller
            selects = sum(1 for s in plan if s[0] == "s")
            in_wire = in_u32 = 0
            all_pay = True
            for p in range(world):
                if p == r:
                    continue
                ps, pc = mg.shard_rows(rows, world, p)
                k = int(((want >= ps) & (want < ps + pc)).sum())
                if cap:
                    k = min(k, cap + cap % 2)
                in_u32 += 4 * k
                in_wire += mg.wire_bytes(pc, k) if (compact and mg.wire_pays(pc, k)) else 4 * k
                all_pay = all_pay and bool(mg.wire_pays(pc, k))
            assert wire[1:] == [selects * in_wire, selects * in_u32], (name, r, wire, selects, in_wire, in_u32)
            if compact and all_pay and name in ("q_a_compact", "r1_dense", "dense_grows", "q_b_ring6") and in_u32:
                assert in_wire < 0.55 * in_u32, (name, in_wire, in_u32)      # (worlds of 2 / 3: 44 000 IDs and more per rank, above the floor too)
            if name in ("r1_dense", "dense_grows"):
                assert all_pay or not compact or world > 8
            if floor is None and name == "s1_sparse":
                assert in_wire == in_u32, (name, in_wire, in_u32)            # below the floor: as they are
            assert len(results) == sum(1 for s in plan if s[0] in "rk"), (name, r)
            for entry in results:
                if name == "overflow":
                    assert isinstance(entry, str) and "overflow" in entry, (name, r, str(entry)[:200])
                elif entry[0] == "count":
                    assert entry[1] == len(want) and entry[2] == mine, (name, r)
                else:
                    arr = np.load(entry[0])
                    assert entry[1] == mine and np.array_equal(arr, want), (name, r, len(arr), len(want))


STALL_WORKER = textwrap.dedent("""
    import ctypes as C, json, os, sys, threading, time, traceback
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import qpelib as q
    pq, mg = q.pq, q.pq_merge()
    L = pq.lib()
    path = LOOPBACK.encode()
    world = int(os.environ["WORLD"])
    n = int(os.environ["ROWS"])
    chain = q.chain_from_jsonable(json.loads(os.environ["CHAIN"]))
    ndev = L.pqps_device_count()
    gate = threading.Barrier(world)
    ident = [None]
    report = [dict() for _ in range(world)]
    lists = [None] * world
    failed = [False] * world

    def rank_main(rank):
        try:
            ctx = pq.Context(rank % ndev)
            start, count = mg.shard_rows(n, world, rank)
            dev = pq.SyntheticTable(ctx, count, seed=21, row0=start)
            if rank == 0:
                buf = C.create_string_buffer(128)
                pq.check(L.pqps_exchange_unique_id(path, buf), "unique id")
                ident[0] = buf.raw
            h = C.c_void_p()
            pq.check(L.pqps_exchange_prepare(ctx.h, path, world, rank, count + 16, 6, C.byref(h)), "prepare")
            gate.wait()
            pq.check(L.pqps_exchange_connect(h, C.create_string_buffer(ident[0], 128)), "connect")
            xch = mg.ShardExchange(pq, ctx, h, world, rank, 6)
            pred, cols, nc, _ = dev.bind(chain)
            t0 = time.monotonic()
            err = None
            try:
                for k in range(12):                                          # the payload group of query 0 (op 4 of the communicator) stalls
                    xch.select(cols, nc, count, start, C.byref(pred), k % 6, None)
                xch.sync()
                xch.result(5)
            except pq.PqpsError as e:
                err = str(e)
            report[rank]["error"] = err
            report[rank]["seconds_to_error"] = time.monotonic() - t0
            failed[rank] = err is not None
            # every later call fails at once (no second wait)
            t1 = time.monotonic()
            try:
                xch.select(cols, nc, count, start, C.byref(pred), 0, None)
                report[rank]["second_call"] = "ok"
            except pq.PqpsError as e:
                report[rank]["second_call"] = str(e)
            report[rank]["second_call_seconds"] = time.monotonic() - t1
            gate.wait()                                                      # the agreement every rank reaches (bench.py: an all-reduce over torch.distributed)
            everyone_ok = not any(failed)
            t2 = time.monotonic()
            xch.close()                                                      # tearing down a dead exchange must not hang either
            report[rank]["close_seconds"] = time.monotonic() - t2
            if not everyone_ok:
                # the fallback: the rank's own scan, the lists put together by the host (here: the threads' shared memory)
                ids_dev, cnt_dev = ctx.malloc(4 * (count + 16)), ctx.malloc(64)
                pq.check(L.pqps_filter_scan(ctx.h, cols, nc, count, start, C.byref(pred), ids_dev, count + 16, cnt_dev, None), "fallback scan")
                ctx.sync()
                k = C.c_uint64()
                ctx.download(C.byref(k), cnt_dev, 8)
                mine = np.zeros(k.value, dtype=np.uint32)
                if k.value:
                    ctx.download(mine.ctypes.data, ids_dev, 4 * k.value)
                lists[rank] = mine
                gate.wait()
                merged = np.concatenate(lists)
                np.save(os.path.join(os.environ["OUT_DIR"], f"fallback_{rank}.npy"), merged)
            report[rank]["total_seconds"] = time.monotonic() - t0
            gate.wait()
            dev.free()
            ctx.close()
        except BaseException:
            traceback.print_exc()
            sys.stderr.flush()
            os._exit(3)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads: t.start()
    for t in threads: t.join()
    with open(os.environ["OUT_FILE"], "w") as f:
        json.dump(report, f)
    print("OK")
""")


@pytest.mark.parametrize("world,stall", [(2, "1:4"), (3, "0:4"), (2, "0:2")])
def test_a_stalled_payload_group_ends_in_abort_and_fallback(tmp_path, world, stall):
    """One rank's payload group (op 4 of the communicator; op 2: the sizes all-gather of the second query) never finishes on the device.  Every rank's bounded wait runs out (PQPS_EXCHANGE_TIMEOUT_S),
    the communicator is aborted, later calls fail at once, the exchange tears down, and all ranks finish on a fallback --
    the whole thing within seconds, no rank left behind."""
    assert LOOPBACK.exists()
    rows, chain = 1_500_001, [("risk_level", ">", "3")]
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {str(q.ROOT)!r}\nLOOPBACK = {str(LOOPBACK)!r}\n" + STALL_WORKER)
    env = dict(os.environ, WORLD=str(world), ROWS=str(rows), CHAIN=json.dumps(q.chain_to_jsonable(chain)), OUT_FILE=str(tmp_path / "out.json"),
               OUT_DIR=str(tmp_path), OMP_NUM_THREADS="1", LOOPBACK_STALL=stall, PQPS_EXCHANGE_TIMEOUT_S="2")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), (p.stdout[-1500:], p.stderr[-3000:])
    report = json.loads((tmp_path / "out.json").read_text())
    want = q.HostSynth(rows, seed=21).oracle_scan(chain)
    for r in range(world):
        rep = report[r]
        assert rep["error"] is not None, (r, rep)                            # nobody got through the stalled exchange
        assert rep["seconds_to_error"] < 20, rep                             # the bound (2 s) + the queries issued before it struck
        assert rep["second_call"] != "ok" and rep["second_call_seconds"] < 0.5, rep
        assert rep["close_seconds"] < 10 and rep["total_seconds"] < 40, rep
        assert np.array_equal(np.load(tmp_path / f"fallback_{r}.npy"), want), r
    assert any("did not finish within" in (rep["error"] or "") for rep in report), report      # at least one rank's own wait ran out
