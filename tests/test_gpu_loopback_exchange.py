"""The shim-driven exchange (pqps_exchange_prepare / _connect / _select / _count / _result) with a world of TWO and
THREE on a one-GPU box: the ranks are threads of one process and the ten nccl* calls the exchange resolves with
dlsym come from tests/loopback/libloopback_rccl.so (a stand-in with RCCL's stream semantics, see its header) instead
of librccl.so.  What runs is the product's own world > 1 code: capacities all-gather at connect, the per-query sizes
all-gather, displacements, the held-back send / recv group (hold = 2 with a ring of 5 or more), ragged capacities, a
gathered list that has to grow, a rank without rows, a local slot that overflows, COUNT(*) all-reduces in between.
Every rank's gathered list and every count must equal the oracle's answer for the whole table
(engine/mpi/executeEngine-mpi.c:745-765 is the shape being reproduced).

With two or more GPUs the ranks spread over them (peer copies); tests/test_gpu_two_ranks.py is the same through the
real RCCL, one process per GPU, and needs two cards."""
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
}


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_world_of_several_through_the_loopback(tmp_path, world):
    assert LOOPBACK.exists(), "build it first: make -C tests/loopback (python __graft_entry__.py does)"
    cases = {k: (v[0], q.chain_to_jsonable(v[1]), v[2], v[3], v[4]) for k, v in CASES.items()}
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {str(q.ROOT)!r}\nLOOPBACK = {str(LOOPBACK)!r}\n" + WORKER)
    env = dict(os.environ, WORLD=str(world), CASES=json.dumps(cases), OUT_FILE=str(tmp_path / "out.json"), OUT_DIR=str(tmp_path),
               OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), (p.stdout[-1500:], p.stderr[-3000:])
    got = json.loads((tmp_path / "out.json").read_text())
    mg = q.pq_merge()
    for name, (rows, chain, cap, ring, plan) in CASES.items():
        want = q.HostSynth(rows, seed=21).oracle_scan(chain)
        for r in range(world):
            start, count = mg.shard_rows(rows, world, r)
            mine = int(((want >= start) & (want < start + count)).sum())
            results = got[r][name]
            assert len(results) == sum(1 for s in plan if s[0] in "rk"), (name, r)
            for entry in results:
                if name == "overflow":
                    assert isinstance(entry, str) and "overflow" in entry, (name, r, str(entry)[:200])
                elif entry[0] == "count":
                    assert entry[1] == len(want) and entry[2] == mine, (name, r)
                else:
                    arr = np.load(entry[0])
                    assert entry[1] == mine and np.array_equal(arr, want), (name, r, len(arr), len(want))
