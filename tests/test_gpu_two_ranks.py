"""The shim-driven exchange with a real world of two: one process per GPU, RCCL between them
(pqps_exchange_prepare / _connect / _select / _count).  Needs two devices -- skipped on a one-GPU box, where
tests/test_gpu_parity.py::test_native_exchange_world_of_one covers the same calls with a world of one and
tests/test_merge_gloo.py the displacement arithmetic with worlds of two and three.

Every rank filters its row range of a seeded synthetic table; the gathered ID list and the COUNT must equal the
single-table oracle answer on BOTH ranks.  Cases: skewed shards (all matches on one rank), an empty shard (a
table of one row), an empty result, a dense result larger than the first allocation of the gathered list, a
ring of one, and a local slot that is too small (reported as an error, on the rank that overflowed and on its
peer alike)."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

import qpelib as q

pq = q.pq
pytestmark = pytest.mark.gpu

WORKER = textwrap.dedent("""
    import ctypes as C, json, os, sys
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import qpelib as q
    pq, mg = q.pq, q.pq_merge()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)           # bootstrap only
    ctx = pq.Context(rank)
    cases = json.loads(os.environ["CASES"])
    out = {}
    for name, (n, chain, cap, ring) in cases.items():
        chain = q.chain_from_jsonable(chain)
        start, count = mg.shard_rows(n, world, rank)
        dev = pq.SyntheticTable(ctx, count, seed=21, row0=start)
        xch = mg.ShardExchange.open(pq, ctx, torch, dist, world, rank, cap if cap else count + 16, ring=ring)
        assert xch is not None, "exchange did not come up"
        pred, cols, nc, _ = dev.bind(chain)
        got = []
        for k in range(3):                                                  # the ring goes round
            xch.select(cols, nc, count, start, C.byref(pred), k % ring, None)
            if ring == 1 or k == 2:
                try:
                    ids, local = xch.result(k % ring)
                    got.append([ids.tolist(), local])
                except pq.PqpsError as e:
                    got.append("error: " + str(e))
        xch.count(cols, nc, count, C.byref(pred), 0, None)
        total, mine = xch.count_result(0)
        out[name] = {"ids": got, "count": [total, mine]}
        xch.close()
        dev.free()
    with open(os.environ["OUT_FILE"] + str(rank), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()
""")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_exchange_between_two_gpus(tmp_path):
    if pq.lib().pqps_device_count() < 2:
        pytest.skip("needs two GPUs")
    world = 2
    n = 3_000_001
    cases = {
        "q_b": (n, [("sudo_used", "=", "TRUE"), "AND", ("risk_level", ">", "2")], 0, 2),
        "first_only": (n, [("command_id", "<=", "1000000")], 0, 2),
        "last_only": (n, [("command_id", ">", "2900000")], 0, 3),
        "dense": (n, [("sudo_used", "=", "FALSE")], 0, 2),                  # > 2^20 IDs: the gathered list grows
        "none": (n, [("risk_level", ">", "9")], 0, 1),
        "one_row": (1, [("risk_level", ">=", "0")], 0, 1),                  # rank 1 owns no rows
        "overflow": (n, [("risk_level", ">=", "1")], 4096, 2),
    }
    cases = {k: (v[0], q.chain_to_jsonable(v[1]), v[2], v[3]) for k, v in cases.items()}
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {str(q.ROOT)!r}\n" + WORKER)
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   CASES=json.dumps(cases), OUT_FILE=str(tmp_path / "out"), OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    try:
        outs = [p.communicate(timeout=600) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so[-1500:], se[-3000:])
    got = [json.loads((tmp_path / f"out{r}").read_text()) for r in range(world)]
    mg = q.pq_merge()
    for name, (rows, chain, cap, ring) in cases.items():
        want = q.HostSynth(rows, seed=21).oracle_scan(q.chain_from_jsonable(chain)).tolist()
        for r in range(world):
            start, count = mg.shard_rows(rows, world, r)
            mine = sum(1 for i in want if start <= i < start + count) if len(want) < 10**6 else None
            res = got[r][name]
            assert res["count"][0] == len(want), (name, r)
            if mine is not None:
                assert res["count"][1] == mine, (name, r)
            for entry in res["ids"]:
                if name == "overflow":
                    assert isinstance(entry, str) and "overflow" in entry, (name, r)
                else:
                    assert entry[0] == want, (name, r)
                    if mine is not None:
                        assert entry[1] == mine, (name, r)
