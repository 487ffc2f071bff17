"""The N > 1 path on CPU: row-range sharding (mpi:703-715) + the all-gatherv of the matching IDs as sizes first,
then exactly-sized point-to-point payload at displacements (merge.py, mirror of mpi:753-765), world_size 2 and 3
over gloo.

Each rank filters ITS shard with the oracle (the checker stands in for the GPU filter,
which needs a device), the shards' ascending ID lists are merged on every rank, and
the result must equal the whole-table oracle answer bit for bit.  Covers skewed shards (one rank's rows all
match while the others' none do), empty shards / empty results, a local buffer that is too small (reported),
several queries in flight (begin k before finish k-1), and the all-or-none bring-up of the shim-driven
exchange with a failure injected on one rank.

Round 4: the payload travels in the COMPACT wire form wherever that is smaller (low 16 bits of every row number + one
u32 per 65 536-row group; merge.wire_pack_numpy / wire_expand_numpy are the CPU twins of the shim's kernels) -- every
case runs with it, a few-per-cent answer (Q_A, Q_B) and dense ones (risk_level > 1, every row) over shards of several
groups included, and once more with u32 IDs only; the bytes that travelled are checked against the model.
"""
import os
import pathlib
import socket
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

ROOT = q.ROOT

WORKER = textwrap.dedent("""
    import importlib.util, os, sys, json
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import qpelib as q
    spec = importlib.util.spec_from_file_location("pqps_merge", os.path.join(ROOT, "parallel-query-processing-system_amd", "merge.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cases = json.loads(os.environ["CASES"])
    compact = os.environ.get("COMPACT", "1") == "1"
    out, wire = {}, {}
    for name, (n, chain, cap) in cases.items():
        start, count = mg.shard_rows(n, world, rank)
        host = q.HostSynth(count, seed=11, row0=start)
        local = host.oracle_scan(q.chain_from_jsonable(chain), id_base=start)
        m = mg.IdMerger(torch, dist, world, rank, cap + 500 * rank if cap else count + 16, torch.device("cpu"),
                        shard=(count, start), compact=compact)                       # capacities differ by rank
        m.set_local(local)
        m.merge()
        wire[name] = [m.wire_bytes_in, m.u32_bytes_in]
        try:
            out[name] = m.result().tolist()
            assert m.totals[0] == len(out[name]) and m.merged.numel() >= m.totals[0]
        except RuntimeError as e:
            # a rank's own buffer was too small: reported, and what every rank DID hold is still gathered in order
            out[name] = {"error": str(e), "held": m.merged[:m.totals[0]].numpy().view("uint32").tolist(), "caps": m.caps}
    # several queries in flight, as bench.py drives them: sizes of query k go out before the payload of k-1
    names = [k for k in cases if k != "overflow"]
    ring = [mg.IdMerger(torch, dist, world, rank, 120_000, torch.device("cpu"), shard=(0, 0), compact=compact) for _ in range(2)]
    piped = {}
    for i, name in enumerate(names):
        n, chain, _ = cases[name]
        start, count = mg.shard_rows(n, world, rank)
        m = ring[i % 2]
        if i >= 2:
            piped[names[i - 2]] = m.result().tolist()
        m.shard = (count, start)                              # (the ring's mergers serve shards of different tables in turn)
        m.set_local(q.HostSynth(count, seed=11, row0=start).oracle_scan(q.chain_from_jsonable(chain), id_base=start))
        m.begin()
        ring[(i - 1) % 2].finish()
    for i in range(max(len(names) - 2, 0), len(names)):
        piped[names[i]] = ring[i % 2].result().tolist()
    assert piped == {k: out[k] for k in names}, "pipelined exchange differs"
    out["_wire"] = wire
    if rank == world - 1:
        with open(os.environ["OUT_FILE"], "w") as f:           # (a pipe would fill up: the parent reads the ranks one by one)
            json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()
""")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,compact", [(2, True), (3, True), (2, False)])
def test_sharded_merge_equals_whole_table(world, compact, tmp_path):
    import json
    cases = {
        # shards of two and more 65 536-row groups: the compact form's group offsets matter
        "q_a_groups": (200_003, q.chain_to_jsonable([("risk_level", ">", "3")]), 0),
        "q_b_groups": (200_003, q.chain_to_jsonable([("sudo_used", "=", "TRUE"), "AND", ("risk_level", ">", "2")]), 0),
        "r1_groups": (200_003, q.chain_to_jsonable([("risk_level", ">", "1")]), 0),
        "s1_groups": (400_001, q.chain_to_jsonable([("sudo_used", "=", "FALSE"), "AND", ("user_name", "=", "student1030")]), 0),   # a few IDs per group
        "q_b": (50_001, q.chain_to_jsonable([("sudo_used", "=", "TRUE"), "AND", ("risk_level", ">", "2")]), 8192),
        "s1": (50_001, q.chain_to_jsonable([("sudo_used", "=", "FALSE"), "AND", ("user_name", "=", "student1030")]), 4096),
        "none": (10_000, q.chain_to_jsonable([("risk_level", ">", "9")]), 4096),
        "tiny": (2, q.chain_to_jsonable([("risk_level", ">=", "1")]), 4096),       # some ranks own zero rows
        # skewed: every row of the first shard matches, none of the others -- and the other way round
        "first_dense": (30_000, q.chain_to_jsonable([("command_id", "<=", str(30_000 // world))]), 0),
        "last_dense": (30_000, q.chain_to_jsonable([("command_id", ">", str(30_000 - 30_000 // world))]), 0),
        "all": (30_001, q.chain_to_jsonable([("risk_level", ">=", "0")]), 0),       # every shard 100 % dense
        "overflow": (50_001, q.chain_to_jsonable([("risk_level", ">=", "1")]), 4096),
    }
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {str(ROOT)!r}\n" + WORKER)
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   CASES=json.dumps(cases), OMP_NUM_THREADS="1", OUT_FILE=str(tmp_path / "result.json"), COMPACT="1" if compact else "0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    got = json.loads((tmp_path / "result.json").read_text())
    mg = q.pq_merge()
    for name, (n, chain, cap) in cases.items():
        want = q.HostSynth(n, seed=11).oracle_scan(q.chain_from_jsonable(chain)).tolist()
        # what the last rank took in: per peer the compact form where it is smaller, u32 IDs otherwise
        model_wire = model_u32 = 0
        for r in range(world - 1):
            start, count = mg.shard_rows(n, world, r)
            k = sum(1 for i in want if start <= i < start + count)
            if cap:
                k = min(k, cap + 500 * r + (cap + 500 * r) % 2)
            model_u32 += 4 * k
            model_wire += mg.wire_bytes(count, k) if (compact and mg.wire_pays(count, k)) else 4 * k
        assert got["_wire"][name] == [model_wire, model_u32], (name, got["_wire"][name], model_wire, model_u32)
        if compact and name in ("q_a_groups", "q_b_groups", "r1_groups", "all"):
            assert model_wire < 0.6 * model_u32, name                 # about half the bytes
        assert model_wire <= model_u32                                # never more than the IDs themselves
        if name in ("s1", "tiny"):
            assert model_wire == model_u32                            # a handful of IDs travel as they are
        if name == "overflow":
            assert "overflow" in got[name]["error"]
            caps = got[name]["caps"]
            assert caps == [4096 + 500 * r + (4096 + 500 * r) % 2 for r in range(world)]
            held = []
            for r in range(world):
                start, count = q.pq_merge().shard_rows(n, world, r)
                held += [i for i in want if start <= i < start + count][:caps[r]]
            assert got[name]["held"] == held
        else:
            assert got[name] == want, name


def test_wire_floor_keeps_small_lists_as_they_are(monkeypatch):
    """Below 32 768 IDs a list travels as u32 whatever its density; PQPS_WIRE_MIN_IDS moves the floor (csrc/pqps_hip.hip: wire_pays)."""
    mg = q.pq_merge()
    monkeypatch.delenv("PQPS_WIRE_MIN_IDS", raising=False)
    rows = 4 * 65536
    assert not mg.wire_pays(rows, 32767) and mg.wire_pays(rows, 32768)
    assert not mg.wire_pays(1 << 30, 32768)                            # 2 IDs per group: the offsets cost more than they save
    monkeypatch.setenv("PQPS_WIRE_MIN_IDS", "0")
    assert mg.wire_pays(rows, 100) and not mg.wire_pays(rows, 3)
    monkeypatch.setenv("PQPS_WIRE_MIN_IDS", "1000000")
    assert not mg.wire_pays(rows, 200000)


BRINGUP_WORKER = textwrap.dedent("""
    import importlib.util, os, sys, json
    import torch, torch.distributed as dist
    spec = importlib.util.spec_from_file_location("pqps_merge", os.path.join(ROOT, "parallel-query-processing-system_amd", "merge.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fail_at, fail_rank = os.environ["FAIL_AT"], int(os.environ["FAIL_RANK"])
    log = []

    def make_id():
        if fail_at == "id":
            raise RuntimeError("injected: no RCCL library")
        return b"x" * 128

    def prepare():
        log.append("prepare")
        if fail_at == "prepare" and rank == fail_rank:
            raise RuntimeError("injected: hipMalloc failed")
        return {"rank": rank}

    def connect(h, ident):
        # stands in for ncclCommInitRank: a collective that returns only when EVERY rank has called it
        log.append("connect")
        assert ident == b"x" * 128
        dist.barrier()
        if fail_at == "connect" and rank == fail_rank:
            raise RuntimeError("injected: communicator refused")

    def close(h):
        log.append("close")

    h = mg.open_exchange(dist, world, rank, make_id, prepare, connect, close, control_device="cpu", torch=torch)
    print("RESULT " + json.dumps({"rank": rank, "up": h is not None, "log": log}))
    dist.barrier()
    dist.destroy_process_group()
""")


@pytest.mark.parametrize("world,fail_at,fail_rank", [(2, "none", 0), (2, "prepare", 1), (3, "prepare", 0), (2, "connect", 1),
                                                     (3, "id", 0)])
def test_exchange_comes_up_on_every_rank_or_on_none(world, fail_at, fail_rank, tmp_path):
    """merge.open_exchange: a rank that fails before the communicator is built must not leave the others
    blocked inside it (the stand-in connect() is a barrier: it would hang), and a failure of any rank at
    any stage sends ALL ranks to the fallback."""
    import json
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {str(ROOT)!r}\n" + BRINGUP_WORKER)
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FAIL_AT=fail_at, FAIL_RANK=str(fail_rank), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]                # a hang would trip this timeout
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    res = sorted((json.loads(ln[len("RESULT "):]) for so, _ in outs for ln in so.splitlines() if ln.startswith("RESULT ")),
                 key=lambda d: d["rank"])
    assert len(res) == world
    assert all(d["up"] == (fail_at == "none") for d in res)
    for d in res:
        if fail_at == "id":
            assert d["log"] == []
        elif fail_at == "prepare":
            assert "connect" not in d["log"]                       # nobody entered the communicator
            assert ("close" in d["log"]) == (d["rank"] != fail_rank)
        elif fail_at == "connect":
            assert d["log"] == ["prepare", "connect", "close"]
        else:
            assert d["log"] == ["prepare", "connect"]


def test_shard_rows_is_the_mpi_partition():
    import importlib.util
    spec = importlib.util.spec_from_file_location("pqps_merge", q.PKG / "merge.py")
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    import ctypes as C
    orc = q.load_oracle()
    for n in (0, 1, 7, 100, 10**9 + 7):
        for world in (1, 2, 3, 8):
            for r in range(world):
                s, c = C.c_uint64(), C.c_uint64()
                orc.orc_partition(n, world, r, C.byref(s), C.byref(c))
                assert mg.shard_rows(n, world, r) == (s.value, c.value)


INDEX_WORKER = textwrap.dedent("""
    import importlib.util, os, sys, json
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import qpelib as q
    spec = importlib.util.spec_from_file_location("pqps_merge", os.path.join(ROOT, "parallel-query-processing-system_amd", "merge.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, cases = 60_001, json.loads(os.environ["CASES"])
    start, count = mg.shard_rows(n, world, rank)
    host = q.HostSynth(count, seed=13, row0=start)
    out = {}
    for name, (col, lo, hi, chain) in cases.items():
        perm = q.host_index_order(host.arr[col])
        local = q.host_index_select(host, {col: perm}, [(col, lo, hi)], q.chain_from_jsonable(chain), id_base=start)
        keys = host.arr[col][(local - start).astype(np.int64)].astype(np.int64)
        ukeys = (keys.astype(np.uint64) ^ np.uint64(0x80000000)) if col != "command_id" else keys.astype(np.uint64)   # order-preserving image
        m = mg.IndexMerger(torch, dist, world, rank, n // world + 16, torch.device("cpu"))    # same capacity on every rank
        m.set_local(local, ukeys)
        m.merge()
        out[name] = m.result().tolist()
    if rank == 0:
        print("RESULT " + json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()
""")


@pytest.mark.parametrize("world", [2, 3])
def test_index_mode_merge_equals_whole_table_leaf_order(world, tmp_path):
    """SURVEY 8(e), index mode across shards: shard-local (key asc, row desc) lists merged by
    (key asc, row desc) equal the single-table index-path answer (numpy restatement of serial:358-474)."""
    import json
    cases = {
        "risk": ("risk_level", 4, 2**31 - 1, q.chain_to_jsonable([("risk_level", ">", "3"), "AND", ("exit_code", "=", "0")])),
        "user": ("user_id", 1001, 1004, q.chain_to_jsonable([("user_id", ">=", "1001"), "AND", ("user_id", "<=", "1004")])),
        "cid": ("command_id", 30_000, 2**62, q.chain_to_jsonable([("command_id", ">=", "30000"), "AND", ("sudo_used", "=", "FALSE")])),
        "none": ("risk_level", 9, 2**31 - 1, q.chain_to_jsonable([("risk_level", ">", "8")])),
    }
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {str(ROOT)!r}\n" + INDEX_WORKER)
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   CASES=json.dumps(cases), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    line = [ln for so, _ in outs for ln in so.splitlines() if ln.startswith("RESULT ")]
    assert len(line) == 1
    got = json.loads(line[0][len("RESULT "):])
    whole = q.HostSynth(60_001, seed=13)
    for name, (col, lo, hi, chain) in cases.items():
        want = q.host_index_select(whole, {col: q.host_index_order(whole.arr[col])}, [(col, lo, hi)], q.chain_from_jsonable(chain))
        assert got[name] == want.tolist(), name
        if name != "none":
            assert len(want) > 50
