"""The N > 1 path on CPU: row-range sharding (mpi:703-715) + [count | IDs] slot all-gather
+ rank-order compaction (merge.py), world_size 2 and 3 over gloo.

Each rank filters ITS shard with the oracle (the checker stands in for the GPU filter,
which needs a device), the shards' ascending ID lists are merged on every rank, and
the result must equal the whole-table oracle answer bit for bit.  Also covers the
slot-overflow report and the empty-shard / empty-result edges.
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
    out = {}
    for name, (n, chain, cap) in cases.items():
        start, count = mg.shard_rows(n, world, rank)
        host = q.HostSynth(count, seed=11, row0=start)
        local = host.oracle_scan(q.chain_from_jsonable(chain), id_base=start)
        m = mg.IdMerger(torch, dist, world, rank, cap, torch.device("cpu"))
        m.set_local(local)
        m.merge()
        try:
            out[name] = m.result().tolist()
        except RuntimeError as e:
            out[name] = "overflow: " + str(e)
    if rank == world - 1:
        print("RESULT " + json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()
""")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_merge_equals_whole_table(world, tmp_path):
    import json
    cases = {
        "q_b": (50_001, q.chain_to_jsonable([("sudo_used", "=", "TRUE"), "AND", ("risk_level", ">", "2")]), 8192),
        "s1": (50_001, q.chain_to_jsonable([("sudo_used", "=", "FALSE"), "AND", ("user_name", "=", "student1030")]), 4096),
        "none": (10_000, q.chain_to_jsonable([("risk_level", ">", "9")]), 4096),
        "tiny": (2, q.chain_to_jsonable([("risk_level", ">=", "1")]), 4096),       # some ranks own zero rows
        "overflow": (50_001, q.chain_to_jsonable([("risk_level", ">=", "1")]), 4096),
    }
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {str(ROOT)!r}\n" + WORKER)
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
    for name, (n, chain, cap) in cases.items():
        want = q.HostSynth(n, seed=11).oracle_scan(q.chain_from_jsonable(chain)).tolist()
        if name == "overflow":
            assert isinstance(got[name], str) and got[name].startswith("overflow")
        else:
            assert got[name] == want, name


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
