"""The oracle (oracle/qpe_oracle.c) against the golden vectors the REAL reference
produced (tests/golden/, made by make_golden.py from oracle/_ref), plus the
known-answer cases the reference's own tests hold:
  tests/executeEngine-serial-test.c:29-114   three evaluateWhereClause asserts
  tests/duplicate-test.c:37-54               duplicate keys: counts 3 and 1
  tests/bplus-serial-test.c:40-43            inclusive ranges 10-30 -> {15,25}, 5-45 -> all
CPU only.
"""
import base64
import ctypes as C
import hashlib
import json
import zlib

import pytest

import qpelib as q

GOLD = q.GOLDEN
SELECT = (json.loads((GOLD / "select_golden.json").read_text())
          + json.loads((GOLD / "select_random_golden.json").read_text())     # + seeded random WHERE trees, same reference
          + json.loads((GOLD / "select_wide_golden.json").read_text()))       # + lists of more than 32 comparisons, reference engine API
INDEX_CONFIGS = {
    "none": [],
    "default": q.DEFAULT_INDEXES,
    "cmdid": [("command_id", 0)],
    "risk": [("risk_level", 1)],
    "risk_twice": [("risk_level", 1), ("risk_level", 1)],
}


def sha_rows(rows):
    h = hashlib.sha256()
    for r in rows:
        for c in r:
            h.update(c.encode("latin-1"))
            h.update(b"\x1f")
        h.update(b"\x1e")
    return h.hexdigest()


_tables = {}


def table(csv, cfg):
    key = (csv, cfg)
    if key not in _tables:
        _tables[key] = q.OracleTable(GOLD / csv, INDEX_CONFIGS[cfg])
    return _tables[key]


@pytest.mark.parametrize("case", SELECT, ids=[f"{c['csv'][:4]}-{c['name']}-{c['indexes']}" for c in SELECT])
def test_select_matches_reference(case):
    t = table(case["csv"], case["indexes"])
    chain = q.chain_from_jsonable(case["where"])
    ids, count, cand = t.select_ids(chain)
    assert count == case["num_records"]
    assert cand == case["candidates"]
    if q.case_ids(case) is not None:
        assert ids == q.case_ids(case)
    sql = case["sql"]
    sel = sql[len("SELECT "):sql.index(" FROM ")]
    cols = None if sel.strip() == "*" else [c.strip() for c in sel.split(",")]
    rows = t.project(ids, cols)
    assert sha_rows(rows) == case["rows_sha256"]
    if case.get("first_rows"):
        assert rows[:len(case["first_rows"])] == case["first_rows"]


def test_golden_is_mostly_pinned():
    unpinned = [c for c in SELECT if not c["pinned"]]
    assert len(unpinned) <= 8 and len(SELECT) >= 150
    # every unpinned case is one the reference cannot run (candidate overflow)
    for c in unpinned:
        assert c["candidates"] > table(c["csv"], c["indexes"]).n


def test_csv_records_match_reference():
    gold = json.loads((GOLD / "records_golden.json").read_text())
    for name, g in gold.items():
        t = table(name, "none")
        assert t.n == g["num_records"]
        blob = zlib.decompress(base64.b64decode(g["zlib_b64"]))
        for i in range(g["dumped"]):
            mine = bytes(t.rows[i])
            assert mine == blob[i * 1040:(i + 1) * 1040], f"{name} row {i}"


def test_index_leaf_order_matches_reference():
    gold = json.loads((GOLD / "index_order_golden.json").read_text())
    lib = q.load_oracle()
    for name, per_attr in gold.items():
        t = table(name, "none")
        for attr, order in per_attr.items():
            perm = (C.c_int * max(1, t.n))()
            assert lib.orc_index_build(t.rows, t.n, attr.encode(), perm) == 0
            assert list(perm[:t.n]) == order, f"{name}:{attr}"


# ---- known-answer tests restated from the reference's tests ---------------
def _admin_record():
    r = q.Record()
    r.command_id = 100
    r.risk_level = 5
    r.user_id = 10
    r.user_name = b"admin"
    r.sudo_used = True
    r.exit_code = 0
    r.raw_command = b"ls -la"
    r.base_command = b"ls"
    r.shell_type = b"bash"
    r.timestamp = b"2023-01-01"
    r.working_directory = b"/home/admin"
    r.host_name = b"localhost"
    return r


def test_kat_evaluate_where_clause():
    lib = q.load_oracle()
    r = _admin_record()
    for chain, want in [
        ([("risk_level", ">", "3")], True),
        ([[("risk_level", ">", "3"), "AND", ("user_id", "=", "10")]], True),
        ([[("risk_level", ">", "10")], "OR", [("user_id", "=", "10")]], True),
        ([[("risk_level", ">", "10")], "AND", [("user_id", "=", "10")]], False),
    ]:
        wl = q.WhereList(chain)
        assert lib.orc_eval_where(C.byref(r), wl.ptr) is want


def test_kat_duplicate_keys(tmp_path):
    p = tmp_path / "dup.csv"
    p.write_text(
        "command_id,raw_command,base_command,shell_type,exit_code,timestamp,sudo_used,working_directory,user_id,user_name,host_name,risk_level\n"
        "1,cmd1,base,bash,0,ts,0,wd,1001,user,host,1\n"
        "2,cmd2,base,bash,0,ts,0,wd,1001,user,host,1\n"
        "3,cmd3,base,bash,0,ts,0,wd,1001,user,host,2\n"
        "4,cmd4,base,bash,0,ts,0,wd,1001,user,host,1\n")
    t = q.OracleTable(p, [("risk_level", 1)])
    assert t.n == 4
    ids, count, _ = t.select_ids([("risk_level", "=", "1")])
    assert count == 3 and ids == [3, 1, 0]          # newest duplicate first
    ids, count, _ = t.select_ids([("risk_level", "=", "2")])
    assert count == 1 and ids == [2]


def test_kat_inclusive_ranges(tmp_path):
    p = tmp_path / "r.csv"
    p.write_text("h\n" + "".join(f"{k},c,b,bash,0,ts,0,wd,1,u,h,1\n" for k in (5, 15, 25, 35, 45)))
    t = q.OracleTable(p, [("command_id", 0)])
    ids, _, _ = t.select_ids([("command_id", ">=", "10"), "AND", ("command_id", "<=", "30")])
    # two probes (>=10: 4 rows, <=30: 3 rows) concatenated then re-filtered, like QPESeq
    assert [t.rows[i].command_id for i in ids] == [15, 25, 15, 25]
    ids, _, _ = t.select_ids([("command_id", ">=", "5")])
    assert [t.rows[i].command_id for i in ids] == [5, 15, 25, 35, 45]


def test_partition_formula():
    lib = q.load_oracle()
    for n in (0, 1, 7, 8, 9, 1000, 10**9 + 3):
        for world in (1, 2, 3, 8):
            pos = 0
            for r in range(world):
                s, c = C.c_uint64(), C.c_uint64()
                lib.orc_partition(n, world, r, C.byref(s), C.byref(c))
                assert s.value == pos
                assert c.value in (n // world, n // world + 1)
                pos += c.value
            assert pos == n


BOOLPROBE = json.loads((GOLD / "select_boolprobe_golden.json").read_text())
BOOLPROBE_INDEX_CONFIGS = {
    "default": q.DEFAULT_INDEXES,
    "bool_only": [("sudo_used", 3)],
    "bool_twice": [("sudo_used", 3), ("risk_level", 1), ("sudo_used", 3)],
}


def test_oracle_boolprobe_matches_qpeomp_golden():
    """The OpenMP / MPI engines' row selection (BOOL indexes probed too, engine/omp/executeEngine-omp.c:362-494, one
    thread): orc_select_ids_v(..., probe_bool = 1) against the compiled reference's answers
    (tests/golden/make_golden.py --boolprobe-only); and the serial walk differs exactly where the fixture says so."""
    assert len(BOOLPROBE) > 100
    tables = {}
    differing = 0
    for case in BOOLPROBE:
        key = (case["csv"], case["indexes"])
        if key not in tables:
            tables[key] = q.OracleTable(GOLD / case["csv"], BOOLPROBE_INDEX_CONFIGS[case["indexes"]])
        orc = tables[key]
        chain = q.chain_from_jsonable(case["where"])
        ids, k, cand = orc.select_ids(chain, probe_bool=True)
        assert k == case["num_records"] and ids == q.case_ids(case), case["name"]
        assert cand == case["candidates"], case["name"]
        assert sha_rows(orc.project(ids, case["columns"])) == case["rows_sha256"], case["name"]
        serial_ids, _, _ = orc.select_ids(chain)
        assert (serial_ids != ids) == case["differs_from_qpeseq"], case["name"]
        differing += serial_ids != ids
    assert differing > 30
