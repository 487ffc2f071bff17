"""BASELINE configs[0] at its stated size: SELECT with WHERE on a 50 k-row commands_* CSV, the file size the reference's own
SELECT test and dispatcher use (tests/serial-SELECT-test.c:12, include/connectEngine.h:11).

The CSV is regenerated from the repository's seeded scripts/make_csv.py (never committed; its SHA-256 is pinned in
tests/golden/commands_50k_golden.json) and every answer is compared with what the COMPILED REFERENCE gave for that very
file (make_golden.py --50k-only): per query the count, the SHA-256 of the row numbers and of every projected cell, under
the default five indexes and without indexes; the QPEHIP driver's stdout and the CSV it leaves behind against QPESeq's,
for both of the reference's sample-query files.  CPU: the oracle against the golden.  GPU: the HIP engine + driver."""
import hashlib
import json
import re
import shutil
import subprocess
import sys

import pytest

import qpelib as q

pq = q.pq
GOLD = json.loads((q.GOLDEN / "commands_50k_golden.json").read_text())
INDEX_CONFIGS = {"none": [], "default": q.DEFAULT_INDEXES}


@pytest.fixture(scope="module")
def csv50k(tmp_path_factory):
    path = tmp_path_factory.mktemp("c50k") / "commands_50k.csv"
    subprocess.run([sys.executable, str(q.ROOT / "scripts" / "make_csv.py"), str(GOLD["rows"]), str(path)], check=True)
    data = path.read_bytes()
    assert len(data) == GOLD["csv_bytes"] and hashlib.sha256(data).hexdigest() == GOLD["csv_sha256"], "the generator no longer writes the file the golden was made from"
    return path


def ids_sha(ids):
    return hashlib.sha256(b"".join(int(i).to_bytes(4, "little") for i in ids)).hexdigest()


def sha_rows(rows):
    h = hashlib.sha256()
    for r in rows:
        for c in r:
            h.update(c.encode("latin-1"))
            h.update(b"\x1f")
        h.update(b"\x1e")
    return h.hexdigest()


def columns_of(sql):
    sel = sql[len("SELECT "):sql.index(" FROM ")]
    return None if sel.strip() == "*" else [c.strip() for c in sel.split(",")]


def test_oracle_answers_the_50k_file_like_the_reference(csv50k):
    for cfg, indexes in INDEX_CONFIGS.items():
        orc = q.OracleTable(csv50k, indexes)
        for case in (c for c in GOLD["select"] if c["indexes"] == cfg):
            ids, count, _cand = orc.select_ids(q.chain_from_jsonable(case["where"]))
            assert count == case["num_records"] and ids[:5] == case["first_ids"], case["name"]
            assert ids_sha(ids) == case["ids_sha256"], (case["name"], cfg)
            assert sha_rows(orc.project(ids, columns_of(case["sql"]))) == case["rows_sha256"], (case["name"], cfg)


@pytest.mark.gpu
def test_hip_engine_answers_the_50k_file_like_the_reference(csv50k):
    for cfg, indexes in INDEX_CONFIGS.items():
        eng = pq.HipEngine(csv50k, indexes)
        assert eng.n == GOLD["rows"]
        for case in (c for c in GOLD["select"] if c["indexes"] == cfg):
            chain = q.chain_from_jsonable(case["where"])
            ids = eng.select_ids(chain)
            assert len(ids) == case["num_records"] and ids[:5] == case["first_ids"], (case["name"], cfg)
            assert ids_sha(ids) == case["ids_sha256"], (case["name"], cfg)
            res = eng.select(columns_of(case["sql"]), chain)
            assert res["success"] and res["numRecords"] == case["num_records"] and res["columns"] == case["columns"]
            assert sha_rows(res["rows"]) == case["rows_sha256"], (case["name"], cfg)
        eng.close()


def normalize(text):
    text = text.split("\x1b[36m=======")[0]
    text = re.sub(r"Query Time: [0-9.]+ seconds", "Query Time: X seconds", text)
    text = re.sub(r"Execution Time: [0-9.]+", "Execution Time: X", text)
    return text


@pytest.mark.gpu
@pytest.mark.parametrize("queries", ["sample-queries.txt", "sample-queries-FULL.txt"])
def test_qpehip_on_the_50k_file_prints_what_qpeseq_printed(csv50k, tmp_path, queries):
    exe = q.PKG / "QPEHIP"
    assert exe.exists(), "build the driver first (make -C parallel-query-processing-system_amd)"
    shutil.copy(csv50k, tmp_path / "data.csv")
    shutil.copy(q.GOLDEN / queries, tmp_path / "sample-queries.txt")
    run = subprocess.run([str(exe), "data.csv"], cwd=tmp_path, capture_output=True, timeout=300)
    assert run.returncode == 0, run.stderr.decode()[-2000:]
    want = GOLD["driver"][queries]
    got = normalize(run.stdout.decode("latin-1"))
    assert got.count("\n") == want["stdout_lines"]
    assert hashlib.sha256(got.encode("latin-1")).hexdigest() == want["stdout_sha256"]
    left = (tmp_path / "data.csv").read_bytes()
    assert len(left) == want["csv_left_bytes"] and hashlib.sha256(left).hexdigest() == want["csv_left_sha256"]
