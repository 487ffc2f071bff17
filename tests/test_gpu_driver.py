"""End-to-end on a real MI355X: the QPEHIP driver (tokenizer -> connectEngine -> HIP engine ->
printTable) must print what the REAL reference's QPESeq printed for the same CSV and the
reference's own sample-queries.txt (tests/golden/qpeseq_stdout.txt, made by make_golden.py),
timings and the summary block aside.  Also printTable() text goldens, INSERT / DELETE, and
the no-GPU failure mode of the engine."""
import ctypes as C
import json
import os
import re
import shutil
import subprocess

import pytest

import qpelib as q

pq = q.pq
pytestmark = pytest.mark.gpu


def normalize(text):
    text = text.split("\x1b[36m=======")[0]
    text = re.sub(r"Query Time: [0-9.]+ seconds", "Query Time: X seconds", text)
    text = re.sub(r"Execution Time: [0-9.]+", "Execution Time: X", text)
    return text


@pytest.mark.parametrize("queries,golden", [("sample-queries.txt", "qpeseq_stdout"), ("sample-queries-FULL.txt", "qpeseq_full_stdout")])
def test_qpehip_prints_what_qpeseq_prints(tmp_path, queries, golden):
    """The reference's own query files through both drivers: same stdout, and the same CSV left
    behind (Sample 5 appends a row; Sample 6 of the FULL file deletes it and rewrites the file
    without its header, like the reference does)."""
    import hashlib
    exe = q.PKG / "QPEHIP"
    assert exe.exists(), "build the driver first (make -C parallel-query-processing-system_amd)"
    shutil.copy(q.GOLDEN / "commands_2k.csv", tmp_path / "data.csv")
    shutil.copy(q.GOLDEN / queries, tmp_path / "sample-queries.txt")         # the driver opens this name (QPESeq.c:40)
    run = subprocess.run([str(exe), "data.csv"], cwd=tmp_path, capture_output=True, timeout=300)
    assert run.returncode == 0, run.stderr.decode()[-2000:]
    got = normalize(run.stdout.decode("latin-1"))
    want = (q.GOLDEN / (golden + ".txt")).read_text(encoding="latin-1")
    assert got == want
    left = (tmp_path / "data.csv").read_bytes()
    sha, size = (q.GOLDEN / (golden + "_csv.sha256")).read_text().split()
    assert len(left) == int(size) and hashlib.sha256(left).hexdigest() == sha
    if queries == "sample-queries.txt":
        assert left.split(b"\n")[-2].startswith(b"999999,echo 'test insert',echo,bash,0,")


def test_print_table_text_matches_reference():
    gold = json.loads((q.GOLDEN / "print_golden.json").read_text())
    L = pq.lib()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    eng = pq.HipEngine(q.GOLDEN / "commands_2k.csv", pq.DEFAULT_INDEXES)
    import tempfile
    for case in gold:
        sql = case["sql"]
        sel = sql[len("SELECT "):sql.index(" FROM ")]
        cols = None if sel.strip() == "*" else [c.strip() for c in sel.split(",")]
        where = sql.split(" WHERE ", 1)[1] if " WHERE " in sql else None
        # the where list as the front end of this repository parses it
        dump = q.call_text(L.hipDumpParse, sql.encode())
        chain = q.parse_where_dump(dump.split(q.RS, 1)[1])
        wl = pq.WhereList(chain)
        items = (C.c_char_p * max(1, len(cols or [])))(*[c.encode() for c in (cols or [])])
        rs = L.executeQuerySelectHIP(eng.e, items if cols else None, len(cols or []), b"Commands", wl.ptr)
        rs.contents.queryTime = 0.0
        with tempfile.NamedTemporaryFile(suffix=".txt") as tf:
            f = libc.fopen(tf.name.encode(), b"w")
            L.printTable(f, rs, case["limit"])
            libc.fclose(f)
            text = open(tf.name, encoding="latin-1").read()
        L.freeResultSet(rs)
        assert text == case["text"], case["name"]
    eng.close()


def test_insert_then_delete_roundtrip(tmp_path):
    """Sample 5 / Sample 6 of sample-queries-FULL.txt: INSERT 999999 then DELETE it."""
    csv = tmp_path / "data.csv"
    shutil.copy(q.GOLDEN / "commands_2k.csv", csv)
    L = pq.lib()
    eng = pq.HipEngine(csv, pq.DEFAULT_INDEXES)
    n0 = eng.n
    r = pq.Record()
    r.command_id, r.exit_code, r.user_id, r.risk_level, r.sudo_used = 999999, 0, 1000, 1, False
    r.raw_command, r.base_command, r.shell_type = b"echo 'test insert'", b"echo", b"bash"
    r.timestamp, r.working_directory, r.user_name, r.host_name = b"2025-12-01T12:00:00.000Z", b"/home/test", b"testuser", b"test-host"
    assert L.executeQueryInsertHIP(eng.e, b"Commands", C.byref(r))
    assert eng.e.contents.num_records == n0 + 1
    assert eng.select_ids([("command_id", "=", "999999")]) == [n0]
    assert eng.select_ids([("user_name", "=", "testuser")]) == [n0]          # dictionary was rebuilt
    bad = pq.Record()
    assert not L.executeQueryInsertHIP(eng.e, b"Commands", C.byref(bad))     # missing fields are rejected
    wl = pq.WhereList([("command_id", "=", "999999")])
    rs = L.executeQueryDeleteHIP(eng.e, b"Commands", wl.ptr)
    assert rs.contents.success and rs.contents.numRecords == 1
    L.freeResultSet(rs)
    assert eng.e.contents.num_records == n0
    assert eng.select_ids([("command_id", "=", "999999")]) == []
    # survivors keep their order; the file was rewritten without a header, like the reference does
    orc = q.OracleTable(q.GOLDEN / "commands_2k.csv", [])
    orc_idx = q.OracleTable(q.GOLDEN / "commands_2k.csv", pq.DEFAULT_INDEXES)
    chain = [("risk_level", ">", "3")]
    assert eng.select_ids(chain) == orc_idx.select_ids(chain)[0]                # indexes were rebuilt too
    assert len(csv.read_bytes().split(b"\n")) == n0 + 1
    wl2 = pq.WhereList([("risk_level", ">=", "4"), "OR", ("shell_type", "=", "fish")])
    want = [i for i in range(orc.n) if not q.load_oracle().orc_eval_where(C.byref(orc.rows[i]), wl2.ptr)]
    rs = L.executeQueryDeleteHIP(eng.e, b"Commands", wl2.ptr)
    assert rs.contents.numRecords == n0 - len(want)
    L.freeResultSet(rs)
    assert [eng.record(i).command_id for i in range(eng.e.contents.num_records)] == [orc.rows[i].command_id for i in want]
    # the device columns were compacted in place (no rebuild): every query must answer exactly like an
    # engine built from scratch over the surviving rows -- scan mode, index mode, strings, ranges
    lines = (q.GOLDEN / "commands_2k.csv").read_bytes().split(b"\n")
    body = [ln for ln in lines[1:] if ln.strip()]
    fresh_csv = tmp_path / "survivors.csv"
    fresh_csv.write_bytes(b"\n".join([lines[0]] + [body[i] for i in want]) + b"\n")
    fresh = pq.HipEngine(fresh_csv, pq.DEFAULT_INDEXES)
    assert fresh.n == eng.e.contents.num_records == len(want)
    for chain in ([("risk_level", "<", "3")], [("user_id", ">=", "1100"), "AND", ("exit_code", "=", "0")],
                  [("user_name", ">", "student1100"), "OR", ("shell_type", "=", "zsh")], [("sudo_used", "=", "TRUE")],
                  [("command_id", "<", "50")], [("base_command", "=", "ls"), "AND", ("host_name", "<=", "labpc-08")],
                  [("shell_type", "=", "fish")], [("risk_level", ">=", "4")], []):
        assert eng.select_ids(chain) == fresh.select_ids(chain), chain
        assert eng.count(chain) == fresh.count(chain)
    got = eng.select(["command_id", "user_name", "raw_command"], [("exit_code", "!=", "0")])
    ref = fresh.select(["command_id", "user_name", "raw_command"], [("exit_code", "!=", "0")])
    assert got["rows"] == ref["rows"] and got["numRecords"] > 0
    fresh.close()
    # delete everything, then nothing
    rs = L.executeQueryDeleteHIP(eng.e, b"Commands", pq.WhereList([("risk_level", "<", "4")]).ptr)
    L.freeResultSet(rs)
    assert eng.e.contents.num_records == 0 and eng.select_ids([]) == [] and eng.count([("risk_level", "=", "1")]) == 0
    rs = L.executeQueryDeleteHIP(eng.e, b"Commands", pq.WhereList([]).ptr)
    assert rs.contents.numRecords == 0
    L.freeResultSet(rs)
    eng.close()


def test_columnar_select_equals_the_string_result_set():
    """executeQuerySelectColumnarHIP (device-gathered typed columns) against executeQuerySelectHIP on golden
    queries: same rows in the same order (scan and index mode, SELECT *, unknown columns), every cell's text
    identical; hipColumnarHead + printTable prints what the reference printed (print_golden.json)."""
    import tempfile
    L = pq.lib()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    eng = pq.HipEngine(q.GOLDEN / "commands_2k.csv", pq.DEFAULT_INDEXES)
    cases = [(None, [("risk_level", "=", "5")]),
             (["command_id", "raw_command", "user_name", "risk_level", "timestamp"], [("sudo_used", "=", "TRUE"), "AND", ("risk_level", ">", "2")]),
             (["bogus", "command_id", "sudo_used", "exit_code"], [("risk_level", ">", "3")]),
             (["user_name", "working_directory", "base_command", "host_name", "shell_type"],
              [("user_id", "=", "1001"), "OR", [("user_name", "=", "student1002"), "AND", ("shell_type", "=", "zsh")]]),
             (["command_id"], [("risk_level", ">", "9")]),
             (None, [])]
    for cols, chain in cases:
        want = eng.select(cols, chain)
        got = eng.select_columnar(cols, chain)
        assert got["success"] and got["numRecords"] == want["numRecords"] and got["columns"] == want["columns"]
        assert got["rows"] == want["rows"], (cols, chain)
        eng.free_columnar(got)
    gold = json.loads((q.GOLDEN / "print_golden.json").read_text())
    for case in gold:
        if case["indexes"] != "default":
            continue
        sql = case["sql"]
        sel = sql[len("SELECT "):sql.index(" FROM ")]
        cols = None if sel.strip() == "*" else [c.strip() for c in sel.split(",")]
        chain = q.parse_where_dump(q.call_text(L.hipDumpParse, sql.encode()).split(q.RS, 1)[1])
        got = eng.select_columnar(cols, chain, text=False)
        got["handle"].contents.queryTime = 0.0
        head = L.hipColumnarHead(got["handle"], case["limit"])
        with tempfile.NamedTemporaryFile(suffix=".txt") as tf:
            f = libc.fopen(tf.name.encode(), b"w")
            L.printTable(f, head, case["limit"])
            libc.fclose(f)
            text = open(tf.name, encoding="latin-1").read()
        L.freeResultSetHead(head, case["limit"])
        eng.free_columnar(got)
        assert text == case["text"], case["name"]
    eng.close()


def test_engine_is_safe_under_concurrent_callers(tmp_path):
    """The reference's OpenMP driver calls one engine from several threads (QPEOMP.c:234-291).
    8 reader threads (scan mode, index mode, COUNT, projection) run against a writer that keeps
    INSERTing and DELETEing a row no reader's predicate matches: every answer must equal the
    single-threaded one."""
    import threading
    csv = tmp_path / "data.csv"
    shutil.copy(q.GOLDEN / "commands_2k.csv", csv)
    L = pq.lib()
    eng = pq.HipEngine(csv, pq.DEFAULT_INDEXES)
    n0 = eng.n
    chains = [[("risk_level", ">=", "4")], [("sudo_used", "=", "TRUE"), "AND", ("risk_level", ">", "2")],
              [("user_name", "=", "student1030"), "OR", ("shell_type", "=", "fish")],
              [("exit_code", "!=", "0"), "AND", ("user_id", ">=", "1100")], [("command_id", "<", "100")],
              [("risk_level", ">=", "4"), "AND", ("exit_code", "=", "0")]]
    want_ids = [eng.select_ids(c) for c in chains]
    want_cnt = [eng.count(c) for c in chains]
    want_rows = [eng.select(["command_id", "user_name", "raw_command"], c)["rows"] for c in chains]
    errors, stop = [], threading.Event()

    def reader(k):
        try:
            for it in range(40):
                i = (k + it) % len(chains)
                if eng.select_ids(chains[i]) != want_ids[i]:
                    errors.append(("ids", k, i))
                if eng.count(chains[i]) != want_cnt[i]:
                    errors.append(("count", k, i))
                if it % 4 == 0 and eng.select(["command_id", "user_name", "raw_command"], chains[i])["rows"] != want_rows[i]:
                    errors.append(("rows", k, i))
        except Exception as e:                                      # noqa: BLE001
            errors.append(("exception", k, repr(e)))

    def writer():
        r = pq.Record()
        r.command_id, r.exit_code, r.user_id, r.risk_level, r.sudo_used = 999999, 0, 1000, 1, False
        r.raw_command, r.base_command, r.shell_type = b"echo x", b"echo", b"bash"
        r.timestamp, r.working_directory, r.user_name, r.host_name = b"2025-12-01T12:00:00.000Z", b"/tmp", b"zz_writer", b"zz-host"
        wl = pq.WhereList([("command_id", "=", "999999")])
        try:
            while not stop.is_set():
                if not L.executeQueryInsertHIP(eng.e, b"Commands", C.byref(r)):
                    errors.append(("insert failed",))
                rs = L.executeQueryDeleteHIP(eng.e, b"Commands", wl.ptr)
                if rs.contents.numRecords != 1:
                    errors.append(("delete count", rs.contents.numRecords))
                L.freeResultSet(rs)
        except Exception as e:                                      # noqa: BLE001
            errors.append(("writer exception", repr(e)))

    threads = [threading.Thread(target=reader, args=(k,)) for k in range(8)]
    w = threading.Thread(target=writer)
    w.start()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    stop.set()
    w.join()
    assert not errors, errors[:5]
    assert eng.e.contents.num_records == n0
    assert [eng.select_ids(c) for c in chains] == want_ids
    eng.close()


def test_engine_fails_loudly_without_a_device(tmp_path):
    code = ("import sys; sys.path.insert(0, %r); import qpelib as q; "
            "q.pq.HipEngine(%r, [])" % (str(q.ROOT / "tests"), str(q.GOLDEN / "commands_2k.csv")))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    run = subprocess.run(["python", "-c", code], env=env, capture_output=True, timeout=300)
    assert run.returncode != 0
    assert b"HIP engine" in run.stderr and b"no CPU fallback" in run.stderr
