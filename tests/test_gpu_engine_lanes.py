"""The engine API over device-resident tables and with several queries in flight (include/executeEngine-hip.h):

  * initializeEngineSyntheticHIP / initializeEngineColumnsHIP -- engines without host rows -- answer like the oracle
    (scan mode, index mode, COUNT, strings, columnar), and INSERT / DELETE keep working on them;
  * concurrent callers: several host threads issue mixed SELECT / COUNT / columnar / string queries through the
    synchronous API, the way the reference's OpenMP driver uses an engine (QPEOMP.c:234-291) -- every answer must
    be the oracle's, whichever lane it ran on;
  * asynchronous tickets: several queries in flight from one thread, results read on the device.

The oracle (tests/qpelib.HostSynth + oracle/qpe_oracle.c) is the checker, on the host twin of the same seeded table."""
import ctypes as C
import random
import threading

import numpy as np
import pytest

import qpelib as q

pq = q.pq
pytestmark = pytest.mark.gpu

N = 300_007
SEED = 0xBEEF

CHAINS = {
    "S1": [("sudo_used", "=", "FALSE"), "AND", ("user_name", "=", "student1030")],
    "Q_A": [("risk_level", ">", "3")],
    "Q_B": [("sudo_used", "=", "TRUE"), "AND", ("risk_level", ">", "2")],
    "Q_C": [("exit_code", "!=", "0"), "AND", ("user_id", ">=", "1500"), "OR", ("risk_level", "=", "5")],
    "S7": [("sudo_used", "=", "TRUE"), "OR", [("risk_level", "=", "5"), "AND", ("shell_type", "=", "bash")]],
    "dense": [("sudo_used", "=", "FALSE")],
    "u8": [("shell_type", "=", "zsh")],
    "all": [],
    "none": [("risk_level", ">", "9")],
    # the single-valued columns: decided when the WHERE is compiled, no column is read
    "const_eq": [("raw_command", "=", "cmd"), "AND", ("risk_level", "=", "4")],
    "const_ne": [("working_directory", "!=", "/home/u"), "OR", ("risk_level", "=", "5")],
    "const_lt": [("timestamp", "<", "2026"), "AND", ("host_name", ">=", "labpc-07")],
    "const_only": [("timestamp", ">", "2026")],
    "ids": [("command_id", ">=", "1000"), "AND", ("command_id", "<", "1100")],
}


@pytest.fixture(scope="module")
def host():
    return q.HostSynth(N, seed=SEED, full=True)


@pytest.fixture(scope="module")
def engine():
    eng = pq.HipEngine.synthetic(N, seed=SEED)
    yield eng
    eng.close()


def test_synthetic_dictionaries_are_the_python_twins():
    L = pq.lib()
    for name, want in list(pq.SYNTH_CONSTANTS.items()) + [("shell_type", pq.SYNTH_SHELLS), ("user_name", pq.SYNTH_USERS_DICT),
                                                            ("host_name", pq.SYNTH_HOSTS), ("base_command", pq.SYNTH_BASES)]:
        k = C.c_int()
        d = L.hipSyntheticDictionary(pq.COL[name], C.byref(k))
        want = [want] if isinstance(want, bytes) else list(want)
        assert [d[i] for i in range(k.value)] == want, name


def test_synthetic_engine_scan_and_count_match_the_oracle(engine, host):
    assert engine.n == N and not engine.e.contents.all_records
    for name, chain in CHAINS.items():
        want = host.oracle_scan(chain).tolist()
        assert engine.select_ids(chain) == want, name
        assert engine.count(chain) == len(want), name


def test_synthetic_engine_strings_and_columnar(engine, host):
    cols = ["command_id", "user_name", "sudo_used", "risk_level", "raw_command", "host_name", "nonsense"]
    for name in ("S1", "ids", "const_ne", "none"):
        chain = CHAINS[name]
        want_ids = host.oracle_scan(chain)
        want = [[host.cell(int(r), c) if c != "nonsense" else "NULL" for c in cols] for r in want_ids]
        res = engine.select(cols, chain)
        assert res["success"] and res["numRecords"] == len(want_ids) and res["rows"] == want, name
        col = engine.select_columnar(cols, chain)
        assert col["success"] and col["rows"] == want, name
        engine.free_columnar(col)
    star = engine.select(None, CHAINS["ids"])                      # SELECT *: all 12 columns
    assert star["columns"] == pq.COLUMNS
    assert star["rows"] == [[host.cell(int(r), c) for c in pq.COLUMNS] for r in host.oracle_scan(CHAINS["ids"])]


def test_synthetic_engine_index_mode(host):
    """Index probes on a device-only engine: (key asc, row desc) per probed condition, duplicates and all."""
    eng = pq.HipEngine.synthetic(N, seed=SEED, indexes=[("command_id", 0), ("user_id", 1), ("risk_level", 1)])
    perms = {c: q.host_index_order(host.arr[c]) for c in ("command_id", "user_id", "risk_level")}
    imin, imax = -2**31, 2**31 - 1
    cases = [
        ([("risk_level", ">", "3")], [("risk_level", 4, imax)]),
        ([("user_id", "=", "1001")], [("user_id", 1001, 1001)]),
        ([("command_id", ">=", str(N - 5000))], [("command_id", N - 5000, 2**64 - 1)]),
        ([("risk_level", ">=", "4"), "AND", ("user_id", "<", "1200")], [("risk_level", 4, imax), ("user_id", imin, 1199)]),
        ([("risk_level", "=", "5"), "OR", ("user_name", "=", "student1030")], [("risk_level", 5, 5)]),
        ([("sudo_used", "=", "TRUE"), "AND", ("risk_level", "!=", "1")], [("risk_level", imin, imax)]),
    ]
    try:
        for chain, probes in cases:
            want = q.host_index_select(host, perms, probes, chain).tolist()
            assert eng.select_ids(chain) == want, chain
    finally:
        eng.close()


def test_synthetic_engine_probes_bool_indexes_on_request(host):
    """hipEngineProbeBoolIndexes on a device-only engine (the OpenMP / MPI engines' row selection, omp:424-459): a
    top-level condition on the BOOL index is probed like the int ones -- rows in (key asc, row desc) order, once per
    probe -- and without the switch it is not (QPESeq's selection).  Checker: the numpy index restatement that
    tests/test_index_checker_pinned.py pins against the oracle, with the BOOL windows of omp:424-459 spelled out."""
    eng = pq.HipEngine.synthetic(N, seed=SEED, indexes=[("sudo_used", 3), ("risk_level", 1)])
    perms = {c: q.host_index_order(host.arr[c]) for c in ("sudo_used", "risk_level")}
    imin, imax = -2**31, 2**31 - 1
    cases = [
        # (chain, probes with BOOL indexes probed, probes of the serial engine)
        ([("sudo_used", "=", "TRUE")], [("sudo_used", 1, 1)], []),
        ([("sudo_used", "!=", "TRUE"), "AND", ("risk_level", ">", "3")], [("sudo_used", 0, 0), ("risk_level", 4, imax)], [("risk_level", 4, imax)]),
        ([("risk_level", "=", "5"), "AND", ("sudo_used", "=", "1")], [("risk_level", 5, 5), ("sudo_used", 1, 1)], [("risk_level", 5, 5)]),
        ([("sudo_used", "=", "TRUE"), "OR", ("user_name", "=", "student1030")], [("sudo_used", 1, 1)], []),
        ([("sudo_used", ">", "TRUE"), "OR", ("risk_level", "=", "5")], [("sudo_used", 1, 0), ("risk_level", 5, 5)], [("risk_level", 5, 5)]),
        ([("sudo_used", "<=", "FALSE")], [("sudo_used", 0, 0)], []),
    ]
    try:
        assert eng.probe_bool_indexes(True) == 0
        for chain, probes, _serial in cases:
            want = q.host_index_select(host, perms, probes, chain).tolist()
            assert eng.select_ids(chain) == want, chain
        assert eng.probe_bool_indexes(False) == 1
        for chain, _probes, serial in cases:
            want = (q.host_index_select(host, perms, serial, chain) if serial else host.oracle_scan(chain)).tolist()
            assert eng.select_ids(chain) == want, chain
    finally:
        eng.close()


def test_dense_answers_move_the_stream_to_one_lane_and_back(engine, host):
    """awaitQueryHIP tells the query stream what the answer looked like (pqps_qstream_hint_answer): while answers hold a
    quarter of the rows or more, ID queries run on one lane.  Which lane a query ran on is not observable from here;
    what is checked is that a run of dense, sparse and dense-again queries with several tickets in flight answers right."""
    seq = ["dense", "dense", "S1", "dense", "Q_A", "S1", "dense", "dense", "Q_C"]
    ctx = pq.Context(0)
    tickets = []

    def check(name0, t0):
        n, res = engine.await_ticket(t0)
        want = host.oracle_scan(CHAINS[name0])
        got = np.zeros(max(n, 1), dtype=np.uint32)
        if n:
            ctx.download(got.ctypes.data, res.ids_dev, 4 * n)
        assert n == len(want) and np.array_equal(got[:n], want), name0
        engine.release_ticket(t0)
    try:
        for name in seq + seq:
            tickets.append((name, engine.select_async(CHAINS[name])))
            if len(tickets) == 3:
                check(*tickets.pop(0))
        for name0, t0 in tickets:
            check(name0, t0)
    finally:
        ctx.close()


def test_columns_engine_equals_synthetic_engine(engine, host):
    cols = {}
    for i, name in enumerate(pq.COLUMNS):
        if pq.COLUMN_KIND[i] == pq.KIND_DICT:
            single = name in pq.SYNTH_CONSTANTS
            cols[name] = (None if single else host.arr[name], list(host.values[name]))
        else:
            cols[name] = host.arr[name]
    eng = pq.HipEngine.from_columns(N, cols, indexes=[("user_id", 1)])
    try:
        for name in ("S1", "Q_B", "S7", "const_lt", "u8"):                 # (no top-level condition on user_id: scan mode on both)
            assert eng.select_ids(CHAINS[name]) == engine.select_ids(CHAINS[name]), name
        chain = [("user_id", "=", "1777")]
        want = q.host_index_select(host, {"user_id": q.host_index_order(host.arr["user_id"])}, [("user_id", 1777, 1777)], chain).tolist()
        assert eng.select_ids(chain) == want
    finally:
        eng.close()


def test_insert_and_delete_on_an_engine_without_host_rows(host):
    eng = pq.HipEngine.synthetic(N, seed=SEED, indexes=[("risk_level", 1)])
    L = pq.lib()
    try:
        r = pq.Record()
        r.command_id, r.raw_command, r.base_command, r.shell_type = 424242424242, b"cmd", b"cmd005", b"zsh"
        r.exit_code, r.timestamp, r.sudo_used, r.working_directory = 77, b"2025-01-01T00:00:00.000Z", True, b"/home/u"
        r.user_id, r.user_name, r.host_name, r.risk_level = 4242, b"student2000", b"labpc-03", 9
        assert L.executeQueryInsertHIP(eng.e, b"commands", C.byref(r))
        assert eng.e.contents.num_records == N + 1
        assert eng.select_ids([("exit_code", "=", "77")]) == [N]
        assert eng.select_ids([("risk_level", ">", "5")]) == [N]               # through the index (re-sorted)
        got = eng.select(["command_id", "user_name", "risk_level"], [("command_id", "=", "424242424242")])
        assert got["rows"] == [["424242424242", "student2000", "9"]]
        # a value that is new to its dictionary keeps the codes order-preserving
        r.command_id, r.user_name = 424242424243, b"student1030x"
        assert L.executeQueryInsertHIP(eng.e, b"commands", C.byref(r))
        assert eng.select_ids([("user_name", "=", "student1030x")]) == [N + 1]
        assert eng.select_ids([("user_name", "=", "student1031")]) == host.oracle_scan([("user_name", "=", "student1031")]).tolist()
        # a new value for a single-valued column needs a rebuild: refused, the table is unchanged
        r.command_id, r.raw_command = 424242424244, b"other"
        assert not L.executeQueryInsertHIP(eng.e, b"commands", C.byref(r))
        assert eng.e.contents.num_records == N + 2
        # DELETE: flags + compaction on the device, the row numbers close up
        chain = [("risk_level", "=", "2")]
        gone = host.oracle_scan(chain)
        wl = pq.WhereList(chain)
        rs = L.executeQueryDeleteHIP(eng.e, b"commands", wl.ptr)
        assert rs.contents.success and rs.contents.numRecords == len(gone)
        L.freeResultSet(rs)
        assert eng.e.contents.num_records == N + 2 - len(gone)
        # numpy model of the table as it now stands: the N generated rows + the two inserted ones, minus the deleted
        risk = np.concatenate([host.arr["risk_level"], [9, 9]])
        sudo = np.concatenate([host.arr["sudo_used"], [1, 1]])
        user = np.concatenate([host.arr["user_id"], [4242, 4242]])
        keep = risk != 2
        risk, sudo, user = risk[keep], sudo[keep], user[keep]
        rows = np.arange(len(risk))
        assert eng.select_ids([("sudo_used", "=", "TRUE")]) == rows[sudo == 1].tolist()             # scan mode: ascending rows
        assert eng.select_ids([("user_id", "<", "1005")]) == rows[user < 1005].tolist()
        hit = rows[risk > 3]                                                                     # index mode: key asc, row desc
        want = sorted(hit.tolist(), key=lambda i: (int(risk[i]), -i))
        assert eng.select_ids([("risk_level", ">", "3")]) == want
        assert eng.count([("risk_level", "=", "2")]) == 0
    finally:
        eng.close()


def _expected(host, chains):
    return {name: host.oracle_scan(chain) for name, chain in chains.items()}


def test_concurrent_callers_on_the_lanes(engine, host):
    """Six host threads, mixed SELECT (ids, strings, columnar) and COUNT, through the synchronous engine API."""
    want = _expected(host, CHAINS)
    names = sorted(CHAINS)
    errors = []

    def worker(seed):
        rng = random.Random(seed)
        try:
            for _ in range(40):
                name = rng.choice(names)
                chain, w = CHAINS[name], want[name]
                kind = rng.randrange(4)
                if kind == 0:
                    got = engine.select_ids(chain)
                    assert got == w.tolist(), (name, "ids")
                elif kind == 1:
                    assert engine.count(chain) == len(w), (name, "count")
                elif kind == 2 and len(w) < 5000:
                    res = engine.select(["command_id", "risk_level"], chain)
                    assert res["rows"] == [[host.cell(int(r), "command_id"), host.cell(int(r), "risk_level")] for r in w], (name, "strings")
                else:
                    res = engine.select_columnar(["command_id"], chain, text=False)
                    vals = res["values"][0]
                    assert res["numRecords"] == len(w) and (len(w) == 0 or np.array_equal(vals, host.arr["command_id"][w])), (name, "columnar")
                    engine.free_columnar(res)
        except Exception as e:                                       # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_concurrent_callers_on_a_csv_engine_with_indexes():
    """The same on the 2 k-row golden CSV with the default five indexes: scan mode and index mode side by side."""
    csv = q.GOLDEN / "commands_2k.csv"
    eng = pq.HipEngine(csv, pq.DEFAULT_INDEXES)
    orc = q.OracleTable(csv, pq.DEFAULT_INDEXES)
    chains = [
        [("sudo_used", "=", "FALSE"), "AND", ("user_name", "=", "student1030")],
        [("risk_level", ">", "3")],
        [("risk_level", ">=", "4"), "AND", ("exit_code", "=", "0")],
        [("user_id", "=", "1001")],
        [("command_id", "<", "10")],
        [("sudo_used", "=", "TRUE"), "OR", [("risk_level", "=", "5"), "AND", ("shell_type", "=", "bash")]],
        [],
    ]
    want = [orc.select_ids(c)[0] for c in chains]
    scan_counts = []
    plain = q.OracleTable(csv, ())
    for c in chains:
        scan_counts.append(plain.select_ids(c)[1])
    errors = []

    def worker(seed):
        rng = random.Random(seed)
        try:
            for _ in range(30):
                i = rng.randrange(len(chains))
                if rng.random() < 0.7:
                    assert eng.select_ids(chains[i]) == want[i], i
                else:
                    assert eng.count(chains[i]) == scan_counts[i], i       # COUNT(*) is the scan-mode count
        except Exception as e:                                       # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(5)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    eng.close()
    assert not errors, errors[:3]


def test_async_tickets_keep_several_queries_in_flight(engine, host):
    """Three tickets outstanding at any time (the engine has four lanes), results read ON THE DEVICE."""
    ctx = pq.Context(0)
    names = sorted(CHAINS)
    want = _expected(host, CHAINS)
    ring = []
    checked = 0
    try:
        for k in range(40):
            name = names[k % len(names)]
            count_only = k % 5 == 4
            ring.append((name, count_only, engine.select_async(CHAINS[name], count_only=count_only)))
            if len(ring) == 3:
                name0, c0, t0 = ring.pop(0)
                n, res = engine.await_ticket(t0)
                assert n == len(want[name0]) == res.count, name0
                if not c0 and n:
                    got = np.zeros(n, dtype=np.uint32)
                    ctx.download(got.ctypes.data, res.ids_dev, 4 * n)
                    assert np.array_equal(got, want[name0]), name0
                    checked += 1
                engine.release_ticket(t0)
        for name0, c0, t0 in ring:
            n, res = engine.await_ticket(t0)
            assert n == len(want[name0])
            engine.release_ticket(t0)
    finally:
        ctx.close()
    assert checked > 10


def test_a_failed_query_leaves_the_lanes_usable(engine, host):
    """A WHERE that cannot be compiled for this table (a > 12-column pass is impossible here, so use an absent feature:
    a NULL attribute is fine -- it is constant false; what fails is nothing on a synthetic table) -- so instead check that
    tickets of failed and of good queries release their lanes: more queries than lanes, one after the other."""
    for _ in range(12):
        t = engine.select_async(CHAINS["Q_A"])
        engine.release_ticket(t)                                     # released without await: waits internally
    assert engine.count(CHAINS["Q_A"]) == len(host.oracle_scan(CHAINS["Q_A"]))
