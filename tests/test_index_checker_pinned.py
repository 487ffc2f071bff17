"""Pins the CHECKER of the scale tests.  tests/test_gpu_index_scale.py and tests/test_gpu_engine_lanes.py compare the
device's index mode at 300 k - 100 M rows with a numpy restatement of the serial engine's index path
(qpelib.host_index_order / host_index_select, serial:358-474) because the oracle works on 1040-byte records and would
need 100 GB there.  Here that numpy restatement is compared with the oracle itself (oracle/qpe_oracle.c:
orc_index_build, orc_select_ids -- the restatement that IS pinned against the compiled reference by
tests/golden/index_order_golden.json and select*_golden.json) on the golden CSV and on a 1 M-row synthetic table
materialised as records.  No GPU involved."""
import ctypes as C

import numpy as np
import pytest

import qpelib as q

pq = q.pq

REC_DT = np.dtype({"names": ["command_id", "raw_command", "base_command", "shell_type", "exit_code", "timestamp",
                             "sudo_used", "working_directory", "user_id", "user_name", "host_name", "risk_level"],
                   "formats": ["<u8", "S512", "S100", "S20", "<i4", "S30", "u1", "S200", "<i4", "S50", "S100", "<i4"],
                   "offsets": [0, 8, 520, 620, 640, 644, 674, 675, 876, 880, 930, 1032], "itemsize": 1040})
IMIN, IMAX, UMAX = -2**31, 2**31 - 1, 2**64 - 1


class RecordTable:
    """Oracle-side table over a numpy record array (what OracleTable is for a CSV)."""

    def __init__(self, recs, indexes):
        self.lib = q.load_oracle()
        self.recs, self.n = recs, len(recs)
        self.rows = C.cast(recs.ctypes.data, C.POINTER(q.Record))
        self.indexes = list(indexes)
        self.perms = []
        for attr, _t in self.indexes:
            p = (C.c_int * max(1, self.n))()
            assert self.lib.orc_index_build(self.rows, self.n, attr.encode(), p) == 0
            self.perms.append(p)
        self._names = q.c_str_array([a for a, _ in self.indexes])
        self._types = q.c_int_array([t for _, t in self.indexes])
        self._permptrs = (C.POINTER(C.c_int) * max(1, len(self.perms)))()
        for i, p in enumerate(self.perms):
            self._permptrs[i] = C.cast(p, C.POINTER(C.c_int))

    def select_ids(self, chain):
        wl = q.WhereList(chain)
        cap = 4 * self.n + 16
        out = np.zeros(cap, dtype=np.uint32)
        cand = C.c_longlong(0)
        k = self.lib.orc_select_ids(self.rows, self.n, len(self.indexes), self._names, self._types, self._permptrs, wl.ptr,
                                    out.ctypes.data_as(C.POINTER(C.c_uint32)), cap, C.byref(cand))
        assert k <= cap
        return out[:k]


def synthetic_records(host):
    n = host.n
    recs = np.zeros(n, dtype=REC_DT)
    a = host.arr
    recs["command_id"], recs["exit_code"], recs["user_id"], recs["risk_level"] = a["command_id"], a["exit_code"], a["user_id"], a["risk_level"]
    recs["sudo_used"] = a["sudo_used"]
    recs["shell_type"] = np.array(pq.SYNTH_SHELLS, dtype="S20")[a["shell_type"]]
    recs["user_name"] = np.array(pq.SYNTH_USERS_DICT, dtype="S50")[a["user_name"]]
    recs["host_name"] = np.array(pq.SYNTH_HOSTS, dtype="S100")[a["host_name"]]
    recs["base_command"] = np.array(pq.SYNTH_BASES, dtype="S100")[a["base_command"]]
    for name, value in pq.SYNTH_CONSTANTS.items():
        recs[name] = value
    return recs


class ArrHost:
    """The two things host_index_select needs from a table, over arbitrary numpy columns + an oracle scan."""

    def __init__(self, arr, scan):
        self.arr, self._scan = arr, scan
        self.n = len(next(iter(arr.values())))

    def oracle_scan(self, chain):
        return self._scan(chain)


CASES = [
    # (chain, probes in the order the serial engine makes them: top-level conditions on u64 / int indexes, serial:358-433)
    ([("risk_level", ">", "3")], [("risk_level", 4, IMAX)]),
    ([("user_id", "=", "1001")], [("user_id", 1001, 1001)]),
    ([("risk_level", "!=", "2")], [("risk_level", IMIN, IMAX)]),
    ([("command_id", "<", "10")], [("command_id", 0, 9)]),
    ([("command_id", ">=", "1990")], [("command_id", 1990, UMAX)]),
    ([("risk_level", ">=", "4"), "AND", ("exit_code", "=", "0")], [("risk_level", 4, IMAX), ("exit_code", 0, 0)]),       # duplicates
    ([("risk_level", "=", "5"), "OR", ("user_name", "=", "student1030")], [("risk_level", 5, 5)]),                      # OR loss
    ([("sudo_used", "=", "TRUE"), "AND", ("user_id", "<=", "1010")], [("user_id", IMIN, 1010)]),                        # BOOL index not probed
    ([("exit_code", "<", "1"), "AND", [("risk_level", ">", "2"), "OR", ("user_id", "=", "1003")]], [("exit_code", IMIN, 0)]),   # nested: no probe inside
]
INDEXES = [("command_id", 0), ("user_id", 1), ("risk_level", 1), ("exit_code", 1), ("sudo_used", 3)]


def check_table(rt, host_like):
    perms = {}
    for (attr, _t), p in zip(rt.indexes, rt.perms):
        if attr in host_like.arr:
            want = np.frombuffer(p, dtype=np.int32, count=rt.n)
            got = q.host_index_order(host_like.arr[attr])
            assert np.array_equal(got, want), attr                         # leaf order: key asc, row desc
            perms[attr] = got
    for chain, probes in CASES:
        want = rt.select_ids(chain)
        got = q.host_index_select(host_like, perms, probes, chain)
        assert np.array_equal(got, want), chain


def test_numpy_index_restatement_equals_the_oracle_on_the_golden_csv():
    orc = q.OracleTable(q.GOLDEN / "commands_2k.csv", INDEXES)
    n = orc.n
    recs = np.frombuffer(C.string_at(C.addressof(orc.rows.contents), n * 1040), dtype=REC_DT).copy()
    rt = RecordTable(recs, INDEXES)
    plain = q.OracleTable(q.GOLDEN / "commands_2k.csv", ())
    arr = {k: recs[k] for k in ("command_id", "user_id", "risk_level", "exit_code")}
    host_like = ArrHost(arr, lambda chain: np.array(plain.select_ids(chain)[0], dtype=np.uint32))
    # the record-array table is the CSV table
    for i, (attr, _t) in enumerate(INDEXES):
        assert list(rt.perms[i][:n]) == orc.index_order(i), attr
    check_table(rt, host_like)


@pytest.mark.parametrize("n", [1_000_003])
def test_numpy_index_restatement_equals_the_oracle_on_a_synthetic_table(n):
    host = q.HostSynth(n, seed=21)
    recs = synthetic_records(q.HostSynth(n, seed=21, full=True))
    rt = RecordTable(recs, INDEXES)
    cases_backup = list(CASES)
    try:
        CASES[3] = ([("command_id", "<", "10")], [("command_id", 0, 9)])
        CASES[4] = ([("command_id", ">=", str(n - 10))], [("command_id", n - 10, UMAX)])
        check_table(rt, host)
    finally:
        CASES[:] = cases_backup
