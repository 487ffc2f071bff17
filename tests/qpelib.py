"""Shared ctypes plumbing for the tests, the golden-vector maker and bench.py.

Three shared objects are involved:

* ``oracle/libqpe_oracle.so``      -- our CPU restatement (checker only)
* ``oracle/_ref/libqpeseq_ref.so`` -- the real reference, built in the authoring
  container from /root/reference (git-ignored, travels to the GPU box prebuilt)
* ``parallel-query-processing-system_amd/libpqps_hip.so`` -- the product (HIP engine + C-ABI shim)

The struct classes mirror include/logType.h and include/executeEngine-serial.h
(layout-identical to the reference's headers of the same name).
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib
import subprocess

ROOT = pathlib.Path(__file__).resolve().parent.parent
PKG = ROOT / "parallel-query-processing-system_amd"
ORACLE_DIR = ROOT / "oracle"
GOLDEN = ROOT / "tests" / "golden"

US, RS = "\x1f", "\x1e"

COLUMNS = ["command_id", "raw_command", "base_command", "shell_type", "exit_code", "timestamp",
           "sudo_used", "working_directory", "user_id", "user_name", "host_name", "risk_level"]
FIELD_UINT64, FIELD_INT, FIELD_STRING, FIELD_BOOL = 0, 1, 2, 3
SCHEMA_TYPES = dict(command_id=FIELD_UINT64, exit_code=FIELD_INT, user_id=FIELD_INT,
                    risk_level=FIELD_INT, sudo_used=FIELD_BOOL)
DEFAULT_INDEXES = [("command_id", 0), ("user_id", 1), ("risk_level", 1), ("exit_code", 1), ("sudo_used", 3)]


class Record(C.Structure):
    _fields_ = [
        ("command_id", C.c_ulonglong),
        ("raw_command", C.c_char * 512),
        ("base_command", C.c_char * 100),
        ("shell_type", C.c_char * 20),
        ("exit_code", C.c_int),
        ("timestamp", C.c_char * 30),
        ("sudo_used", C.c_bool),
        ("working_directory", C.c_char * 200),
        ("user_id", C.c_int),
        ("user_name", C.c_char * 50),
        ("host_name", C.c_char * 100),
        ("risk_level", C.c_int),
    ]


assert C.sizeof(Record) == 1040


class WhereClause(C.Structure):
    pass


WhereClause._fields_ = [
    ("attribute", C.c_char_p),
    ("operator", C.c_char_p),
    ("value", C.c_char_p),
    ("value_type", C.c_int),
    ("next", C.POINTER(WhereClause)),
    ("logical_op", C.c_char_p),
    ("sub", C.POINTER(WhereClause)),
]
assert C.sizeof(WhereClause) == 56


class ResultSet(C.Structure):
    _fields_ = [
        ("numRecords", C.c_int),
        ("numColumns", C.c_int),
        ("columnNames", C.POINTER(C.c_char_p)),
        ("columnTypes", C.POINTER(C.c_int)),
        ("data", C.POINTER(C.POINTER(C.c_char_p))),
        ("queryTime", C.c_double),
        ("success", C.c_bool),
    ]


assert C.sizeof(ResultSet) == 48


class Engine(C.Structure):
    _fields_ = [
        ("tableName", C.c_char_p),
        ("bplus_tree_roots", C.c_void_p),
        ("num_indexes", C.c_int),
        ("indexed_attributes", C.POINTER(C.c_char_p)),
        ("attribute_types", C.POINTER(C.c_int)),
        ("all_records", C.POINTER(C.POINTER(Record))),
        ("num_records", C.c_int),
        ("datafile", C.c_char_p),
        ("record_block", C.c_void_p),
    ]


assert C.sizeof(Engine) == 72


# --------------------------------------------------------------------------
# WHERE lists.  A python "chain" is a list alternating items and the strings
# "AND"/"OR";  an item is ("attr", "op", "value") or a nested chain (list).
#   [("sudo_used","=","TRUE"), "OR", [("risk_level","=","5"), "AND", ("shell_type","=","bash")]]
# --------------------------------------------------------------------------
class WhereList:
    """Owns the ctypes nodes of one whereClauseS list (keeps them alive)."""

    def __init__(self, chain):
        self._keep = []
        self.head = self._build(chain)

    def _build(self, chain):
        if not chain:
            return None
        items = chain[0::2]
        ops = chain[1::2]
        nodes = []
        for it in items:
            n = WhereClause()
            if isinstance(it, list):
                n.attribute = None
                n.operator = None
                n.value = None
                sub = self._build(it)
                n.sub = C.pointer(sub) if sub is not None else None
            else:
                a, o, v = it[:3]
                n.attribute = a.encode() if a is not None else None
                n.operator = o.encode() if o is not None else None
                n.value = v.encode("latin-1") if v is not None else None
                n.value_type = it[3] if len(it) > 3 else 0
            self._keep.append(n)
            nodes.append(n)
        for i, n in enumerate(nodes):
            if i + 1 < len(nodes):
                n.next = C.pointer(nodes[i + 1])
                op = ops[i]
                n.logical_op = op.encode() if op is not None else None
            else:
                n.logical_op = None
        return nodes[0]

    @property
    def ptr(self):
        return C.byref(self.head) if self.head is not None else None


def parse_where_dump(text: str):
    """Inverse of ref_harness.c:dump_where / host dump: text -> python chain."""
    pos = 0

    def chain():
        nonlocal pos
        assert text[pos] == "["
        pos += 1
        out = []
        while text[pos] != "]":
            if text[pos] == "(":
                pos += 1
                item = chain()
                assert text[pos] == ")"
                pos += 1
            else:
                end = text.index(RS, pos)
                parts = text[pos:end].split(US)
                item = (parts[0], parts[1], parts[2], int(parts[3]))
                # the logic op is the last field of this record
                pos = end - len(parts[4]) - 1
            assert text[pos] == US
            end = text.index(RS, pos)
            logic = text[pos + 1:end]
            pos = end + 1
            out.append(item)
            out.append(None if logic == "<end>" else logic)
        pos += 1
        if out:
            out.pop()          # trailing logic of the last node is always <end>
        return out

    return chain()


def chain_to_jsonable(chain):
    """Chain -> JSON: leaves become {"a","o","v","t"}, nested chains {"sub": [...]}."""
    out = []
    for i, x in enumerate(chain):
        if i % 2 == 1:
            out.append(x)
        elif isinstance(x, list):
            out.append({"sub": chain_to_jsonable(x)})
        else:
            out.append({"a": x[0], "o": x[1], "v": x[2], "t": x[3] if len(x) > 3 else 0})
    return out


def chain_from_jsonable(obj):
    out = []
    for i, x in enumerate(obj):
        if i % 2 == 1:
            out.append(x)
        elif "sub" in x:
            out.append(chain_from_jsonable(x["sub"]))
        else:
            out.append((x["a"], x["o"], x["v"], x.get("t", 0)))
    return out


# --------------------------------------------------------------------------
# Library loaders
# --------------------------------------------------------------------------
def build_oracle():
    subprocess.run(["make", "-s", "-C", str(ORACLE_DIR)], check=True, stdout=subprocess.DEVNULL)


_oracle = None


def load_oracle():
    global _oracle
    if _oracle is not None:
        return _oracle
    so = ORACLE_DIR / "libqpe_oracle.so"
    if not so.exists():
        build_oracle()
    lib = C.CDLL(str(so))
    W = C.POINTER(WhereClause)
    lib.orc_eval_where.restype = C.c_bool
    lib.orc_eval_where.argtypes = [C.POINTER(Record), W]
    lib.orc_check_condition.restype = C.c_bool
    lib.orc_check_condition.argtypes = [C.POINTER(Record), W]
    lib.orc_linear_search.restype = C.c_int
    lib.orc_linear_search.argtypes = [C.POINTER(C.POINTER(Record)), C.c_int, W, C.POINTER(C.c_int)]
    lib.orc_fill_record.restype = None
    lib.orc_fill_record.argtypes = [C.POINTER(Record), C.c_char_p]
    lib.orc_load_csv.restype = C.c_int
    lib.orc_load_csv.argtypes = [C.c_char_p, C.POINTER(C.POINTER(Record))]
    lib.orc_index_build.restype = C.c_int
    lib.orc_index_build.argtypes = [C.POINTER(Record), C.c_int, C.c_char_p, C.POINTER(C.c_int)]
    lib.orc_select_ids.restype = C.c_longlong
    lib.orc_select_ids.argtypes = [C.POINTER(Record), C.c_int, C.c_int, C.POINTER(C.c_char_p),
                                   C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_int)), W,
                                   C.POINTER(C.c_uint32), C.c_longlong, C.POINTER(C.c_longlong)]
    lib.orc_attr_string.restype = None
    lib.orc_attr_string.argtypes = [C.POINTER(Record), C.c_char_p, C.c_char_p, C.c_size_t]
    lib.orc_partition.restype = None
    lib.orc_partition.argtypes = [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.orc_scan_columns.restype = C.c_longlong
    lib.orc_scan_columns.argtypes = [C.c_void_p, W, C.c_uint32, C.POINTER(C.c_uint32), C.c_longlong, C.c_int]
    _oracle = lib
    return lib


_ref = False


def load_ref():
    """The real reference (+ our harness); None when oracle/_ref was never built."""
    global _ref
    if _ref is not False:
        return _ref
    so = ORACLE_DIR / "_ref" / "libqpeseq_ref.so"
    if not so.exists():
        _ref = None
        return None
    lib = C.CDLL(str(so))
    W = C.POINTER(WhereClause)
    lib.refh_open.restype = C.c_void_p
    lib.refh_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
    lib.refh_close.argtypes = [C.c_void_p]
    lib.refh_close.restype = None
    lib.refh_num_records.argtypes = [C.c_void_p]
    lib.refh_get_record.argtypes = [C.c_void_p, C.c_int, C.POINTER(Record)]
    lib.refh_get_record.restype = None
    for name in ("refh_tokens", "refh_parse"):
        f = getattr(lib, name)
        f.restype = C.c_longlong
        f.argtypes = [C.c_char_p, C.c_char_p, C.c_longlong]
    lib.refh_select.restype = C.c_longlong
    lib.refh_select.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_longlong]
    lib.refh_print.restype = C.c_int
    lib.refh_print.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p]
    lib.refh_index_order.restype = C.c_int
    lib.refh_index_order.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib.refh_find_range.restype = C.c_int
    lib.refh_find_range.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.POINTER(C.c_int)]
    lib.evaluateWhereClause.restype = C.c_bool
    lib.evaluateWhereClause.argtypes = [C.POINTER(Record), W]
    lib.getRecordFromLine.restype = C.POINTER(Record)
    lib.getRecordFromLine.argtypes = [C.c_char_p]
    lib.get_attribute_string_value.restype = C.c_void_p
    lib.get_attribute_string_value.argtypes = [C.POINTER(Record), C.c_char_p]
    _ref = lib
    return lib


def call_text(fn, *args, cap=1 << 16):
    """Calls a (…, char *out, long long cap) -> needed-bytes function, growing the buffer."""
    while True:
        buf = C.create_string_buffer(cap)
        need = fn(*args, buf, cap)
        if need < 0:
            return None
        if need < cap:
            return buf.raw[:need].decode("latin-1")
        cap = need + 16


def c_str_array(strings):
    arr = (C.c_char_p * max(1, len(strings)))()
    for i, s in enumerate(strings):
        arr[i] = s.encode()
    return arr


def c_int_array(vals):
    arr = (C.c_int * max(1, len(vals)))()
    for i, v in enumerate(vals):
        arr[i] = v
    return arr


class RefEngine:
    """QPESeq engine from oracle/_ref (in-container pinning only)."""

    def __init__(self, csv_path, indexes):
        self.lib = load_ref()
        self._names = c_str_array([a for a, _ in indexes])
        self._types = c_int_array([t for _, t in indexes])
        self.h = self.lib.refh_open(str(csv_path).encode(), len(indexes), self._names, self._types)
        self.n = self.lib.refh_num_records(self.h)

    def select(self, sql):
        text = call_text(self.lib.refh_select, self.h, sql.encode("latin-1"))
        if text is None:
            return None
        recs = text.split(RS)[:-1]
        head = recs[0].split(US)
        names = recs[1].split(US)[:-1]
        rows = [r.split(US)[:-1] for r in recs[2:]]
        assert int(head[0]) == len(rows)
        return dict(numRecords=int(head[0]), numColumns=int(head[1]), success=head[2] == "1",
                    columns=names, rows=rows)

    def record(self, i):
        r = Record()
        self.lib.refh_get_record(self.h, i, C.byref(r))
        return r

    def index_order(self, idx):
        out = (C.c_int * max(1, self.n))()
        k = self.lib.refh_index_order(self.h, idx, out)
        return list(out[:k])

    def close(self):
        if self.h:
            self.lib.refh_close(self.h)
            self.h = None


class OracleTable:
    """CSV loaded by the oracle + emulated indexes."""

    def __init__(self, csv_path, indexes=()):
        self.lib = load_oracle()
        rows = C.POINTER(Record)()
        self.n = self.lib.orc_load_csv(str(csv_path).encode(), C.byref(rows))
        if self.n < 0:
            raise FileNotFoundError(csv_path)
        self.rows = rows
        self.set_indexes(indexes)

    def set_indexes(self, indexes):
        self.indexes = list(indexes)
        self._perms = []
        for attr, _t in self.indexes:
            p = (C.c_int * max(1, self.n))()
            rc = self.lib.orc_index_build(self.rows, self.n, attr.encode(), p)
            assert rc == 0, attr
            self._perms.append(p)
        self._names = c_str_array([a for a, _ in self.indexes])
        self._types = c_int_array([t for _, t in self.indexes])
        self._permptrs = (C.POINTER(C.c_int) * max(1, len(self._perms)))()
        for i, p in enumerate(self._perms):
            self._permptrs[i] = C.cast(p, C.POINTER(C.c_int))

    def index_order(self, i):
        return list(self._perms[i][:self.n])

    def select_ids(self, chain, cap=None):
        wl = WhereList(chain)
        cap = cap if cap is not None else 8 * self.n + 16
        out = (C.c_uint32 * max(1, cap))()
        cand = C.c_longlong(0)
        k = self.lib.orc_select_ids(self.rows, self.n, len(self.indexes), self._names, self._types,
                                    self._permptrs, wl.ptr, out, cap, C.byref(cand))
        return list(out[:min(k, cap)]), k, cand.value

    def cell(self, row, attr):
        buf = C.create_string_buffer(1200)
        self.lib.orc_attr_string(C.byref(self.rows[row]), attr.encode(), buf, 1200)
        return buf.value.decode("latin-1")

    def project(self, ids, columns):
        cols = columns if columns else COLUMNS
        return [[self.cell(r, c) for c in cols] for r in ids]
