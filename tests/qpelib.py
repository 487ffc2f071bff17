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
import importlib.util
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
PKG = ROOT / "parallel-query-processing-system_amd"
ORACLE_DIR = ROOT / "oracle"
GOLDEN = ROOT / "tests" / "golden"


def load_package():
    """Imports parallel-query-processing-system_amd/ (not a valid identifier) as `pqps_amd`."""
    if "pqps_amd" in sys.modules:
        return sys.modules["pqps_amd"]
    spec = importlib.util.spec_from_file_location("pqps_amd", PKG / "__init__.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["pqps_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


pq = load_package()
# struct mirrors and WHERE-list builder live with the product binding
Record, WhereClause, ResultSet, Engine, WhereList = pq.Record, pq.WhereClause, pq.ResultSet, pq.EngineS, pq.WhereList
COLUMNS, DEFAULT_INDEXES = pq.COLUMNS, pq.DEFAULT_INDEXES
FIELD_UINT64, FIELD_INT, FIELD_STRING, FIELD_BOOL = 0, 1, 2, 3

US, RS = "\x1f", "\x1e"


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
    lib.orc_select_ids_v.restype = C.c_longlong
    lib.orc_select_ids_v.argtypes = [C.POINTER(Record), C.c_int, C.c_int, C.POINTER(C.c_char_p),
                                     C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_int)), C.POINTER(WhereClause),
                                     C.POINTER(C.c_uint32), C.c_longlong, C.POINTER(C.c_longlong), C.c_int]
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


_merge = None


def pq_merge():
    """parallel-query-processing-system_amd/merge.py as a module (no torch import of its own)."""
    global _merge
    if _merge is None:
        import importlib.util
        spec = importlib.util.spec_from_file_location("pqps_merge", PKG / "merge.py")
        _merge = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_merge)
    return _merge


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
    if hasattr(lib, "refh_select_where"):
        lib.refh_select_where.restype = C.c_longlong
        lib.refh_select_where.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, W, C.c_char_p, C.c_longlong]
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


_ref_omp = False


def load_ref_omp():
    """The reference's OpenMP engine (+ oracle/ref_harness_omp.c); None when oracle/_ref was never built."""
    global _ref_omp
    if _ref_omp is not False:
        return _ref_omp
    so = ORACLE_DIR / "_ref" / "libqpeomp_ref.so"
    if not so.exists():
        _ref_omp = None
        return None
    lib = C.CDLL(str(so))
    lib.refo_open.restype = C.c_void_p
    lib.refo_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
    lib.refo_close.argtypes = [C.c_void_p]
    lib.refo_close.restype = None
    lib.refo_num_records.argtypes = [C.c_void_p]
    lib.refo_select_where.restype = C.c_longlong
    lib.refo_select_where.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(WhereClause), C.c_char_p, C.c_longlong]
    _ref_omp = lib
    return lib


class RefOmpEngine:
    """QPEOMP's engine from oracle/_ref (in-container pinning only; run it with OMP_NUM_THREADS=1)."""

    def __init__(self, csv_path, indexes):
        self.lib = load_ref_omp()
        self._names = c_str_array([a for a, _ in indexes])
        self._types = c_int_array([t for _, t in indexes])
        self.h = self.lib.refo_open(str(csv_path).encode(), len(indexes), self._names, self._types)
        self.n = self.lib.refo_num_records(self.h)

    def select_where(self, columns, chain):
        wl = WhereList(chain)
        items = c_str_array(columns or [])
        return RefEngine._result(call_text(self.lib.refo_select_where, self.h, items, len(columns or []), wl.ptr))

    def close(self):
        if self.h:
            self.lib.refo_close(self.h)
            self.h = None


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
        return self._result(call_text(self.lib.refh_select, self.h, sql.encode("latin-1")))

    @staticmethod
    def _result(text):
        if text is None:
            return None
        recs = text.split(RS)[:-1]
        head = recs[0].split(US)
        names = recs[1].split(US)[:-1]
        rows = [r.split(US)[:-1] for r in recs[2:]]
        assert int(head[0]) == len(rows)
        return dict(numRecords=int(head[0]), numColumns=int(head[1]), success=head[2] == "1",
                    columns=names, rows=rows)

    def select_where(self, columns, chain):
        """executeQuerySelectSerial on a chain built here (no parser: any number of conditions per level)."""
        wl = WhereList(chain)
        items = c_str_array(columns or [])
        text = call_text(self.lib.refh_select_where, self.h, items, len(columns or []), wl.ptr)
        return self._result(text)

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

    def select_ids(self, chain, cap=None, probe_bool=False):
        """probe_bool: the OpenMP / MPI engines' walk (BOOL indexes probed too), one thread."""
        wl = WhereList(chain)
        cap = cap if cap is not None else 8 * self.n + 16
        out = (C.c_uint32 * max(1, cap))()
        cand = C.c_longlong(0)
        k = self.lib.orc_select_ids_v(self.rows, self.n, len(self.indexes), self._names, self._types,
                                      self._permptrs, wl.ptr, out, cap, C.byref(cand), 1 if probe_bool else 0)
        return list(out[:min(k, cap)]), k, cand.value

    def cell(self, row, attr):
        buf = C.create_string_buffer(1200)
        self.lib.orc_attr_string(C.byref(self.rows[row]), attr.encode(), buf, 1200)
        return buf.value.decode("latin-1")

    def project(self, ids, columns):
        cols = columns if columns else COLUMNS
        return [[self.cell(r, c) for c in cols] for r in ids]


# --------------------------------------------------------------------------
# Synthetic tables on the host (CPU twin of pqps_synth_generate) + the oracle's
# columnar scan over them.
# --------------------------------------------------------------------------
class OrcColumns(C.Structure):
    _fields_ = [
        ("n_rows", C.c_uint64),
        ("command_id", C.c_void_p),
        ("exit_code", C.c_void_p), ("user_id", C.c_void_p), ("risk_level", C.c_void_p),
        ("sudo_used", C.c_void_p),
        ("str_code", C.c_void_p * 7),
        ("str_code_width", C.c_int * 7),
        ("str_dict", C.POINTER(C.c_char_p) * 7),
    ]


# oracle string-column order: raw_command, base_command, shell_type, timestamp, working_directory, user_name, host_name
ORC_STR = ["raw_command", "base_command", "shell_type", "timestamp", "working_directory", "user_name", "host_name"]


class HostSynth:
    """Host copy of rows [row0, row0+n) of the synthetic table (numpy arrays), same bits as the device."""

    def __init__(self, n, seed=0x5EED, row0=0, full=False):
        import numpy as np
        self.n, self.seed, self.row0 = n, seed, row0
        L = pq.lib()
        cdf, shell = pq.synth_user_tables(seed)
        self.arr = {}
        sc = pq.SynthCols()
        dt = dict(command_id=np.uint64, exit_code=np.int32, user_id=np.int32, risk_level=np.int32,
                  sudo_used=np.uint8, shell_type=np.uint8, user_name=np.uint16, host_name=np.uint8, base_command=np.uint8)
        for name, field, _w in pq.SYNTH_LAYOUT:
            a = np.zeros(max(n, 1), dtype=dt[name])
            self.arr[name] = a
            setattr(sc, field, a.ctypes.data)
        L.pqps_synth_generate_host(seed, row0, n, cdf, shell, C.byref(sc))
        for k in self.arr:
            self.arr[k] = self.arr[k][:n]
        self._dicts = {}
        self.values = {"shell_type": pq.SYNTH_SHELLS, "user_name": pq.SYNTH_USERS_DICT, "host_name": pq.SYNTH_HOSTS,
                       "base_command": pq.SYNTH_BASES}
        if full:                                        # the single-valued columns of the synthetic ENGINE table
            for name, value in pq.SYNTH_CONSTANTS.items():
                self.arr[name] = np.zeros(n, dtype=np.uint8)
                self.values[name] = [value]
        for name, vals in self.values.items():
            arr = (C.c_char_p * len(vals))(*vals)
            self._dicts[name] = arr

    def cell(self, row, column):
        """Text of one cell as get_attribute_string_value (serial:216-248) prints it."""
        v = self.arr[column][row]
        if column in self.values:
            return self.values[column][int(v)].decode("latin-1")
        if column == "sudo_used":
            return "true" if v else "false"
        return str(int(v))

    def orc_columns(self):
        oc = OrcColumns()
        oc.n_rows = self.n
        for name in ("command_id", "exit_code", "user_id", "risk_level", "sudo_used"):
            setattr(oc, name, self.arr[name].ctypes.data)
        for k, name in enumerate(ORC_STR):
            if name in self._dicts:
                oc.str_code[k] = self.arr[name].ctypes.data
                oc.str_code_width[k] = self.arr[name].dtype.itemsize
                oc.str_dict[k] = C.cast(self._dicts[name], C.POINTER(C.c_char_p))
        return oc

    def oracle_scan(self, chain, id_base=0, nthreads=1):
        import numpy as np
        lib = load_oracle()
        wl = WhereList(chain)
        oc = self.orc_columns()
        out = np.zeros(max(self.n, 1), dtype=np.uint32)
        k = lib.orc_scan_columns(C.byref(oc), wl.ptr, id_base, out.ctypes.data_as(C.POINTER(C.c_uint32)), self.n, nthreads)
        return out[:k]


# ---- numpy restatement of the serial engine's index path (serial:358-474) over a HostSynth table ----
def host_index_order(keys):
    """(key asc, row desc): stable sort of the rows fed in descending order = B+-tree leaf order."""
    import numpy as np
    n = len(keys)
    rev = np.arange(n - 1, -1, -1, dtype=np.int64)
    return rev[np.argsort(keys[rev], kind="stable")]


def host_index_select(host, perms, probes, chain, id_base=0):
    """probes: [(column, key_lo, key_hi)] in chain order; candidates of each probe appended, then
    re-filtered by the complete WHERE, order kept.  `perms[column]` = host_index_order of that column."""
    import numpy as np
    full = np.zeros(host.n, dtype=bool)
    full[host.oracle_scan(chain)] = True
    out = []
    for name, lo, hi in probes:
        perm = perms[name]
        k = host.arr[name][perm]
        b, e = np.searchsorted(k, lo, "left"), np.searchsorted(k, hi, "right")
        cand = perm[b:max(b, e)]
        out.append(cand[full[cand]] + id_base)
    return np.concatenate(out).astype(np.uint32) if out else np.zeros(0, np.uint32)


# ---- golden ID lists: plain JSON lists (select_golden.json) or zlib + base64 of the uint32 array ----
def pack_ids(ids):
    import base64, zlib
    import numpy as np
    return base64.b64encode(zlib.compress(np.asarray(ids, dtype=np.uint32).tobytes(), 9)).decode()


def case_ids(case):
    """The expected row-ID list of a golden SELECT case, or None if the case carries none."""
    if "ids" in case:
        return list(case["ids"])
    if "ids_zlib_b64" in case:
        import base64, zlib
        import numpy as np
        return np.frombuffer(zlib.decompress(base64.b64decode(case["ids_zlib_b64"])), dtype=np.uint32).tolist()
    return None
