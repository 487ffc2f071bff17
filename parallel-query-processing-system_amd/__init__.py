"""ctypes binding of the MI355X SELECT/WHERE backend (libpqps_hip.so).

The product is C: `csrc/pqps_hip.hip` (kernels + C-ABI shim, include/pqps_hip.h)
and the C11 engine under `engine/hip/` + `host/` (include/executeEngine-hip.h).
This module only mirrors those headers for Python callers (tests, bench.py);
it contains no compute and no fallback: if the shared library is missing or
no GPU is visible, calls fail loudly.

The directory name contains '-', so import it with::

    import importlib.util, sys
    spec = importlib.util.spec_from_file_location(
        "pqps_amd", "<repo>/parallel-query-processing-system_amd/__init__.py")
    pqps_amd = importlib.util.module_from_spec(spec); sys.modules["pqps_amd"] = pqps_amd
    spec.loader.exec_module(pqps_amd)
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib
import subprocess

PKG_DIR = pathlib.Path(__file__).resolve().parent
# (PQPS_LIB: a development build of the same library, e.g. -DPQPS_DEV_SHAPES / -DPQPS_STAMPS, for tuning runs)
LIB_PATH = pathlib.Path(os.environ.get("PQPS_LIB") or (PKG_DIR / "libpqps_hip.so"))

MAX_COLUMNS = 12
MAX_LEAVES = 32
TT_LEAVES = 6
TILE_ROWS = 4096
SLOT_HEADER_WORDS = 4
ACCEPT, REJECT = 0xFE, 0xFF
SYNTH_USERS = 2000

COLUMNS = ["command_id", "raw_command", "base_command", "shell_type", "exit_code", "timestamp",
           "sudo_used", "working_directory", "user_id", "user_name", "host_name", "risk_level"]
COL = {name: i for i, name in enumerate(COLUMNS)}
KIND_U64, KIND_I32, KIND_BOOL, KIND_DICT = 0, 1, 2, 3
COLUMN_KIND = [KIND_U64, KIND_DICT, KIND_DICT, KIND_DICT, KIND_I32, KIND_DICT,
               KIND_BOOL, KIND_DICT, KIND_I32, KIND_DICT, KIND_DICT, KIND_I32]
FIELD_UINT64, FIELD_INT, FIELD_STRING, FIELD_BOOL = 0, 1, 2, 3
DEFAULT_INDEXES = [("command_id", 0), ("user_id", 1), ("risk_level", 1), ("exit_code", 1), ("sudo_used", 3)]


# ---- include/logType.h, include/executeEngine-serial.h ---------------------
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


class ColumnarResult(C.Structure):
    """struct hipColumnarResult (include/executeEngine-hip.h)."""
    _fields_ = [("numRecords", C.c_int), ("numColumns", C.c_int), ("columnNames", C.POINTER(C.c_char_p)),
                ("columnKinds", C.POINTER(C.c_int)), ("values", C.POINTER(C.c_void_p)),
                ("dictionaries", C.POINTER(C.POINTER(C.c_char_p))), ("dictionarySizes", C.POINTER(C.c_int)),
                ("queryTime", C.c_double), ("success", C.c_bool)]


class ColumnData(C.Structure):
    """struct hipColumnData (include/executeEngine-hip.h)."""
    _fields_ = [("values", C.c_void_p), ("width", C.c_uint), ("on_device", C.c_int),
                ("dictionary", C.POINTER(C.c_char_p)), ("dictionary_count", C.c_int)]


class DeviceResult(C.Structure):
    """struct hipDeviceResult (include/executeEngine-hip.h)."""
    _fields_ = [("count", C.c_longlong), ("ids_dev", C.c_void_p), ("device", C.c_int), ("n_shards", C.c_int),
                ("shard_count", C.c_ulonglong * 16)]


class BenchResult(C.Structure):
    """struct hipBenchResult (include/engineBench.h)."""
    _fields_ = [("seconds", C.c_double), ("queries", C.c_longlong), ("matches", C.c_longlong), ("mismatches", C.c_longlong),
                ("issue_seconds", C.c_double), ("await_seconds", C.c_double),
                ("want_checksum", C.c_int), ("have_checksum", C.c_int), ("checksum", C.c_ulonglong * 2)]


class EngineS(C.Structure):
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


# ---- include/pqps_hip.h -------------------------------------------------------
class Column(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_uint32), ("reserved", C.c_uint32)]


class Leaf(C.Structure):
    _fields_ = [("column", C.c_uint32), ("negate", C.c_uint32), ("lo", C.c_uint64), ("span", C.c_uint64)]


class Predicate(C.Structure):
    _fields_ = [
        ("n_leaves", C.c_uint32),
        ("n_columns", C.c_uint32),
        ("truth", C.c_uint64),
        ("leaf", Leaf * MAX_LEAVES),
        ("on_true", C.c_uint8 * MAX_LEAVES),
        ("on_false", C.c_uint8 * MAX_LEAVES),
        ("order", C.c_uint8 * MAX_LEAVES),
    ]


class Pass(C.Structure):
    """struct hipPass (include/hipPredicate.h)."""
    _fields_ = [("pred", Predicate), ("column_ids", C.c_int * MAX_COLUMNS)]


class Plan(C.Structure):
    _fields_ = [("n_passes", C.c_int), ("passes", C.POINTER(Pass))]


class SynthCols(C.Structure):
    _fields_ = [
        ("command_id", C.c_void_p), ("exit_code", C.c_void_p), ("user_id", C.c_void_p),
        ("risk_level", C.c_void_p), ("sudo_used", C.c_void_p), ("shell_code", C.c_void_p),
        ("user_code", C.c_void_p), ("host_code", C.c_void_p), ("base_code", C.c_void_p),
    ]


# ---- include/hipPredicate.h ----------------------------------------------------
class ColumnInfo(C.Structure):
    _fields_ = [("present", C.c_int), ("kind", C.c_int), ("width", C.c_uint32),
                ("dict_count", C.c_int), ("dict", C.POINTER(C.c_char_p))]


class Schema(C.Structure):
    _fields_ = [("col", ColumnInfo * MAX_COLUMNS)]


assert C.sizeof(Record) == 1040 and C.sizeof(WhereClause) == 56
assert C.sizeof(ResultSet) == 48 and C.sizeof(EngineS) == 72


class PqpsError(RuntimeError):
    pass


# ---- WHERE lists ------------------------------------------------------------------
# A python "chain" alternates items and "AND"/"OR"; an item is (attr, op, value[, value_type])
# or a nested chain (list):
#   [("sudo_used","=","TRUE"), "OR", [("risk_level","=","5"), "AND", ("shell_type","=","bash")]]
class WhereList:
    """Owns the ctypes nodes of one whereClauseS list."""

    def __init__(self, chain):
        self._keep = []
        self.head = self._build(chain)

    def _build(self, chain):
        if not chain:
            return None
        items, ops = chain[0::2], chain[1::2]
        nodes = []
        for it in items:
            n = WhereClause()
            if isinstance(it, list):
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
        for i, n in enumerate(nodes[:-1]):
            n.next = C.pointer(nodes[i + 1])
            n.logical_op = ops[i].encode() if ops[i] is not None else None
        return nodes[0]

    @property
    def ptr(self):
        return C.byref(self.head) if self.head is not None else None


# ---- library --------------------------------------------------------------------------
_lib = None


def build_library():
    """`make` in the package directory (hipcc --offload-arch=gfx950 + gcc)."""
    subprocess.run(["make", "-s", "-C", str(PKG_DIR)], check=True)


_bench_lib = None


def bench_lib():
    """libpqps_bench.so: the lab bench above the engine API (host/engineBench.c) -- not part of the product library."""
    global _bench_lib
    if _bench_lib is None:
        lib()                                                    # the product first: the bench library links against it
        path = PKG_DIR / "libpqps_bench.so"
        if not path.exists():
            raise PqpsError(f"{path} is missing: build it with `make -C {PKG_DIR}`")
        B = C.CDLL(str(path))
        B.hipEngineBench.argtypes = [C.POINTER(C.POINTER(EngineS)), C.c_int, C.POINTER(WhereClause), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(BenchResult)]
        _bench_lib = B
    return _bench_lib


def lib():
    """The product shared object.  Never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise PqpsError(f"{LIB_PATH} is missing: build it with `make -C {PKG_DIR}` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(str(LIB_PATH))
    W = C.POINTER(WhereClause)
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    L.pqps_last_error.restype = C.c_char_p
    L.pqps_last_kernel.restype = C.c_char_p
    L.pqps_ctx_device.argtypes = [vp]
    L.pqps_copy_peer.argtypes = [vp, vp, vp, vp, C.c_size_t, vp]
    L.pqps_device_count.restype = C.c_int
    L.pqps_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.pqps_ctx_destroy.argtypes = [vp]
    L.pqps_ctx_destroy.restype = None
    L.pqps_ctx_sync.argtypes = [vp, vp]
    L.pqps_ctx_reserve.argtypes = [vp, u64]
    L.pqps_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    L.pqps_ctx_set_timing.argtypes = [vp, C.c_int]
    L.pqps_ctx_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.pqps_device_info.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int), C.POINTER(u64)]
    L.pqps_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.pqps_free.argtypes = [vp, vp]
    L.pqps_memset.argtypes = [vp, vp, C.c_int, C.c_size_t, vp]
    L.pqps_upload.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.pqps_download.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.pqps_filter_scan.argtypes = [vp, C.POINTER(Column), u32, u64, u32, C.POINTER(Predicate), vp, u64, vp, vp]
    L.pqps_filter_count.argtypes = [vp, C.POINTER(Column), u32, u64, C.POINTER(Predicate), vp, vp]
    L.pqps_filter_flags.argtypes = [vp, C.POINTER(Column), u32, u64, C.POINTER(Predicate), vp, vp, vp]
    L.pqps_filter_gather.argtypes = [vp, C.POINTER(Column), u32, vp, vp, u64, u32, C.POINTER(Predicate), vp, u64, vp, vp]
    L.pqps_index_build.argtypes = [vp, C.POINTER(Column), u64, C.c_int, vp, vp, vp]
    L.pqps_index_probe.argtypes = [vp, vp, u32, C.c_int, u64, u64, u64, vp, vp]
    L.pqps_index_select.argtypes = [vp, C.POINTER(Column), u32, C.POINTER(Column), vp, vp, C.c_int, u64, u64, u64, u32, C.POINTER(Predicate), vp, vp, u64, vp, vp]
    L.pqps_partition.argtypes = [u64, C.c_int, C.c_int, C.POINTER(u64), C.POINTER(u64)]
    L.pqps_partition.restype = None
    L.pqps_synth_user_tables.argtypes = [u64, vp, vp]
    L.pqps_synth_user_tables.restype = None
    L.pqps_synth_generate.argtypes = [vp, u64, u64, u64, vp, vp, C.POINTER(SynthCols), vp]
    L.pqps_synth_generate_host.argtypes = [u64, u64, u64, vp, vp, C.POINTER(SynthCols)]
    L.pqps_synth_generate_host.restype = None
    L.pqps_bump_codes.argtypes = [vp, vp, u32, u64, u32, vp]
    L.pqps_compact_rows.argtypes = [vp, C.POINTER(Column), u32, u64, vp, C.POINTER(u64), vp]
    L.pqps_project_column.argtypes = [vp, C.POINTER(Column), vp, vp, u64, u32, vp, vp]
    L.pqps_gather_keys.argtypes = [vp, C.POINTER(Column), C.c_int, vp, vp, u64, u32, vp, vp]
    L.pqps_merge_index_slots.argtypes = [vp, vp, vp, u32, u64, vp, u64, vp, vp]
    L.pqps_merge_slots.argtypes = [vp, vp, u32, u64, vp, u64, vp, vp]
    L.pqps_qstream_create.argtypes = [vp, u32, C.POINTER(vp)]
    L.pqps_qstream_scan.argtypes = [vp, C.POINTER(Column), u32, u64, u32, C.POINTER(Predicate), vp, u64, vp, vp]
    L.pqps_qstream_count.argtypes = [vp, C.POINTER(Column), u32, u64, C.POINTER(Predicate), vp, vp]
    L.pqps_qstream_sync.argtypes = [vp]
    L.pqps_qstream_hint_answer.argtypes = [vp, C.c_uint64, C.c_uint64]
    L.pqps_qstream_hint_answer.restype = None
    L.pqps_qstream_scan_slot.argtypes = [vp, u32, C.POINTER(Column), u32, u64, u32, C.POINTER(Predicate), vp, u64, vp, vp]
    L.pqps_qstream_count_slot.argtypes = [vp, u32, C.POINTER(Column), u32, u64, C.POINTER(Predicate), vp, vp]
    L.pqps_qstream_wait.argtypes = [vp, u32]
    L.pqps_qstream_lane.argtypes = [vp, u32, u64, vp, C.POINTER(vp), C.POINTER(vp)]
    L.pqps_qstream_mark.argtypes = [vp, u32]
    L.pqps_qstream_set_timing.argtypes = [vp, C.c_int]
    L.pqps_qstream_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.pqps_qstream_wait_ns.argtypes = [vp, C.c_int]
    L.pqps_qstream_wait_ns.restype = u64
    L.pqps_exchange_wait_ns.argtypes = [vp, C.c_int]
    L.pqps_exchange_wait_ns.restype = u64
    L.pqps_qstream_destroy.argtypes = [vp]
    L.pqps_exchange_unique_id.argtypes = [C.c_char_p, vp]
    L.pqps_exchange_create.argtypes = [vp, C.c_char_p, vp, u32, u32, u64, u32, C.POINTER(vp)]
    L.pqps_exchange_prepare.argtypes = [vp, C.c_char_p, u32, u32, u64, u32, C.POINTER(vp)]
    L.pqps_exchange_connect.argtypes = [vp, vp]
    L.pqps_exchange_select.argtypes = [vp, C.POINTER(Column), u32, u64, u32, C.POINTER(Predicate), u32, vp]
    L.pqps_exchange_count.argtypes = [vp, C.POINTER(Column), u32, u64, C.POINTER(Predicate), u32, vp]
    L.pqps_exchange_result.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    L.pqps_exchange_sync.argtypes = [vp]
    L.pqps_exchange_wire_bytes.argtypes = [vp, C.POINTER(u64), C.c_int]
    L.pqps_wire_bytes.argtypes = [u64, u64]
    L.pqps_wire_bytes.restype = u64
    L.pqps_wire_pays.argtypes = [u64, u64]
    L.pqps_wire_pack.argtypes = [vp, vp, u64, u64, u32, C.c_int, vp, vp, vp]
    L.pqps_wire_expand.argtypes = [vp, vp, u64, u32, vp, vp]
    L.pqps_exchange_wire_bytes.restype = None
    L.pqps_exchange_eager.argtypes = [vp, C.POINTER(u64), C.c_int]
    L.pqps_exchange_eager.restype = None
    L.pqps_exchange_destroy.argtypes = [vp]
    L.hipCompileWhere.argtypes = [C.POINTER(Schema), W, C.POINTER(Predicate), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
    L.hipCompileWherePlan.argtypes = [C.POINTER(Schema), W, C.POINTER(Plan), C.c_char_p, C.c_size_t]
    L.hipPlanFree.argtypes = [C.POINTER(Plan)]
    L.hipPlanFree.restype = None
    L.hipColumnId.argtypes = [C.c_char_p]
    for name in ("hipDumpTokens", "hipDumpParse"):
        f = getattr(L, name)
        f.restype = C.c_longlong
        f.argtypes = [C.c_char_p, C.c_char_p, C.c_longlong]
    # engine API (include/executeEngine-hip.h)
    E = C.POINTER(EngineS)
    L.initializeEngineHIP.restype = E
    L.initializeEngineHIP.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_char_p, C.c_char_p]
    L.initializeEngineSyntheticHIP.restype = E
    L.initializeEngineSyntheticHIP.argtypes = [C.c_ulonglong, C.c_ulonglong, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_char_p]
    L.initializeEngineColumnsHIP.restype = E
    L.initializeEngineColumnsHIP.argtypes = [C.c_ulonglong, C.POINTER(ColumnData), C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_char_p]
    L.hipSyntheticDictionary.restype = C.POINTER(C.c_char_p)
    L.hipSyntheticDictionary.argtypes = [C.c_int, C.POINTER(C.c_int)]
    for name in ("executeQuerySelectAsyncHIP", "executeQueryCountAsyncHIP"):
        f = getattr(L, name)
        f.restype = vp
        f.argtypes = [E, W]
    L.awaitQueryHIP.restype = C.c_longlong
    L.awaitQueryHIP.argtypes = [vp, C.POINTER(DeviceResult)]
    L.releaseQueryHIP.argtypes = [vp]
    L.releaseQueryHIP.restype = None
    L.hipEngineLanes.argtypes = [E]
    L.initializeEngineSyntheticRankHIP.restype = E
    L.initializeEngineSyntheticRankHIP.argtypes = [C.c_ulonglong, C.c_ulonglong, C.c_int, C.c_int, C.c_char_p]
    L.hipEngineRcclIdHIP.argtypes = [C.c_char_p, vp]
    L.hipEngineJoinRanksHIP.argtypes = [E, C.c_char_p, vp]
    L.hipEngineJoinPrepareHIP.argtypes = [E, C.c_char_p]
    L.hipEngineJoinConnectHIP.argtypes = [E, vp]
    L.hipEngineLeaveRanksHIP.argtypes = [E]
    L.hipEngineWireBytesHIP.argtypes = [E, C.POINTER(C.c_ulonglong), C.c_int]
    L.hipEngineEagerQueriesHIP.argtypes = [E, C.POINTER(C.c_ulonglong), C.c_int]
    L.hipQueryChecksumHIP.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.pqps_ids_checksum.argtypes = [vp, vp, u64, C.POINTER(u64), vp]
    L.pqps_qstream_reserve.argtypes = [vp, u64]
    L.pqps_qstream_test_fail_slot.argtypes = [vp, u32]
    L.hipEngineProbeBoolIndexes.argtypes = [E, C.c_int]
    L.hipEngineKernelTiming.argtypes = [E, C.c_int]
    L.hipEngineKernelTime.argtypes = [E, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.pqps_malloc_mapped.argtypes = [vp, C.c_size_t, C.POINTER(vp), C.POINTER(vp)]
    L.pqps_free_mapped.argtypes = [vp, vp]
    L.destroyEngineHIP.argtypes = [E]
    L.destroyEngineHIP.restype = None
    L.executeQuerySelectHIP.restype = C.POINTER(ResultSet)
    L.executeQuerySelectHIP.argtypes = [E, C.POINTER(C.c_char_p), C.c_int, C.c_char_p, W]
    L.executeQuerySelectIdsHIP.restype = C.c_longlong
    L.executeQuerySelectIdsHIP.argtypes = [E, W, C.POINTER(C.POINTER(C.c_uint)), C.POINTER(C.c_double)]
    L.hipEngineShards.restype = C.c_int
    L.hipEngineShards.argtypes = [E, C.POINTER(C.c_ulonglong), C.c_int]
    L.executeQueryCountHIP.restype = C.c_longlong
    L.executeQueryCountHIP.argtypes = [E, W]
    CR = C.POINTER(ColumnarResult)
    L.executeQuerySelectColumnarHIP.restype = CR
    L.executeQuerySelectColumnarHIP.argtypes = [E, C.POINTER(C.c_char_p), C.c_int, W]
    L.freeColumnarResultHIP.argtypes = [CR]
    L.hipColumnarCellText.restype = vp
    L.hipColumnarCellText.argtypes = [CR, C.c_int, C.c_int]
    L.hipColumnarHead.restype = C.POINTER(ResultSet)
    L.hipColumnarHead.argtypes = [CR, C.c_int]
    L.freeResultSetHead.argtypes = [C.POINTER(ResultSet), C.c_int]
    L.executeQueryDeleteHIP.restype = C.POINTER(ResultSet)
    L.executeQueryDeleteHIP.argtypes = [E, C.c_char_p, W]
    L.executeQueryInsertHIP.restype = C.c_bool
    L.executeQueryInsertHIP.argtypes = [E, C.c_char_p, C.POINTER(Record)]
    L.addAttributeIndexHIP.restype = C.c_bool
    L.addAttributeIndexHIP.argtypes = [E, C.c_char_p, C.c_char_p, C.c_int]
    L.freeResultSet.argtypes = [C.POINTER(ResultSet)]
    L.freeResultSet.restype = None
    L.evaluateWhereClause.restype = C.c_bool
    L.evaluateWhereClause.argtypes = [C.POINTER(Record), W]
    L.linearSearchRecords.restype = C.POINTER(C.POINTER(Record))
    L.linearSearchRecords.argtypes = [C.POINTER(C.POINTER(Record)), C.c_int, W, C.POINTER(C.c_int)]
    L.isAttributeIndexed.argtypes = [E, C.c_char_p]
    L.printTable.argtypes = [vp, C.POINTER(ResultSet), C.c_int]
    L.printTable.restype = None
    L.run_test_query.argtypes = [E, C.c_char_p, C.c_int]
    L.run_test_query.restype = None
    L.free.argtypes = [vp]
    L.free.restype = None
    _lib = L
    return L


def check(rc, what="pqps call"):
    if rc != 0:
        raise PqpsError(f"{what} failed ({rc}): {lib().pqps_last_error().decode()}")


# ---- predicate compile --------------------------------------------------------------------
class SchemaSpec:
    """Python-side description of which columns exist (width, dictionary)."""

    def __init__(self):
        self.schema = Schema()
        self._keep = []
        for i, kind in enumerate(COLUMN_KIND):
            self.schema.col[i].kind = kind

    def set_numeric(self, name, width):
        c = self.schema.col[COL[name]]
        c.present, c.width = 1, width
        return self

    def set_dict(self, name, width, values):
        """values: list of bytes, ascending in strcmp (= bytes) order."""
        arr = (C.c_char_p * max(1, len(values)))()
        for i, v in enumerate(values):
            arr[i] = v
        self._keep.append(arr)
        c = self.schema.col[COL[name]]
        c.present, c.width, c.dict_count, c.dict = 1, width, len(values), arr
        return self


def compile_plan(spec: SchemaSpec, chain):
    """-> [(Predicate, [column id per slot]), ...]: the passes of hipCompileWherePlan.  A column id >= MAX_COLUMNS
    names the flags of pass (id - MAX_COLUMNS); the last pass is the query's result."""
    wl = WhereList(chain)
    plan = Plan()
    err = C.create_string_buffer(200)
    rc = lib().hipCompileWherePlan(C.byref(spec.schema), wl.ptr, C.byref(plan), err, 200)
    if rc != 0:
        raise PqpsError("hipCompileWherePlan: " + err.value.decode())
    out = []
    for k in range(plan.n_passes):
        pred = Predicate()
        C.memmove(C.byref(pred), C.byref(plan.passes[k].pred), C.sizeof(Predicate))
        out.append((pred, list(plan.passes[k].column_ids[:pred.n_columns])))
    lib().hipPlanFree(C.byref(plan))
    return out


def compile_where(spec: SchemaSpec, chain):
    """-> (Predicate, [column id per slot]).  Raises PqpsError when not expressible."""
    wl = WhereList(chain)
    pred = Predicate()
    ids = (C.c_int * MAX_COLUMNS)()
    err = C.create_string_buffer(200)
    rc = lib().hipCompileWhere(C.byref(spec.schema), wl.ptr, C.byref(pred), ids, err, 200)
    if rc != 0:
        raise PqpsError("hipCompileWhere: " + err.value.decode())
    return pred, list(ids[:pred.n_columns])


# ---- device objects ----------------------------------------------------------------------------
class Context:
    def __init__(self, device=0):
        self.h = C.c_void_p()
        check(lib().pqps_ctx_create(device, C.byref(self.h)), "pqps_ctx_create")

    def info(self):
        name = C.create_string_buffer(64)
        cus, hbm = C.c_int(), C.c_uint64()
        check(lib().pqps_device_info(self.h, name, C.byref(cus), C.byref(hbm)))
        return name.value.decode(), cus.value, hbm.value

    def malloc(self, nbytes):
        p = C.c_void_p()
        check(lib().pqps_malloc(self.h, nbytes, C.byref(p)), "pqps_malloc")
        return p.value

    def free(self, p):
        check(lib().pqps_free(self.h, p))

    def upload(self, dptr, buf, nbytes, stream=None):
        check(lib().pqps_upload(self.h, dptr, buf, nbytes, stream), "pqps_upload")

    def download(self, host, dptr, nbytes, stream=None):
        check(lib().pqps_download(self.h, host, dptr, nbytes, stream), "pqps_download")

    def memset(self, dptr, value, nbytes, stream=None):
        check(lib().pqps_memset(self.h, dptr, value, nbytes, stream))

    def sync(self, stream=None):
        check(lib().pqps_ctx_sync(self.h, stream), "pqps_ctx_sync")

    def set_timing(self, on=True):
        check(lib().pqps_ctx_set_timing(self.h, 1 if on else 0))

    def set_option(self, name, value):
        check(lib().pqps_ctx_set_option(self.h, name.encode(), int(value)), "pqps_ctx_set_option")

    def ids_checksum(self, ids_dev, count, stream=None):
        """(sum of ids, sum of ids[i] * (2 i + 1)) mod 2^64 of a device-resident list (pqps_ids_checksum)."""
        out = (C.c_uint64 * 2)()
        check(lib().pqps_ids_checksum(self.h, ids_dev, int(count), out, stream), "pqps_ids_checksum")
        return int(out[0]), int(out[1])

    def kernel_time(self):
        """-> (ms in the scan kernel, ms in the whole query, launches) summed since the last call (an ID query is one launch: both equal)."""
        ev, tot, k = C.c_double(), C.c_double(), C.c_int()
        check(lib().pqps_ctx_kernel_time(self.h, C.byref(ev), C.byref(tot), C.byref(k)))
        return ev.value, tot.value, k.value

    def close(self):
        if self.h:
            lib().pqps_ctx_destroy(self.h)
            self.h = C.c_void_p()


def column_array(pairs):
    """[(device_ptr, width), ...] -> ctypes array of pqps_column."""
    arr = (Column * max(1, len(pairs)))()
    for i, (p, w) in enumerate(pairs):
        arr[i].data, arr[i].width = p, w
    return arr


SYNTH_SHELLS = [b"bash", b"fish", b"sh", b"zsh"]
SYNTH_HOSTS = sorted([b"labpc-01", b"labpc-02", b"labpc-03", b"labpc-04", b"labpc-05", b"labpc-06",
                      b"labpc-07", b"labpc-08", b"labpc-09", b"labpc-10", b"vm-ubuntu-01", b"vm-ubuntu-02",
                      b"cs-lab-01", b"cs-lab-02", b"personal-laptop", b"remote-ssh-01"])
SYNTH_USERS_DICT = [b"student%d" % (1000 + i) for i in range(SYNTH_USERS)]
SYNTH_BASES = sorted(b"cmd%03d" % i for i in range(111))
# the single-valued string columns of the synthetic ENGINE table (initializeEngineSyntheticHIP)
SYNTH_CONSTANTS = {"raw_command": b"cmd", "timestamp": b"2025-01-01T00:00:00.000Z", "working_directory": b"/home/u"}
# (record column, SynthCols field, bytes per row)
SYNTH_LAYOUT = [("command_id", "command_id", 8), ("exit_code", "exit_code", 4), ("user_id", "user_id", 4),
                ("risk_level", "risk_level", 4), ("sudo_used", "sudo_used", 1), ("shell_type", "shell_code", 1),
                ("user_name", "user_code", 2), ("host_name", "host_code", 1), ("base_command", "base_code", 1)]


def synth_schema():
    s = SchemaSpec()
    for name, w in (("command_id", 8), ("exit_code", 4), ("user_id", 4), ("risk_level", 4), ("sudo_used", 1)):
        s.set_numeric(name, w)
    s.set_dict("shell_type", 1, SYNTH_SHELLS)
    s.set_dict("user_name", 2, SYNTH_USERS_DICT)
    s.set_dict("host_name", 1, SYNTH_HOSTS)
    s.set_dict("base_command", 1, SYNTH_BASES)
    return s


def synth_user_tables(seed):
    cdf = (C.c_uint32 * SYNTH_USERS)()
    shell = (C.c_uint8 * SYNTH_USERS)()
    lib().pqps_synth_user_tables(seed, cdf, shell)
    return cdf, shell


class SyntheticTable:
    """Rows [row0, row0+n) of the seeded commands_* table, resident on the device.

    alloc(nbytes) -> device pointer lets the caller own the memory (bench.py passes a
    torch allocator); default is pqps_malloc.  Only `columns` are materialised."""

    def __init__(self, ctx: Context, n_rows, seed=0x5EED, row0=0, columns=None, alloc=None, stream=None):
        self.ctx, self.n, self.seed, self.row0 = ctx, n_rows, seed, row0
        self.schema = synth_schema()
        wanted = set(columns) if columns is not None else {c for c, _, _ in SYNTH_LAYOUT}
        cap = (n_rows + TILE_ROWS - 1) // TILE_ROWS * TILE_ROWS + TILE_ROWS
        self._own = alloc is None
        alloc = alloc or ctx.malloc
        self.ptr, self.width = {}, {}
        sc = SynthCols()
        for name, field, w in SYNTH_LAYOUT:
            if name not in wanted:
                self.schema.schema.col[COL[name]].present = 0
                continue
            p = alloc(cap * w)
            self.ptr[name], self.width[name] = p, w
            setattr(sc, field, p)
        for name in COLUMNS:
            if name not in self.ptr:
                self.schema.schema.col[COL[name]].present = 0
        cdf, shell = synth_user_tables(seed)
        self._cdf_dev, self._shell_dev = ctx.malloc(4 * SYNTH_USERS), ctx.malloc(SYNTH_USERS)
        ctx.upload(self._cdf_dev, cdf, 4 * SYNTH_USERS)
        ctx.upload(self._shell_dev, shell, SYNTH_USERS)
        check(lib().pqps_synth_generate(ctx.h, seed, row0, n_rows, self._cdf_dev, self._shell_dev, C.byref(sc), stream),
              "pqps_synth_generate")
        ctx.sync(stream)

    def bind(self, chain):
        """-> (Predicate, pqps_column array, n_cols, algorithmic bytes per row)."""
        pred, ids = compile_where(self.schema, chain)
        cols = column_array([(self.ptr[COLUMNS[i]], self.width[COLUMNS[i]]) for i in ids])
        return pred, cols, len(ids), sum(self.width[COLUMNS[i]] for i in ids)

    def free(self):
        if self._own:
            for p in self.ptr.values():
                self.ctx.free(p)
        self.ctx.free(self._cdf_dev)
        self.ctx.free(self._shell_dev)
        self.ptr = {}


class HipEngine:
    """initializeEngineHIP / executeQuerySelectHIP / destroyEngineHIP from Python."""

    def __init__(self, csv_path, indexes=()):
        L = lib()
        names = (C.c_char_p * max(1, len(indexes)))(*[a.encode() for a, _ in indexes])
        types = (C.c_int * max(1, len(indexes)))(*[t for _, t in indexes])
        self.e = L.initializeEngineHIP(len(indexes), names, types, str(csv_path).encode(), b"commands")
        self.n = self.e.contents.num_records

    @classmethod
    def _index_args(cls, indexes):
        names = (C.c_char_p * max(1, len(indexes)))(*[a.encode() for a, _ in indexes])
        types = (C.c_int * max(1, len(indexes)))(*[t for _, t in indexes])
        return names, types

    @classmethod
    def synthetic(cls, n_rows, seed=0x5EED, indexes=()):
        """initializeEngineSyntheticHIP: the seeded synthetic table behind the engine API, no host rows."""
        self = cls.__new__(cls)
        names, types = cls._index_args(indexes)
        self.e = lib().initializeEngineSyntheticHIP(n_rows, seed, len(indexes), names, types, b"commands")
        if not self.e:
            raise PqpsError("initializeEngineSyntheticHIP failed")
        self.n = self.e.contents.num_records
        return self

    @classmethod
    def synthetic_rank(cls, rows_total, world, rank, seed=0x5EED):
        """initializeEngineSyntheticRankHIP: this process's rows of a table that `world` processes hold between them
        (not joined yet: join_ranks / join_prepare + join_connect)."""
        self = cls.__new__(cls)
        self.e = lib().initializeEngineSyntheticRankHIP(rows_total, seed, world, rank, b"commands")
        if not self.e:
            raise PqpsError("initializeEngineSyntheticRankHIP failed")
        self.n = self.e.contents.num_records
        return self

    @staticmethod
    def rccl_id(rccl_library):
        ident = C.create_string_buffer(128)
        if lib().hipEngineRcclIdHIP(str(rccl_library).encode(), ident) != 0:
            raise PqpsError("hipEngineRcclIdHIP failed: " + lib().pqps_last_error().decode())
        return ident.raw

    def join_prepare(self, rccl_library):
        if lib().hipEngineJoinPrepareHIP(self.e, str(rccl_library).encode()) != 0:
            raise PqpsError("hipEngineJoinPrepareHIP failed: " + lib().pqps_last_error().decode())

    def join_connect(self, ident):
        if lib().hipEngineJoinConnectHIP(self.e, C.create_string_buffer(ident, 128)) != 0:
            raise PqpsError("hipEngineJoinConnectHIP failed: " + lib().pqps_last_error().decode())

    def join_ranks(self, rccl_library, ident):
        self.join_prepare(rccl_library)
        self.join_connect(ident)

    def leave_ranks(self):
        lib().hipEngineLeaveRanksHIP(self.e)

    def wire_bytes(self, reset=False):
        out = (C.c_ulonglong * 2)()
        lib().hipEngineWireBytesHIP(self.e, out, 1 if reset else 0)
        return int(out[0]), int(out[1])

    def eager_queries(self, reset=False):
        """(SELECTs answered in the sizes all-gather alone, SELECTs finished, IDs a rank's block has room for) of this rank's exchange."""
        out = (C.c_ulonglong * 3)()
        lib().hipEngineEagerQueriesHIP(self.e, out, 1 if reset else 0)
        return int(out[0]), int(out[1]), int(out[2])

    def ticket_checksum(self, ticket):
        """(sum of the row numbers, sum of id[i] * (2 i + 1)) mod 2^64 of the ticket's answer, computed on the device."""
        out = (C.c_ulonglong * 2)()
        if lib().hipQueryChecksumHIP(ticket, out) != 0:
            raise PqpsError("hipQueryChecksumHIP failed")
        return int(out[0]), int(out[1])

    @classmethod
    def from_columns(cls, n_rows, columns, indexes=()):
        """initializeEngineColumnsHIP.  columns: {name: numpy array} for numeric columns,
        {name: (codes array or None, [bytes, ...] dictionary)} for string columns (all 12 of them)."""
        import numpy as np
        self = cls.__new__(cls)
        arr = (ColumnData * MAX_COLUMNS)()
        keep = []
        for i, name in enumerate(COLUMNS):
            v = columns[name]
            if COLUMN_KIND[i] == KIND_DICT:
                codes, values = v
                d = (C.c_char_p * max(1, len(values)))(*values)
                keep.append(d)
                arr[i].dictionary, arr[i].dictionary_count = d, len(values)
                if codes is not None:
                    codes = np.ascontiguousarray(codes)
                    keep.append(codes)
                    arr[i].values, arr[i].width = codes.ctypes.data, codes.dtype.itemsize
            else:
                a = np.ascontiguousarray(v)
                keep.append(a)
                arr[i].values, arr[i].width = a.ctypes.data, a.dtype.itemsize
        names, types = cls._index_args(indexes)
        self.e = lib().initializeEngineColumnsHIP(n_rows, arr, len(indexes), names, types, b"commands")
        if not self.e:
            raise PqpsError("initializeEngineColumnsHIP failed")
        self.n = self.e.contents.num_records
        return self

    def select_async(self, chain, count_only=False):
        """-> ticket (opaque).  The WHERE list only has to live until this call returns."""
        wl = WhereList(chain)
        f = lib().executeQueryCountAsyncHIP if count_only else lib().executeQuerySelectAsyncHIP
        return f(self.e, wl.ptr)

    def await_ticket(self, ticket):
        """-> (count, DeviceResult)."""
        res = DeviceResult()
        k = lib().awaitQueryHIP(ticket, C.byref(res))
        return k, res

    def release_ticket(self, ticket):
        lib().releaseQueryHIP(ticket)

    def select_ids(self, chain):
        wl = WhereList(chain)
        ids = C.POINTER(C.c_uint)()
        qt = C.c_double()
        k = lib().executeQuerySelectIdsHIP(self.e, wl.ptr, C.byref(ids), C.byref(qt))
        if k < 0:
            raise PqpsError("executeQuerySelectIdsHIP failed")
        out = list(ids[:k])
        lib().free(ids)
        return out

    def probe_bool_indexes(self, enable=True):
        """Index mode follows QPEOMP / QPEMPI (BOOL indexes probed too, omp:424-459) instead of QPESeq; -> previous setting."""
        return lib().hipEngineProbeBoolIndexes(self.e, 1 if enable else 0)

    def shards(self):
        """Rows held by each device shard (one entry unless PQPS_DEVICES names several devices)."""
        rows = (C.c_ulonglong * 16)()
        k = lib().hipEngineShards(self.e, rows, 16)
        return list(rows[:k])

    def count(self, chain):
        wl = WhereList(chain)
        return lib().executeQueryCountHIP(self.e, wl.ptr)

    def select(self, columns, chain):
        wl = WhereList(chain)
        items = (C.c_char_p * max(1, len(columns or [])))(*[c.encode() for c in (columns or [])])
        rs = lib().executeQuerySelectHIP(self.e, items if columns else None, len(columns or []), b"commands", wl.ptr)
        r = rs.contents
        names = [r.columnNames[j].decode() for j in range(r.numColumns)]
        rows = [[r.data[i][j].decode("latin-1") for j in range(r.numColumns)] for i in range(r.numRecords)]
        out = dict(numRecords=r.numRecords, numColumns=r.numColumns, columns=names, rows=rows,
                   success=bool(r.success), queryTime=r.queryTime)
        lib().freeResultSet(rs)
        return out

    def select_columnar(self, columns, chain, text=True):
        """executeQuerySelectColumnarHIP: typed per-column arrays gathered on the device; with text=True
        also every cell as the string hipColumnarCellText makes of it."""
        import numpy as np
        L = lib()
        wl = WhereList(chain)
        items = (C.c_char_p * max(1, len(columns or [])))(*[c.encode() for c in (columns or [])])
        res = L.executeQuerySelectColumnarHIP(self.e, items if columns else None, len(columns or []), wl.ptr)
        r = res.contents
        n, m = r.numRecords, r.numColumns
        dtypes = {0: np.uint64, 1: np.int32, 2: np.uint8, 3: np.uint32}
        out = dict(numRecords=n, numColumns=m, columns=[r.columnNames[j].decode() for j in range(m)],
                   kinds=[r.columnKinds[j] for j in range(m)], values=[], success=bool(r.success), handle=res)
        for j in range(m):
            k = r.columnKinds[j]
            if k < 0 or n == 0 or not r.values[j]:
                out["values"].append(None)
            else:
                dt = np.dtype(dtypes[k])
                out["values"].append(np.frombuffer(C.string_at(r.values[j], n * dt.itemsize), dtype=dt).copy())
        if text:
            rows = []
            for i in range(n):
                row = []
                for j in range(m):
                    p = L.hipColumnarCellText(res, i, j)
                    row.append(C.string_at(p).decode("latin-1"))
                    L.free(p)
                rows.append(row)
            out["rows"] = rows
        return out

    def free_columnar(self, out):
        lib().freeColumnarResultHIP(out.pop("handle"))

    def record(self, i):
        return self.e.contents.all_records[i].contents

    def close(self):
        if self.e:
            lib().destroyEngineHIP(self.e)
            self.e = None
