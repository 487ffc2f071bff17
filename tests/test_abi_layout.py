"""The drop-in boundary without a GPU: struct layouts equal the reference's
(sizes measured from /root/reference/include with gcc),
and libpqps_hip.so exports every function include/*.h declares."""
import ctypes as C
import pathlib
import re
import subprocess

import pytest

import qpelib as q

PROBE = r"""
#include <stdio.h>
#include <stddef.h>
#include "executeEngine-hip.h"
#include "buildEngine-hip.h"
#include "connectEngine.h"
#include "hipPredicate.h"
#include "printHelper.h"
#include "sql.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(record), sizeof(struct engineS),
           sizeof(struct resultSetS), sizeof(struct whereClauseS), sizeof(KEY_T), sizeof(node),
           sizeof(ParsedSQL), sizeof(Token), sizeof(Condition), sizeof(FieldInfo));
    printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", offsetof(record, command_id),
           offsetof(record, raw_command), offsetof(record, base_command), offsetof(record, shell_type),
           offsetof(record, exit_code), offsetof(record, timestamp), offsetof(record, sudo_used),
           offsetof(record, working_directory), offsetof(record, user_id), offsetof(record, user_name),
           offsetof(record, host_name), offsetof(record, risk_level));
    printf("%zu %zu %zu %zu\n", offsetof(ParsedSQL, logic_ops), offsetof(ParsedSQL, num_conditions),
           sizeof(pqps_predicate), sizeof(pqps_leaf));
    return 0;
}
"""


def test_struct_layouts_match_reference(tmp_path):
    src = tmp_path / "probe.c"
    src.write_text(PROBE)
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c11", f"-I{q.ROOT / 'include'}", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    assert out[0].split() == "1040 72 48 56 16 40 6336 260 336 24".split()
    assert out[1].split() == "0 8 520 620 640 644 674 675 876 880 930 1032".split()
    logic_off, ncond_off, pred_size, leaf_size = map(int, out[2].split())
    assert ncond_off == logic_off + 16          # logic_ops[4] aliases num_conditions (parser quirk)
    assert pred_size == C.sizeof(q.pq.Predicate) and leaf_size == C.sizeof(q.pq.Leaf)


REFERENCE_INCLUDE = pathlib.Path("/root/reference/include")
# what only this repository has: the HIP engine's own headers (everything else of include/ restates a reference header)
HIP_ONLY_HEADERS = ["executeEngine-hip.h", "buildEngine-hip.h", "hipPredicate.h", "pqps_hip.h"]

SHARED_PROBE = r"""
#include <stdio.h>
#include <stddef.h>
#include "executeEngine-serial.h"
#include "connectEngine.h"
#include "printHelper.h"
#include "bplus.h"
#include "sql.h"
#define F(T, f) printf(#T "." #f " %zu %zu\n", offsetof(T, f), sizeof(((T *)0)->f))
int main(void) {
    printf("sizes %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(record), sizeof(struct engineS), sizeof(struct resultSetS),
           sizeof(struct whereClauseS), sizeof(KEY_T), sizeof(node), sizeof(ParsedSQL), sizeof(Token), sizeof(Condition), sizeof(FieldInfo));
    F(record, command_id); F(record, raw_command); F(record, base_command); F(record, shell_type); F(record, exit_code);
    F(record, timestamp); F(record, sudo_used); F(record, working_directory); F(record, user_id); F(record, user_name);
    F(record, host_name); F(record, risk_level);
    F(struct engineS, tableName); F(struct engineS, bplus_tree_roots); F(struct engineS, num_indexes); F(struct engineS, indexed_attributes);
    F(struct engineS, attribute_types); F(struct engineS, all_records); F(struct engineS, num_records); F(struct engineS, datafile);
    F(struct engineS, record_block);
    F(struct resultSetS, numRecords); F(struct resultSetS, numColumns); F(struct resultSetS, columnNames); F(struct resultSetS, columnTypes);
    F(struct resultSetS, data); F(struct resultSetS, queryTime); F(struct resultSetS, success);
    F(struct whereClauseS, attribute); F(struct whereClauseS, operator); F(struct whereClauseS, value); F(struct whereClauseS, value_type);
    F(struct whereClauseS, next); F(struct whereClauseS, logical_op); F(struct whereClauseS, sub);
    F(ParsedSQL, command); F(ParsedSQL, table); F(ParsedSQL, columns); F(ParsedSQL, num_columns); F(ParsedSQL, conditions);
    F(ParsedSQL, logic_ops); F(ParsedSQL, num_conditions);
    F(Condition, column); F(Condition, op); F(Condition, value); F(Condition, is_numeric);
    F(Token, type); F(Token, value);
    printf("enums %d %d %d %d %d %d\n", (int)FIELD_UINT64, (int)FIELD_INT, (int)FIELD_STRING, (int)FIELD_BOOL, MAX_TOKENS, ROW_LIMIT);
    return 0;
}
"""


def _run_probe(tmp_path, name, include_dirs):
    src = tmp_path / f"{name}.c"
    src.write_text(SHARED_PROBE)
    exe = tmp_path / name
    subprocess.run(["gcc", "-std=c11", *[f"-I{d}" for d in include_dirs], str(src), "-o", str(exe)], check=True)
    return subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout


@pytest.mark.skipif(not REFERENCE_INCLUDE.exists(), reason="the reference tree exists in the authoring container only")
def test_struct_layouts_equal_the_reference_headers_themselves(tmp_path):
    """The same probe compiled against /root/reference/include and against include/: every size, every field offset
    and width of the contract structs, the enum values and the two macros the bridge uses."""
    ours = _run_probe(tmp_path, "ours", [q.ROOT / "include"])
    theirs = _run_probe(tmp_path, "theirs", [REFERENCE_INCLUDE])
    assert ours == theirs
    assert ours.splitlines()[0].split()[1:] == "1040 72 48 56 16 40 6336 260 336 24".split()


@pytest.mark.skipif(not REFERENCE_INCLUDE.exists(), reason="the reference tree exists in the authoring container only")
def test_engine_sources_compile_against_the_reference_headers(tmp_path):
    """engine/hip/*.c drop into the reference's tree -- they compile against the REFERENCE's own
    headers (executeEngine-serial.h, bplus.h, recordSchema.h, logType.h, sql.h ...) plus the four headers only this
    repository has.  -Werror: a prototype that drifted from the reference's would be a conflicting declaration."""
    hip_only = tmp_path / "hip_only"
    hip_only.mkdir()
    for h in HIP_ONLY_HEADERS:
        (hip_only / h).write_text((q.ROOT / "include" / h).read_text())
    sources = sorted((q.PKG / "engine" / "hip").glob("*.c"))
    assert len(sources) >= 3
    for src in sources:
        p = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Werror", "-Wno-stringop-truncation", f"-I{REFERENCE_INCLUDE}", f"-I{hip_only}", str(src)],
                           capture_output=True, text=True)
        assert p.returncode == 0, (src.name, p.stderr[-2000:])


def _declared_functions(header):
    text = re.sub(r"/\*.*?\*/", "", pathlib.Path(header).read_text(), flags=re.S)
    text = re.sub(r"static inline[^{]*\{.*?\n\}", "", text, flags=re.S)
    names = set()
    for m in re.finditer(r"^[A-Za-z_][\w \*]*?\b([A-Za-z_]\w*)\s*\([^;{]*\)\s*;", text, flags=re.M | re.S):
        if "typedef" in m.group(0):
            continue
        names.add(m.group(1))
    return names


def test_library_exports_every_declared_symbol():
    lib = q.pq.lib()
    headers = ["pqps_hip.h", "executeEngine-hip.h", "buildEngine-hip.h", "hipPredicate.h", "connectEngine.h",
               "printHelper.h", "sql.h", "recordSchema.h", "executeEngine-serial.h"]
    declared = set()
    for h in headers:
        declared |= _declared_functions(q.ROOT / "include" / h)
    # the lab bench is a library of its own beside the product (round 4): libpqps_bench.so exports engineBench.h, the product does not
    bench_declared = _declared_functions(q.ROOT / "include" / "engineBench.h")
    assert bench_declared == {"hipEngineBench"}
    assert hasattr(q.pq.bench_lib(), "hipEngineBench") and not hasattr(lib, "hipEngineBench")
    assert {"pqps_filter_scan", "pqps_filter_gather", "pqps_filter_count", "pqps_filter_flags", "pqps_index_build",
            "pqps_index_probe", "executeQuerySelectHIP", "initializeEngineHIP", "destroyEngineHIP",
            "linearSearchRecords", "evaluateWhereClause", "tokenize", "parse_tokens", "run_test_query",
            "printTable", "hipCompileWhere", "initializeEngineSyntheticHIP", "initializeEngineColumnsHIP",
            "executeQuerySelectAsyncHIP", "executeQueryCountAsyncHIP", "awaitQueryHIP", "releaseQueryHIP",
            "pqps_qstream_scan_slot", "pqps_qstream_wait", "pqps_copy_peer", "pqps_last_kernel",
            # round 4: ranks behind the engine API, the compact wire form, bounded waits, per-launch status, checks
            "initializeEngineSyntheticRankHIP", "hipEngineJoinRanksHIP", "hipEngineJoinPrepareHIP", "hipEngineJoinConnectHIP",
            "hipEngineLeaveRanksHIP", "hipEngineRcclIdHIP", "hipEngineWireBytesHIP", "hipEngineEagerQueriesHIP", "hipEngineLanes", "hipQueryChecksumHIP",
            "pqps_wire_pack", "pqps_wire_expand", "pqps_wire_bytes", "pqps_wire_pays", "pqps_exchange_wire_bytes", "pqps_exchange_eager",
            "pqps_ids_checksum", "pqps_qstream_reserve", "pqps_qstream_test_fail_slot", "pqps_ctx_set_option"} <= declared
    assert all(hasattr(lib, n) for n in ("hipTableLaneCount", "hipTableLocksCreate", "hipTableLocksDestroy", "hipTableAcquireLane"))
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing


def test_partition_matches_mpi_formula():
    lib = q.pq.lib()
    orc = q.load_oracle()
    for n in (0, 5, 8, 1000003, 10**9):
        for world in (1, 2, 4, 8):
            for r in range(world):
                s1, c1, s2, c2 = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
                lib.pqps_partition(n, world, r, C.byref(s1), C.byref(c1))
                orc.orc_partition(n, world, r, C.byref(s2), C.byref(c2))
                assert (s1.value, c1.value) == (s2.value, c2.value)


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing in the package or the public headers may include,
    import, link or load it (and there is no CPU fallback to route through)."""
    import re
    import subprocess
    pat = re.compile(r"qpe_oracle|libqpe_oracle|oracle/|orc_[a-z_]+\(|_ref/|libqpeseq_ref")
    offenders = []
    for base in (q.PKG, q.ROOT / "include"):
        for path in base.rglob("*"):
            if path.is_file() and path.suffix in {".c", ".h", ".hpp", ".hip", ".py", ""} and "build" not in path.parts \
                    and path.name not in {"QPEHIP"} and path.suffix != ".so":
                try:
                    text = path.read_text(encoding="latin-1")
                except OSError:
                    continue
                if path.name == "Makefile" or path.suffix:
                    for i, line in enumerate(text.splitlines(), 1):
                        if pat.search(line) and "checker" not in line:
                            offenders.append(f"{path.relative_to(q.ROOT)}:{i}: {line.strip()[:100]}")
    assert not offenders, offenders
    lib = q.PKG / "libpqps_hip.so"
    if lib.exists():
        needed = subprocess.run(["readelf", "-d", str(lib)], capture_output=True, text=True).stdout
        assert "oracle" not in needed and "qpeseq" not in needed
