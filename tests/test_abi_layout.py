"""The drop-in boundary without a GPU: struct layouts equal the reference's
(sizes measured from /root/reference/include with gcc),
and libpqps_hip.so exports every function include/*.h declares."""
import ctypes as C
import pathlib
import re
import subprocess

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
    assert {"pqps_filter_scan", "pqps_filter_gather", "pqps_filter_count", "pqps_filter_flags", "pqps_index_build",
            "pqps_index_probe", "executeQuerySelectHIP", "initializeEngineHIP", "destroyEngineHIP",
            "linearSearchRecords", "evaluateWhereClause", "tokenize", "parse_tokens", "run_test_query",
            "printTable", "hipCompileWhere"} <= declared
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
