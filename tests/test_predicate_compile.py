"""engine/hip/hipPredicate.c (WHERE list -> window leaves + truth/jump table)
checked on the CPU: a numpy model of the kernel's arithmetic (kernel_model.py)
applied to the compiled predicate must select exactly the rows the oracle
selects in scan mode.  Covers every golden WHERE, random trees (incl. > 6
leaves => jump-table path) and literal edge cases."""
import json
import random

import numpy as np
import pytest

import kernel_model as km
import qpelib as q

pq = q.pq
SELECT = (json.loads((q.GOLDEN / "select_golden.json").read_text())
          + json.loads((q.GOLDEN / "select_random_golden.json").read_text())      # + seeded random WHERE trees, same reference
          + json.loads((q.GOLDEN / "select_wide_golden.json").read_text()))       # + lists of more than 32 comparisons (several passes)
_cache = {}


def setup(csv):
    if csv not in _cache:
        t = q.OracleTable(q.GOLDEN / csv, [])
        spec, arrays = km.columns_from_records(t.rows, t.n)
        _cache[csv] = (t, spec, arrays)
    return _cache[csv]


def model_ids(spec, arrays, chain):
    """The compiled plan through the numpy model: passes before the last leave one flag byte per row."""
    passes = pq.compile_plan(spec, chain)
    n = len(arrays["command_id"])
    flags = []
    for pred, ids in passes:
        cols = [flags[i - pq.MAX_COLUMNS] if i >= pq.MAX_COLUMNS else arrays[pq.COLUMNS[i]] for i in ids]
        assert len(cols) <= pq.MAX_COLUMNS and pred.n_leaves <= pq.MAX_LEAVES
        mask = km.evaluate(pred, cols)
        flags.append((np.ones(n, dtype=np.uint8) if mask else np.zeros(n, dtype=np.uint8)) if isinstance(mask, bool)
                     else mask.astype(np.uint8))
    if len(passes) == 1:
        assert pq.compile_where(spec, chain)[1] == ids          # one pass: the plan is hipCompileWhere's own result
    if isinstance(mask, bool):
        return list(range(n)) if mask else [], pred
    return list(np.nonzero(mask)[0]), pred


SCAN_CASES = [c for c in SELECT if c["indexes"] == "none"]


@pytest.mark.parametrize("case", SCAN_CASES, ids=[f"{c['csv'][:4]}-{c['name']}" for c in SCAN_CASES])
def test_compiled_predicate_selects_golden_rows(case):
    t, spec, arrays = setup(case["csv"])
    chain = q.chain_from_jsonable(case["where"])
    got, _ = model_ids(spec, arrays, chain)
    want, count, _ = t.select_ids(chain)
    assert got == want and len(got) == case["num_records"]


LEAVES = [
    ("risk_level", ["=", "!=", ">", "<", ">=", "<="], ["0", "1", "2", "3", "4", "5", "9", "abc", ""]),
    ("exit_code", ["=", "!=", ">", "<", ">=", "<="], ["0", "1", "2", "126", "127", "130", "200", "2147483647", "-1"]),
    ("user_id", ["=", "!=", ">", "<", ">=", "<="], ["1000", "1001", "1040", "1088", "999", "5000", "-2147483648"]),
    ("command_id", ["=", "!=", ">", "<", ">=", "<="], ["0", "1", "77", "1000", "1999", "2000", "18446744073709551615", "-1", "x"]),
    ("sudo_used", ["=", "!=", ">", "<="], ["TRUE", "FALSE", "true", "1", "0", "yes"]),
    ("shell_type", ["=", "!=", ">", "<", ">=", "<="], ["bash", "zsh", "fish", "sh", "", "c", "zzz", "bas"]),
    ("user_name", ["=", "!=", ">", "<", ">=", "<="], ["student1030", "student1000", "student1", "student9", "a", "zz"]),
    ("host_name", ["=", "!=", "<", ">="], ["labpc-01", "labpc-05", "vm-ubuntu-02", "m"]),
    ("base_command", ["=", "<=", ">"], ["ls", "cat", "sudo", "a"]),
    ("working_directory", ["=", "<", ">="], ["/tmp", "/home", "/"]),
    ("timestamp", ["<", ">=", "="], ["2026-01-01", "2025-12-31T23:59:59.999Z", "3"]),
    ("raw_command", ["=", ">", "<="], ["pwd", "ls -la", "sudo", "z"]),
    ("nonexistent", ["=", "!="], ["5"]),
    ("risk_level", ["~", "=="], ["3"]),          # unknown operators
]


def random_chain(rng, depth, max_items):
    n = rng.randint(1, max_items)
    out = []
    for i in range(n):
        if depth > 0 and rng.random() < 0.25:
            out.append(random_chain(rng, depth - 1, max_items))
        else:
            a, ops, vals = rng.choice(LEAVES)
            out.append((a, rng.choice(ops), rng.choice(vals)))
        if i + 1 < n:
            out.append(rng.choice(["AND", "OR", "AND", "OR", None, "XOR"]))
    return out


def count_leaves(chain):
    return sum(count_leaves(x) if isinstance(x, list) else 1 for x in chain[0::2])


@pytest.mark.parametrize("seed", range(12))
def test_random_where_trees(seed):
    rng = random.Random(1234 + seed)
    t, spec, arrays = setup("commands_2k.csv")
    big = 0
    for _ in range(60):
        chain = random_chain(rng, depth=3, max_items=5)
        got, pred = model_ids(spec, arrays, chain)               # any size: large trees become several passes
        big += pred.n_leaves > pq.TT_LEAVES
        want, _, _ = t.select_ids(chain)
        assert got == want, chain
    assert big >= 1          # the jump-table path was exercised


def test_edge_csv_random_trees():
    rng = random.Random(99)
    t, spec, arrays = setup("edge_cases.csv")
    for _ in range(300):
        chain = random_chain(rng, depth=2, max_items=4)
        got, _ = model_ids(spec, arrays, chain)
        want, _, _ = t.select_ids(chain)
        assert got == want, chain


def test_wide_lists_become_several_passes():
    t, spec, arrays = setup("commands_2k.csv")
    wide = [c for c in SCAN_CASES if c.get("leaves", 0) > pq.MAX_LEAVES]
    assert len(wide) >= 10
    several = 0
    for case in wide:
        passes = pq.compile_plan(spec, q.chain_from_jsonable(case["where"]))
        several += len(passes) > 1
        for pred, ids in passes[:-1]:
            assert pred.n_leaves <= pq.MAX_LEAVES and len(ids) <= pq.MAX_COLUMNS
    assert several >= 8          # the rest fold down to one pass (constant leaves, unreachable branches)
    # a clause of any size compiles: 600 conditions in one flat list
    chain = []
    for i in range(600):
        chain += [("command_id", "=", str(3 * i + 1)), "OR"]
    chain += [("risk_level", ">", "9")]
    got, _ = model_ids(spec, arrays, chain)
    assert got == t.select_ids(chain)[0] and len(got) > 100


def test_constant_predicates_fold():
    _, spec, _ = setup("commands_2k.csv")
    for chain, truth in [
        ([], 1), ([("nonexistent", "=", "5")], 0), ([("shell_type", "=", "nosuchshell")], 0),
        ([("shell_type", "!=", "nosuchshell")], 1), ([("sudo_used", ">", "FALSE")], 0),
        ([("command_id", ">", "18446744073709551615")], 0), ([("risk_level", "<", "-2147483648")], 0),
        ([("nonexistent", "=", "1"), "OR", ("shell_type", "!=", "q")], 1),
    ]:
        pred, ids = pq.compile_where(spec, chain)
        assert pred.n_leaves == 0 and pred.n_columns == 0 and pred.truth == truth, chain


def test_unreachable_leaves_are_dropped():
    _, spec, _ = setup("commands_2k.csv")
    # (false AND risk) OR exit  ->  only exit_code is ever read
    pred, ids = pq.compile_where(spec, [[("nonexistent", "=", "1"), "AND", ("risk_level", "=", "5")], "OR", ("exit_code", "=", "0")])
    assert [pq.COLUMNS[i] for i in ids] == ["exit_code"] and pred.n_leaves == 1


def test_absent_column_is_an_error():
    spec = pq.synth_schema()
    with pytest.raises(pq.PqpsError):
        pq.compile_where(spec, [("timestamp", ">", "2026")])
    # ... but not when the leaf can never be evaluated
    pred, _ = pq.compile_where(spec, [("nonexistent", "=", "1"), "AND", ("timestamp", ">", "2026")])
    assert pred.n_leaves == 0 and pred.truth == 0


def test_flag_attributes_from_callers_are_unknown_attributes():
    """'\\x01<n>' names the flag buffer of pass n INSIDE the planner only (hipPredicate.c:node_column).  A list that comes in
    through the engine API with such a name must compile like any unknown attribute -- constant false (serial:278-281) --
    and never become an index into flag buffers that do not exist."""
    spec = pq.synth_schema()
    for name in ("\x010", "\x011", "\x017", "\x01-1"):
        pred, ids = pq.compile_where(spec, [(name, "=", "1")])
        assert pred.n_leaves == 0 and pred.truth == 0 and ids == []
        plan = pq.compile_plan(spec, [("risk_level", ">", "3"), "OR", (name, "=", "1", 0x7F1A6)])     # even with the planner's mark
        assert len(plan) == 1 and plan[0][0].n_leaves == 1 and all(i < pq.MAX_COLUMNS for i in plan[0][1])


def test_single_valued_dictionary_column_is_folded():
    """A string column with ONE value needs no device buffer: a comparison on it is decided when the WHERE is compiled."""
    spec = pq.SchemaSpec().set_numeric("risk_level", 4).set_dict("raw_command", 0, [b"cmd"])
    cases = [("=", "cmd", True), ("!=", "cmd", False), ("<", "cmd", False), ("<=", "cmd", True), (">", "cm", True), (">=", "cmd", True),
             ("=", "other", False), ("!=", "other", True), ("<", "z", True), (">", "z", False)]
    for op, lit, truth in cases:
        pred, ids = pq.compile_where(spec, [("raw_command", op, lit)])
        assert pred.n_leaves == 0 and pred.truth == (1 if truth else 0), (op, lit)
        pred, ids = pq.compile_where(spec, [("raw_command", op, lit), "AND", ("risk_level", "=", "5")])
        assert (pred.n_leaves, len(ids)) == ((1, 1) if truth else (0, 0)), (op, lit)
