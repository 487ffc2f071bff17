#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the REAL reference.

Runs only in the authoring container: it needs oracle/_ref/libqpeseq_ref.so
(the reference compiled from /root/reference by oracle/Makefile) and, for the
one-off creation of commands_2k.csv, the reference's own data generator
(/root/reference/data-generation/generate_commands.py, run as a subprocess).
The generator is unseeded, so the CSV it wrote is COMMITTED and never
regenerated; everything else here is deterministic given the two CSVs.

Outputs (all data: inputs + what the reference answered):
  commands_2k.csv, edge_cases.csv          inputs
  select_golden.json     SELECT cases: where-list as parsed by the reference,
                         result row numbers, sha256 of every projected cell
  parse_golden.json      SQL text -> reference token stream / ParsedSQL / where list
  records_golden.json    CSV -> the 1040-byte records the reference loaded
  index_order_golden.json  leaf order of the reference's B+ trees
  print_golden.json      printTable() text for a few queries
Cases whose index candidates exceed num_records are never sent to the
reference (it overflows its buffer, executeEngine-serial.c:342,447); they are
recorded with "pinned": false and the oracle's answer.
"""
import base64
import hashlib
import json
import pathlib
import subprocess
import sys
import tempfile
import zlib

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
import qpelib as q  # noqa: E402

REF_GEN = pathlib.Path("/root/reference/data-generation/generate_commands.py")
HEADER = ("command_id,raw_command,base_command,shell_type,exit_code,timestamp,sudo_used,"
          "working_directory,user_id,user_name,host_name,risk_level")

INDEX_CONFIGS = {
    "none": [],
    "default": q.DEFAULT_INDEXES,                      # connectEngine.c:48-62
    "cmdid": [("command_id", 0)],                      # tests/serial-SELECT-test.c:149
    "risk": [("risk_level", 1)],                       # tests/duplicate-test.c:29-31
    "risk_twice": [("risk_level", 1), ("risk_level", 1)],
}

# (select list, where text or None).  Shapes: sample-queries.txt (S1..S8),
# tests/serial-SELECT-test.c:169-185 (T*), SURVEY App. A (A*), quirks (Q*).
SELECT_CASES = [
    ("S1", "command_id, base_command, sudo_used, user_name, timestamp", 'sudo_used = FALSE AND user_name = "student1030"'),
    ("S2", "command_id, raw_command, user_name, risk_level, timestamp", "sudo_used = TRUE AND risk_level > 2"),
    ("S3", "raw_command, exit_code, timestamp, sudo_used, user_name, risk_level", "risk_level > 3"),
    ("S4", "*", "risk_level = 5"),
    ("S7", "command_id, raw_command, risk_level, exit_code", 'sudo_used = TRUE OR (risk_level = 5 AND shell_type = "bash")'),
    ("S8", "user_name, working_directory, base_command", 'user_id = 1001 OR (user_name = "student1002" AND shell_type = "zsh")'),
    ("T_all", "*", None),
    ("T_proj", "command_id, user_name", None),
    ("T_risk", "*", "risk_level > 2"),
    ("T_shell", "command_id, shell_type", "shell_type = 'bash'"),
    ("T_and", "command_id, risk_level, sudo_used", "risk_level > 2 AND sudo_used = 1"),
    ("T_exit", "command_id, exit_code", "exit_code = 0"),
    ("A_dup", "command_id, risk_level, exit_code", "risk_level >= 4 AND exit_code = 0"),
    ("A_orloss1", "command_id, risk_level, user_name", 'risk_level = 5 OR user_name = "student1030"'),
    ("A_orloss2", "command_id, risk_level, user_name", 'user_name = "student1030" OR risk_level = 5'),
    ("A_neq", "command_id, risk_level", "risk_level != 3"),
    ("A_cid_lt", "command_id", "command_id < 10"),
    ("A_cid_ge", "command_id", "command_id >= 1990"),
    ("A_cid_eq", "command_id, raw_command", "command_id = 77"),
    ("A_cid_ne", "command_id", "command_id != 5"),
    ("A_cid_gt_le", "command_id", "command_id > 100 AND command_id <= 110"),
    ("A_uid", "command_id, user_id", "user_id = 1001"),
    ("A_uid_range", "command_id, user_id", "user_id >= 1010 AND user_id < 1020"),
    ("A_str_range", "command_id, user_name", 'user_name >= "student1050" AND user_name < "student1060"'),
    ("A_ts", "command_id, timestamp", 'timestamp > "2026-01-01"'),
    ("A_base_le", "command_id, base_command", 'base_command <= "cat"'),
    ("A_host_ne", "command_id, host_name", 'host_name != "labpc-01"'),
    ("A_wd", "command_id, working_directory", 'working_directory = "/tmp"'),
    ("A_raw", "command_id, raw_command", 'raw_command = "pwd"'),
    ("A_raw_gt", "command_id", 'raw_command > "sudo"'),
    ("A_shell_lt", "command_id, shell_type", 'shell_type < "fish"'),
    ("Q_uid_str", "command_id", 'user_id = "1001"'),
    ("Q_sudo_1", "command_id", "sudo_used = 1"),
    ("Q_sudo_true_lc", "command_id", "sudo_used = true"),
    ("Q_sudo_yes", "command_id", "sudo_used = 'yes'"),
    ("Q_sudo_gt", "command_id", "sudo_used > FALSE"),
    ("Q_sudo_ne", "command_id", "sudo_used != TRUE"),
    ("Q_cid_max", "command_id", "command_id = 18446744073709551615"),
    ("Q_cid_big", "command_id", "command_id <= 99999999999999999999"),
    ("Q_exit_gt", "command_id, exit_code", "exit_code > 100"),
    ("Q_exit_ne0", "command_id, exit_code", "exit_code != 0"),
    ("Q_unknown_attr", "command_id", "nonexistent = 5"),
    ("Q_unknown_or", "command_id", "nonexistent = 5 OR risk_level = 5"),
    ("Q_unknown_col", "bogus, command_id", "risk_level = 5"),
    ("Q_noop", "command_id", "risk_level 5"),                       # missing operator -> "="
    ("N_paren_all", "command_id", "(risk_level > 3 AND user_id = 1001)"),
    ("N_two_subs", "command_id", "(risk_level > 3) OR (user_id = 1001)"),
    ("N_mid", "command_id", "sudo_used = TRUE AND (risk_level = 4 OR risk_level = 5) AND exit_code != 0"),
    ("N_deep", "command_id", "((risk_level = 5))"),
    ("N_lead_sub", "command_id", '(risk_level = 5 OR risk_level = 4) AND shell_type = "zsh"'),
    ("N_sub_or_idx", "command_id", "(sudo_used = TRUE AND exit_code = 0) OR risk_level = 5"),
    ("R_and_or", "command_id", "risk_level = 1 AND exit_code != 0 OR sudo_used = TRUE"),
    ("R_or_and", "command_id", "sudo_used = TRUE OR risk_level = 1 AND exit_code != 0"),
    ("R_three_or", "command_id", "risk_level = 5 OR risk_level = 4 OR user_id = 1001"),
    ("R_four", "command_id", 'sudo_used = FALSE AND risk_level >= 2 AND exit_code = 0 AND shell_type = "zsh"'),
    ("W_order_by", "command_id", "risk_level = 5 ORDER BY command_id DESC"),
    ("W_none_match", "command_id", "risk_level > 7"),
    ("W_all_match", "command_id", "risk_level >= 1"),
]

# configs each case runs under on commands_2k.csv
CASE_CONFIGS = ["none", "default"]
EXTRA_CONFIGS = {"T_all": ["cmdid"], "T_risk": ["cmdid", "risk", "risk_twice"], "T_shell": ["cmdid"],
                 "T_and": ["cmdid"], "T_exit": ["cmdid"], "S3": ["risk", "risk_twice"], "A_neq": ["risk"]}

PARSE_CASES = [
    "SELECT * FROM Commands",
    "select command_id from commands where risk_level > 3",
    "SELECT a, b ,c FROM t WHERE x >= 10 AND y <= 'abc' OR z != \"q r\";",
    "# -- Sample 1:\nSELECT command_id, base_command, sudo_used, user_name, timestamp\nFROM Commands\nWHERE sudo_used = FALSE AND user_name = \"student1030\"",
    "SELECT * FROM c WHERE risk_level = 1 and exit_code = 0",          # lowercase and: not a keyword
    "SELECT * FROM c WHERE risk_level = 1 or exit_code = 0",           # lowercase or: keyword
    "SELECT * FROM c WHERE a = TRUE AND b = false AND c = True",
    "SELECT * FROM c WHERE a = -5",                                     # no negative literals
    "SELECT * FROM c WHERE a = 3.14",
    "SELECT * FROM c WHERE a = 'it''s'",
    "SELECT * FROM c WHERE a = 'unterminated",
    "SELECT * FROM c WHERE (a = 1 AND (b = 2 OR c = 3)) OR d = 4",
    "SELECT * FROM c WHERE a = 1 AND b = 2 AND c = 3 AND d = 4",
    "SELECT * FROM c WHERE ((a = 1) AND (b = 2)) AND ((c = 3))",
    "SELECT * FROM c WHERE a 5",
    "SELECT * FROM c WHERE a = 1 ORDER BY b DESC",
    "SELECT * FROM c WHERE a = 1 ORDER BY b ASC",
    "SELECT * FROM c ORDER BY b",
    "SELECT x FROM c WHERE a = 1 -- trailing comment\n AND b = 2",
    "INSERT INTO Commands VALUES (999999, \"echo 'test insert'\", \"echo\", \"bash\", 0, \"2025-12-01T12:00:00.000Z\", \"FALSE\", \"/home/test\", 1000, \"testuser\", \"test-host\", 1)",
    "DELETE FROM Commands WHERE command_id = 999999",
    "DESCRIBE Commands",
    "UPDATE Commands",
    "",
    "   ",
    "SELECT",
    "SELECT * FROM",
    "SELECT * FROM c WHERE",
    "SELECT *, a FROM c",
    "SELECT a b FROM c",
    "SELECT * FROM c WHERE a > = 5",
    "SELECT * FROM c WHERE a >= 5 AND b <= 6 AND c != 7 AND d < 8 AND e > 9",
    "SELECT * FROM c WHERE a=1 AND(b=2)",
    "SELECT * FROM c WHERE risk_level=5;SELECT 1",
    "SELECT * FROM c WHERE name = \"semi;colon\"",
    "SELECT * FROM c WHERE a = @ 5",
    "SELECT * FROM c WHERE _x9 = 5 AND y_ = 'z'",
]


def make_edge_csv(path):
    long_raw = "x" * 700 + "," + "y" * 600          # physical line > 1023 bytes -> split rows
    lines = [
        HEADER,
        '1,"echo ""hi"", there",echo,bash,0,2025-01-01T00:00:00.000Z,true,/home/a,1001,alice,host-1,3',
        "2,plain,ls,zsh,1,ts,FALSE,/tmp,1002,bob,host-2,1",
        "3,,,,,,,,,,,",
        "4,short",
        '5,"quoted"tail,cat,sh,-1,ts,1,/,  -5 ,carol,h,+2',
        "",
        "6,cmd,base,bash,0,ts,True,/x,1003,dave,host,2\r",
        "7,cmd,base,12345678901234567890,65,ts,TRUE,/x,1003,dave,host,2",      # shell_type fills all 20 bytes
        "8,cmd,base,bash,0,123456789012345678901234567890,0,/x,1004,erin,host,4",  # timestamp fills 30 bytes, sudo false
        "9,cmd,base,bash,0,123456789012345678901234567890,1,/x,1004,erin,host,4",  # ... sudo true (byte 0x01 follows)
        "18446744073709551615,max,base,fish,130,ts,false,/m,2147483647,u,h,5",
        "18446744073709551616,sat,base,fish,126,ts,false,/m,-2147483648,u,h,5",
        "-1,neg,base,fish,127,ts,false,/m,1005,u,h,5",
        "abc,nan,base,fish,2,ts,false,/m,12abc,u,h,x",
        "10,dup,base,bash,0,ts,false,/d,1001,alice,host-1,3",
        "10,dup,base,bash,0,ts,false,/d,1001,alice,host-1,3",
        "10,dup2,base,bash,0,ts,true,/d,1001,alice,host-1,1",
        f"11,{long_raw},base,bash,0,ts,false,/l,1006,long,host,2",
        '12,"multi, comma, field","b,c",zsh,0,ts,false,"/w,d",1007,"na,me","ho,st",1',
        "13,trailing,comma,bash,0,ts,false,/t,1008,tc,host,2,extra,fields",
        " 14, lead space, ls ,bash, 0,ts, true,/s,1009,sp,host, 3",
        "15,noeol,base,bash,0,ts,false,/n,1010,ne,host,1",           # file ends without newline
    ]
    data = "\n".join(lines)
    path.write_bytes(data.encode("latin-1"))


EDGE_SELECTS = [
    ("E_all", "*", None),
    ("E_risk3", "*", "risk_level = 3"),
    ("E_shell20", "command_id, shell_type, exit_code", 'shell_type = "12345678901234567890"'),
    ("E_shell20A", "command_id, shell_type", 'shell_type = "12345678901234567890A"'),
    ("E_ts30", "command_id, timestamp, sudo_used", 'timestamp = "123456789012345678901234567890"'),
    ("E_ts30_1", "command_id, timestamp, sudo_used", 'timestamp > "123456789012345678901234567890"'),
    ("E_cid_max", "command_id, raw_command", "command_id = 18446744073709551615"),
    ("E_cid0", "command_id, raw_command", "command_id = 0"),
    ("E_uid_neg", "command_id, user_id", "user_id < 0"),
    ("E_uid_max", "command_id, user_id", "user_id >= 2147483647"),
    ("E_dup", "command_id, raw_command", "command_id = 10"),
    ("E_empty_str", "command_id", 'raw_command = ""'),
    ("E_sudo", "command_id, sudo_used", "sudo_used = TRUE"),
    ("E_comma", "*", 'user_name = "na,me"'),
    ("E_quote", "command_id, raw_command", "command_id = 1"),
    ("E_lead", "command_id, raw_command, base_command, risk_level", 'base_command = " ls "'),
]


def sha_rows(rows):
    h = hashlib.sha256()
    for r in rows:
        for c in r:
            h.update(c.encode("latin-1"))
            h.update(b"\x1f")
        h.update(b"\x1e")
    return h.hexdigest()


def compose(select_list, where):
    sql = f"SELECT {select_list} FROM Commands"
    if where:
        sql += f" WHERE {where}"
    return sql


def parse_with_ref(ref, sql):
    text = q.call_text(ref.refh_parse, sql.encode("latin-1"))
    head, wd = text.split(q.RS, 1)
    return head, q.parse_where_dump(wd)


def normalize_driver_output(text):
    """Driver stdout without what legitimately differs between engines / runs: timings and the
    summary block."""
    import re
    text = text.split("\x1b[36m=======")[0]
    text = re.sub(r"Query Time: [0-9.]+ seconds", "Query Time: X seconds", text)
    text = re.sub(r"Execution Time: [0-9.]+", "Execution Time: X", text)
    return text


def main():
    ref = q.load_ref()
    if ref is None:
        sys.exit("oracle/_ref/libqpeseq_ref.so missing: run `make -C oracle` in the authoring container")
    q.build_oracle()

    csv2k = HERE / "commands_2k.csv"
    if not csv2k.exists():
        subprocess.run([sys.executable, str(REF_GEN), "2000", str(csv2k)], check=True)
    edge = HERE / "edge_cases.csv"
    make_edge_csv(edge)

    # ---- parse golden -------------------------------------------------
    parse_out = []
    all_sql = list(PARSE_CASES) + [compose(s, w) for _, s, w in SELECT_CASES]
    for sql in all_sql:
        toks = q.call_text(ref.refh_tokens, sql.encode("latin-1"))
        parsed = q.call_text(ref.refh_parse, sql.encode("latin-1"))
        parse_out.append({"sql": sql, "tokens": toks, "parse": parsed})
    (HERE / "parse_golden.json").write_text(json.dumps(parse_out, indent=0))

    # ---- records golden -----------------------------------------------
    rec_out = {}
    for path, limit in ((edge, None), (csv2k, 8)):
        eng = q.RefEngine(path, [])
        n = eng.n if limit is None else min(limit, eng.n)
        blob = b"".join(bytes(eng.record(i)) for i in range(n))
        rec_out[path.name] = {"num_records": eng.n, "dumped": n,
                              "zlib_b64": base64.b64encode(zlib.compress(blob, 9)).decode()}
        eng.close()
    (HERE / "records_golden.json").write_text(json.dumps(rec_out, indent=0))

    # ---- index order golden -------------------------------------------
    idx_attrs = [("command_id", 0), ("user_id", 1), ("risk_level", 1), ("exit_code", 1), ("sudo_used", 3),
                 ("shell_type", 2), ("user_name", 2)]
    idx_out = {}
    for path in (csv2k, edge):
        eng = q.RefEngine(path, idx_attrs)
        idx_out[path.name] = {a: eng.index_order(i) for i, (a, _) in enumerate(idx_attrs)}
        eng.close()
    (HERE / "index_order_golden.json").write_text(json.dumps(idx_out, separators=(",", ":")))

    # ---- select golden -------------------------------------------------
    sel_out = []

    def run_cases(path, cases, configs_for, ids_from_command_id):
        engines = {}
        oracles = {}
        for name, sel, where in cases:
            for cfg in configs_for(name):
                if cfg not in engines:
                    engines[cfg] = q.RefEngine(path, INDEX_CONFIGS[cfg])
                    oracles[cfg] = q.OracleTable(path, INDEX_CONFIGS[cfg])
                eng, orc = engines[cfg], oracles[cfg]
                sql = compose(sel, where)
                head, chain = parse_with_ref(ref, sql)
                o_ids, o_count, cand = orc.select_ids(chain)
                case = {"name": name, "csv": path.name, "indexes": cfg, "sql": sql,
                        "where": q.chain_to_jsonable(chain), "candidates": cand}
                cols = None if sel.strip() == "*" else [c.strip() for c in sel.split(",")]
                if cand > eng.n:
                    # would overflow the reference's candidate buffer: never run it there
                    case.update(pinned=False, num_records=o_count, ids=o_ids,
                                rows_sha256=sha_rows(orc.project(o_ids, cols)))
                    sel_out.append(case)
                    continue
                res = eng.select(sql)
                case.update(pinned=True, num_records=res["numRecords"], columns=res["columns"],
                            rows_sha256=sha_rows(res["rows"]), first_rows=res["rows"][:3])
                if ids_from_command_id:
                    r2 = eng.select(compose("command_id", where))
                    case["ids"] = [int(r[0]) for r in r2["rows"]]
                sel_out.append(case)
        for e in engines.values():
            e.close()

    run_cases(csv2k, SELECT_CASES, lambda n: CASE_CONFIGS + EXTRA_CONFIGS.get(n, []), True)
    run_cases(edge, EDGE_SELECTS, lambda n: ["none", "default"], False)
    (HERE / "select_golden.json").write_text(json.dumps(sel_out, separators=(",", ":")))

    # ---- printTable golden ----------------------------------------------
    prt = []
    eng = q.RefEngine(csv2k, q.DEFAULT_INDEXES)
    for name, limit in (("S1", 20), ("S4", 20), ("S7", 5), ("T_proj", 3), ("W_none_match", 20), ("A_cid_lt", 0)):
        sel, where = next((s, w) for n_, s, w in SELECT_CASES if n_ == name)
        sql = compose(sel, where)
        with tempfile.NamedTemporaryFile(suffix=".txt") as tf:
            rc = ref.refh_print(eng.h, sql.encode(), limit, tf.name.encode())
            assert rc == 0
            prt.append({"name": name, "sql": sql, "limit": limit, "indexes": "default",
                        "text": pathlib.Path(tf.name).read_text(encoding="latin-1")})
    eng.close()
    (HERE / "print_golden.json").write_text(json.dumps(prt, indent=0))

    driver_goldens(csv2k)
    fifty_k_goldens()

    unpinned = [c["name"] + "/" + c["indexes"] for c in sel_out if not c["pinned"]]
    print(f"select cases: {len(sel_out)} ({len(unpinned)} not sent to the reference: {unpinned})")
    print(f"parse cases: {len(parse_out)}")


def random_where_cases(csv2k, n_cases, seed):
    """Seeded random WHERE trees over all 12 columns: literals drawn from the table itself (so that
    predicates match something), every operator, AND / OR mixes without precedence, parenthesised
    sub-chains two levels deep, the 5-conditions-per-level quirk.  Only tokens the reference's parser
    handles deterministically: non-negative digit runs, double-quoted strings without quote / comment
    characters, TRUE / FALSE / 1 / 0."""
    import csv as csvmod
    import random
    import re
    rng = random.Random(seed)
    with open(csv2k, newline="", encoding="latin-1") as f:
        rows = list(csvmod.DictReader(f))
    numeric = ["command_id", "exit_code", "user_id", "risk_level"]
    strings = ["raw_command", "base_command", "shell_type", "timestamp", "working_directory", "user_name", "host_name"]
    safe = re.compile(r"^[A-Za-z0-9_./: @=+,~-]+$")
    pool = {c: sorted({r[c] for r in rows if safe.match(r[c]) and "--" not in r[c]}) for c in strings}
    pool.update({c: sorted({r[c] for r in rows}, key=int) for c in numeric})
    ops = ["=", "!=", "<", "<=", ">", ">="]

    def condition():
        kind = rng.random()
        if kind < 0.45:
            c = rng.choice(numeric)
            v = rng.choice(pool[c]) if rng.random() < 0.8 else str(rng.randrange(0, 3000))
            return f"{c} {rng.choice(ops)} {v}"
        if kind < 0.85:
            c = rng.choice(strings)
            v = rng.choice(pool[c])
            if rng.random() < 0.15:
                v = v[:max(1, len(v) // 2)]                           # a prefix: range semantics of strcmp
            return f'{c} {rng.choice(ops)} "{v}"'
        return f"sudo_used {rng.choice(['=', '=', '!=', '<', '>='])} {rng.choice(['TRUE', 'FALSE', '1', '0'])}"

    def chain(depth):
        k = rng.choice([1, 2, 2, 3, 3, 4, 5] if depth == 0 else [1, 2, 2, 3])
        parts = []
        for i in range(k):
            if depth < 2 and rng.random() < 0.25:
                parts.append("(" + chain(depth + 1) + ")")
            else:
                parts.append(condition())
            if i + 1 < k:
                parts.append(rng.choice(["AND", "OR"]))
        return " ".join(parts)

    all_cols = numeric + strings + ["sudo_used"]
    cases = []
    for i in range(n_cases):
        sel = "*" if rng.random() < 0.2 else ", ".join(rng.sample(all_cols, rng.randint(1, 5)))
        cases.append((f"R{i:03d}", sel, chain(0)))
    return cases


def random_goldens(csv2k, n_cases=220, seed=20260101):
    """select_random_golden.json: the random WHERE trees through the REAL reference (scan mode and the
    default five indexes).  Cases whose candidate list would overflow the reference's buffer
    (heap corruption there) are dropped, so every entry is pinned."""
    ref = q.load_ref()
    q.build_oracle()
    out, dropped = [], 0
    engines = {cfg: q.RefEngine(csv2k, INDEX_CONFIGS[cfg]) for cfg in ("none", "default")}
    oracles = {cfg: q.OracleTable(csv2k, INDEX_CONFIGS[cfg]) for cfg in ("none", "default")}
    for name, sel, where in random_where_cases(csv2k, n_cases, seed):
        sql = compose(sel, where)
        head, chain = parse_with_ref(ref, sql)
        for cfg in ("none", "default"):
            eng, orc = engines[cfg], oracles[cfg]
            o_ids, o_count, cand = orc.select_ids(chain)
            if cand > eng.n:
                dropped += 1
                continue
            res = eng.select(sql)
            r2 = eng.select(compose("command_id", where))
            out.append({"name": name, "csv": csv2k.name, "indexes": cfg, "sql": sql, "where": q.chain_to_jsonable(chain),
                        "candidates": cand, "pinned": True, "num_records": res["numRecords"], "columns": res["columns"],
                        "rows_sha256": sha_rows(res["rows"]),
                        "ids_zlib_b64": q.pack_ids([int(r[0]) for r in r2["rows"]])})
    for e in engines.values():
        e.close()
    (HERE / "select_random_golden.json").write_text(json.dumps(out, separators=(",", ":")))
    nonempty = sum(1 for c in out if c["num_records"] > 0)
    print(f"random select cases: {len(out)} pinned ({nonempty} with matches), {dropped} dropped (candidate overflow)")


def wide_where_cases(csv2k, seed=20260202):
    """WHERE lists with more than 32 comparisons (the device predicate's size: the HIP engine splits them into
    passes).  The reference's PARSER takes at most 5 conditions per level, its engine API any list -- these are
    built as whereClauseS lists, below the parser.  Literals from the table, like random_where_cases."""
    import csv as csvmod
    import random
    rng = random.Random(seed)
    with open(csv2k, newline="", encoding="latin-1") as f:
        rows = list(csvmod.DictReader(f))
    numeric = ["command_id", "exit_code", "user_id", "risk_level"]
    strings = ["raw_command", "base_command", "shell_type", "timestamp", "working_directory", "user_name", "host_name"]
    every = numeric + strings + ["sudo_used"]
    pool = {c: sorted({r[c] for r in rows}) for c in strings}
    pool.update({c: sorted({r[c] for r in rows}, key=int) for c in numeric})
    ops = ["=", "!=", "<", "<=", ">", ">="]

    def condition(columns=None):
        c = rng.choice(columns or every)
        if c in numeric:
            return (c, rng.choice(ops), rng.choice(pool[c]), 0)
        if c in strings:
            return (c, rng.choice(ops), rng.choice(pool[c]), 1)
        return ("sudo_used", rng.choice(["=", "!="]), rng.choice(["TRUE", "FALSE"]), 1)

    def join(parts, glue=None):
        out = []
        for i, part in enumerate(parts):
            out.append(part)
            if i + 1 < len(parts):
                out.append(glue or rng.choice(["AND", "OR"]))
        return out

    def flat(k, columns=None, glue=None):
        return join([condition(columns) for _ in range(k)], glue)

    def window():
        v = int(rng.choice(pool["command_id"]))
        return [("command_id", ">=", str(v), 0), "AND", ("command_id", "<=", str(v + rng.randrange(0, 40)), 0)]

    cases = []
    for i in range(3):       # a: one flat list of 40 / 47 / 70 conditions, no parentheses at all
        cases.append((f"Wa{i}", ["command_id", "risk_level"], flat([40, 47, 70][i])))
    for i in range(2):       # b: 5 x (5 x pair) = 50 leaves
        cases.append((f"Wb{i}", ["command_id"], join([join([flat(2) for _ in range(5)]) for _ in range(5)])))
    for i in range(2):       # c: one parenthesised element of 45 leaves (too large even alone) between plain conditions
        big = join([join([flat(3) for _ in range(3)]) for _ in range(5)])
        cases.append((f"Wc{i}", None, join([condition(), big, condition(), condition()])))
    for i in range(2):       # d: all 12 columns inside each of 5 elements (the 12-column budget of a pass), 60 leaves
        def twelve():
            cols = every[:]
            rng.shuffle(cols)
            return join([join([condition([c]) for c in cols[k:k + 3]]) for k in range(0, 12, 3)])
        cases.append((f"Wd{i}", ["command_id"], join([twelve() for _ in range(5)], "OR")))
    for i in range(2):       # e: OR of 36 narrow command_id windows (72 leaves on one column, few rows)
        cases.append((f"We{i}", ["command_id", "user_name"], join([window() for _ in range(36)], "OR")))
    # f: 4 levels deep, 5 x 5 x 2 x 2 = 100 leaves
    cases.append(("Wf0", ["command_id"], join([join([join([flat(2) for _ in range(2)]) for _ in range(5)]) for _ in range(5)])))
    # g: 33 leaves exactly -- a 31-leaf parenthesised element, then two conditions
    cases.append(("Wg0", ["command_id"], [flat(31), "AND", condition(), "OR", condition()]))
    # h: a long AND list that narrows to few rows, then an OR tail
    loose = join([(c, "!=", rng.choice(pool[c]), 0) for c in rng.choices(["exit_code", "user_id", "command_id"], k=34)], "AND")
    cases.append(("Wh0", None, loose + ["AND", ("risk_level", ">", "3", 0), "OR"] + window()))
    return cases


def count_chain_leaves(chain):
    return sum(count_chain_leaves(e) if isinstance(e, list) else 1 for e in chain[0::2])


def render_chain(chain):
    out = []
    for i, e in enumerate(chain):
        if i % 2 == 1:
            out.append(e)
        elif isinstance(e, list):
            out.append("(" + render_chain(e) + ")")
        else:
            out.append(f'{e[0]} {e[1]} "{e[2]}"' if e[3] else f"{e[0]} {e[1]} {e[2]}")
    return " ".join(out)


def wide_goldens(csv2k):
    """select_wide_golden.json: the > 32-comparison lists through the REAL reference's engine API
    (executeQuerySelectSerial on the whereClauseS list: oracle/ref_harness.c refh_select_where), scan mode and
    the default five indexes; same record layout as select_random_golden.json ("sql" is a rendering for the
    reader: the reference's parser would drop conditions past the fifth of a level)."""
    q.load_ref()
    q.build_oracle()
    out = []
    engines = {cfg: q.RefEngine(csv2k, INDEX_CONFIGS[cfg]) for cfg in ("none", "default")}
    oracles = {cfg: q.OracleTable(csv2k, INDEX_CONFIGS[cfg]) for cfg in ("none", "default")}
    for name, cols, chain in wide_where_cases(csv2k):
        leaves = count_chain_leaves(chain)
        assert leaves > 32, (name, leaves)
        for cfg in ("none", "default"):
            eng, orc = engines[cfg], oracles[cfg]
            o_ids, o_count, cand = orc.select_ids(chain)
            if cand > eng.n:
                print(f"{name}/{cfg}: candidate overflow in the reference, dropped")
                continue
            res = eng.select_where(cols, chain)
            r2 = eng.select_where(["command_id"], chain)
            ids = [int(r[0]) for r in r2["rows"]]
            assert ids == o_ids, (name, cfg)
            out.append({"name": name, "csv": csv2k.name, "indexes": cfg,
                        "sql": compose(", ".join(cols) if cols else "*", render_chain(chain)), "where": q.chain_to_jsonable(chain),
                        "leaves": leaves, "candidates": cand, "pinned": True, "num_records": res["numRecords"],
                        "columns": res["columns"], "rows_sha256": sha_rows(res["rows"]), "ids_zlib_b64": q.pack_ids(ids)})
            print(f"{name}/{cfg}: {leaves} leaves, {cand} candidates, {res['numRecords']} rows")
    for e in engines.values():
        e.close()
    (HERE / "select_wide_golden.json").write_text(json.dumps(out, separators=(",", ":")))


BOOLPROBE_INDEX_CONFIGS = {
    "default": q.DEFAULT_INDEXES,                      # connectEngine.c:48-62: sudo_used is a BOOL index
    "bool_only": [("sudo_used", 3)],
    "bool_twice": [("sudo_used", 3), ("risk_level", 1), ("sudo_used", 3)],
}


def boolprobe_cases():
    """WHERE lists for the OpenMP / MPI engines' row selection (omp:362-494): every operator on the BOOL index with
    every kind of literal, beside int / u64 probes, under OR, nested (no probe there), twice."""
    cases = [
        ("S1", [("sudo_used", "=", "FALSE"), "AND", ("user_name", "=", "student1030")]),
        ("S2", [("sudo_used", "=", "TRUE"), "AND", ("risk_level", ">", "2")]),
        ("S7", [("sudo_used", "=", "TRUE"), "OR", [("risk_level", "=", "5"), "AND", ("shell_type", "=", "bash")]]),
        ("or_loses_a_side", [("sudo_used", "=", "TRUE"), "OR", ("user_name", "=", "student1030")]),
        ("nested_is_not_probed", [("risk_level", ">", "3"), "AND", [("sudo_used", "=", "TRUE"), "OR", ("exit_code", "!=", "0")]]),
        ("nested_only", [[("sudo_used", "=", "TRUE"), "AND", ("risk_level", ">", "1")]]),
        ("twice", [("sudo_used", "=", "TRUE"), "AND", ("sudo_used", "!=", "FALSE")]),
        ("bool_then_int", [("sudo_used", "=", "TRUE"), "AND", ("exit_code", "=", "0")]),
        ("int_then_bool", [("risk_level", ">=", "4"), "AND", ("sudo_used", "!=", "TRUE")]),
        ("u64_and_bool", [("command_id", "<", "700"), "AND", ("sudo_used", "=", "1")]),
        ("no_bool_condition", [("risk_level", ">", "3")]),
        ("string_only", [("user_name", "=", "student1030")]),
    ]
    for op in ("=", "!=", ">", ">=", "<", "<="):
        for lit in ("TRUE", "FALSE", "true", "1", "0", "maybe"):
            cases.append((f"op_{op}_{lit}", [("sudo_used", op, lit), "AND", ("risk_level", ">", "0")] if lit == "maybe"
                          else [("sudo_used", op, lit)]))
    return cases


def boolprobe_goldens(csv2k):
    """select_boolprobe_golden.json: the row selection of executeQuerySelectOMP (oracle/_ref/libqpeomp_ref.so =
    engine/omp + oracle/ref_harness_omp.c) with ONE thread.  Must be run with OMP_NUM_THREADS=1: with more, the
    reference appends the candidates of different indexes in whatever order its threads arrive (omp:366,481)."""
    import os
    assert os.environ.get("OMP_NUM_THREADS") == "1", "run with OMP_NUM_THREADS=1"
    if q.load_ref_omp() is None:
        sys.exit("oracle/_ref/libqpeomp_ref.so missing: run `make -C oracle` in the authoring container")
    q.build_oracle()
    out = []
    cols = ["command_id", "sudo_used", "risk_level", "user_name"]
    for cfg, indexes in BOOLPROBE_INDEX_CONFIGS.items():
        eng, orc = q.RefOmpEngine(csv2k, indexes), q.OracleTable(csv2k, indexes)
        for name, chain in boolprobe_cases():
            o_ids, o_count, cand = orc.select_ids(chain, probe_bool=True)
            if cand > eng.n:                                    # would overflow the reference's candidate buffer (omp:346)
                print(f"{name}/{cfg}: {cand} candidates for {eng.n} rows -- not sent to the reference")
                continue
            res = eng.select_where(cols, chain)
            ids = [int(r[0]) for r in res["rows"]]
            assert ids == o_ids, (name, cfg, len(ids), len(o_ids))
            serial_ids, _, _ = orc.select_ids(chain)
            out.append({"name": name, "csv": csv2k.name, "indexes": cfg, "where": q.chain_to_jsonable(chain), "candidates": cand,
                        "num_records": res["numRecords"], "columns": res["columns"], "rows_sha256": sha_rows(res["rows"]),
                        "ids_zlib_b64": q.pack_ids(ids), "differs_from_qpeseq": ids != serial_ids})
        eng.close()
    (HERE / "select_boolprobe_golden.json").write_text(json.dumps(out, separators=(",", ":")))
    print(f"bool-probe cases: {len(out)} pinned, {sum(c['differs_from_qpeseq'] for c in out)} of them differ from QPESeq's answer")


def driver_goldens(csv2k):
    """End-to-end driver goldens: the reference's QPESeq on its own sample-queries.txt and
    sample-queries-FULL.txt (which adds Sample 6, the DELETE).  The driver always opens
    "sample-queries.txt" in its working directory (QPESeq.c:40); INSERT appends to the CSV and DELETE
    rewrites it, so every run works on a scratch copy, and the CSV the run leaves behind is part of
    the golden (as a SHA-256)."""
    import hashlib
    import shutil
    qpeseq = q.ORACLE_DIR / "_ref" / "QPESeq_ref"
    for src, out_name in (("sample-queries.txt", "qpeseq_stdout.txt"), ("sample-queries-FULL.txt", "qpeseq_full_stdout.txt")):
        sample = HERE / src
        if not sample.exists():
            shutil.copy("/root/reference/" + src, sample)                   # input data file of the reference
        with tempfile.TemporaryDirectory() as td:
            shutil.copy(csv2k, pathlib.Path(td) / "data.csv")
            shutil.copy(sample, pathlib.Path(td) / "sample-queries.txt")
            out = subprocess.run([str(qpeseq), "data.csv"], cwd=td, capture_output=True, check=True).stdout.decode("latin-1")
            left = (pathlib.Path(td) / "data.csv").read_bytes()
        (HERE / out_name).write_text(normalize_driver_output(out), encoding="latin-1")
        (HERE / (out_name[:-4] + "_csv.sha256")).write_text(hashlib.sha256(left).hexdigest() + f" {len(left)}\n")


FIFTY_K_ROWS = 50_000


def fifty_k_goldens():
    """BASELINE configs[0] at its stated size: the reference's own SELECT test and dispatcher read commands_50k.csv
    (tests/serial-SELECT-test.c:12, include/connectEngine.h:11).  The reference's file is an LFS stub and its generator is
    unseeded, so the 50 k-row CSV comes from the repository's own seeded scripts/make_csv.py -- regenerated wherever it is
    needed, never committed: its SHA-256 is in the golden, and so is, per query of sample-queries.txt / the SELECT test, what
    the compiled reference answered (count, SHA-256 of the row numbers, SHA-256 of every projected cell) under the default
    five indexes and without indexes, plus the SHA-256 of QPESeq's normalised stdout for both sample files and of the CSV
    each run leaves behind."""
    import hashlib
    import shutil
    ref = q.load_ref()
    if ref is None:
        sys.exit("oracle/_ref/libqpeseq_ref.so missing")
    qpeseq = q.ORACLE_DIR / "_ref" / "QPESeq_ref"
    out = {"rows": FIFTY_K_ROWS, "generator": "scripts/make_csv.py 50000 <out.csv> (seed 0x5EED)", "select": [], "driver": {}}
    with tempfile.TemporaryDirectory() as td:
        td = pathlib.Path(td)
        csv = td / "commands_50k.csv"
        subprocess.run([sys.executable, str(q.ROOT / "scripts" / "make_csv.py"), str(FIFTY_K_ROWS), str(csv)], check=True)
        out["csv_sha256"] = hashlib.sha256(csv.read_bytes()).hexdigest()
        out["csv_bytes"] = csv.stat().st_size
        names = ["S1", "S2", "S3", "S4", "S7", "S8", "T_all", "T_proj", "T_risk", "T_shell", "T_and", "T_exit"]
        for cfg in ("default", "none"):
            eng = q.RefEngine(csv, INDEX_CONFIGS[cfg])
            for name in names:
                sel, where = next((s_, w) for n_, s_, w in SELECT_CASES if n_ == name)
                sql = compose(sel, where)
                _head, chain = parse_with_ref(ref, sql)
                res = eng.select(sql)
                ids = [int(r[0]) for r in eng.select(compose("command_id", where))["rows"]]     # command_id = row number in this table
                out["select"].append({"name": name, "indexes": cfg, "sql": sql, "where": q.chain_to_jsonable(chain),
                                      "num_records": res["numRecords"], "columns": res["columns"],
                                      "rows_sha256": sha_rows(res["rows"]),
                                      "ids_sha256": hashlib.sha256(b"".join(i.to_bytes(4, "little") for i in ids)).hexdigest(),
                                      "first_ids": ids[:5]})
            eng.close()
        for src in ("sample-queries.txt", "sample-queries-FULL.txt"):
            run = td / ("run_" + src)
            run.mkdir()
            shutil.copy(csv, run / "data.csv")
            shutil.copy(HERE / src, run / "sample-queries.txt")
            text = subprocess.run([str(qpeseq), "data.csv"], cwd=run, capture_output=True, check=True).stdout.decode("latin-1")
            left = (run / "data.csv").read_bytes()
            out["driver"][src] = {"stdout_sha256": hashlib.sha256(normalize_driver_output(text).encode("latin-1")).hexdigest(),
                                  "stdout_lines": normalize_driver_output(text).count("\n"),
                                  "csv_left_sha256": hashlib.sha256(left).hexdigest(), "csv_left_bytes": len(left)}
    (HERE / "commands_50k_golden.json").write_text(json.dumps(out, indent=0))
    print(f"50 k-row goldens: {len(out['select'])} SELECT cases, 2 driver runs")


if __name__ == "__main__":
    import sys
    if "--50k-only" in sys.argv:
        fifty_k_goldens()
    elif "--driver-only" in sys.argv:
        driver_goldens(HERE / "commands_2k.csv")
    elif "--wide-only" in sys.argv:
        wide_goldens(HERE / "commands_2k.csv")
    elif "--boolprobe-only" in sys.argv:
        boolprobe_goldens(HERE / "commands_2k.csv")
    elif "--random-only" in sys.argv:
        random_goldens(HERE / "commands_2k.csv")
    else:
        main()
