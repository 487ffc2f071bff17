"""host/tokenizer.c + host/connectEngine.c:convert_conditions against what the
reference's tokenizer/parser produced for the same SQL text
(tests/golden/parse_golden.json, from oracle/_ref).  CPU only."""
import json

import pytest

import qpelib as q

CASES = json.loads((q.GOLDEN / "parse_golden.json").read_text())


@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_front_end_matches_reference(case):
    lib = q.pq.lib()
    sql = case["sql"].encode("latin-1")
    assert q.call_text(lib.hipDumpTokens, sql) == case["tokens"]
    assert q.call_text(lib.hipDumpParse, sql) == case["parse"]


def test_live_against_reference_if_present():
    ref = q.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    lib = q.pq.lib()
    extra = [
        "SELECT * FROM c WHERE a=1 AND b=2 AND c=3 AND d=4 AND e=5 AND f=6",
        "SELECT * FROM c WHERE a=1 OR b=2 OR c=3 OR d=4 OR e=5 OR f=6 OR g=7",
        "SELECT * FROM c WHERE a=1 AND b=2 AND c=3 AND d=4 AND e=5",
        "SELECT * FROM c WHERE (a=1 AND b=2 AND c=3 AND d=4 AND e=5 AND f=6) OR g=7",
        "SELECT a,b,c,d,e,f,g,h,i,j FROM c WHERE x='1'",
        "DELETE FROM c WHERE (a = 1 OR b = 2) AND c = 3",
        "SELECT * FROM c WHERE a != 'x' AND b >= \"y z\" OR (c < 3 AND d <= 4)",
    ]
    for sql in extra:
        s = sql.encode()
        assert q.call_text(lib.hipDumpTokens, s) == q.call_text(ref.refh_tokens, s), sql
        assert q.call_text(lib.hipDumpParse, s) == q.call_text(ref.refh_parse, s), sql
