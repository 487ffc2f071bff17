#!/bin/bash
# End-to-end: the real reference driver (oracle/_ref/QPESeq_ref) vs QPEHIP on the same synthetic CSV and
# the reference's sample-queries.txt (QUERIES=sample-queries-FULL.txt for the file with the DELETE).  usage: tests/e2e_compare.sh <rows>
rows=${1:-1000000}
work=$(mktemp -d)
python3 $GRAFT_REPO_ROOT/scripts/make_csv.py $rows $work/data.csv
cp $GRAFT_REPO_ROOT/tests/golden/${QUERIES:-sample-queries.txt} $work/sample-queries.txt
cd $work
cp data.csv data_ref.csv; cp data.csv data_hip.csv
echo "== QPESeq (reference, 1 core)"; ( time $GRAFT_REPO_ROOT/oracle/_ref/QPESeq_ref data_ref.csv > ref.out ) 2>&1 | grep real; grep -a -E "Initialization|Query Execution|Total Execution" ref.out | sed 's/\x1b\[[0-9;]*m//g'
echo "== QPEHIP"; ( time $GRAFT_REPO_ROOT/parallel-query-processing-system_amd/QPEHIP data_hip.csv > hip.out ) 2>&1 | grep real; grep -a -E "Initialization|Query Execution|Total Execution" hip.out | sed 's/\x1b\[[0-9;]*m//g'
norm() { sed -E 's/Query Time: [0-9.]+ seconds/Query Time: X seconds/; s/Execution Time: [0-9.]+/Execution Time: X/' "$1" | sed '/Execution Summary/,$d'; }
norm ref.out > ref.norm; norm hip.out > hip.norm
if cmp -s ref.norm hip.norm; then echo "OUTPUT IDENTICAL ($(wc -l < ref.norm) lines)"; else echo "OUTPUT DIFFERS"; diff ref.norm hip.norm | head -20; fi
if cmp -s data_ref.csv data_hip.csv; then echo "CSV LEFT BEHIND IDENTICAL ($(wc -c < data_ref.csv) bytes)"; else echo "CSV LEFT BEHIND DIFFERS"; fi
grep -a "Query Time" ref.out | head -8; echo --; grep -a "Query Time" hip.out | head -8
rm -rf $work
