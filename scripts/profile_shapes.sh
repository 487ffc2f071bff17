#!/bin/bash
# Round-4 profile set: per shape one `rocprofv3 --kernel-trace --stats` run and the FETCH_SIZE / WRITE_SIZE passes (one counter
# group per run, `--kernel-trace` only), every launch of the scan kernel one whole query (`--level shim --no-pipeline`).
# usage (on the GPU box): scripts/profile_shapes.sh <set>      set = a (100 M rows) | b (1 G rows) | "tag:query:rows:copies ..."
# -> gpurun_out/prof_r04_<tag>/, gpurun_out/pmc_r04_<tag>_*/; then, here: scripts/summarize_profiles.py r04 r04_<tag> ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
case "$1" in
  a) SET="s1_100m:S1:100000000:2 qa_100m:Q_A:100000000:2 qb_100m:Q_B:100000000:2 u8_100m:Q_u8:100000000:2 u16_100m:Q_u16:100000000:2 r2_100m:Q_r2:100000000:2 r1_100m:Q_r1:100000000:2";;
  b) SET="s1_1b:S1:1000000000:1 qa_1b:Q_A:1000000000:1 qb_1b:Q_B:1000000000:1 qc_1b:Q_C:1000000000:1 r1_1b:Q_r1:1000000000:1";;
  *) SET="$1";;
esac
for spec in $SET; do
  IFS=: read tag q rows copies <<< "$spec"
  ARGS="--level shim --no-pipeline --steps 50 --warmup 5 --reps 1 --no-cpu-baseline --no-extras --query $q --rows $rows --copies $copies"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04_$tag -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_r04_$tag.json 2> $R/gpurun_out/prof_r04_$tag.err
  echo "stats $tag rc=$?"
  for grp in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r04_${tag}_$grp -- python3 $R/bench.py ${ARGS/--steps 50 --warmup 5/--steps 6 --warmup 2} > /dev/null 2> $R/gpurun_out/pmc_r04_${tag}_$grp.err
    echo "pmc $tag $grp rc=$?"
  done
done
