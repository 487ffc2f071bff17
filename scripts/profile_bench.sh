#!/bin/bash
# rocprofv3 kernel-trace stats of one bench.py run; CSVs land in gpurun_out/prof_<tag>/
# (pass --no-pipeline to have every launch of the scan kernel be one whole query, as in bench.py's roofline leg)
# usage: scripts/profile_bench.sh <tag> [extra bench.py args]
tag=${1:-r01}; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --reps 1 --no-cpu-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.err
echo "rocprof rc=$?"
