#!/bin/bash
# rocprofv3 kernel-trace stats + the PMC passes of an arbitrary python script (one counter group per run).
# usage: scripts/profile_cmd.sh <tag> <script.py> [args...]      -> gpurun_out/prof_<tag>/, gpurun_out/pmc_<tag>_*/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/"$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.err
echo "rocprof $tag rc=$?"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$name -- python3 $GRAFT_REPO_ROOT/"$@" > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$name.err
  echo "pmc $tag $name rc=$?"
done
