#!/bin/bash
# rocprofv3 PMC passes (one counter group per run) over a short bench.py run.
# usage: scripts/profile_pmc.sh <tag> <rows> [extra bench args]
tag=$1; rows=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --reps 1 --no-cpu-baseline --no-extras --rows $rows "$@" > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$name.err
  echo "pmc $name rc=$?"
done
