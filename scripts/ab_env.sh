#!/bin/bash
# A/B of one tuning switch of the shim on the GPU box: runs bench.py once per value and prints K1 time,
# roofline fraction and whole-job time.  usage: scripts/ab_env.sh VAR value [value ...] -- [bench.py args]
#   PQPS_NT_LOADS=0|1            plain / streaming loads (default: by scan footprint, 320 MiB)
#   PQPS_K1_ITERS=<float>        grid of K1 as iterations per wave (default: 1 when streaming, 1.5 otherwise)
#   PQPS_K1_BLOCKS_PER_CU=<int>  grid cap of K1 per CU
#   PQPS_CHAIN_MULTI=0|1         several steps per loop iteration for a lone 1-byte column
#   PQPS_SUM_LAG, PQPS_EXPAND_LAG, PQPS_EXPAND_SPIN_LIMIT, PQPS_QSTREAM_LANES: see scripts/README.md
var=$1; shift
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
for v in "${vals[@]}"; do
  env "$var=$v" python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null |
    python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$var=$v', 'K1 us', round(r['avg_kernel_ms']*1e3,1), 'frac', round(r['frac'],3), 'us/query', round(d['ms_per_step']*1e3,1))"
done
