#!/bin/bash
# Run ON THE GPU BOX: the matrix-core front end with its roles meeting through LDS counters (variant library "flags":
# PYSDR_TUNING=1 PYSDR_MFMA_FLAGS=-DMM_FLAGS=1 python -m pysdr_amd.build --variant flags) against the shipped barrier-per-tile
# form, alternating on one box; first the parity tests of the matrix-core path under the variant.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PYSDR_TUNING=1 PYSDR_LIB_VARIANT=${TESTV:-flags} timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c1_am or long_prototype or raw_chunk or non_finite or c4_wbfm or matrix_core or full_size_batch" 2>&1 | tail -4
B="--no-cpu-baseline --no-host-fed --no-other-configs --full-line"
for rep in $(seq 1 ${REPS:-3}); do
 for w in ${WLS:-c1 c4 c1synch}; do
  for v in main ${VARIANTS:-flags}; do
   if [ "$v" = main ]; then e="PYSDR_X=0"; else e="PYSDR_TUNING=1 PYSDR_LIB_VARIANT=$v"; fi
   env $e python3 bench.py --workload $w $B 2>&1 | tail -1 | python3 -c "
import sys,json
try:
    j=json.loads(sys.stdin.read().strip()); r=j.get('roofline_mixdec') or {}
    print('$v $w', round(j['value']/1e3,1),'GS/s', round(j['ms_per_step'],4),'ms; front', round(r.get('avg_launch_ms',0),4), 'ms frac', round(r.get('frac',0),3), 'job', round(j['roofline_job']['frac'],3), 'verify', j.get('verify_worst_rel'), j.get('verified_ranks'))
except Exception as e:
    print('$v $w FAILED', e)
"
  done
 done
done
