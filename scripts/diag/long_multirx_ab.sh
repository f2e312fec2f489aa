#!/bin/bash
# Run ON THE GPU BOX: the long-prototype multi-RX shapes of mixdec.hip, variant libraries built beforehand
# (python -m pysdr_amd.build --variant NAME) against the shipped one, alternating, on one box.
#   VARIANTS="main tpb512 nh3" WLS="ft8tri test2rx" REPS=2 bash scripts/diag/long_multirx_ab.sh
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-host-fed --no-other-configs --full-line"
for rep in $(seq 1 ${REPS:-2}); do
 for w in ${WLS:-ft8tri test2rx}; do
  for v in ${VARIANTS:-main}; do
   args="--workload $w"
   case $w in c2|rx6|c1|c4) args="--workload $w";; c3_nopsd) args="--workload c3 --no-psd";; c3_1001) args="--workload c3 --ntaps 1001 --no-psd";; rx6_1001) args="--workload rx6 --ntaps 1001";; c2_1001) args="--workload c2 --ntaps 1001";; esac
   if [ "$v" = main ]; then e="PYSDR_X=0"; else e="PYSDR_TUNING=1 PYSDR_LIB_VARIANT=$v"; fi
   env $e python3 bench.py $args $B 2>&1 | tail -1 | python3 -c "
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
