#!/bin/bash
# Run ON THE GPU BOX with the diagnostic library built (python -m pysdr_amd.build --diag): the mix + decimate kernel of the
# long-prototype multi-RX workloads with parts of its tile loop skipped (PYSDR_DEBUG_FLAGS: 1 no dot products, 2 no tile
# copies, 4 no raw-peak scan, 8 no output stores) -- results WRONG by design, timing only.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-host-fed --no-other-configs --no-verify --full-line"
for w in ${WLS:-ft8tri test2rx}; do
 for f in ${FLAGS:-0 1 2 4 8 5 6 3 7}; do
  PYSDR_TUNING=1 PYSDR_USE_DIAG_LIB=1 PYSDR_DEBUG_FLAGS=$f python3 bench.py --workload $w $B 2>/dev/null | tail -1 | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip())
print('$w flags $f: front %.4f ms  step %.4f ms' % (j['kernel_ms']['front'], j['ms_per_step']))"
 done
done
