#!/bin/bash
# Run ON THE GPU BOX: what a flush of the mix + decimate kernel's output stage costs -- the front end's time against the number of
# tiles between flushes (PYSDR_MIXDEC_YFLUSH caps it; the host's plan gives 16 / 10 / 5 at 1 / 4 / 6 RX, 255 taps).
#   bash scripts/diag/yflush_sweep.sh            (profiles/r06_yflush.txt)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-host-fed --no-other-configs --full-line --no-verify"
for w in "c2" "c3 --no-psd" "rx6" "ft8tri"; do
  for yf in ${YFS:-0 1 2 3 5 8 16}; do
    if [ "$yf" = 0 ]; then e="PYSDR_X=0"; else e="PYSDR_TUNING=1 PYSDR_MIXDEC_YFLUSH=$yf"; fi
    env $e python3 bench.py --workload $w $B 2>&1 | tail -1 | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip()); r=j.get('roofline_mixdec') or {}
print('$w yflush=$yf', 'front', round(r.get('avg_launch_ms',0),4), 'ms frac', round(r.get('frac',0),3))
"
  done
done
