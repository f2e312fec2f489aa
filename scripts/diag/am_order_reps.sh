#!/bin/bash
# c1synch overlapped: the previous call's tail BEHIND the start of this call's walks (default for AM-Synch: walks beside the
# tail, then beside the next front end) against IN FRONT of it (PYSDR_OVERLAP_ORDER=1: walks beside the next front end only),
# alternating, N repetitions -- measured again now that the walks are 65 us instead of 100
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYSDR_TUNING=1
N=${1:-6}
for rep in $(seq 1 $N); do
  for d in 0 1; do
    PYSDR_OVERLAP_ORDER=$d python3 bench.py --workload c1synch --no-cpu-baseline --no-host-fed --no-other-configs --no-verify > /tmp/o.json 2>/tmp/o.err
    python3 - $d <<'PY'
import json,sys
d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
print("tail_first=%s %7.1f GS/s %.3f ms front %.3f" % (sys.argv[1], d['value']/1e3, d['ms_per_step'], d['kernel_ms']['front']))
PY
  done
done | tee /tmp/reps.txt
python3 - <<'PY'
import re, statistics as st
v={'0':[], '1':[]}
for l in open('/tmp/reps.txt'):
    m=re.match(r"tail_first=(\d)\s+([\d.]+)", l)
    if m: v[m.group(1)].append(float(m.group(2)))
for k in '01': print("tail_first=%s: median %.1f mean %.1f min %.1f max %.1f (n=%d)" % (k, st.median(v[k]), st.mean(v[k]), min(v[k]), max(v[k]), len(v[k])))
PY
