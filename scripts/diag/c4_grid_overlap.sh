#!/bin/bash
# C4 overlapped with the IF decimator on fewer workgroups than CUs: do the pilot-loop waves then find SIMDs without MFMA waves?
cd "$GRAFT_REPO_ROOT"
export PYSDR_TUNING=1
for g in 256 248 240 224 192; do
  PYSDR_MIXDEC_GRID=$g python3 bench.py --workload c4 --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
  python3 - $g <<'PY'
import json,sys
d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
print("grid %s %7.1f GS/s %.3f ms  %s  verify %.2g" % (sys.argv[1], d['value']/1e3, d['ms_per_step'], {k:(round(v,3) if v else v) for k,v in d['kernel_ms'].items()}, d.get('verify_worst_rel',-1)))
PY
done
