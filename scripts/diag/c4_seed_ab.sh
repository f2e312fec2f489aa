#!/bin/bash
# C4: the pilot loop's segments from Newton-in-time seeds (default) against the 13-tau warm-ups (PYSDR_WFM_SEED=0), overlapped and single-stream
cd "$GRAFT_REPO_ROOT"
export PYSDR_TUNING=1
for rep in 1 2; do
for seed in 1 0 "1,1024"; do
  for ov in "" "--no-overlap"; do
    PYSDR_WFM_SEED=$seed python3 bench.py --workload c4 $ov --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
    python3 - "$seed" "$ov" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
    print("seed %-7s %-13s %7.1f GS/s %.3f ms  %s  verify %.2g  %s" % (sys.argv[1], sys.argv[2] or "overlapped", d['value']/1e3, d['ms_per_step'], {k:(round(v,3) if v else v) for k,v in d['kernel_ms'].items()}, d.get('verify_worst_rel',-1), json.dumps(d['pilot_pll'])[:150]))
except Exception as e:
    print("FAILED", sys.argv[1:], e, open('/tmp/o.err').read()[-800:])
PY
  done
done; done
