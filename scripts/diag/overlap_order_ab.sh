#!/bin/bash
# overlapped calls: the previous call's tail behind (0, default) or in front of (1) the start of this call's loop walks
cd "$GRAFT_REPO_ROOT"
export PYSDR_TUNING=1
for rep in 1 2; do
for w in c4 c1synch; do
  for o in 0 1; do
    PYSDR_OVERLAP_ORDER=$o python3 bench.py --workload $w --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
    python3 - $o $w <<'PY'
import json,sys
d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
print("order %s %-8s %7.1f GS/s %.3f ms  %s  verify %.2g" % (sys.argv[1], sys.argv[2], d['value']/1e3, d['ms_per_step'], {k:(round(v,3) if v else v) for k,v in d['kernel_ms'].items()}, d.get('verify_worst_rel',-1)))
PY
  done
  python3 bench.py --workload $w --no-overlap --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
  python3 - off $w <<'PY'
import json,sys
d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
print("order %s %-8s %7.1f GS/s %.3f ms  %s  verify %.2g" % (sys.argv[1], sys.argv[2], d['value']/1e3, d['ms_per_step'], {k:(round(v,3) if v else v) for k,v in d['kernel_ms'].items()}, d.get('verify_worst_rel',-1)))
PY
done; done
