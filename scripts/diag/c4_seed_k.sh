#!/bin/bash
# seeded pilot loop: segments cost no warm-up any more, so more of them shorten every chain for free -- until the SIMDs' issue rate binds
cd "$GRAFT_REPO_ROOT"
export PYSDR_TUNING=1
for cfg in default 20,13,4,3,2048,2048 20,13,4,3,3072,1024 20,13,4,3,4096,1024; do
  if [ "$cfg" != default ]; then export PYSDR_WFM_PLL=$cfg; else unset PYSDR_WFM_PLL; fi
  for ov in "" "--no-overlap"; do
  python3 bench.py --workload c4 $ov --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
  python3 - "$cfg" "$ov" <<'PY'
import json,sys
d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
print("%-22s %-13s %7.1f GS/s %.3f ms verify %.2g %s" % (sys.argv[1], sys.argv[2] or "overlapped", d['value']/1e3, d['ms_per_step'], d.get('verify_worst_rel',-1), json.dumps(d['pilot_pll'])[:110]))
PY
  done
done
