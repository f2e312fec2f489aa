cd "$GRAFT_REPO_ROOT"
for cfg in default 20,13,4,3,2304,1024 20,13,4,3,3072,1024 20,13,4,3,1024,2048; do
  if [ "$cfg" != default ]; then export PYSDR_TUNING=1 PYSDR_WFM_PLL=$cfg; else unset PYSDR_TUNING PYSDR_WFM_PLL; fi
  python3 bench.py --workload c4 --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
  python3 - <<'PY'
import json
d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
print("%8.1f GS/s %.3f ms verify %.2g %s" % (d['value']/1e3, d['ms_per_step'], d.get('verify_worst_rel',-1), json.dumps(d['pilot_pll'])[:120]))
PY
done
