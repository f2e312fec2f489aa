cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
for t in _r04 .; do
  (cd $t && python3 bench.py --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err; python3 - $t <<'PY'
import json,sys
d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
print(sys.argv[1], "%.1f GS/s %.3f ms" % (d['value']/1e3, d['ms_per_step']), d['kernel_ms'])
PY
  )
done
done
