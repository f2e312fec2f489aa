cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T0=$(date +%s)
python3 bench.py > gpurun_out/bench_live_line.json 2> gpurun_out/bench_live.err; echo rc=$?
echo "wall $(( $(date +%s) - T0 )) s"; wc -c gpurun_out/bench_live_line.json
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_live_line.json").read().strip().splitlines()[-1])
print(d["roofline"]); print(d.get("roofline_mixdec"))
f=json.load(open(d["full"])); print(f.get("live_traffic_bytes_per_launch")); print(f["roofline_mixdec"].get("traffic"), f["roofline_mixdec"].get("traffic_source"))
PY
