#!/bin/bash
# AM-Synch: warm-ups by the linear solve (default) against walked (PYSDR_AM_SEED=0); arg y on the walks' stream (default) against
# the front stream (PYSDR_AM_PHASE_STREAM=0); overlapped and single-stream; then the kernel averages
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYSDR_TUNING=1
line() {
python3 - "$@" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
    print("%-34s %7.1f GS/s %.3f ms  %s  job %.3f verify %.2g  pll %s" % (" ".join(sys.argv[1:]), d['value']/1e3, d['ms_per_step'], {k:(round(v,3) if v else v) for k,v in d['kernel_ms'].items()}, d['roofline_job']['frac'], d.get('verify_worst_rel',-1), d.get('carrier_pll')))
except Exception as e:
    print("FAILED", sys.argv[1:], e, open('/tmp/o.err').read()[-600:])
PY
}
for rep in 1 2; do
for cfg in "1 1" "0 1" "1 0" "0 0"; do
  set -- $cfg
  PYSDR_AM_SEED=$1 PYSDR_AM_PHASE_STREAM=$2 python3 bench.py --workload c1synch --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
  line seed=$1 phase_on_2=$2 overlapped
done
for s in 1 0; do
  PYSDR_AM_SEED=$s python3 bench.py --workload c1synch --no-overlap --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
  line seed=$s single-stream
done
done
python3 bench.py --workload c1 --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
line c1 plain AM
for s in 1 0; do
  O=gpurun_out/am_seed_kt_$s; rm -rf $O; mkdir -p $O
  PYSDR_AM_SEED=$s rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --workload c1synch --no-overlap --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 20 --warmup 4 > $O.log 2>&1
  echo "seed=$s single-stream kernel averages:"
  python3 - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:7]:
        print("   %-50s calls %4s avg %9.1f us  %5.1f %%" % (r["Name"].replace("(anonymous namespace)::", "").replace("pysdr::", "").replace("void ", "")[:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
