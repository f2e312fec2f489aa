#!/bin/bash
# AM-Synch: a block's first guess by the direct linear solve (default) against the free-running line (PYSDR_AM_DIRECT=0), with
# and without the linear warm-ups (PYSDR_AM_SEED); overlapped and single-stream; kernel averages; the live one-chunk call
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYSDR_TUNING=1
line() {
python3 - "$@" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
    p=d.get('carrier_pll') or {}
    print("%-40s %7.1f GS/s %.3f ms  %s  job %.3f verify %.2g  seg %s patched %s join %s lin %s" % (" ".join(sys.argv[1:]), d['value']/1e3, d['ms_per_step'], {k:(round(v,3) if v else v) for k,v in d['kernel_ms'].items()}, d['roofline_job']['frac'], d.get('verify_worst_rel',-1), p.get('segments'), p.get('patched_serially'), (p.get('widest_join') or {}).get('phase_words_of_2^32'), p.get('linear_starts')))
except Exception as e:
    print("FAILED", sys.argv[1:], e, open('/tmp/o.err').read()[-600:])
PY
}
for rep in 1 2; do
for cfg in "1 1" "0 1" "1 0" "0 0"; do
  set -- $cfg
  PYSDR_AM_DIRECT=$1 PYSDR_AM_SEED=$2 python3 bench.py --workload c1synch --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
  line direct=$1 seed=$2 overlapped
  PYSDR_AM_DIRECT=$1 PYSDR_AM_SEED=$2 python3 bench.py --workload c1synch --no-overlap --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
  line direct=$1 seed=$2 single-stream
done
done
python3 bench.py --workload c1 --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
line c1 plain AM
for s in "1 1" "0 0"; do
  set -- $s
  O=gpurun_out/am_direct_kt_$1$2; rm -rf $O; mkdir -p $O
  PYSDR_AM_DIRECT=$1 PYSDR_AM_SEED=$2 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --workload c1synch --no-overlap --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 20 --warmup 4 > $O.log 2>&1
  echo "direct=$1 seed=$2 single-stream kernel averages:"
  python3 - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:7]:
        print("   %-50s calls %4s avg %9.1f us  %5.1f %%" % (r["Name"].replace("(anonymous namespace)::", "").replace("pysdr::", "").replace("void ", "")[:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
[ -f scripts/diag/live_latency.py ] && { PYSDR_AM_DIRECT=1 python3 scripts/diag/live_latency.py 2>&1 | tail -6; echo "-- direct=0"; PYSDR_AM_DIRECT=0 python3 scripts/diag/live_latency.py 2>&1 | tail -6; }
