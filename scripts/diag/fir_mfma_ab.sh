#!/bin/bash
# AF FIR on the matrix cores (default there) against the packed-FMA form (PYSDR_FIR_MFMA=0): every workload, same box; then the
# (the matrix-core kernel was removed after this measurement; it is in commit 28d40af: check that out to re-run)
# kernel averages of C1 and 6 RX by rocprofv3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYSDR_TUNING=1
for w in c1 c2 rx6 c1synch c4mono c4 c3; do
  for m in 1 0; do
    PYSDR_FIR_MFMA=$m python3 bench.py --workload $w --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
    python3 - $w $m <<'PY'
import json,sys
try:
    d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
    print("%-8s fir_mfma %s %7.1f GS/s %.3f ms  %s  job %.3f verify %.2g" % (sys.argv[1], sys.argv[2], d['value']/1e3, d['ms_per_step'], {k:(round(v,3) if v else v) for k,v in d['kernel_ms'].items()}, d['roofline_job']['frac'], d.get('verify_worst_rel',-1)))
except Exception as e:
    print("FAILED", sys.argv[1:], e, open('/tmp/o.err').read()[-600:])
PY
  done
done
for w in c1 rx6; do
  O=gpurun_out/fir_mfma_kt_$w; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --workload $w --no-overlap --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 20 --warmup 4 > $O.log 2>&1
  python3 - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("   %-50s calls %4s avg %9.1f us  %5.1f %%" % (r["Name"].replace("(anonymous namespace)::", "").replace("pysdr::", "").replace("void ", "")[:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
