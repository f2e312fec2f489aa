#!/bin/bash
# timeline of the kernels of a few steps (which kernels of the second half run beside the next front end?)
#   scripts/diag/overlap_trace.sh <workload> [bench flags...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=${1:-c4}; shift
O=gpurun_out/overlap_trace_$W; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 bench.py --workload $W --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 8 --warmup 4 "$@" > $O.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
rows = rows[-45:]
t0 = rows[0][0]
for s, e, n, q, st in rows:
    n = n.replace("pysdr::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:34]
    print("%10.1f us .. %10.1f  (%8.1f us)  q=%s s=%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, st, n))
PY
