#!/bin/bash
# timeline of the kernels of a few steps (do stage 2 and the next front end overlap?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/overlap_trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 bench.py --no-psd --no-cpu-baseline --no-host-fed --steps 6 --warmup 2 "$@" > $O.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0 = rows[0][0]
for s, e, n, q, st in rows[-40:]:
    n = n.replace("pysdr::(anonymous namespace)::", "").split("(")[0][-28:]
    print("%10.1f us  +%8.1f us  q=%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
PY
