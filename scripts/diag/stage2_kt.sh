#!/bin/bash
# per-kernel times of the demod-only C3 step: stage2_kt.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/stage2_kt; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --no-psd --no-cpu-baseline --no-host-fed --steps 20 "$@" > $O.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if int(r["Calls"]) >= 20: print("   %-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
