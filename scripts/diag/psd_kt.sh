#!/bin/bash
# per-kernel times of the PSD harness variants: psd_kt.sh name...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  O=gpurun_out/psd_kt/$v; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- scripts/experiments/psdvar_$v.bin 2048 3 > $O.log 2>&1
  echo "== $v: $(grep two-kernel $O.log)"
  python3 - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "psd_" in n: print("   %-28s calls %4s avg %9.1f us" % (n.split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
