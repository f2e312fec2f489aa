#!/bin/bash
# per-kernel times of the C4 step (rocprofv3 --kernel-trace --stats), optionally under a PYSDR_WFM_PLL setting: c4_kt.sh [cfg...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "${@:-default}"; do
  O=gpurun_out/c4_kt/$(echo $cfg | tr ',' '_'); rm -rf $O; mkdir -p $O
  if [ "$cfg" != default ]; then export PYSDR_TUNING=1 PYSDR_WFM_PLL=$cfg; else unset PYSDR_TUNING PYSDR_WFM_PLL; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --workload c4 --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 20 --warmup 4 > $O.log 2>&1
  echo "== $cfg"
  python3 - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:8]:
        print("   %-40s calls %4s avg %9.1f us  %5.1f %%" % (r["Name"].replace("(anonymous namespace)::", "").replace("pysdr::", "").replace("void ", "")[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
