#!/bin/bash
# per-kernel times of the c1synch step (rocprofv3 --kernel-trace --stats), optionally under a PYSDR_AM_PLL setting
# ("taus,taus_exact,coarse_sweeps,kmax,tmin"): amsynch_kt.sh [cfg...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "${@:-default}"; do
  O=gpurun_out/amsynch_kt/$(echo $cfg | tr ',' '_'); rm -rf $O; mkdir -p $O
  if [ "$cfg" != default ]; then export PYSDR_TUNING=1 PYSDR_AM_PLL=$cfg; else unset PYSDR_TUNING PYSDR_AM_PLL; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --workload c1synch --no-cpu-baseline --no-host-fed --no-other-configs --steps 20 --warmup 4 > $O.log 2>&1
  echo "== $cfg"
  python3 - $O $O.log <<'PY'
import csv, glob, json, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:9]:
        print("   %-40s calls %4s avg %9.1f us  %5.1f %%" % (r["Name"].replace("(anonymous namespace)::", "").replace("pysdr::", "").replace("void ", "")[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print("   step %.3f ms, %.1f GS/s, verify %.2g, carrier_pll %s" % (d["ms_per_step"], d["value"] / 1e3, d.get("verify_worst_rel", -1), json.dumps(d["carrier_pll"])))
except Exception as e:
    print("   no bench line:", e)
PY
done
