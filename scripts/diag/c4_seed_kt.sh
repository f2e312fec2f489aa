cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c4_seed_kt; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --workload c4 --no-overlap --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 20 --warmup 4 > $O.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("   %-50s calls %4s avg %9.1f us  %5.1f %%" % (r["Name"].replace("(anonymous namespace)::", "").replace("pysdr::", "").replace("void ", "")[:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
