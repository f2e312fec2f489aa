#!/bin/bash
# c1synch single-stream: kernel averages by rocprofv3 + bench lines (overlapped, single-stream)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYSDR_TUNING=1
for o in "" "--no-overlap"; do
  python3 bench.py --workload c1synch $o --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
  python3 - "$o" <<'PY'
import json,sys
d=json.loads([l for l in open("/tmp/o.json") if l.startswith("{")][-1]); p=d["carrier_pll"]
print("%-13s %.1f GS/s %.4f ms %s verify %.2g lin %s join %s patched %s" % (sys.argv[1] or "overlapped", d["value"]/1e3, d["ms_per_step"], {k:(round(v,3) if v else v) for k,v in d["kernel_ms"].items()}, d["verify_worst_rel"], p["linear_starts"], p["widest_join"]["phase_words_of_2^32"], p["patched_serially"]))
PY
done
O=gpurun_out/am_kt; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --workload c1synch --no-overlap --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 20 --warmup 4 "$@" > $O.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:7]:
        print("   %-50s calls %4s avg %9.1f us  %5.1f %%" % (r["Name"].replace("(anonymous namespace)::", "").replace("pysdr::", "").replace("void ", "")[:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
