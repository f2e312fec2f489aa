#!/bin/bash
# C4 stereo: bench lines (overlapped, single-stream) + kernel averages of the single-stream form by rocprofv3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYSDR_TUNING=1
for rep in 1 2; do
for o in "" "--no-overlap"; do
  python3 bench.py --workload c4 $o --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
  python3 - "$o" <<'PY'
import json,sys
d=json.loads([l for l in open("/tmp/o.json") if l.startswith("{")][-1]); p=d["pilot_pll"]
print("%-13s %.1f GS/s %.4f ms %s verify %.2g join %s patched %s" % (sys.argv[1] or "overlapped", d["value"]/1e3, d["ms_per_step"], {k:(round(v,3) if v else v) for k,v in d["kernel_ms"].items()}, d["verify_worst_rel"], p["widest_join"]["phase_words_of_2^32"], p["patched_serially"]))
PY
done
done
O=gpurun_out/c4_kt; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --workload c4 --no-overlap --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 20 --warmup 4 "$@" > $O.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("   %-50s calls %4s avg %9.1f us  %5.1f %%  min %.1f max %.1f" % (r["Name"].replace("(anonymous namespace)::", "").replace("pysdr::", "").replace("void ", "")[:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"]), float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
