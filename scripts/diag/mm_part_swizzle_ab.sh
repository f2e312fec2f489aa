#!/bin/bash
# matrix-core front end: the partial-tile exchange XOR-swizzled (shipped) against the plain [col][4 rows] layout
# (-DMM_PART_PLAIN), same box, alternating; then SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of both for C1 and C4
export PYSDR_TUNING=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp pysdr_amd/libpysdr_hip.so /tmp/keep.so
PYSDR_MFMA_FLAGS="-DMM_PART_PLAIN" python -m pysdr_amd.build --force > /tmp/build.log 2>&1 || { echo "build failed"; grep -i error /tmp/build.log | head; exit 1; }
cp pysdr_amd/libpysdr_hip.so /tmp/plain.so
for rep in 1 2 3 4; do
  for v in keep plain; do
    cp /tmp/$v.so pysdr_amd/libpysdr_hip.so
    for w in c1 c4mono; do
      python3 bench.py --workload $w --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
      python3 - "$v" "$w" <<'PY'
import json,sys
d=json.loads([l for l in open("/tmp/o.json") if l.startswith("{")][-1])
print("%-8s %-7s %.1f GS/s %.4f ms front %.4f verify %.2g hash %s" % ("swizzle" if sys.argv[1]=="keep" else "plain", sys.argv[2], d["value"]/1e3, d["ms_per_step"], d["kernel_ms"]["front"], d["verify_worst_rel"], d["tuning"].get("build_flags_hash")))
PY
    done
  done
done
for v in keep plain; do
  cp /tmp/$v.so pysdr_amd/libpysdr_hip.so
  for w in c1 c4mono; do
    O=gpurun_out/mm_part_pmc_${v}_$w; rm -rf $O; mkdir -p $O
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O -- python3 bench.py --workload $w --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 3 --warmup 1 > $O.log 2>&1
    python3 - $O "$v" "$w" <<'PY'
import csv, glob, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "mixdec_mfma" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k,c in acc.items():
    print("%-8s %-7s conflict %.3e  lds active %.3e  (%.0f %%)  busy %.3e" % ("swizzle" if sys.argv[2]=="keep" else "plain", sys.argv[3], c["SQ_LDS_BANK_CONFLICT"], c["SQ_LDS_IDX_ACTIVE"], 100*c["SQ_LDS_BANK_CONFLICT"]/max(1,c["SQ_LDS_IDX_ACTIVE"]), c["SQ_BUSY_CYCLES"]))
PY
  done
done
cp /tmp/keep.so pysdr_amd/libpysdr_hip.so
