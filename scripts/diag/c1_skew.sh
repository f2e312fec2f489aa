#!/bin/bash
# A/B of the skewed tap schedule (mixdec.hip) on the configurations whose DOWN is a multiple of 32:
#   bench lines with PYSDR_MIXDEC_SKEW=0 / 1, and SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE /
#   SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES of the mixdec kernel in both (PMC pass of its own).
# usage (on the GPU box): bash scripts/diag/c1_skew.sh gpurun_out/c1skew [workload]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=${1:-gpurun_out/c1skew}; W=${2:-c1}
rm -rf $O; mkdir -p $O
for k in 0 1; do
  PYSDR_MIXDEC_SKEW=$k python3 bench.py --workload $W --no-cpu-baseline --no-host-fed > $O/bench_skew$k.json 2> $O/bench_skew$k.err
  PYSDR_MIXDEC_SKEW=$k rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc$k -- python3 bench.py --workload $W --no-cpu-baseline --no-host-fed --steps 3 --warmup 1 > $O/pmc$k.json 2> $O/pmc$k.err
  python3 - $O $k <<'PY'
import csv, glob, json, sys, collections
O, k = sys.argv[1], sys.argv[2]
d = json.loads(open(f"{O}/bench_skew{k}.json").read().strip().splitlines()[-1])
print(f"skew={k}: {d['value']/1e3:.1f} GS/s, front {d['kernel_ms']['front']:.4f} ms, roofline_mixdec.frac {d['roofline_mixdec']['frac']:.3f}")
acc = collections.defaultdict(list)
for f in glob.glob(f"{O}/pmc{k}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "mixdec" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {c: sum(v) / len(v) for c, v in acc.items()}
print("   mixdec PMC per launch:", {c: round(v) for c, v in sorted(out.items())})
json.dump(dict(skew=int(k), workload=d["config"]["workload"], value_MSps=d["value"], front_ms=d["kernel_ms"]["front"],
               roofline_mixdec=d["roofline_mixdec"], pmc_per_launch=out, source_sha256=d["source_sha256"]),
          open(f"{O}/summary_skew{k}.json", "w"), indent=1)
PY
done
find $O -name "*.csv" -size +2M -delete
