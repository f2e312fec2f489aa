#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fir_pmc; rm -rf $O; mkdir -p $O
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- python3 bench.py --workload ${WL:-c2} --no-cpu-baseline --steps 3 --warmup 1 > $O/$name.log 2>&1
  python3 - "$O/$name" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("demod_fir", "agc_scan", "apply_kernel"):
            if key in k: acc[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:10s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
}
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES
run p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
