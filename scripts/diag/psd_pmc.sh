#!/bin/bash
# PMC passes over the PSD test harness (separate passes, no trace domains)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/psd_pmc; rm -rf $O; mkdir -p $O
B=scripts/experiments/psd_frame_test.bin
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- $B 1024 2 > $O/$name.log 2>&1
  python3 - "$O/$name" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "frame" if "psd_frame" in k else ("cols" if "psd_cols" in k else ("rows" if "psd_rows" in k else None))
        if k: acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:6s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
}
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run p3 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC
run p4 FETCH_SIZE
run p5 WRITE_SIZE
run p6 GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
tail -3 $O/p1.log
