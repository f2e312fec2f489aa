#!/bin/bash
# how busy the vector pipe is under the loop walks (single-stream): SQ counters of am_pll_seg_kernel (c1synch) and wfm_pll_seg_kernel (c4)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in c1synch c4; do
  O=gpurun_out/pll_walk_pmc_$w; rm -rf $O; mkdir -p $O
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O -- python3 bench.py --workload $w --no-overlap --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 3 --warmup 1 > $O.log 2>&1
  python3 - $O $w <<'PY'
import csv, glob, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("pysdr::(anonymous namespace)::","").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k,r["Counter_Name"])] += 1
for k,c in sorted(acc.items()):
    if not any(x in k for x in ("pll_seg","demod_fir","am_phase","seed_reduce","mixdec_mfma")): continue
    L=max(1,n[(k,"SQ_WAVE_CYCLES")])
    print("%-8s %-44s launches %3d  VALU insts/launch %.3e  active-VALU cycles %.3e  wave cycles %.3e  busy %.3e  gui_active %.3e  -> VALU active / (4 x busy) = %.2f" % (sys.argv[2], k[:44], L, c["SQ_INSTS_VALU"]/L, c["SQ_ACTIVE_INST_VALU"]/L, c["SQ_WAVE_CYCLES"]/L, c["SQ_BUSY_CYCLES"]/L, c["GRBM_GUI_ACTIVE"]/L, c["SQ_ACTIVE_INST_VALU"]/max(1,4*c["SQ_BUSY_CYCLES"])))
PY
done
