#!/bin/bash
# PMC passes over scripts/diag/l2_writeback_test.bin (separate passes, no trace domains)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/l2wb; rm -rf $O; mkdir -p $O
B=scripts/diag/l2_writeback_test.bin
$B > $O/times.txt 2>&1
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/$c -- $B > $O/$c.log 2>&1
  python3 - "$O/$c" $c <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{sys.argv[2]:10s} {k:20s} n={len(v)} mean={sum(v)/len(v):.1f} KiB")
PY
done
cat $O/times.txt
