# per-kernel timeline of one bench step (rocprofv3 --kernel-trace): where the gaps between the kernels of a step are
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=${1:-c1}
O=gpurun_out/timeline_$W; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 bench.py --workload $W --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 6 --warmup 2 > $O/b.json 2> $O/err.txt
python3 - "$O" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-60:]
t0 = int(rows[0]['Start_Timestamp'])
prev_end = None
for r in rows[-28:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print('%9.1f us  +%6.1f gap  dur %8.1f  %s' % ((s - t0) / 1e3, gap, (e - s) / 1e3, r['Kernel_Name'].split('(')[0][-60:]))
    prev_end = e
PY
rm -rf $O
