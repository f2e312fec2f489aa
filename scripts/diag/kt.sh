# rocprofv3 kernel stats of bench.py for the workloads given as arguments (per-kernel average durations)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in "$@"; do
  O=gpurun_out/r04_kt/$w
  rm -rf $O && mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --workload $w --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 10 --warmup 3 > $O/bench.json 2> $O/err.txt
  python3 - "$O" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:12]:
        print('%-90s calls %4s avg us %9.1f  %5.1f %%' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
  find $O -name "*kernel_trace.csv" -delete
done
