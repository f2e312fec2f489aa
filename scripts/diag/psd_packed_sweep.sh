# packed PSD pair: group size x streams (bench.py --no-demod: PSD only), then per-kernel times on one stream
for st in 1 2 3; do
  for grp in 384 448 512 576 640; do
    PYSDR_TUNING=1 PYSDR_PSD_STREAMS=$st PYSDR_PSD_GROUP=$grp python bench.py --no-demod --no-cpu-baseline --no-host-fed --no-other-configs --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streams $st group $grp:', 'psd ms %.4f' % d['kernel_ms']['psd_call'], 'frac %.3f' % d['roofline_psd']['frac'])"
  done
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_psd_kt; rm -rf $O; mkdir -p $O
PYSDR_TUNING=1 PYSDR_PSD_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --no-demod --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 5 --warmup 2 > $O/b.json 2> $O/err.txt
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r04_psd_kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'psd' in r['Name']: print(r['Name'][:60], r['Calls'], 'avg us', float(r['AverageNs'])/1e3)
PY
find $O -name "*kernel_trace.csv" -delete
