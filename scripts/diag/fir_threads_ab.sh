# AF FIR: threads per workgroup (x 8 outputs = tile) 128 / 256 / 512, kernel averages on C1 / C2 / C3 (no PSD) / C4
export PYSDR_TUNING=1   # build.py reads PYSDR_*_FLAGS only under the tuning master switch (round 5)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp pysdr_amd/libpysdr_hip.so /tmp/keep.so
for t in 256 128 512 256; do
  PYSDR_STAGE2_FLAGS="-DFIRX_THREADS=$t" python -m pysdr_amd.build --force > /tmp/build.log 2>&1 || { echo "build failed: $t"; grep error /tmp/build.log | head -3; continue; }
  for w in c1 c2 "c3 --no-psd" c4; do
    O=gpurun_out/fir_kt; rm -rf $O; mkdir -p $O
    rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --workload $w --no-cpu-baseline --no-host-fed --no-other-configs --steps 10 --warmup 3 > $O/bench.json 2> $O/err.txt
    python3 - "$O" "$t" "$w" <<'PY'
import csv, glob, sys, json
ver = json.loads(open(sys.argv[1] + '/bench.json').read().strip().splitlines()[-1]).get('verify_worst_rel')
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'demod_fir' in r['Name']:
            print('threads %-4s %-12s %-26s avg us %7.1f  verify %s' % (sys.argv[2], sys.argv[3], r['Name'].split('::')[-1][:26], float(r['AverageNs']) / 1e3, ver))
PY
  done
done
cp /tmp/keep.so pysdr_amd/libpysdr_hip.so
