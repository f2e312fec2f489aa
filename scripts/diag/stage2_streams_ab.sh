cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYSDR_TUNING=1   # build.py reads PYSDR_*_FLAGS only under the tuning master switch (round 5)
cp pysdr_amd/libpysdr_hip.so /tmp/keep.so
for fl in "" "-DS2X_PLAIN_STREAMS" "" "-DS2X_PLAIN_STREAMS"; do
  PYSDR_STAGE2_FLAGS="$fl" python -m pysdr_amd.build --force > /tmp/build.log 2>&1 || { echo "build failed: $fl"; continue; }
  for w in c3 c4; do
    O=gpurun_out/s2_kt/$w; rm -rf $O; mkdir -p $O
    rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --workload $w --no-psd --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 10 --warmup 3 > $O/bench.json 2> $O/err.txt
    python3 - "$O" "$fl" "$w" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'apply_kernel' in r['Name'] or 'demod_fir' in r['Name']:
            print('%-22s %s %-28s avg us %7.1f' % (sys.argv[2], sys.argv[3], r['Name'].split('::')[-1][:28], float(r['AverageNs']) / 1e3))
PY
  done
done
cp /tmp/keep.so pysdr_amd/libpysdr_hip.so
