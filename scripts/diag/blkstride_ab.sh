# distance between the per-block accumulators (words): 64 (shipped) against 32 / 16, AF FIR + gains kernels on C1 / C3 (no PSD)
export PYSDR_TUNING=1   # build.py reads PYSDR_*_FLAGS only under the tuning master switch (round 5)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp pysdr_amd/libpysdr_hip.so /tmp/keep.so
for st in ${STRIDES:-64 32 16 64}; do
  PYSDR_STAGE2_FLAGS="-DPYSDR_BLK_STRIDE=$st" PYSDR_API_FLAGS="-DPYSDR_BLK_STRIDE=$st" python -m pysdr_amd.build --force > /tmp/build.log 2>&1 || { echo "build failed: $st"; grep error /tmp/build.log | head -3; continue; }
  for w in c1 "c3 --no-psd"; do
    O=gpurun_out/blk_kt; rm -rf $O; mkdir -p $O
    rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --workload $w --no-cpu-baseline --no-host-fed --no-other-configs --steps 10 --warmup 3 > $O/bench.json 2> $O/err.txt
    python3 - "$O" "$st" "$w" <<'PY'
import csv, glob, sys, json
ver = json.loads(open(sys.argv[1] + '/bench.json').read().strip().splitlines()[-1]).get('verify_worst_rel')
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'demod_fir' in r['Name'] or 'agc_scan' in r['Name']:
            print('stride %-3s %-12s %-22s avg us %7.1f  verify %s' % (sys.argv[2], sys.argv[3], r['Name'].split('::')[-1][:22], float(r['AverageNs']) / 1e3, ver))
PY
  done
done
cp /tmp/keep.so pysdr_amd/libpysdr_hip.so
