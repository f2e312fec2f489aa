# where agc_scan_kernel's time goes: builds of stage2.hip with parts switched off (results WRONG), kernel averages from rocprofv3
grep -q "MM_NO_CONS" pysdr_amd/csrc/mixdec_mfma.hip || { echo "the ablation branches are not in the sources: patch -p1 < scripts/experiments/ablation_switches.patch.txt first (and git checkout pysdr_amd/csrc afterwards)"; exit 1; }
export PYSDR_TUNING=1   # build.py reads PYSDR_*_FLAGS only under the tuning master switch (round 5)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp pysdr_amd/libpysdr_hip.so /tmp/keep.so
for fl in "" "-DAGCX_NO_CHAIN" "-DAGCX_NO_ZERO" "-DAGCX_NO_LOAD" "-DAGCX_NO_CHAIN -DAGCX_NO_ZERO -DAGCX_NO_LOAD"; do
  PYSDR_STAGE2_FLAGS="-DPYSDR_ABLATE $fl" python -m pysdr_amd.build --force > /tmp/build.log 2>&1 || { echo "build failed: $fl"; grep error /tmp/build.log | head -3; continue; }
  for w in ${WL:-c1 c2}; do
    O=gpurun_out/agc_kt/$w; rm -rf $O; mkdir -p $O
    rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --workload $w --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 10 --warmup 3 > $O/bench.json 2> $O/err.txt
    python3 - "$O" "$fl" "$w" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'agc_scan' in r['Name'] or 'epilogue' in r['Name'] or 'hist_roll' in r['Name']:
            print('%-50s %s %-20s avg us %7.1f' % (sys.argv[2], sys.argv[3], r['Name'].split('::')[-1][:20], float(r['AverageNs']) / 1e3))
PY
  done
done
cp /tmp/keep.so pysdr_amd/libpysdr_hip.so
