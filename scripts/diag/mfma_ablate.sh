# A/B builds of mixdec_mfma.hip (flag sets as arguments) timed with bench.py c1 and c4 (front-end ms; C4 also per kernel)
grep -q "MM_NO_CONS" pysdr_amd/csrc/mixdec_mfma.hip || { echo "the ablation branches are not in the sources: patch -p1 < scripts/experiments/ablation_switches.patch.txt first (and git checkout pysdr_amd/csrc afterwards)"; exit 1; }
export PYSDR_TUNING=1   # build.py reads PYSDR_*_FLAGS only under the tuning master switch (round 5)
cp pysdr_amd/libpysdr_hip.so /tmp/keep.so
for fl in "$@"; do
  PYSDR_MFMA_FLAGS="-DPYSDR_ABLATE $fl" python -m pysdr_amd.build --force > /tmp/build.log 2>&1 || { echo "build failed: $fl"; grep -i "error" /tmp/build.log | head -3; continue; }
  for w in ${WL:-c1 c4}; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-host-fed --no-verify --no-other-configs --steps 15 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-50s' % '$fl', '$w', 'GS/s %.1f' % (d['value'] / 1e3), 'front ms %.4f' % d['kernel_ms']['front'], 'mixdec frac %.3f' % d['roofline_mixdec']['frac'])
"
  done
done
cp /tmp/keep.so pysdr_amd/libpysdr_hip.so
