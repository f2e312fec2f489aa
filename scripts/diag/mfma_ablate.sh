# ablation builds of mixdec_mfma.hip timed with bench.py c1 (front-end kernel ms); arguments = flag sets
cp pysdr_amd/libpysdr_hip.so /tmp/keep.so
for fl in "$@"; do
  PYSDR_MFMA_FLAGS="$fl" python -m pysdr_amd.build --force > /tmp/build.log 2>&1 || { echo "build failed: $fl"; tail -5 /tmp/build.log; continue; }
  timeout 300 python bench.py --workload c1 --no-cpu-baseline --no-host-fed --steps 15 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s' % '$fl', 'front ms %.4f' % d['kernel_ms']['front'], 'mixdec frac %.3f' % d['roofline_mixdec']['frac'])
"
done
cp /tmp/keep.so pysdr_amd/libpysdr_hip.so
