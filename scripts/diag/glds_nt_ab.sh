# LDS-DMA loads of the two mix + decimate kernels: nontemporal against plain, all four configurations (+ demod-only C3), two rounds
export PYSDR_TUNING=1   # build.py reads PYSDR_*_FLAGS only under the tuning master switch (round 5)
cp pysdr_amd/libpysdr_hip.so /tmp/keep.so
for rep in 1 2; do
for fl in "plain" "nt"; do
  if [ $fl = nt ]; then export PYSDR_MIXDEC_FLAGS="-DPYSDR_GLDS_NT" PYSDR_MFMA_FLAGS="-DPYSDR_GLDS_NT"; else export PYSDR_MIXDEC_FLAGS="" PYSDR_MFMA_FLAGS=""; fi
  python -m pysdr_amd.build --force > /tmp/build.log 2>&1 || { echo "build failed: $fl"; grep -i "error" /tmp/build.log | head -3; continue; }
  for w in c1 c2 c3 "c3 --no-psd" c4; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-host-fed --no-other-configs --steps 15 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-6s' % '$fl', '%-12s' % '$w', 'GS/s %.1f' % (d['value'] / 1e3), 'ms/step %.4f' % d['ms_per_step'], 'front ms %.4f' % d['kernel_ms']['front'], 'mixdec frac %.3f' % d['roofline_mixdec']['frac'], 'verify %.1e' % d.get('verify_worst_rel', -1))
"
  done
done
done
cp /tmp/keep.so pysdr_amd/libpysdr_hip.so
