# A/B of compile-time variants of psdfft.hip ON the GPU box: for every flag set given as an argument ("" = shipped) rebuild
grep -q "MM_NO_CONS" pysdr_amd/csrc/mixdec_mfma.hip || { echo "the ablation branches are not in the sources: patch -p1 < scripts/experiments/ablation_switches.patch.txt first (and git checkout pysdr_amd/csrc afterwards)"; exit 1; }
export PYSDR_TUNING=1   # build.py reads PYSDR_*_FLAGS only under the tuning master switch (round 5)
# the library and time the PSD alone (bench.py --no-demod) and inside C3.
#   bash scripts/diag/psd_variants.sh "" "-DPSDX_NO_SCALE" "-DPSDX_NO_LO"
BASE="-fno-slp-vectorize -fno-signed-zeros"
for fl in "$@"; do
  echo "=== flags: '$fl'"
  PYSDR_PSD_FLAGS="$BASE -DPYSDR_ABLATE $fl" python -m pysdr_amd.build --force > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; continue; }
  for mode in "--no-demod" ""; do
    timeout 300 python bench.py $mode --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 20 --warmup 4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('mode \"$mode\":', 'GS/s %.1f' % (d['value'] / 1e3), 'ms/step %.4f' % d['ms_per_step'], 'psd ms %.4f' % d['kernel_ms']['psd_call'], 'psd frac %.3f' % d['roofline_psd']['frac'])
"
  done
done
PYSDR_PSD_FLAGS="$BASE" python -m pysdr_amd.build --force > /tmp/build.log 2>&1
