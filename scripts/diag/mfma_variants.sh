# A/B of compile-time variants of mixdec_mfma.hip ON the GPU box: for every flag set given as an argument ("" = shipped)
export PYSDR_TUNING=1   # build.py reads PYSDR_*_FLAGS only under the tuning master switch (round 5)
# rebuild the diagnostic library, print the stamp summary of workgroup 3, and the C1 / C4 kernel times from bench.py
# (the diagnostic library serves both: without PYSDR_DEBUG_FLAGS it skips nothing).
#   bash scripts/diag/mfma_variants.sh "" "-DMM_NO_PK" "-DMM_PROD_PRIO=3"
for fl in "$@"; do
  echo "=== flags: '$fl'"
  PYSDR_MFMA_FLAGS="$fl" python -m pysdr_amd.build --diag > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; continue; }
  PYSDR_USE_DIAG_LIB=1 PYSDR_DEBUG_FLAGS=256 timeout 300 python scripts/diag/mfma_stamps.py 2>&1 | sed -n 1,9p
  for w in c1 c4; do
    PYSDR_USE_DIAG_LIB=1 timeout 300 python bench.py --workload $w --no-cpu-baseline --no-host-fed --steps 15 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', 'GS/s %.1f' % (d['value'] / 1e3), 'ms/step %.4f' % d['ms_per_step'], 'front ms %.4f' % d['kernel_ms']['front'], 'mixdec frac %.3f' % d['roofline_mixdec']['frac'])
"
  done
done
