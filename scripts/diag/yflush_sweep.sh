for yf in 0 1 2 4 8; do
  for w in c3 c2; do
  if [ $yf = 0 ]; then unset PYSDR_TUNING PYSDR_MIXDEC_YFLUSH; else export PYSDR_TUNING=1 PYSDR_MIXDEC_YFLUSH=$yf; fi
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-host-fed --no-other-configs --no-psd --steps 15 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('yflush $yf', '$w', 'GS/s %.1f' % (d['value'] / 1e3), 'ms/step %.4f' % d['ms_per_step'], 'front ms %.4f' % d['kernel_ms']['front'], 'mixdec frac %.3f' % d['roofline_mixdec']['frac'])
"
  done
done
