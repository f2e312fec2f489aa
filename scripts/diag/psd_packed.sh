# the PSD pair with the 24-bit packed intermediate against the float2 form: parity tests, then C3 and PSD-only timings
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spectrum or psd or c3_four or waterfall" 2>&1 | tail -2
for pk in 1 0; do
  for grp in 448 576; do
    PYSDR_TUNING=1 PYSDR_PSD_PACKED=$pk PYSDR_PSD_GROUP=$grp python bench.py --no-cpu-baseline --no-host-fed --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('packed $pk group $grp:', 'GS/s %.1f' % (d['value'] / 1e3), 'ms %.4f' % d['ms_per_step'], 'psd ms %.4f' % d['kernel_ms']['psd_call'], 'psd frac %.3f' % d['roofline_psd']['frac'], 'verify %.2e' % d['verify_worst_rel'])"
  done
done
