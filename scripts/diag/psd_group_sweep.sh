for grp in 320 384 448 512 576 640 448; do
  PYSDR_TUNING=1 PYSDR_PSD_GROUP=$grp python bench.py --no-cpu-baseline --no-host-fed --no-other-configs --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('group $grp:', 'GS/s %.1f' % (d['value'] / 1e3), 'ms %.4f' % d['ms_per_step'], 'psd ms %.4f' % d['kernel_ms']['psd_call'], 'frac %.3f' % d['roofline_psd']['frac'])"
done
