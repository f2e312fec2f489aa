for i in 1 2 3; do
for v in new old; do
  if [ $v = old ]; then export PYSDR_USE_DIAG_LIB=1; else unset PYSDR_USE_DIAG_LIB; fi
  python bench.py --no-demod --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 20 --warmup 4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v psd only: ms/step %.4f' % d['ms_per_step'], 'psd %.4f' % d['kernel_ms']['psd_call'])"
  python bench.py --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --steps 20 --warmup 4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v c3 GS/s %.1f' % (d['value'] / 1e3), 'ms/step %.4f' % d['ms_per_step'], 'psd %.4f' % d['kernel_ms']['psd_call'])"
done
done
