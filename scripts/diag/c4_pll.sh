# pilot-loop settings: parity of the WFM tests (worst relerr printed by -rA is not available: pass/fail) and the C4 step
for cfg in "" "20,13,5,3,2048,2048,0" "20,13,5,3,2048,2048,4" "20,13,5,3,2048,2048,6"; do
  echo "== PYSDR_WFM_PLL='$cfg'"
  if [ -n "$cfg" ]; then export PYSDR_TUNING=1 PYSDR_WFM_PLL=$cfg; else unset PYSDR_TUNING PYSDR_WFM_PLL; fi
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pll.py -x -q -m gpu -k "c4 or wbfm or wfm" 2>&1 | tail -1
  python bench.py --workload c4 --no-cpu-baseline --no-host-fed 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('GS/s %.1f' % (d['value'] / 1e3), 'ms %.4f' % d['ms_per_step'], 'job %.3f' % d['roofline_job']['frac'], 'verify %.2e' % d.get('verify_worst_rel', -1), d.get('pilot_pll'))"
done
