# pilot-loop settings after the sweeps lost a third of their instructions (round 4): segments / exact tail / cap again.
#   PYSDR_WFM_PLL = W tau (no mean), W tau (mean at hand), exact tail tau, coarse sweeps, max segments, min samples per segment, exact cap
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pll.py -x -q -m gpu -k "c4 or wbfm or wfm" 2>&1 | tail -1
for cfg in "" "20,13,5,3,3072,2048,5" "20,13,5,3,4096,1024,5" "20,13,5,3,1536,2048,5" "20,13,4,3,2048,2048,5" "20,13,5,3,2048,2048,6" "20,12,5,3,2048,2048,5" ""; do
  echo "== PYSDR_WFM_PLL='$cfg'"
  if [ -n "$cfg" ]; then export PYSDR_TUNING=1 PYSDR_WFM_PLL=$cfg; else unset PYSDR_TUNING PYSDR_WFM_PLL; fi
  python bench.py --workload c4 --no-cpu-baseline --no-host-fed --no-other-configs --steps 20 --warmup 4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('GS/s %.1f' % (d['value'] / 1e3), 'ms %.4f' % d['ms_per_step'], 'job %.3f' % d['roofline_job']['frac'], 'verify %.2e' % d.get('verify_worst_rel', -1), d.get('pilot_pll'))"
done
