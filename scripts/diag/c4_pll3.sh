# staged coarse warm-up of the pilot loop: PYSDR_WFM_PLL fields 8, 9 = time constants at coarse_sweeps / coarse_sweeps - 1 in front of
# the exact tail (the rest of the warm-up at coarse_sweeps - 2): C4 step, joins patched, verification against the oracle
for cfg in "" "20,13,5,3,1536,2048,5,4,4" "20,13,5,3,1536,2048,5,2,3" "20,13,5,3,1536,2048,5,3,3" "20,13,5,3,1536,2048,5,2,6" "20,13,5,3,1536,2048,5,4,0" "20,13,5,3,1536,2048,5,2,2" ""; do
  echo "== PYSDR_WFM_PLL='$cfg'"
  if [ -n "$cfg" ]; then export PYSDR_TUNING=1 PYSDR_WFM_PLL=$cfg; else unset PYSDR_TUNING PYSDR_WFM_PLL; fi
  python bench.py --workload c4 --no-cpu-baseline --no-host-fed --no-other-configs --steps 20 --warmup 4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('GS/s %.1f' % (d['value'] / 1e3), 'ms %.4f' % d['ms_per_step'], 'job %.3f' % d['roofline_job']['frac'], 'verify %.2e' % d.get('verify_worst_rel', -1), d.get('pilot_pll'))"
done
