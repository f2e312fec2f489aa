# pilot-loop warm-up settings judged by the WIDEST JOIN they leave (pysdr_pll_join_margin; tolerance 512 words / 1e-9 rad per sample):
#   PYSDR_WFM_PLL = W tau, W tau (mean at hand), exact tail tau, coarse sweeps, max segments, min samples, exact cap, tau at coarse sweeps, tau at one fewer, tail cap
for cfg in "" "20,13,5,3,1536,2048,5,4,4,0" "20,13,5,3,1536,2048,5,3,5,0" "20,13,5,3,1536,2048,5,0,0,4" "20,13,5,3,1536,2048,5,4,4,4" "20,13,4,3,1536,2048,5,0,0,0" "20,12,5,3,1536,2048,5,0,0,0" "20,14,5,3,1536,2048,5,0,0,0" ""; do
  echo "== PYSDR_WFM_PLL='$cfg'"
  if [ -n "$cfg" ]; then export PYSDR_TUNING=1 PYSDR_WFM_PLL=$cfg; else unset PYSDR_TUNING PYSDR_WFM_PLL; fi
  python bench.py --workload c4 --no-cpu-baseline --no-host-fed --no-other-configs --steps 20 --warmup 4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
p = d.get('pilot_pll')
print('GS/s %.1f' % (d['value'] / 1e3), 'ms %.4f' % d['ms_per_step'], 'verify %.2e' % d.get('verify_worst_rel', -1), 'segments', p['segments'], 'patched', p['patched_serially'], 'widest join', p['widest_join']['phase_words_of_2^32'], 'words', '%.2e' % p['widest_join']['integrator_rad_per_sample'], 'rad/sample')"
done
