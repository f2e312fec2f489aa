# soak: every single-GPU configuration, long timed loops, the last step of each run verified against the oracle; three rounds
for rep in 1 2 3; do
  for w in c1 c2 c3 c4 c4mono rx6 c1synch ft8tri test2rx; do
    python3 bench.py --workload $w --no-cpu-baseline --no-host-fed --no-other-configs --full-line --steps 150 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $rep %-7s GS/s %7.1f  verify %.2e  ranks verified %s  pll %s' % ('$w', d['value'] / 1e3, d.get('verify_worst_rel', -1), d.get('verified_ranks'), (d.get('pilot_pll') or {}).get('patched_serially')))"
  done
done
