# bench.py --workload c4 under PYSDR_WFM_PLL = "taus,taus_fast,taus_exact,coarse_sweeps,kmax,tmin" settings given as arguments
for cfg in "$@"; do
PYSDR_TUNING=1 PYSDR_WFM_PLL=$cfg python bench.py --workload c4 --no-cpu-baseline --no-host-fed --steps 12 --warmup 3 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['value']/1e3,1), 'GS/s front', round(d['kernel_ms']['front'],3), d['pilot_pll'])"
done
