# producer waves per mixdec workgroup: A/B over the BASELINE configurations (bench lines only)
for np in ${PRODS:-0 1 2 4}; do
 for w in ${WLS:-c1 c2 c3 c4}; do
  PYSDR_MIXDEC_PROD=$np python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-host-fed 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip()); r=j.get('roofline_mixdec') or {}
print('nprod=$np $w', round(j['value']/1e3,1),'GS/s', round(j['ms_per_step'],3),'ms; mixdec', round(r.get('avg_launch_ms',0),4), 'ms frac', round(r.get('frac',0),3))
"
 done
done
