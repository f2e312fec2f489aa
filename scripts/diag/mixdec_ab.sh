# A/B on ONE box: the library in the diag slot (PYSDR_USE_DIAG_LIB=1, here: a build of an older tree) against the
# current one, alternating, REPS times per configuration (boxes of the pool differ by +-3-5 %)
for rep in $(seq 1 ${REPS:-3}); do
 for w in ${WLS:-c1 c2 c3 c4}; do
  for v in old new ${EXTRA}; do
   case $v in old) e="PYSDR_USE_DIAG_LIB=1";; new) e="PYSDR_X=0";; *) e="$v";; esac
   env $e python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-host-fed 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip()); r=j.get('roofline_mixdec') or {}
print('$v $w', round(j['value']/1e3,1),'GS/s', round(j['ms_per_step'],3),'ms; mixdec', round(r.get('avg_launch_ms',0),4), 'ms frac', round(r.get('frac',0),3))
"
  done
 done
done
