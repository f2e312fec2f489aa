for cfg in "1 1024" "2 512" "4 256" "3 320"; do set -- $cfg
 for w in c1 c4 c2; do
  echo "== wgs=$1 threads=$2 $w"
  PYSDR_TUNING=1 PYSDR_MIXDEC_WGS=$1 timeout 200 python bench.py --workload $w --threads $2 --steps 10 --warmup 3 --no-cpu-baseline --no-host-fed 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
  j=json.loads(l); print(j['value'], j['ms_per_step'], j.get('roofline_mixdec'))
except Exception as e: print('ERR', l[-300:])
"
 done
done
