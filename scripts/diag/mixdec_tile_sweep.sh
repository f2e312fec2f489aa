# the vector mix + decimate kernel after its copies went nontemporal: tile size and workgroups per CU again (C2, C3 without PSD)
for cfg in "0 1 1024" "49152 1 1024" "40960 1 1024" "32768 1 1024" "32768 2 512" "24576 2 512" "0 1 1024"; do set -- $cfg
 for w in c2 "c3 --no-psd"; do
  PYSDR_TUNING=1 PYSDR_MIXDEC_WGS=$2 timeout 200 python bench.py --workload $w --tile-bytes $1 --threads $3 --steps 16 --warmup 4 --no-cpu-baseline --no-host-fed --no-other-configs 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip())
print('tile %6s wgs %s threads %4s %-12s' % ('$1', '$2', '$3', '$w'), 'GS/s %.1f' % (j['value'] / 1e3), 'front ms %.4f' % j['kernel_ms']['front'], 'mixdec frac %.3f' % j['roofline_mixdec']['frac'], 'verify %.1e' % j.get('verify_worst_rel', -1))
"
 done
done
