#!/bin/bash
# Run ON THE GPU BOX: the long-prototype multi-RX workloads as the round started (vector form, taps in LDS).
set -u
OUT=${1:-gpurun_out/r06_base}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT" && mkdir -p "$OUT"
B="--no-cpu-baseline --no-host-fed --no-other-configs --full-line"
for w in ft8tri test2rx; do
  python3 bench.py --workload $w $B > "$OUT/$w.json" 2> "$OUT/$w.err"
done
python3 bench.py --workload c2 --ntaps 1001 $B > "$OUT/c2_1001.json" 2> "$OUT/c2_1001.err"
python3 bench.py --workload c3 --ntaps 1001 --no-psd $B > "$OUT/c3_1001_nopsd.json" 2> "$OUT/c3_1001_nopsd.err"
python3 bench.py --workload rx6 --ntaps 1001 $B > "$OUT/rx6_1001.json" 2> "$OUT/rx6_1001.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ft8tri_kt" -- python3 bench.py --workload ft8tri $B --no-verify > "$OUT/ft8tri_kt.json" 2> "$OUT/ft8tri_kt.err"
find "$OUT" -name "*kernel_trace.csv" -delete
for f in ft8tri test2rx c2_1001 c3_1001_nopsd rx6_1001; do
  echo "$f: $(python3 -c "
import json
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1])
print('GS/s %.1f ms %.4f front %.4f ms frac %.3f job %.3f verify %s' % (d['value']/1e3, d['ms_per_step'], d['kernel_ms']['front'], d['roofline_mixdec']['frac'], d['roofline_job']['frac'], d.get('verify_worst_rel')))" 2>&1 | tail -1)"
done
