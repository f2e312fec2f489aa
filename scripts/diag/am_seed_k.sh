#!/bin/bash
# AM-Synch: more, shorter segments now that a warm-up is a linear solve -- K = 2048 / 4096 / 8192 segments per 4096-chunk call
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYSDR_TUNING=1
line() {
python3 - "$@" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
    p=d.get('carrier_pll') or {}
    print("%-40s %7.1f GS/s %.3f ms  %s  job %.3f verify %.2g  seg %s patched %s join %s lin %s" % (" ".join(sys.argv[1:]), d['value']/1e3, d['ms_per_step'], {k:(round(v,3) if v else v) for k,v in d['kernel_ms'].items()}, d['roofline_job']['frac'], d.get('verify_worst_rel',-1), p.get('segments'), p.get('patched_serially'), (p.get('widest_join') or {}).get('phase_words_of_2^32'), p.get('linear_starts')))
except Exception as e:
    print("FAILED", sys.argv[1:], e, open('/tmp/o.err').read()[-600:])
PY
}
for rep in 1 2; do
for k in 2048 4096 8192; do
  for s in 1 0; do
    PYSDR_AM_SEED=$s PYSDR_AM_PLL=16,5,4,$k,512 python3 bench.py --workload c1synch --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
    line K=$k seed=$s overlapped
    PYSDR_AM_SEED=$s PYSDR_AM_PLL=16,5,4,$k,512 python3 bench.py --workload c1synch --no-overlap --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
    line K=$k seed=$s single-stream
  done
done
done
