#!/bin/bash
# wave priority of the PLL segment walks (s_setprio PLLX_PRIO at kernel start) with the calls overlapped: 0 / 1 / 3
export PYSDR_TUNING=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp pysdr_amd/libpysdr_hip.so /tmp/keep.so
for pr in 3 0 1; do
  PYSDR_STAGE2_FLAGS="-DPLLX_PRIO=$pr" python -m pysdr_amd.build --force > /tmp/build.log 2>&1 || { echo "build failed: $pr"; grep -i "error" /tmp/build.log | head -3; continue; }
  for w in c1synch c4; do
    python3 bench.py --workload $w --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
    python3 - $pr $w <<'PY'
import json,sys
d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
print("prio %s %-8s %7.1f GS/s %.3f ms  %s  verify %.2g" % (sys.argv[1], sys.argv[2], d['value']/1e3, d['ms_per_step'], {k:(round(v,3) if v else v) for k,v in d['kernel_ms'].items()}, d.get('verify_worst_rel',-1)))
PY
  done
done
cp /tmp/keep.so pysdr_amd/libpysdr_hip.so
