#!/bin/bash
# C1 (MFMA front end, 4 LDS images = 156 KB per CU) with 3 images (121 KB: an AF-FIR workgroup fits beside it) and the calls
# overlapped: does the AF stage of call k then run beside the mix + decimate of call k + 1, and what does it cost the front end?
export PYSDR_TUNING=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp pysdr_amd/libpysdr_hip.so /tmp/keep.so
for fl in "" "-DMM_C1_NBUF=3"; do
  PYSDR_MFMA_FLAGS="$fl" PYSDR_API_FLAGS="$fl" python -m pysdr_amd.build --force > /tmp/build.log 2>&1 || { echo "build failed: $fl"; grep -i "error" /tmp/build.log | head -3; continue; }
  for w in c1 c1synch; do
  for mode in --no-overlap --overlap-all; do
    python3 bench.py --workload $w $mode --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
    python3 - "$fl" $w $mode <<'PY'
import json,sys
d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
print("[%s] %-8s %-13s %7.1f GS/s %.3f ms  %s  verify %.2g hash %s" % (sys.argv[1], sys.argv[2], sys.argv[3], d['value']/1e3, d['ms_per_step'], {k:(round(v,3) if v else v) for k,v in d['kernel_ms'].items()}, d.get('verify_worst_rel',-1), d['tuning'].get('build_flags_hash')))
PY
  done; done
done
cp /tmp/keep.so pysdr_amd/libpysdr_hip.so
