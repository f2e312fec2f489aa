#!/bin/bash
# the Newton seeds' affine maps in fp64 (default) against fp32 (-DSEEDX_FLOAT): time beside the front end, and the joins they leave
export PYSDR_TUNING=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp pysdr_amd/libpysdr_hip.so /tmp/keep.so
for fl in "${@:-}"; do
  PYSDR_SEED_FLAGS="$fl" python -m pysdr_amd.build --force > /tmp/build.log 2>&1 || { echo "build failed: $fl"; grep -i "error" /tmp/build.log | head -3; continue; }
  echo "== [$fl]"
  for ov in "" "--no-overlap"; do
    python3 bench.py --workload c4 $ov --no-cpu-baseline --no-host-fed --no-other-configs > /tmp/o.json 2>/tmp/o.err
    python3 - "$ov" <<'PY'
import json,sys
d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
print("%-13s %7.1f GS/s %.3f ms  %s  verify %.2g  %s" % (sys.argv[1] or "overlapped", d['value']/1e3, d['ms_per_step'], {k:(round(v,3) if v else v) for k,v in d['kernel_ms'].items()}, d.get('verify_worst_rel',-1), json.dumps(d['pilot_pll'])[:170]))
PY
  done
  scripts/diag/overlap_trace.sh c4 2>&1 | grep "seed_reduce\|mixdec_mfma" | head -4
done
cp /tmp/keep.so pysdr_amd/libpysdr_hip.so
