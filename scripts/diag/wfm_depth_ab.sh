#!/bin/bash
# pilot walk: mpx loads a group of 8 blocks ahead (default) against one block ahead (-DWFMX_DEPTH=1), same box, alternating
export PYSDR_TUNING=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp pysdr_amd/libpysdr_hip.so /tmp/keep.so
PYSDR_STAGE2_FLAGS="-DWFMX_DEPTH=1" python -m pysdr_amd.build --force > /tmp/build.log 2>&1 || { echo "build failed"; grep -i error /tmp/build.log | head; exit 1; }
cp pysdr_amd/libpysdr_hip.so /tmp/d1.so
for rep in 1 2 3 4; do
  for v in keep d1; do
    cp /tmp/$v.so pysdr_amd/libpysdr_hip.so
    for o in "" "--no-overlap"; do
      python3 bench.py --workload c4 $o --no-cpu-baseline --no-host-fed --no-other-configs --no-verify > /tmp/o.json 2>/tmp/o.err
      python3 - "$v" "$o" <<'PY'
import json,sys
d=json.loads([l for l in open("/tmp/o.json") if l.startswith("{")][-1])
print("%-5s %-13s %.1f GS/s %.4f ms front %.3f hash %s" % ("depth8" if sys.argv[1]=="keep" else "depth1", sys.argv[2] or "overlapped", d["value"]/1e3, d["ms_per_step"], d["kernel_ms"]["front"], d["tuning"].get("build_flags_hash")))
PY
    done
  done
done
cp /tmp/keep.so pysdr_amd/libpysdr_hip.so
