#!/bin/bash
# A/B of the two-stream call pipeline (pysdr_set_overlap): every single-GPU workload with and without it, same box.
#   scripts/diag/overlap_ab.sh [workload ...]
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/overlap_ab
WL=("$@"); [ ${#WL[@]} -eq 0 ] && WL=(c1 c1synch c2 rx6 c4mono c4 c3)
for w in "${WL[@]}"; do
  for mode in on off; do
    flag=""; [ $mode = off ] && flag="--no-overlap"
    python3 bench.py --workload $w $flag --no-cpu-baseline --no-host-fed --no-other-configs > gpurun_out/overlap_ab/${w}_$mode.json 2> gpurun_out/overlap_ab/${w}_$mode.err
    python3 - gpurun_out/overlap_ab/${w}_$mode.json $w $mode <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d["kernel_ms"]
    print("%-8s overlap %-3s  %8.1f GS/s  %.3f ms/step  front %.3f stage2 %s  job frac %.3f  verify %.2g (%s ranks)  %s" % (
        sys.argv[2], sys.argv[3], d["value"] / 1e3, d["ms_per_step"], k["front"] or 0, "%.3f" % k["stage2"] if k["stage2"] else "-",
        d["roofline_job"]["frac"], d.get("verify_worst_rel", -1), d.get("verified_ranks"),
        json.dumps(d.get("pilot_pll") or d.get("carrier_pll") or "")[:110]), flush=True)
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:], flush=True)
PY
  done
done
