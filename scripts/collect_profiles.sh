#!/bin/bash
# Run ON THE GPU BOX (via gpurun):  bash scripts/collect_profiles.sh gpurun_out/prof_r2 [quick]
# Collects what profiles/ is built from, for every BASELINE configuration that fits one GPU:
#   <cfg>/bench.json                 the plain bench line (events only, no profiler attached)
#   <cfg>/kt/...kernel_stats.csv     rocprofv3 --kernel-trace --stats of the same command
#   c3/pmc_fetch, c3/pmc_write, c2/... HBM-traffic PMC counters in their OWN passes (never with a trace)
set -u
OUT=${1:-gpurun_out/prof}
QUICK=${2:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT" && mkdir -p "$OUT"
TAG=${PROFILE_TAG:-r06}
declare -a CFG_NAMES=() CFG_ARGS=()
run_cfg() {  # name, pmc(0/1), bench args...
  local name=$1 pmc=$2; shift 2
  mkdir -p "$OUT/$name"
  CFG_NAMES+=("$name"); CFG_ARGS+=("$* ${PYSDR_PSD_STREAMS:+@1stream}")
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name/kt" -- python3 bench.py "$@" --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --full-line > "$OUT/$name/bench_kt.json" 2> "$OUT/$name/kt.err"
  if [ "$pmc" = 1 ]; then
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$name/pmc_fetch" -- python3 bench.py "$@" --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --full-line --steps 3 --warmup 1 > "$OUT/$name/pmc_fetch.json" 2> "$OUT/$name/pmc_fetch.err"
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$name/pmc_write" -- python3 bench.py "$@" --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --full-line --steps 3 --warmup 1 > "$OUT/$name/pmc_write.json" 2> "$OUT/$name/pmc_write.err"
    # wave-level counters of the same kernels (LDS bank conflicts, VALU / LDS activity, parked cycles)
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d "$OUT/$name/pmc_sq" -- python3 bench.py "$@" --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --full-line --steps 3 --warmup 1 > "$OUT/$name/pmc_sq.json" 2> "$OUT/$name/pmc_sq.err"
    # the matrix pipe (mixdec_mfma.hip: C1, C4)
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/$name/pmc_mfma" -- python3 bench.py "$@" --no-cpu-baseline --no-host-fed --no-other-configs --no-verify --full-line --steps 3 --warmup 1 > "$OUT/$name/pmc_mfma.json" 2> "$OUT/$name/pmc_mfma.err"
  fi
  # keep only the small summaries (the per-dispatch traces are tens of MB)
  find "$OUT/$name" -name "*kernel_trace.csv" -delete
  find "$OUT/$name" -name "*counter_collection.csv" -size +8M -delete
}
run_cfg c3 1
# the PSD's two kernels one after the other (the default deals half-groups over two streams, so the per-kernel
# durations of the trace above overlap): un-overlapped per-kernel times for the same command
PYSDR_TUNING=1 PYSDR_PSD_STREAMS=1 run_cfg c3_psd1stream 0 --no-cpu-baseline
[ -n "$QUICK" ] && exit 0
run_cfg c3_nopsd 0 --no-psd --no-cpu-baseline
run_cfg c2 1 --workload c2
run_cfg c1 1 --workload c1
run_cfg c4 1 --workload c4
run_cfg c4mono 0 --workload c4mono --no-cpu-baseline
run_cfg rx6 0 --workload rx6 --no-cpu-baseline
run_cfg c1synch 0 --workload c1synch --no-cpu-baseline
# the reference's own multi-receiver launch scripts with its default 1001-tap prototype (FT8tri:47-74, TEST:13-32): matrix-core shapes of mixdec.hip
run_cfg ft8tri 1 --workload ft8tri
run_cfg test2rx 1 --workload test2rx
# The plain bench lines (events only, no profiler attached) come LAST: the counters above are first condensed into
# profiles/<tag>_pmc_traffic.json (stamped with the hashes of the kernel sources), so that the lines carry `traffic`.
python3 scripts/summarize_profiles.py "$OUT" profiles "$TAG" > "$OUT/summarize.log" 2>&1
for i in "${!CFG_NAMES[@]}"; do
  name=${CFG_NAMES[$i]}; args=${CFG_ARGS[$i]}
  if [[ "$args" == *@1stream* ]]; then
    PYSDR_TUNING=1 PYSDR_PSD_STREAMS=1 python3 bench.py ${args%@1stream} --no-other-configs --full-line > "$OUT/$name/bench.json" 2> "$OUT/$name/bench.err"
  else
    python3 bench.py $args --no-other-configs --full-line > "$OUT/$name/bench.json" 2> "$OUT/$name/bench.err"
  fi
  echo "$name: $(python3 -c "
import json,sys
d=json.loads(open('$OUT/$name/bench.json').read().strip().splitlines()[-1])
print('GS/s %.1f ms %.4f frac %.3f traffic %s job %.3f' % (d['value']/1e3, d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline_job']['frac']))")"
done
