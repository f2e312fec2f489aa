#!/bin/bash
# Run ON THE GPU BOX (via gpurun):  bash scripts/collect_profiles.sh gpurun_out/r1
# Collects what profiles/ is built from: the default bench line, rocprofv3 kernel stats of the
# same command, and the HBM-traffic PMC counters in their own passes (never with a trace).
set -u
OUT=${1:-gpurun_out/prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT" && mkdir -p "$OUT"
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py --no-cpu-baseline > "$OUT/bench_kt.json" 2> "$OUT/kt.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
tail -c 400 "$OUT/bench_default.json"
