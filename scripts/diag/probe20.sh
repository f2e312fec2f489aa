#!/bin/bash
# 20 in-process ncclCommInitRank + broadcast round trips on this box (C probe with a SIGABRT
# backtrace handler + the Python child the GPU test runs); appends one line per box to
# gpurun_out/rccl_boxes.txt
cd "$(dirname "$0")/../.."
out=gpurun_out/rccl_diag; mkdir -p $out
ok=0; bad=0
for i in $(seq 1 20); do
  if timeout 120 scripts/diag/rccl_init_probe local setdev_first > $out/p.$i.out 2> $out/p.$i.err; then ok=$((ok+1)); rm -f $out/p.$i.out $out/p.$i.err; else bad=$((bad+1)); fi
done
echo "$(date -u +%FT%TZ) host=$(hostname) gpu=$(rocm-smi --showuniqueid 2>/dev/null | grep -m1 -o '0x[0-9a-f]*') c_probe ok=$ok bad=$bad" | tee -a gpurun_out/rccl_boxes.txt
