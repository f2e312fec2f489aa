#!/bin/bash
# Root-cause hunt for the intermittent abort inside ncclCommInitRank.
# Writes everything under gpurun_out/rccl_diag/.
cd "$(dirname "$0")/../.."
out=gpurun_out/rccl_diag
mkdir -p $out
ulimit -c 0
run_variant() {  # name, count, cmd...
  local name=$1 n=$2; shift 2
  local ok=0 bad=0
  for i in $(seq 1 $n); do
    timeout 120 "$@" > $out/$name.$i.out 2> $out/$name.$i.err
    rc=$?
    if [ $rc -eq 0 ]; then ok=$((ok+1)); rm -f $out/$name.$i.out $out/$name.$i.err; else bad=$((bad+1)); echo "rc=$rc" >> $out/$name.$i.err; fi
  done
  echo "$name ok=$ok bad=$bad" | tee -a $out/summary.txt
}
P=scripts/diag/rccl_init_probe
run_variant c_local_first 25 $P local setdev_first
run_variant c_global_first 15 $P global setdev_first
run_variant c_local_late 15 $P local setdev_late
cat > $out/child.py <<'PY'
import ctypes as C, sys, faulthandler
faulthandler.enable()
import numpy as np
sys.path.insert(0, sys.argv[1])
from pysdr_amd import _lib, multi
from pysdr_amd.synth import CONFIGS, synth_iq
from tests.test_gpu_parity import make_gpu_receivers
P, rxs = make_gpu_receivers(CONFIGS['C2'])
ctx = P._pysdr_stream
lib = _lib.lib()
x = synth_iq(CONFIGS['C2'], 4096, 5)
d = C.c_void_p()
_lib.check(lib.pysdr_dev_alloc(0, x.nbytes, C.byref(d)), "alloc")
_lib.check(lib.pysdr_dev_upload(0, d, C.c_void_p(x.ctypes.data), x.nbytes), "upload")
bc = multi.RcclBroadcaster(ctx)
bc.bcast(d.value, x.nbytes, 0)
_lib.check(lib.pysdr_sync(ctx.h), "sync")
bc.close()
print("RCCL_ROUNDTRIP_OK")
PY
run_variant py_child 20 python3 $out/child.py "$PWD"
NCCL_DEBUG=WARN run_variant py_child_warn 10 python3 $out/child.py "$PWD"
ls $out | head -50
for f in $(ls $out/*.err 2>/dev/null | head -6); do echo "=== $f"; tail -40 $f; done
