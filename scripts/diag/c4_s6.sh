cp pysdr_amd/libpysdr_hip.so /tmp/keep.so
export PYSDR_TUNING=1   # build.py reads PYSDR_*_FLAGS only under the tuning master switch (round 5)
for fl in "-DMM_C4_S=6 -DMM_C4_NBUF=4 -DMM_C4_FLAGS=1" "-DMM_C4_S=6 -DMM_C4_NBUF=4 -DMM_C4_FLAGS=0"; do
  PYSDR_MFMA_FLAGS="$fl" PYSDR_API_FLAGS="$fl" python -m pysdr_amd.build --force > /tmp/build.log 2>&1 || { echo "build failed: $fl"; grep -i "error" /tmp/build.log | head -3; continue; }
  echo "== $fl"
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c4 or wbfm" 2>&1 | tail -1
  bash scripts/diag/kt.sh c4 2>&1 | grep "mfma\|pll_seg"
done
cp /tmp/keep.so pysdr_amd/libpysdr_hip.so
