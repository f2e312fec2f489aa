cp pysdr_amd/libpysdr_hip.so /tmp/keep.so
export PYSDR_TUNING=1   # build.py reads PYSDR_*_FLAGS only under the tuning master switch (round 5)
for fl in "" "-DMM_NTX=2 -DMM_C1_FLAGS=3" "" "-DMM_NTX=2 -DMM_C1_FLAGS=3"; do
  echo "=== flags '$fl'"
  PYSDR_MFMA_FLAGS="$fl" python -m pysdr_amd.build --force > /tmp/build.log 2>&1 || { echo build failed; continue; }
  python scripts/long_prototype_rates.py 2>&1 | sed 's/vector form.*//'
done
cp /tmp/keep.so pysdr_amd/libpysdr_hip.so
