#!/bin/bash
# Build the HOST half of libpysdr_hip.so (pysdr_amd/csrc/api.hip, compiled as plain C++) over the fake
# HIP runtime + the checking launch layer, once with AddressSanitizer + UBSan and once with
# ThreadSanitizer, and run the driver.  CPU only (SURVEY.md 5 "sanitizers on the CPU build").
#   tests/host_san/run.sh [asan|tsan|all]
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
OUT=${HOST_SAN_OUT:-/tmp/pysdr_host_san}
mkdir -p "$OUT"
SRC="$ROOT/pysdr_amd/csrc/api.hip $HERE/stub_kernels.cpp $HERE/san_main.cpp"
INC="-I$HERE/fake_hip -I$ROOT/pysdr_amd/csrc"
build() { # name, flags...
  local name=$1; shift
  g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -Wall -Wno-unused-function -x c++ $INC "$@" $SRC -o "$OUT/san_$name" -ldl -lpthread
}
what=${1:-all}
if [ "$what" = asan ] || [ "$what" = all ]; then
  build asan -fsanitize=address,undefined -fno-sanitize-recover=undefined
  ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 "$OUT/san_asan"
  PYSDR_TUNING=1 PYSDR_MIXDEC_MFMA=0 ASAN_OPTIONS=detect_leaks=1 "$OUT/san_asan" > /dev/null   # long single-RX prototypes on the vector form
  # every tuning variable parsed (the staged / capped pilot-loop warm-up plan, the other resampler forms, small grids, groups)
  PYSDR_TUNING=1 PYSDR_WFM_PLL="20,13,4,3,1536,2048,5,4,4,4" PYSDR_RESAMP_PLAIN=2 PYSDR_MIXDEC_GRID=3 PYSDR_PSD_GROUP=64 PYSDR_PSD_STREAMS=3 \
    PYSDR_PSD_PACKED=0 PYSDR_MIXDEC_YFLUSH=2 PYSDR_AM_PLL="14,4,4,1024,256" PYSDR_OVERLAP=1 ASAN_OPTIONS=detect_leaks=1 "$OUT/san_asan" > /dev/null
  ASAN_OPTIONS=detect_leaks=1 "$OUT/san_asan" race
fi
if [ "$what" = tsan ] || [ "$what" = all ]; then
  build tsan -fsanitize=thread
  TSAN_OPTIONS=halt_on_error=1 "$OUT/san_tsan" race
fi
echo HOST_SAN_ALL_OK
