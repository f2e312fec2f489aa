#!/bin/bash
# builds scripts/experiments/psd_frame_test.bin from the product's kernel sources
set -e
cd "$(dirname "$0")/../.."
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off"; C=pysdr_amd/csrc
hipcc $F -c $C/psdfft.hip -o /tmp/psdfft.o
hipcc $F -fno-slp-vectorize -I pysdr_amd/csrc ${PSD_ABL:+-DPSD_ABL=$PSD_ABL} ${PSD_STAMP:+-DPSD_STAMP} -c scripts/experiments/psd_frame.hip -o /tmp/psdreg.o
hipcc $F ${PSD_STAMP:+-DPSD_STAMP} -c scripts/experiments/psd_frame_test.hip -o /tmp/psd_frame_test.o
hipcc --offload-arch=gfx950 /tmp/psd_frame_test.o /tmp/psdfft.o /tmp/psdreg.o -o scripts/experiments/psd_frame_test${PSD_ABL:+_abl$PSD_ABL}${PSD_STAMP:+_stamp}.bin
