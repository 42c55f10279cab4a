#!/bin/bash
# builds build/exp/libphaze_<prefix><table>.so for priority tables given as 1_0_1_2_... (see PV_PT in pv_wave_fft.h); usage: build_pt_variants.sh <file> <prefix> <extra-flags> table...
FILE=$1; PRE=$2; EXTRA=$3; shift 3
cd "$(dirname "$0")/../phaze_amd/csrc"
for t in "$@"; do
  ( make variant NAME=$PRE$t FILE=$FILE EXTRA="-DPV_PT=${t//_/,} $EXTRA" 2>&1 | grep -E " error|Error " -A3 ) &
  while (( $(jobs -r | wc -l) >= 8 )); do wait -n; done
done
wait
