#!/bin/bash
# one-row step of medium-q4_1: cross q inside the cross kernel (shipped) against q by its own launch (developer build, CRISPY_ASR_XQ_FIRST=1)
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/c22_*.log
for rep in 1 2; do
  CRISPY_HIP_LIB=$PWD/crispy_amd/libcrispy_hip_dev.so SPEC=medium:q4_1 FLAVOUR=resident B=1 NEW=17 step 300 c22_in.log python tools/prof_decode_catalog.py
  CRISPY_DEV_KNOBS=1 CRISPY_ASR_XQ_FIRST=1 CRISPY_HIP_LIB=$PWD/crispy_amd/libcrispy_hip_dev.so SPEC=medium:q4_1 FLAVOUR=resident B=1 NEW=17 step 300 c22_first.log python tools/prof_decode_catalog.py
done
echo in-kernel; grep "per generated" $GO/c22_in.log; echo q-first; grep "per generated" $GO/c22_first.log
