#!/bin/bash
# the coarse pitch search on the f16 matrix cores: RNNoise parity tests, then a same-box A/B of the 4096 x 100 step against
# the vector-ALU form (crispy_amd/csrc/build/variants/lib_valu.so = the same tree with -DRN_COARSE_MFMA=0)
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/c15_*.log
step 600 c15_tests.log python -m pytest tests/test_gpu_rnnoise.py -x -q -m gpu -s
for rep in 1 2 3; do
  BS=4096,1024 T=100 step 120 c15_mfma.log python tools/sweep_streams.py
  CRISPY_HIP_LIB=$PWD/crispy_amd/csrc/build/variants/lib_valu.so BS=4096,1024 T=100 step 120 c15_valu.log python tools/sweep_streams.py
done
tail -n 3 $GO/c15_tests.log; echo mfma; grep -v "^==" $GO/c15_mfma.log | tail -n 12; echo valu; grep -v "^==" $GO/c15_valu.log | tail -n 12
