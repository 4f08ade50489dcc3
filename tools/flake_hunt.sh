#!/bin/bash
# developer tool: how often does a selection of tests/test_gpu_rnnoise.py fail?  tools/flake_hunt.sh REPS "k-expr" ...
cd "$(dirname "$0")/.."
reps=$1; shift
for sel in "$@"; do
  fails=0
  for i in $(seq $reps); do
    if [ "$sel" = ALL ]; then timeout -k 10 200 python -m pytest tests/test_gpu_rnnoise.py -q -m gpu > gpurun_out/flake.log 2>&1; else timeout -k 10 200 python -m pytest tests/test_gpu_rnnoise.py -q -m gpu -k "$sel" > gpurun_out/flake.log 2>&1; fi
    rc=$?
    if [ $rc -ne 0 ]; then fails=$((fails+1)); grep -E "^FAILED|^E  .*(stream|staged|fused)" gpurun_out/flake.log | cut -c1-600 | head -4; fi
  done
  echo "== [$sel]: $fails failures in $reps runs"
done
