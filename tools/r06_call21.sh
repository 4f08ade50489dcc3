#!/bin/bash
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/c21_*.log
step 600 c21_tests.log python -m pytest tests/test_gpu_fused_decode.py tests/test_gpu_recording.py -x -q -m gpu --durations=5 -s
tail -n 8 $GO/c21_tests.log | cut -c1-200
