#!/bin/bash
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/c26_*.log
step 600 c26_tests.log python -m pytest tests/test_gpu_gemv_decode.py -x -q -m gpu -s
tail -n 3 $GO/c26_tests.log
