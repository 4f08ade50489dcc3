#!/bin/bash
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/dbg.log
step 900 dbg.log python -m pytest tests/test_gpu_rnnoise.py -x -q -m gpu -s
grep -v "^\.\|pitch index" $GO/dbg.log | tail -n 40
