#!/bin/bash
# log-mel with its tables in LDS and the cheap log10: parity tests, the recording tests, then the default bench line
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/c10_*.log
step 500 c10_tests.log python -m pytest tests/test_gpu_logmel.py tests/test_gpu_recording.py tests/test_gpu_whisper.py tests/test_gpu_pipeline.py -x -q -m gpu --durations=8 -s
step 400 c10_bench.log python bench.py
tail -n 4 $GO/c10_tests.log; tail -n 2 $GO/c10_bench.log | cut -c1-3000
