#!/bin/bash
# beam search decided on the device (captured steps): the scripted-model tests, the fuzz tool, the timing of a beam-5 call against greedy
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/c11_*.log
step 600 c11_tests.log python -m pytest tests/test_gpu_decision.py -x -q -m gpu --durations=8 -s
step 400 c11_fuzz.log python tools/fuzz_beam.py
step 300 c11_time.log python tools/time_beam.py
tail -n 4 $GO/c11_tests.log; tail -n 5 $GO/c11_fuzz.log; tail -n 12 $GO/c11_time.log
