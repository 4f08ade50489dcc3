#!/bin/bash
# the rows of a clip over one copy of its K | V (fused_cross_rows_kernel): decision tests, fused-decode tests, ladder timing, beam timing
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/c17_*.log
step 600 c17_tests.log python -m pytest tests/test_gpu_decision.py tests/test_gpu_fused_decode.py -x -q -m gpu --durations=5 -s
step 200 c17_ladder.log python tools/prof_ladder.py
echo
tail -n 3 $GO/c17_tests.log; grep "clips" $GO/c17_ladder.log; grep "walked" $GO/c17_tests.log
