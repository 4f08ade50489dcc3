#!/bin/bash
# the matrix-vector step of the catalog widths at every row count: the tests, then gemv against skinny per position at 1 ... 128 rows
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/c19_*.log
step 900 c19_tests.log python -m pytest tests/test_gpu_gemv_decode.py tests/test_gpu_resident.py -x -q -m gpu --durations=5 -s
SPEC=small:dense step 300 c19_small.log python tools/time_gemv_rows.py
SPEC=medium:q4_1 step 500 c19_medium.log python tools/time_gemv_rows.py
tail -n 3 $GO/c19_tests.log; grep "rows:" $GO/c19_small.log; grep "rows:" $GO/c19_medium.log
