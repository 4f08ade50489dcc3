#!/bin/bash
# cross q by its own launch at every row count: gemv / resident / catalog tests, then the catalog models per position and the row sweep
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/c23_*.log
step 900 c23_tests.log python -m pytest tests/test_gpu_gemv_decode.py tests/test_gpu_resident.py tests/test_gpu_whisper.py -x -q -m gpu --durations=5 -s
step 600 r06_resident.err python3 tools/bench_resident.py medium:q4_1 large_v3:q5_0
grep '^{' $GO/r06_resident.err > $GO/r06_resident.json
SPEC=medium:q4_1 step 500 c23_medium.log python tools/time_gemv_rows.py
tail -n 3 $GO/c23_tests.log
python3 - <<'P'
import json
for l in open('gpurun_out/r06_resident.json'):
    d=json.loads(l); print(d['model'], {k: round(v['decode_ms_per_position'],3) for k,v in d['flavours'].items()})
P
grep "rows:" $GO/c23_medium.log
