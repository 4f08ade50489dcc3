#!/bin/bash
# round 6, GPU call 8: the matrix-vector decode step of the catalog widths -- parity tests, medium / large-v3 step times, timeline
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/c8_*.log $GO/c8_*.txt
step 900 c8_tests.log python -m pytest tests/test_gpu_gemv_decode.py -x -q -m gpu -s
step 600 c8_tests_b.log python -m pytest tests/test_gpu_resident.py tests/test_gpu_recording.py -x -q -m gpu
for spec in medium:q4_1 large_v3:q5_0 small:f16; do for fl in resident inflated; do
  SPEC=$spec FLAVOUR=$fl step 500 c8_cat_time.log python tools/prof_decode_catalog.py
done; done
cd /tmp
SPEC=large_v3:q5_0 FLAVOUR=resident step 400 c8_prof.log rocprofv3 --kernel-trace --output-format csv -d $GO/c8_trace_medium -- python3 $GRAFT_REPO_ROOT/tools/prof_decode_catalog.py
python3 $GRAFT_REPO_ROOT/tools/dec_timeline.py $GO/c8_trace_medium > $GO/c8_timeline_large_resident.txt 2>&1
cd $GRAFT_REPO_ROOT
rm -rf $GO/c8_trace_*
tail -n 5 $GO/c8_tests.log; tail -n 3 $GO/c8_tests_b.log
grep -h "ms per generated\|gemv rms" $GO/c8_*.log
head -40 $GO/c8_timeline_large_resident.txt
