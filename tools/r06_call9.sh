#!/bin/bash
# round 6, GPU call 9: the matrix-vector decode step of the catalog widths -- parity tests, medium / large-v3 step times, timeline
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/c9_*.log $GO/c9_*.txt
step 900 c9_tests.log python -m pytest tests/test_gpu_gemv_decode.py -x -q -m gpu -s
step 600 c9_tests_b.log python -m pytest tests/test_gpu_resident.py tests/test_gpu_recording.py -x -q -m gpu
for spec in medium:q4_1 large_v3:q5_0 small:f16; do for fl in resident inflated; do
  SPEC=$spec FLAVOUR=$fl step 500 c9_cat_time.log python tools/prof_decode_catalog.py
done; done
cd /tmp
SPEC=large_v3:q5_0 FLAVOUR=resident step 400 c9_prof.log rocprofv3 --kernel-trace --output-format csv -d $GO/c9_trace_medium -- python3 $GRAFT_REPO_ROOT/tools/prof_decode_catalog.py
python3 $GRAFT_REPO_ROOT/tools/dec_timeline.py $GO/c9_trace_medium > $GO/c9_timeline_large_resident.txt 2>&1
cd $GRAFT_REPO_ROOT
rm -rf $GO/c9_trace_*
tail -n 5 $GO/c9_tests.log; tail -n 3 $GO/c9_tests_b.log
grep -h "ms per generated\|gemv rms" $GO/c9_*.log
head -40 $GO/c9_timeline_large_resident.txt
