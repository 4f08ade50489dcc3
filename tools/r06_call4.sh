#!/bin/bash
# round 6, GPU call 4: cross block with K and V requested up front -- parity, step times, timelines at large batches and catalog sizes
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/c4_*.log
step 600 c4_tests_a.log python -m pytest tests/test_gpu_recording.py tests/test_gpu_fused_decode.py -x -q -m gpu -s
for m in tiny base; do for b in 1 64 256 512; do
  MODEL=$m B=$b PREC=1 step 120 c4_dec_time.log python tools/dec_time.py
done; done
cd /tmp
for cfg in "tiny 512" "base 256" "tiny 64" "tiny 1"; do set -- $cfg
  MODEL=$1 B=$2 PREC=1 step 200 c4_prof.log rocprofv3 --kernel-trace --output-format csv -d $GO/c4_dec_$1_$2 -- python3 $GRAFT_REPO_ROOT/tools/prof_decode.py
  python3 $GRAFT_REPO_ROOT/tools/dec_timeline.py $GO/c4_dec_$1_$2 > $GO/c4_timeline_$1_$2.txt 2>&1
done
for fl in resident inflated; do
  SPEC=medium:q4_1 FLAVOUR=$fl step 400 c4_prof.log rocprofv3 --kernel-trace --output-format csv -d $GO/c4_dec_medium_$fl -- python3 $GRAFT_REPO_ROOT/tools/prof_decode_catalog.py
  python3 $GRAFT_REPO_ROOT/tools/dec_timeline.py $GO/c4_dec_medium_$fl > $GO/c4_timeline_medium_$fl.txt 2>&1
done
cd $GRAFT_REPO_ROOT
rm -rf $GO/c4_dec_*          # the traces are large; the timelines are what is read
tail -n 4 $GO/c4_tests_a.log
grep -h "decode\|ms per generated\|chunk by chunk" $GO/c4_*.log
