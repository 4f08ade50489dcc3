#!/bin/bash
# round 6, GPU call 5: cross block A/B (values early or late), encoder attention diet: parity tests, times, timelines
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/c5_*.log $GO/c5_*.txt
step 900 c5_tests_a.log python -m pytest tests/test_gpu_recording.py tests/test_gpu_fused_decode.py tests/test_gpu_mode1.py tests/test_gpu_whisper.py -x -q -m gpu -s
for m in tiny base; do for b in 1 64 128 256 512; do for ev in 0 1; do
  CRISPY_HIP_LIB=$PWD/crispy_amd/libcrispy_hip_dev.so CRISPY_FX_EARLYV=$ev MODEL=$m B=$b PREC=1 step 120 c5_ab_cross.log python tools/dec_time.py
done; done; done
for m in tiny base; do for b in 1 64 256 512; do
  MODEL=$m B=$b PREC=1 step 120 c5_time_release.log python tools/dec_time.py
done; done
PREC=1 B=64 step 120 c5_enc_time.log python tools/enc_time.py
PREC=1 B=256 MODEL=base step 120 c5_enc_time.log python tools/enc_time.py
cd /tmp
B=64 PREC=1 step 200 c5_prof.log rocprofv3 --kernel-trace --output-format csv -d $GO/c5_trace_enc -- python3 $GRAFT_REPO_ROOT/tools/prof_encode.py
python3 $GRAFT_REPO_ROOT/tools/enc_timeline.py $GO/c5_trace_enc > $GO/c5_encoder_timeline.txt 2>&1
cd $GRAFT_REPO_ROOT
rm -rf $GO/c5_trace_*
tail -n 4 $GO/c5_tests_a.log
grep -h "decode" $GO/c5_ab_cross.log | paste - - ; echo; grep -h "decode" $GO/c5_time_release.log; grep -hv "^==" $GO/c5_enc_time.log | tail -4
