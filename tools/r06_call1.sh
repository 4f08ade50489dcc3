#!/bin/bash
# round 6, GPU call 1: the changed decode-path tests, counters of the encoder attention, decode step times by rows
source "$(dirname "$0")/gpu_steps.sh"
rm -f gpurun_out/c1_*.log
step 900 c1_tests.log python -m pytest tests/test_gpu_fused_decode.py tests/test_gpu_resident.py tests/test_gpu_decision.py -x -q -m gpu -s
step 600 c1_pipeline.log python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -s
step 400 c1_pmc64.log bash tools/pmc_attn_enc.sh 64
step 400 c1_pmc256.log bash tools/pmc_attn_enc.sh 256
for m in tiny base; do for b in 1 64 128 256 512; do
  MODEL=$m B=$b PREC=1 step 120 c1_dec_time.log python tools/dec_time.py
done; done
grep -h "decode" gpurun_out/c1_dec_time.log
tail -3 gpurun_out/c1_tests.log gpurun_out/c1_pipeline.log
