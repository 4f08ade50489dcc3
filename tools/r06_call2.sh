#!/bin/bash
# round 6, GPU call 2: recording entry point, int16 transport, fallback groups, fused-decode contract, host-fed bench leg
source "$(dirname "$0")/gpu_steps.sh"
rm -f gpurun_out/c2_*.log
step 600 c2_tests_a.log python -m pytest tests/test_gpu_recording.py tests/test_gpu_fused_decode.py -x -q -m gpu -s
step 600 c2_tests_b.log python -m pytest tests/test_gpu_resident.py tests/test_gpu_decision.py -x -q -m gpu -s
step 600 c2_tests_c.log python -m pytest tests/test_gpu_rnnoise.py -x -q -m gpu -k "int16 or layouts or pipelined or golden"
step 600 c2_tests_d.log python -m pytest tests/test_gpu_bench_paths.py tests/test_gpu_whisper.py -x -q -m gpu -s
tail -4 gpurun_out/c2_tests_a.log gpurun_out/c2_tests_b.log gpurun_out/c2_tests_c.log gpurun_out/c2_tests_d.log
grep -h "host-fed\|chunk by chunk" gpurun_out/c2_tests_*.log
