#!/bin/bash
# round 6, GPU call 3: reruns of call 2's failures, pitch accounting, decode timelines at large batches and catalog sizes
source "$(dirname "$0")/gpu_steps.sh"
rm -f gpurun_out/c3_*.log
step 600 c3_tests_a.log python -m pytest tests/test_gpu_recording.py tests/test_gpu_fused_decode.py -x -q -m gpu -s
step 600 c3_tests_b.log python -m pytest tests/test_gpu_resident.py tests/test_gpu_decision.py -x -q -m gpu -s
step 600 c3_tests_c.log python -m pytest tests/test_gpu_rnnoise.py -x -q -m gpu -s -k "mixed_batch or sampled_oracle or int16"
step 300 c3_hostfed.log python bench.py --steps 5 --no-asr --no-latency --no-live-traffic --no-cpu-baseline --no-cfg45 --sustain-seconds 0
cd /tmp
for cfg in "tiny 512" "base 256" "tiny 64"; do set -- $cfg
  MODEL=$1 B=$2 PREC=1 step 200 c3_prof.log rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c3_dec_$1_$2 -- python3 $GRAFT_REPO_ROOT/tools/prof_decode.py
  python3 $GRAFT_REPO_ROOT/tools/dec_timeline.py $GRAFT_REPO_ROOT/gpurun_out/c3_dec_$1_$2 > $GRAFT_REPO_ROOT/gpurun_out/c3_timeline_$1_$2.txt 2>&1
done
for fl in resident inflated; do
  SPEC=medium:q4_1 FLAVOUR=$fl step 400 c3_prof.log rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c3_dec_medium_$fl -- python3 $GRAFT_REPO_ROOT/tools/prof_decode_catalog.py
  python3 $GRAFT_REPO_ROOT/tools/dec_timeline.py $GRAFT_REPO_ROOT/gpurun_out/c3_dec_medium_$fl > $GRAFT_REPO_ROOT/gpurun_out/c3_timeline_medium_$fl.txt 2>&1
done
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/c3_dec_*          # the traces are large; the timelines are what is read
tail -n 4 gpurun_out/c3_tests_a.log; tail -n 4 gpurun_out/c3_tests_b.log; tail -n 6 gpurun_out/c3_tests_c.log
grep -h "ms per generated\|pitch index\|chunk by chunk" gpurun_out/c3_*.log
python3 -c "
import json
l=[x for x in open('gpurun_out/c3_hostfed.log') if x.startswith('{')][-1]; j=json.loads(l); print('value', j['value'], 'host_fed', j['host_fed'])"
