#!/bin/bash
# Run ON THE GPU BOX (via gpurun), in two calls: everything profiles/r06_* is made from.
#   tools/collect_r06.sh a   kernel-trace statistics + counter passes of the driver's bench command, the stream-count sweep, the bench line
#   tools/collect_r06.sh b   decode-step timelines (tiny 1 / 64 / 512 rows, base 256, medium-q4_1 and large-v3-q5_0 resident), the
#                            encoder timeline, the catalog models resident / inflated, the 64-clip ladder, the cost of a beam position
source "$(dirname "$0")/gpu_steps.sh"
part=${1:-a}
if [ "$part" = a ]; then
  step 900 r06_collect.log tools/collect_profiles.sh r06
  BS=1024,2048,4096,8192,16384 T=100 step 300 r06_sweep.txt python3 tools/sweep_streams.py
  step 600 r06_bench.err python3 bench.py
  grep '^{"metric"' $GO/r06_bench.err | tail -n 1 > $GO/r06_bench_line.json
  tail -c 600 $GO/r06_bench_line.json
else
  cd /tmp
  for spec in tiny:1 tiny:64 tiny:512 base:256; do
    m=${spec%%:*}; b=${spec##*:}
    MODEL=$m B=$b PREC=1 step 200 r06_prof.log rocprofv3 --kernel-trace --output-format csv -d $GO/r06_dec_${m}_$b -- python3 $GRAFT_REPO_ROOT/tools/prof_decode.py
    python3 $GRAFT_REPO_ROOT/tools/dec_timeline.py $GO/r06_dec_${m}_$b < /dev/null > $GO/r06_asr_decode_step_timeline_${m}_$b.txt 2>&1
  done
  for spec in medium:q4_1 large_v3:q5_0; do
    m=${spec%%:*}
    SPEC=$spec FLAVOUR=resident step 400 r06_prof.log rocprofv3 --kernel-trace --output-format csv -d $GO/r06_dec_${m}_resident -- python3 $GRAFT_REPO_ROOT/tools/prof_decode_catalog.py
    python3 $GRAFT_REPO_ROOT/tools/dec_timeline.py $GO/r06_dec_${m}_resident < /dev/null > $GO/r06_asr_decode_step_timeline_${m}_resident.txt 2>&1
  done
  B=64 PREC=1 step 200 r06_prof.log rocprofv3 --kernel-trace --output-format csv -d $GO/r06_enc -- python3 $GRAFT_REPO_ROOT/tools/prof_encode.py
  python3 $GRAFT_REPO_ROOT/tools/enc_timeline.py $GO/r06_enc < /dev/null > $GO/r06_asr_encoder_timeline.txt 2>&1
  step 300 r06_prof.log rocprofv3 --kernel-trace --output-format csv -d $GO/r06_ladder -- python3 $GRAFT_REPO_ROOT/tools/prof_ladder.py
  python3 $GRAFT_REPO_ROOT/tools/dec_breakdown.py $GO/r06_ladder < /dev/null > $GO/r06_asr_batch_ladder_breakdown.txt 2>&1
  cd $GRAFT_REPO_ROOT
  find $GO/r06_dec_* $GO/r06_enc $GO/r06_ladder -name "*.csv" -size +256k -delete
  step 600 r06_resident.err python3 tools/bench_resident.py medium:q4_1 large_v3:q5_0
  grep '^{' $GO/r06_resident.err > $GO/r06_resident.json
  step 200 r06_beam_position_cost.txt python3 tools/time_beam.py
  tail -n 3 $GO/r06_asr_batch_ladder_breakdown.txt; cut -c1-400 $GO/r06_resident.json; grep "clips:" $GO/r06_beam_position_cost.txt
fi
