#!/bin/bash
# Run ON THE GPU BOX (via gpurun): everything profiles/r05_* is made from -- kernel-trace statistics of the driver's bench
# command, the counter passes, the stream-count sweep, the decode-step timelines of the fused and the staged path at 1 and
# 64 clips, the encoder timeline, and the full bench line.  tools/collect_r05.sh
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=$PWD/gpurun_out
tools/collect_profiles.sh r05 > $out/r05_collect.log 2>&1
BS=1024,2048,4096,8192,16384 T=100 python3 tools/sweep_streams.py > $out/r05_sweep.txt 2>&1
for b in 1 64; do
  for d in fused stages; do
    CRISPY_ASR_DECODE=$d B=$b PREC=1 rocprofv3 --kernel-trace --output-format csv -d $out/r05_dec_${d}_$b -- python3 tools/prof_decode.py > /dev/null 2>&1
    python3 tools/dec_timeline.py $out/r05_dec_${d}_$b > $out/r05_asr_decode_step_timeline_${d}_$b.txt 2>&1
  done
done
B=64 PREC=1 rocprofv3 --kernel-trace --output-format csv -d $out/r05_enc -- python3 tools/prof_encode.py > /dev/null 2>&1
python3 tools/enc_timeline.py $out/r05_enc > $out/r05_asr_encoder_timeline.txt 2>&1
python3 bench.py > $out/r05_bench_line.json 2> $out/r05_bench.err
tail -c 1500 $out/r05_bench_line.json
