#!/bin/bash
# the files of profiles/r06_* that name kernels of the decode path, again at the final tree: kernel statistics of the bench
# command, the 64-clip ladder by kernel, the bench line
source "$(dirname "$0")/gpu_steps.sh"
cd /tmp
step 400 r06_trace2.log rocprofv3 --kernel-trace --stats --output-format csv -d $GO/r06_trace3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --no-latency --no-cfg45 --no-host-fed --sustain-seconds 0
step 300 r06_prof.log rocprofv3 --kernel-trace --output-format csv -d $GO/r06_ladder2 -- python3 $GRAFT_REPO_ROOT/tools/prof_ladder.py
cd $GRAFT_REPO_ROOT
f=$(find $GO/r06_trace3 -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && cp "$f" $GO/r06_bench_kernel_stats.csv
python3 tools/dec_breakdown.py $GO/r06_ladder2 < /dev/null > $GO/r06_asr_batch_ladder_breakdown.txt 2>&1
find $GO/r06_trace3 $GO/r06_ladder2 -name "*.csv" -size +512k -delete
step 600 r06_bench.err python3 bench.py
grep '^{"metric"' $GO/r06_bench.err | tail -n 1 > $GO/r06_bench_line.json
head -n 6 $GO/r06_bench_kernel_stats.csv | cut -c1-200; head -n 5 $GO/r06_asr_batch_ladder_breakdown.txt | cut -c1-160; tail -c 300 $GO/r06_bench_line.json
