#!/bin/bash
# the kernel-trace statistics of the driver's command again, without the host-fed leg (its piecewise launches halve the frame kernel's average)
source "$(dirname "$0")/gpu_steps.sh"
cd /tmp
step 400 r06_trace2.log rocprofv3 --kernel-trace --stats --output-format csv -d $GO/r06_trace2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --no-latency --no-cfg45 --no-host-fed --sustain-seconds 0
cd $GRAFT_REPO_ROOT
f=$(find $GO/r06_trace2 -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && cp "$f" $GO/r06_bench_kernel_stats.csv && head -n 8 "$f" | cut -c1-220
find $GO/r06_trace2 -name "*.csv" -size +512k -delete
