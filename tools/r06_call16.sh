#!/bin/bash
source "$(dirname "$0")/gpu_steps.sh"
rm -rf $GO/c16_*
cd /tmp
step 300 c16_dn.log rocprofv3 --kernel-trace --output-format csv -d $GO/c16_dn -- python3 $GRAFT_REPO_ROOT/tools/prof_denoise_1024.py
cd $GRAFT_REPO_ROOT
python tools/kernel_spans.py $GO/c16_dn < /dev/null > $GO/c16_spans.txt 2>&1
find $GO/c16_dn -name "*.csv" -size +1M -delete; BS=1024 T=3001 step 100 c16_sweep.log python tools/sweep_streams.py; tail -n 2 $GO/c16_sweep.log
cat $GO/c16_spans.txt | cut -c1-200; grep "streams x" $GO/c16_dn.log
