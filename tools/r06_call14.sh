#!/bin/bash
# kernel statistics of the 64-clip ladder (one group of 320 rows per pass)
source "$(dirname "$0")/gpu_steps.sh"
rm -rf $GO/c14_*
cd /tmp
step 300 c14_ladder.log rocprofv3 --kernel-trace --output-format csv -d $GO/c14_ladder -- python3 $GRAFT_REPO_ROOT/tools/prof_ladder.py
cd $GRAFT_REPO_ROOT
python tools/dec_breakdown.py $GO/c14_ladder < /dev/null > $GO/c14_breakdown.txt 2>&1
find $GO/c14_ladder -name "*.csv" -size +1M -delete
head -n 30 $GO/c14_breakdown.txt | cut -c1-200; tail -n 1 $GO/c14_breakdown.txt; grep "clips: call" $GO/c14_ladder.log
