#!/bin/bash
# where a wide matrix-vector step of medium-q4_1 spends its time: one position at 64 rows by kernel
source "$(dirname "$0")/gpu_steps.sh"
rm -rf $GO/c20_*
cd /tmp
SPEC=medium:q4_1 FLAVOUR=resident B=64 NEW=9 step 400 c20_prof.log rocprofv3 --kernel-trace --output-format csv -d $GO/c20_trace -- python3 $GRAFT_REPO_ROOT/tools/prof_decode_catalog.py
cd $GRAFT_REPO_ROOT
python3 tools/dec_breakdown.py $GO/c20_trace < /dev/null > $GO/c20_breakdown.txt 2>&1
find $GO/c20_trace -name "*.csv" -size +512k -delete
head -n 16 $GO/c20_breakdown.txt | cut -c1-190
