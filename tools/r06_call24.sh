#!/bin/bash
# the matrix-vector step's timelines again at the final tree (cross q as its own launch)
source "$(dirname "$0")/gpu_steps.sh"
cd /tmp
for spec in medium:q4_1 large_v3:q5_0; do
  m=${spec%%:*}
  rm -rf $GO/r06_dec_${m}_resident
  SPEC=$spec FLAVOUR=resident step 400 r06_prof.log rocprofv3 --kernel-trace --output-format csv -d $GO/r06_dec_${m}_resident -- python3 $GRAFT_REPO_ROOT/tools/prof_decode_catalog.py
  python3 $GRAFT_REPO_ROOT/tools/dec_timeline.py $GO/r06_dec_${m}_resident < /dev/null > $GO/r06_asr_decode_step_timeline_${m}_resident.txt 2>&1
done
cd $GRAFT_REPO_ROOT
find $GO/r06_dec_medium_resident $GO/r06_dec_large_v3_resident -name "*.csv" -size +256k -delete
head -n 12 $GO/r06_asr_decode_step_timeline_medium_resident.txt | cut -c1-150
