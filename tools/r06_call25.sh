#!/bin/bash
source "$(dirname "$0")/gpu_steps.sh"
step 300 c25_smoke.log python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")'
tail -n 3 $GO/c25_smoke.log
