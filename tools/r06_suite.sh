#!/bin/bash
# the whole GPU suite as the driver runs it (-s: what the runtime prints when it aborts a process must reach the log), slowest tests listed
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/suite.log
step 1150 suite.log python -m pytest tests/ -x -q -m gpu --durations=40 -s
grep -v "^\.*$" $GO/suite.log | grep -i "abort\|fault\|error\|HSA\|corrupt\|free()\|malloc\|passed\|failed\|^[0-9.]*s " | tail -n 70
