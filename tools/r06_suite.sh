#!/bin/bash
# the whole GPU suite as the driver runs it, with the slowest tests listed
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/suite.log
step 1150 suite.log python -m pytest tests/ -x -q -m gpu --durations=40
tail -n 60 $GO/suite.log
