#!/bin/bash
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/c12_*.log
step 300 c12_time.log python tools/time_beam.py
tail -n 14 $GO/c12_time.log
