#!/bin/bash
# A/B of two builds on the ASR timing tools (encoder + decode; mode 0 and mode 1; mel)
for lib in crispy_amd/libcrispy_hip.so crispy_amd/csrc/build/variants/lib_noslp_all.so; do
  echo "=== $lib"
  for prec in 1 0; do
    CRISPY_HIP_LIB=$PWD/$lib PREC=$prec timeout -k 10 200 python tools/enc_time.py 2>&1 | tail -1
    CRISPY_HIP_LIB=$PWD/$lib PREC=$prec timeout -k 10 200 python tools/dec_time.py 2>&1 | tail -1
    CRISPY_HIP_LIB=$PWD/$lib PREC=$prec B=1 timeout -k 10 200 python tools/dec_time.py 2>&1 | tail -1
  done
  CRISPY_HIP_LIB=$PWD/$lib PREC=1 timeout -k 10 200 python tools/bench_whisper.py 2>&1 | grep -E "log-mel|encoder|decode" | head -4
done
