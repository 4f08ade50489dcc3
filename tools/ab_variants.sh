#!/bin/bash
# Developer tool: build variants of libcrispy_hip.so with different -D flags and time the frame kernel on each.
#   tools/ab_variants.sh build  name1:"-DFOO=1" name2:"-DFOO=2" ...     (here, cross-compiles)
#   tools/ab_variants.sh run    name1 name2 ...                          (on the GPU box, via gpurun)
set -e
cd "$(dirname "$0")/.."
mode=$1; shift
mkdir -p crispy_amd/csrc/build/variants
if [ "$mode" = build ]; then
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    (cd crispy_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-slp-vectorize $flags -x hip -shared \
       -o build/variants/lib_$name.so api_util.cpp crispy_api.cpp rn_kernels.hip asr_api.cpp mel_kernels.hip resample_kernels.hip whisper_kernels.hip whisper_enc_f16.hip whisper_dec_f16.hip whisper_quant.hip whisper_api.cpp) &
  done
  wait
else
  for name in "$@"; do
    echo "== $name"
    CRISPY_HIP_LIB=$PWD/crispy_amd/csrc/build/variants/lib_$name.so BS=${BS:-4096} T=${T:-100} python tools/sweep_streams.py
  done
fi
