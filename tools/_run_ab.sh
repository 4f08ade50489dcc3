for v in "$@"; do echo "== $v"; export CRISPY_HIP_LIB=$PWD/crispy_amd/csrc/build/variants/lib_$v.so; BS=4096 T=100 python tools/sweep_streams.py 2>&1 | grep "B="; done
export CRISPY_HIP_LIB=$PWD/crispy_amd/csrc/build/variants/lib_$1.so; B=6 T=20 timeout 300 python tools/gpu_parity_debug.py 2>&1 | grep -E "^b[0-9]:|mismatch" | head -8
