for v in "$@"; do echo "== $v"; export CRISPY_HIP_LIB=$PWD/crispy_amd/csrc/build/variants/lib_$v.so; BS=4096 T=100 python tools/sweep_streams.py 2>&1 | grep "B="; done
