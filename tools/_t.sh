for i in 1 2 3 4; do for v in old new; do echo -n "$v "; CRISPY_HIP_LIB=$PWD/crispy_amd/csrc/build/variants/lib_$v.so BS=4096 T=100 python tools/sweep_streams.py 2>&1 | grep "B=" | cut -c20-60; done; done
CRISPY_HIP_LIB=$PWD/crispy_amd/csrc/build/variants/lib_new.so B=6 T=30 timeout 300 python tools/gpu_parity_debug.py 2>&1 | grep -E "^b[0-9]:|mismatch" | head -8
