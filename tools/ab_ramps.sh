# developer tool (GPU box): step time of BASELINE cfg 2 for several sub-chunk ramps / sub-chunk sizes (lib_sub<N>.so built with -DRN_SUB_FRAMES=N)
run() { echo "== lib ${1:-default} ramp $2"; CRISPY_HIP_LIB=$1 CRISPY_RN_RAMP=$2 timeout -k 10 100 python tools/step_timeline.py 2>&1 | grep -E "frame-kernel sum"; }
V=$PWD/crispy_amd/csrc/build/variants
for rep in 1 2; do
run "" "3,4,6,9"
run "" "3,4,5,7,10"
run $V/lib_sub10.so "3,4,6,9"
run $V/lib_sub16.so "3,4,6,9,13"
run $V/lib_sub24.so "3,4,6,9,13,19"
run $V/lib_sub16.so "3,4,6,9"
done
