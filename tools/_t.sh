python -m pytest tests/test_gpu_rnnoise.py -x -q 2>&1 | tail -2
for i in 1 2 3; do BS=4096 T=100 python tools/sweep_streams.py 2>&1 | grep "B=" | cut -c20-60; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/hp1 -- python tools/pmc_frame.py > /dev/null 2>&1
f=$(find gpurun_out/hp1 -name "*kernel_stats.csv" | head -1); grep "rn_" $f | cut -c1-140
