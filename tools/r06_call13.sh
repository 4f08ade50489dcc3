#!/bin/bash
# ladder groups of up to 512 rows: decision tests, then the ASR leg of the bench (batch_ladder)
source "$(dirname "$0")/gpu_steps.sh"
rm -f $GO/c13_*.log
echo skip tests
step 400 c13_bench.log python bench.py --no-cpu-baseline --no-host-fed --no-latency --no-cfg45 --no-live-traffic --sustain-seconds 0

python - <<'P'
import json
l=[x for x in open('gpurun_out/c13_bench.log') if x.startswith('{')][-1]
d=json.loads(l)
def find(o,path=''):
    if isinstance(o,dict):
        for k,v in o.items():
            if 'ladder' in k or 'default' in k: print(path+k, json.dumps(v)[:1800])
            else: find(v,path+k+'.')
find(d)
P
