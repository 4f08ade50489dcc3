#!/bin/bash
source "$(dirname "$0")/gpu_steps.sh"
step 600 r06_bench.err python3 bench.py
grep '^{"metric"' $GO/r06_bench.err | tail -n 1 > $GO/r06_bench_line.json
python3 - <<'P'
import json
d=json.loads(open('gpurun_out/r06_bench_line.json').read())
print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], 'asr', round(d['asr_rtfx']['value']), 'cfg4', d['cfg4']['ms_per_step'], 'cfg5', d['cfg5']['ms_per_step'])
dc=d['asr']['single_clip']['default_options_f16_operand_mode']
print(dc['ms'], dc['one_greedy_pass_per_window']['ms'], dc['batch_ladder']['ms'], dc['batch_ladder']['ratio_to_single_clip'])
P
