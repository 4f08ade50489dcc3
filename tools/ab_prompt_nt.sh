# developer tool (GPU box): the multi-row prompt step with plain (default) or non-temporal (CRISPY_XKV_PROMPT_NT=1) K|V loads
run() { PREC=1 B=$1 NEW=$2 timeout -k 10 100 python tools/dec_time.py 2>&1 | tail -1; }
for rep in 1 2; do
  echo "== prompt step: plain loads"; unset CRISPY_XKV_PROMPT_NT; run 64 1; run 64 32; run 128 1
  echo "== prompt step: non-temporal"; export CRISPY_XKV_PROMPT_NT=1; run 64 1; run 64 32; run 128 1
done
