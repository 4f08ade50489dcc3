# developer tool (GPU box): decode time against the number of generated tokens -> what the prompt (cross K|V + prefill) costs
for b in 64 1; do for n in 1 2 8 32; do PREC=1 B=$b NEW=$n timeout -k 10 100 python tools/dec_time.py 2>&1 | tail -1; done; done
