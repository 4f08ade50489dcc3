#!/bin/bash
# Run ON THE GPU BOX (via gpurun): counter passes over one mode-1 encoder pass, filtered to attn_enc_h_kernel
# (VERDICT r5 next #2).  One `rocprofv3 --pmc` pass per group (8 SQ slots per pass; GRBM apart), python3 straight after `--`.
#   usage: bash tools/pmc_attn_enc.sh [B]      -> gpurun_out/attn_pmc_B<B>/{p1..p4}, gpurun_out/r06_attn_enc_pmc_B<B>.json
set -u
cd "$(dirname "$0")/.."
root=$PWD
B=${1:-64}
out=$root/gpurun_out/attn_pmc_B$B
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# names this rocprofv3 does not list on this device are dropped from a group (an unknown name fails the whole pass)
rocprofv3 -L > $out/counters_available.txt 2>&1 || true
keep() { for c in "$@"; do if grep -qw "$c" $out/counters_available.txt; then printf "%s " "$c"; else echo "not listed: $c" >> $out/dropped.txt; fi; done; }
i=0
for grp in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM" \
  "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  grp=$(keep $grp)
  [ -z "$grp" ] && continue
  B=$B PREC=1 timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -o pmc -- python3 $root/tools/pmc_enc.py > $out/p$i.log 2>&1 || echo "pass $i failed (see $out/p$i.log)"
done
cd $root
python3 tools/pmc_attn_summary.py $out $B > gpurun_out/r06_attn_enc_pmc_B$B.json
tail -c 2500 gpurun_out/r06_attn_enc_pmc_B$B.json
