# usage: bash tools/pmc_passes.sh OUTDIR [group ...] -- separate rocprofv3 --pmc passes over tools/prof_encode.py (env MODEL/B/PREC)
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then
  set -- "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"
fi
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/p$i -o pmc -- python3 $GRAFT_REPO_ROOT/tools/prof_encode.py > $GRAFT_REPO_ROOT/$out/p$i.log 2>&1 || echo "pass $i failed"
done
