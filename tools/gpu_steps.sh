# Sourced by the per-call scripts that run ON THE GPU BOX (via gpurun).  step <seconds> <log> <command...>: the command under
# `timeout -k 10`, output appended to gpurun_out/<log>; a step that timed out or was killed ENDS the call (no further GPU
# step is started after one: the box may be unhealthy), any other failure is recorded and the call goes on.
cd "$(dirname "${BASH_SOURCE[0]}")/.."
export TMPDIR=/tmp
GO=$PWD/gpurun_out          # absolute: a caller may cd (rocprofv3 wants /tmp) between steps
mkdir -p $GO
step() {
  local secs=$1 log=$2; shift 2
  echo "== $(date +%T) $*" >> $GO/$log
  timeout -k 10 $secs "$@" >> $GO/$log 2>&1
  local rc=$?
  echo "== rc $rc" >> $GO/$log
  echo "[$(date +%T)] rc $rc: $*"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out or was killed: ending the call"; exit $rc; fi
  return 0
}
