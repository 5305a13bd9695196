#!/bin/bash
# Runs a list of GPU steps on the box (via gpurun); a step that times out or is killed ends the session
# (no further GPU step after a hang), an ordinary failure is recorded and the next step runs.
#   tools/gpu_session.sh <outdir> "<name>|<timeout s>|<command>" ...
OUT=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; tmo=${rest%%|*}; cmd=${rest#*|}
  echo "=== $name (timeout $tmo s): $cmd" | tee -a $OUT/session.log
  start=$(date +%s)
  timeout -k 10 $tmo bash -c "$cmd" > $OUT/$name.log 2>&1
  rc=$?
  echo "=== $name rc=$rc $(( $(date +%s) - start )) s" | tee -a $OUT/session.log
  tail -n 6 $OUT/$name.log
  # a step that timed out, was killed or died of a signal (134 = SIGABRT: the HIP runtime aborts on a GPU memory fault; 139 = SIGSEGV)
  # ends the session: no further GPU step after a hang or a fault
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "step hung, was killed or aborted (rc $rc): stopping" | tee -a $OUT/session.log; exit 1; fi
done
exit 0
