#!/bin/bash
# Runs a list of GPU steps on the box (via gpurun); a step that times out or is killed ends the session
# (no further GPU step after a hang), an ordinary failure is recorded and the next step runs.
#   tools/gpu_session.sh <outdir> "<name>|<timeout s>|<command>" ...
OUT=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
# steps whose name starts with "tests" run with the HIP runtime reporting its errors (AMD_LOG_LEVEL=1: errors only), so that an
# abort or a fault names itself in the step log; timed steps run without it
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; tmo=${rest%%|*}; cmd=${rest#*|}
  echo "=== $name (timeout $tmo s): $cmd" | tee -a $OUT/session.log
  start=$(date +%s)
  case $name in tests*) export AMD_LOG_LEVEL=1;; *) unset AMD_LOG_LEVEL;; esac
  timeout -k 10 $tmo bash -c "$cmd" > $OUT/$name.log 2>&1
  rc=$?
  # what the kernel driver saw, if this user may read it (a GPU memory fault or a reset is logged there, not in the process)
  if [ $rc -ge 128 ] || [ $rc -eq 124 ]; then (dmesg 2>/dev/null | tail -n 40) > $OUT/$name.dmesg 2>/dev/null; fi
  echo "=== $name rc=$rc $(( $(date +%s) - start )) s" | tee -a $OUT/session.log
  tail -n 6 $OUT/$name.log
  # a step that timed out, was killed or died of a signal (134 = SIGABRT: the HIP runtime aborts on a GPU memory fault; 139 = SIGSEGV)
  # ends the session: no further GPU step after a hang or a fault
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "step hung, was killed or aborted (rc $rc): stopping" | tee -a $OUT/session.log; exit 1; fi
done
exit 0
