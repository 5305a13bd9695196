#!/bin/bash
# Developer helper (GPU box): counters of k_pixel for one configuration under two libraries (base = tools/libsdfr_base.so, work = the tree's).
#   tools/pmc_ab.sh <outdir> <config>
OUT=$1; C=$2
mkdir -p $OUT
export TMPDIR=/tmp
CMD="bench.py --config $C --steps 16 --warmup 2 --no-cpu-baseline --no-second-pass --no-extra-passes"
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
for v in base work; do
  if [ $v = base ]; then export SDFR_LIBRARY=$GRAFT_REPO_ROOT/tools/libsdfr_base.so; else unset SDFR_LIBRARY; fi
  timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --kernel-trace -d $OUT/${v}_valu --output-format csv -- python3 $CMD > $OUT/${v}_valu.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM --kernel-trace -d $OUT/${v}_wait --output-format csv -- python3 $CMD > $OUT/${v}_wait.log 2>&1 || echo "wait set failed" >> $OUT/${v}_wait.log
  timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES --kernel-trace -d $OUT/${v}_icache --output-format csv -- python3 $CMD > $OUT/${v}_icache.log 2>&1 || echo "icache set failed" >> $OUT/${v}_icache.log
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${v}_fetch --output-format csv -- python3 $CMD > $OUT/${v}_fetch.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${v}_write --output-format csv -- python3 $CMD > $OUT/${v}_write.log 2>&1 || exit 1
  for s in valu wait icache fetch write; do
    f=$(find $OUT/${v}_$s -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 tools/pmc_summary.py $f > $OUT/${v}_$s.txt 2>&1
  done
done
grep -h -A12 "k_pixel" $OUT/*_valu.txt $OUT/*_wait.txt $OUT/*_icache.txt $OUT/*_fetch.txt $OUT/*_write.txt 2>/dev/null | head -150
