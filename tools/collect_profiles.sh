#!/bin/bash
# Runs on the GPU box (via gpurun): for one BASELINE configuration, the bench line + rocprofv3 kernel
# stats + PMC passes of the SAME command (bench.py --no-second-pass: one frame in flight, so every
# dispatch in the statistics is an isolated kernel).
#   tools/collect_profiles.sh <round> <config> [pmc]      e.g.  r02 3 pmc     (pmc: also the counter passes)
# Outputs under gpurun_out/<round>/cfg<config>/ ; tools/profiles_to_repo.py turns them into profiles/.
set -o pipefail
R=${1:-r04}
C=${2:-3}
OUT=gpurun_out/$R/cfg$C
mkdir -p $OUT
export TMPDIR=/tmp
CMD="bench.py --config $C --steps 16 --warmup 2 --no-cpu-baseline --no-second-pass --no-extra-passes"
# .git does not travel to the GPU box: the caller stamps the tree (git rev-parse --short HEAD > .build_commit) before gpurun
cp .build_commit $OUT/commit.txt 2>/dev/null || echo unknown > $OUT/commit.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 $CMD > $OUT/stats.log 2>&1 || exit 1
if [ "$3" = "pmc" ]; then
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --kernel-trace -d $OUT/pmc_valu --output-format csv -- python3 $CMD > $OUT/pmc_valu.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch --output-format csv -- python3 $CMD > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write --output-format csv -- python3 $CMD > $OUT/pmc_write.log 2>&1 || exit 1
fi
# the bench line last: bench.py reads profiles/pmc_<round>_cfg<config>.json (executed-work view, HBM
# traffic), which the PMC passes above have just refreshed in this copy of the tree
python3 tools/profiles_to_repo.py $R $C > /dev/null 2>&1
timeout -k 10 400 python3 bench.py --config $C > $OUT/bench.json 2> $OUT/bench.err || exit 1
tail -c 600 $OUT/bench.json
