#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel stats + PMC passes of the same command.
# Outputs under gpurun_out/$1/ ; tools/profiles_to_repo.py turns them into profiles/.
set -o pipefail
R=${1:-r01}
OUT=gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
CMD="bench.py --steps 16 --warmup 2 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 $CMD > $OUT/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --kernel-trace -d $OUT/pmc_valu --output-format csv -- python3 $CMD > $OUT/pmc_valu.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch --output-format csv -- python3 $CMD > $OUT/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write --output-format csv -- python3 $CMD > $OUT/pmc_write.log 2>&1
# the bench lines last: bench.py reads profiles/pmc_*.json (executed-work view, HBM traffic), which
# the PMC passes above have just refreshed in this copy of the tree
python3 tools/profiles_to_repo.py $R > /dev/null 2>&1
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 300 python3 bench.py --schedule wavefront --no-cpu-baseline > $OUT/bench_wavefront.json 2>> $OUT/bench.err
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock Freq" | head -6 > $OUT/device.txt
nproc >> $OUT/device.txt
ls -R $OUT | head -40
