#!/bin/bash
# bench.py --no-cpu-baseline for a list of variants: <lib>[,ENV=VAL...]   lib = "default" (the product build) or a
# name of tools/libsdfr_<name>.so;  e.g.  tools/variant_bench.sh 3 default default,SDFR_PIXEL_PERSISTENT=0 w5,SDFR_PIXEL_BLOCKS_PER_CU=20
C=$1; shift
for spec in "$@"; do
  lib=${spec%%,*}; envs=""
  if [ "$spec" != "$lib" ]; then envs=$(echo "${spec#*,}" | tr ',' ' '); fi
  if [ "$lib" != "default" ]; then envs="$envs SDFR_LIBRARY=$PWD/tools/libsdfr_$lib.so"; fi
  env $envs python bench.py --config $C --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg $C %-44s %8.1f Mrays/s %7.3f ms/frame  kernel %7.3f  two-in-flight %7.3f' % ('$spec', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('two_frames_in_flight',{}).get('ms_per_step',0)))"
done
