#!/bin/bash
# bench.py --no-cpu-baseline for a list of library variants (tools/libsdfr_<name>.so; "default" = the product build)
#   tools/variant_bench.sh <config> <name> ...
C=$1; shift
for v in "$@"; do
  if [ "$v" = "default" ]; then unset SDFR_LIBRARY; else export SDFR_LIBRARY=$PWD/tools/libsdfr_$v.so; fi
  python bench.py --config $C --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg $C %-8s %8.1f Mrays/s %7.3f ms/frame  kernel %7.3f  two-in-flight %7.3f' % ('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('two_frames_in_flight',{}).get('ms_per_step',0)))"
done
