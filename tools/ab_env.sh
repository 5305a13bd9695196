#!/bin/bash
# Developer A/B (GPU box): bench.py of one configuration under several environment settings, interleaved, twice.
#   tools/ab_env.sh <config> "<name>=<ENV=..,ENV=..>" ...     (several settings of one variant: comma-separated) prints ms_per_step / kernel_ms per variant and pass
C=$1; shift
for pass in 1 2; do
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  line=$(env ${envs//,/ } python bench.py --config $C --steps 32 --warmup 6 --no-cpu-baseline --no-second-pass 2>/dev/null | tail -1)
  echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg $C pass $pass %-22s ms_per_step %.4f kernel_ms %.4f Mrays/s %.0f' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms'] if 'roofline' in d else -1, d['value']))"
done
done
