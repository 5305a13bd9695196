#!/bin/bash
# tools/scene_table.py per variant (same syntax as tools/variant_bench.sh)
for spec in "$@"; do
  lib=${spec%%,*}; envs=""
  if [ "$spec" != "$lib" ]; then envs=$(echo "${spec#*,}" | tr ',' ' '); fi
  if [ "$lib" != "default" ]; then envs="$envs SDFR_LIBRARY=$PWD/tools/libsdfr_$lib.so"; fi
  echo "--- $spec"; env $envs python tools/scene_table.py 2>/dev/null | grep -v "host buffer"
done
