#!/bin/bash
# Register / scratch / occupancy report of the per-scene kernels of one scene group (compiler's view).
#   tools/kernel_resources.sh <group = scene index, csrc/sdfr_perpixel.h> [name filter] [extra flags...]   (the scene's own options of buildlib.SCENE_FLAGS are NOT applied here)
G=${1:-2}; F=${2:-k_pixel}; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -x hip -Wno-unused-result -Wno-unknown-pragmas \
  -Isdf_playground_amd/csrc -DSDFR_GROUP=$G "$@" -c sdf_playground_amd/csrc/sdfr_kernels_group.hip -o /tmp/kres_$G.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -A12 "Function Name: .*$F" | grep -E "Function Name|VGPRs:|SGPRs:|Spill|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: //'
