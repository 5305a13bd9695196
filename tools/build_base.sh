#!/bin/bash
# Developer helper: builds the library of a commit (default HEAD) into tools/libsdfr_base.so, for A/B runs against the working tree
# (tools/ab_env.sh with SDFR_LIBRARY=.../tools/libsdfr_base.so).
REV=${1:-HEAD}
OUT=$PWD/tools/libsdfr_base.so
rm -rf /tmp/sdfr_base && git worktree add -f /tmp/sdfr_base $REV -q && (cd /tmp/sdfr_base && python -c "
from sdf_playground_amd import buildlib
print(buildlib.build(force=True, out='$OUT'))") ; git worktree remove --force /tmp/sdfr_base
