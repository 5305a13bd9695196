cd $GRAFT_REPO_ROOT
tools/gpu_session.sh gpurun_out/s28 \
 "table|900|tools/variant_table.sh default w3 w4 w5 w6 w8"
