cd $GRAFT_REPO_ROOT
tools/gpu_session.sh gpurun_out/s27 \
 "tree_default|200|python tools/quickbench.py --scene tree --schedules 1 --frames 4" \
 "tree_w5|200|SDFR_LIBRARY=\$PWD/tools/libsdfr_w5.so python tools/quickbench.py --scene tree --schedules 1 --frames 4" \
 "tree_w4|200|SDFR_LIBRARY=\$PWD/tools/libsdfr_w4.so python tools/quickbench.py --scene tree --schedules 1 --frames 4" \
 "dist_default|200|python tools/quickbench.py --scene distortion --schedules 1 --frames 4" \
 "dist_w5|200|SDFR_LIBRARY=\$PWD/tools/libsdfr_w5.so python tools/quickbench.py --scene distortion --schedules 1 --frames 4" \
 "fuzz_tree|400|python tools/fuzz_parity.py --scenes tree --cases 1500 --seed 9101"
