cd $GRAFT_REPO_ROOT
tools/gpu_session.sh gpurun_out/s29 \
 "tree_default|200|python tools/quickbench.py --scene tree --schedules 1,0 --frames 4" \
 "fuzz_tree|400|python tools/fuzz_parity.py --scenes tree --cases 2500 --seed 9102" \
 "math|300|python -m pytest tests/test_gpu_math.py -x -q" \
 "jit|600|python tools/jit_all_scenes.py" \
 "lense_w8|300|tools/variant_bench.sh 5 default w8 default w8"
