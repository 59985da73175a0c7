cd $GRAFT_REPO_ROOT
bash tools/diag/build_variant.sh mztrace -DGNNMP_MAZE_TRACE > gpurun_out/g32.log 2>&1
GNNMP_LIB=$PWD/gnn-motion-planning_amd/libgnnmp_mztrace.so python tools/diag/maze_trace.py >> gpurun_out/g32.log 2>&1
timeout 1200 python -m pytest tests/test_known_answer_gpu.py tests/test_planner_gpu.py tests/test_planner_evalset_gpu.py tests/test_planner_rounds_gpu.py tests/test_maze_explore_gpu.py -m gpu -x -q 2>&1 | tail -5 >> gpurun_out/g32.log
timeout 900 python tools/diag/planner_chunks.py 1024 >> gpurun_out/g32.log 2>&1
