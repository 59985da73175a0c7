#!/bin/bash
# round 5, first GPU pass: new tests, census, planner parity (incl. bf16 decision level), bench line
mkdir -p gpurun_out/r05
python -m pytest tests/test_status_gpu.py tests/test_dist_gpu.py -x -q > gpurun_out/r05/tests_new.log 2>&1
echo "tests_new rc=$?" >> gpurun_out/r05/tests_new.log
python bench.py > gpurun_out/r05/bench.json 2> gpurun_out/r05/bench.err
python tools/parity_census.py > gpurun_out/r05/parity_census.txt 2>&1
python tools/planner_parity.py > gpurun_out/r05/planner_parity.txt 2>&1
tail -5 gpurun_out/r05/tests_new.log
