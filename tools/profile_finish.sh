#!/bin/bash
# Run ON THE GPU BOX right after tools/profile_round.sh: text summaries of the rocprofv3 databases (the .db files are
# too large to keep), HBM traffic file, parity table.
R=$PWD
python tools/rocprof_summary.py $(find gpurun_out/trace -name "*.db" | head -1) gpurun_out/kernel_stats.txt > /dev/null 2>&1
python tools/pmc_summary.py $(find gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq -name "*.db") -o gpurun_out/pmc_counters.txt > /dev/null 2>&1
python tools/traffic_json.py $(find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.db") > gpurun_out/traffic.log 2>&1   # round 3: tools/profile_r03.sh is the current script
cp profiles/kernel_traffic.json gpurun_out/
python tools/parity_report.py > gpurun_out/parity.txt 2>&1
find gpurun_out -name "*.db" -delete
ls gpurun_out
