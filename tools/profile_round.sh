#!/bin/bash
# Run ON THE GPU BOX (via gpurun): bench + rocprofv3 kernel trace + separate PMC passes for HBM traffic.
# Outputs go to gpurun_out/; summaries are produced afterwards by tools/rocprof_summary.py / tools/pmc_summary.py.
set -u
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python bench.py > $R/gpurun_out/bench.json 2> $R/gpurun_out/bench.err
cat $R/gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_sq
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --pcie-steps 0 --bf16x3-steps 0 > $R/gpurun_out/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --pcie-steps 0 --bf16x3-steps 0 --unique 64 > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --pcie-steps 0 --bf16x3-steps 0 --unique 64 > $R/gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_sq -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --pcie-steps 0 --bf16x3-steps 0 --unique 64 > $R/gpurun_out/pmc_sq.log 2>&1
ls $R/gpurun_out
