#!/bin/bash
# Run ON THE GPU BOX after the last kernel-source edit: re-take the HBM traffic entries (they are stamped with the source hash)
# and the bench lines that quote them.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03f
rm -rf $O; mkdir -p $O
cd $R
BA="--no-cpu-baseline --planner-problems 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0 --inflight-steps 0 --other-configs-steps 0"
bash tools/diag/traffic_pass.sh 'pre_resident_kernel<32, 0, true' 'maze2 N=1000 k1=8 graphs=256 fp32' > $O/traffic_edge_pre.log 2>&1
bash tools/diag/traffic_pass.sh 'mp_fused_kernel<32, 0' 'maze2 N=1000 k1=8 graphs=256 fp32' > $O/traffic_mp_cfg2.log 2>&1
bash tools/diag/traffic_pass.sh 'mp_fused_kernel<64, 1' 'kuka7 N=2000 k1=10 graphs=64 bf16' --env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16 > $O/traffic_mp_cfg3.log 2>&1
bash tools/diag/traffic_pass.sh 'mp_fused_kernel<32, 1' 'kuka14 N=5000 k1=16 graphs=32 bf16' --env kuka14 --nodes 5000 --k1 16 --graphs 32 --mlp-dtype bf16 > $O/traffic_mp_cfg5.log 2>&1
cp profiles/kernel_traffic.json $O/
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py $BA --env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16 2>/dev/null | tail -1 > $O/bench_cfg3_kuka7_bf16.json
timeout 300 python bench.py $BA --env kuka14 --nodes 5000 --k1 16 --graphs 32 --mlp-dtype bf16 2>/dev/null | tail -1 > $O/bench_cfg5_kuka14_bf16.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 $BA > $O/trace.log 2>&1
cd $R
python tools/rocprof_summary.py $(find $O/trace -name "*.db" | head -1) $O/bench_kernel_stats.txt > /dev/null 2>&1
find $O -name "*.db" -delete
ls $O
