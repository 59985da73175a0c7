cd $GRAFT_REPO_ROOT
C5="--env kuka14 --nodes 5000 --k1 16 --graphs 32 --mlp-dtype bf16"
C3="--env kuka7 --nodes 2000 --k1 10 --graphs 64 --mlp-dtype bf16"
BA="--no-cpu-baseline --planner-problems 0 --pcie-steps 0 --dense-steps 0 --bf16x3-steps 0 --single-steps 0"
bash tools/diag/traffic_pass.sh 'pre_resident_kernel<32, 1, true' 'kuka14 N=5000 k1=16 graphs=32 bf16' $C5 > gpurun_out/g40.log 2>&1
bash tools/diag/traffic_pass.sh 'pre_resident_kernel<64, 1, true' 'kuka7 N=2000 k1=10 graphs=64 bf16' $C3 >> gpurun_out/g40.log 2>&1
cp profiles/kernel_traffic.json gpurun_out/g40_kernel_traffic.json
timeout 300 python bench.py $BA $C5 2>/dev/null | tail -1 > gpurun_out/g40_bench_cfg5.json
