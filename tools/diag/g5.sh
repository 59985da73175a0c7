cd $GRAFT_REPO_ROOT
./tools/microbench/stream_read > gpurun_out/g5_stream.txt 2>&1
cat gpurun_out/g5_stream.txt
bash tools/diag/ab_cfg.sh 3 base noasm nouncond old kel2 2>/dev/null > gpurun_out/g5.log
bash tools/diag/ab_cfg.sh 5 base noasm nouncond old kel2 2>/dev/null >> gpurun_out/g5.log
bash tools/diag/ab_cfg.sh 2 base noasm nouncond old kel2 2>/dev/null >> gpurun_out/g5.log
cat gpurun_out/g5.log
