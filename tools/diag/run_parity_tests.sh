python tools/parity_report.py fp32 > gpurun_out/parity_new.txt 2>&1
GNNMP_NODE_F64=0 python tools/parity_report.py fp32 > gpurun_out/parity_allfp32.txt 2>&1
head -17 gpurun_out/parity_new.txt
python -m pytest tests/test_explorer_parity.py tests/test_explorer_fuzz_gpu.py tests/test_full_size_gpu.py tests/test_explorer_bf16x3.py tests/test_full_size_bf16_gpu.py -q 2>&1 | tail -15
python __graft_entry__.py smoke 2>&1 | tail -3
