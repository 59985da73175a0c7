"""Known byte counts for calibrating the L2's memory-side request counters (run under tools/pmc_passes.sh):
a 1 GiB streaming copy (reads 1 GiB, writes 1 GiB per launch) and a gather of 2^23 random 128-byte rows out of a 4 GiB
table (reads 1 GiB of rows + 64 MiB of indices, writes 1 GiB per launch)."""
import torch
dev = torch.device('cuda', 0)
x = torch.rand(1 << 28, device=dev)
table = torch.rand(1 << 25, 32, device=dev)
idx = torch.randperm(1 << 25, device=dev)[: 1 << 23].contiguous()
for _ in range(5):
    y = x.clone()
    z = table.index_select(0, idx)
torch.cuda.synchronize()
