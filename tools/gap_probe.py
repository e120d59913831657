#!/usr/bin/env python3
"""LAB: what does the ~5.6 us between two dependent launches depend on? Pairs (first kernel, tiny second kernel) on one stream under
rocprofv3 --kernel-trace; tools/gap_probe.py prints nothing itself -- read the database with tools/gap_probe_read.py.
    cd /tmp && rocprofv3 --kernel-trace -d <dir> -o gp -- python3 <repo>/tools/gap_probe.py"""
import torch
dev = torch.device("cuda", 0)
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)          # 1 GiB
mid = torch.empty(8 * 1024 * 1024, dtype=torch.float32, device=dev)            # 32 MiB
a = torch.randn(8192, 8192, dtype=torch.bfloat16, device=dev)
b = torch.randn(8192, 8192, dtype=torch.bfloat16, device=dev)
c = torch.empty(8192, 8192, dtype=torch.bfloat16, device=dev)
tiny = torch.zeros(64, device=dev)
torch.cuda.synchronize()
for rep in range(12):
    big.fill_(1.0); tiny.add_(1.0)                 # 1 GiB written, then tiny           (marker: fill -> add)
    tiny.mul_(1.0)
    mid.fill_(2.0); tiny.add_(1.0)                 # 32 MiB written (fits the L2s + MALL)
    tiny.mul_(1.0)
    torch.mm(a, b, out=c); tiny.add_(1.0)          # ~0.5 ms of matrix work, 128 MiB written
    tiny.mul_(1.0)
    s = big.sum(); tiny.add_(1.0)                  # 1 GiB READ, nothing written
    tiny.mul_(1.0)
    torch.cuda.synchronize()
