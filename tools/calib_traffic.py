"""Calibration workload for the FETCH_SIZE / WRITE_SIZE counters: known byte counts in access shapes like the step
kernel's (16-byte-per-lane contiguous records; 4-byte-per-lane strided rows).  Run under rocprofv3 --pmc."""
import torch

dev = torch.device("cuda", 0)
n = 64 * 1024 * 1024  # 256 MiB of float32
x = torch.ones(n, device=dev, dtype=torch.float32)
y = torch.empty_like(x)
torch.cuda.synchronize()
for _ in range(3):
    y.copy_(x)          # reads 256 MiB, writes 256 MiB, 16 B per lane
torch.cuda.synchronize()
z = torch.empty(n // 4, device=dev, dtype=torch.float32)
for _ in range(3):
    z.fill_(2.0)        # writes 64 MiB
torch.cuda.synchronize()
