"""Probe (1 GPU, world_size 1): does an RCCL all-reduce survive hipGraph capture + replay under this torch/ROCm?
Informational -- the multi-GPU path launches eagerly; a world-1 communicator does not prove the N>1 case."""
import os
import sys
import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
dist.init_process_group('nccl', rank=0, world_size=1)
x = torch.ones(1 << 20, device='cuda', dtype=torch.float64)
dist.all_reduce(x)
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        dist.all_reduce(x)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    dist.all_reduce(x)
    y = x * 2
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print('rccl graph probe ok', float(y[0]))
dist.destroy_process_group()
