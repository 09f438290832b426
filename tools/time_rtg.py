"""Dev tool: navsim_rtg_scan timing (hipGraph of 32 launches, HIP events) + bit-exactness vs the oracle on odd shapes."""
import sys, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from navbot_ppo_amd.env import rtg_scan
from oracle import navsim_oracle as O
for T, N in ((512, 4096), (512, 4097), (100, 37), (1300, 640)):
    rew = torch.randn((T, N), device="cuda"); ended = (torch.rand((T, N), device="cuda") < 0.01).to(torch.uint8)
    out = rtg_scan(rew, ended, 0.99)
    ref = O.compute_rtgs_tn(rew.cpu().numpy(), ended.cpu().numpy(), 0.99)
    assert np.array_equal(out.cpu().numpy(), ref), (T, N)
T, N = 512, 4096
rew = torch.randn((T, N), device="cuda"); ended = (torch.rand((T, N), device="cuda") < 0.01).to(torch.uint8); out = torch.empty_like(rew)
for _ in range(10): rtg_scan(rew, ended, 0.99, out=out)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()   # 32 launches per replay: the python launch path is not what is measured
with torch.cuda.graph(g):
    for _ in range(32): rtg_scan(rew, ended, 0.99, out=out)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): g.replay()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 320 * 1e3
print(f"rtg_scan {T}x{N}: {us:.2f} us -> {T*N*9/us/1e3:.1f} GB/s (bit-exact vs oracle on 4 shapes)")
