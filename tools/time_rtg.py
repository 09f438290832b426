"""Dev tool: navsim_rtg_scan timing (hipGraph of 32 launches, HIP events) for the T-split kernel and the serial (bit-exact) one,
plus the share of stores that differ from the oracle (contract: <= 1 float32 ulp)."""
import sys, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from navbot_ppo_amd.env import rtg_scan
from oracle import navsim_oracle as O
for T, N, p in ((512, 4096, .01), (512, 4096, .001), (512, 4096, 0.), (512, 4097, .01), (100, 37, .01), (1300, 640, .002)):
    rew = torch.randn((T, N), device="cuda") * 20; ended = (torch.rand((T, N), device="cuda") < p).to(torch.uint8)
    out = rtg_scan(rew, ended, 0.99).cpu().numpy()
    ref = O.compute_rtgs_tn(rew.cpu().numpy(), ended.cpu().numpy(), 0.99)
    d = np.abs(out.view(np.int32).astype(np.int64) - ref.view(np.int32))
    print(f"T={T} N={N} p_end={p}: max ulp {d.max()}, differing stores {(d != 0).sum()} of {d.size}")
    assert d.max() <= 1
T, N = 512, 4096
rew = torch.randn((T, N), device="cuda"); ended = (torch.rand((T, N), device="cuda") < 0.01).to(torch.uint8); out = torch.empty_like(rew)
for mode in ("0", "1"):
    ex = mode == "1"
    for _ in range(10): rtg_scan(rew, ended, 0.99, out=out, exact=ex)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()   # 32 launches per replay: the python launch path is not what is measured
    with torch.cuda.graph(g):
        for _ in range(32): rtg_scan(rew, ended, 0.99, out=out, exact=ex)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 320 * 1e3
    print(f"rtg_scan {T}x{N} {'serial (exact)' if mode == '1' else 'T-split'}: {us:.2f} us -> {T*N*9/us/1e3:.1f} GB/s")
