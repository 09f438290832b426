#!/usr/bin/env python3
"""Dev tool: the batch-independent part of one mlp64x2 update epoch (navppo_mlp64_update_epoch: mlp64_pass_both + reduce_adam) -- epochs
at small batches, HIP events over 200 back-to-back epochs.  usage: python tools/time_update_fixed.py [lib.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navbot_ppo_amd import _native, nets, ppo
if len(sys.argv) > 1:
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device("cuda"); torch.manual_seed(0)
a, c = nets.make_policy("mlp64x2"); a.to(dev); c.to(dev)
up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="mlp64x2"), None, dev)
st = torch.zeros(8, device=dev)
for n in (32, 2048, 65536, 131072, 262144, 524288):
    obs = torch.rand((n, 16), device=dev); acts = torch.rand((n, 2), device=dev); logp = -torch.rand(n, device=dev) - 1
    rtg = torch.randn(n, device=dev) * 50; adv = torch.randn(n, device=dev)
    for _ in range(5): up._fused_epoch(obs, acts, logp, rtg, adv, 0.8, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): up._fused_epoch(obs, acts, logp, rtg, adv, 0.8, st)
    e1.record(); torch.cuda.synchronize()
    print(f"n={n:7d} tiles per wave and net {n / 32 / 2048:6.2f}: {e0.elapsed_time(e1) / 200 * 1e3:7.1f} us per epoch", flush=True)
