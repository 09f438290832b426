#!/usr/bin/env python3
"""Dev tool: one update epoch of the 16-64-64 heads at the bench batch (2,097,152 samples), native f32 MFMA path vs the split-bf16
path (PPOConfig.update_arith), HIP events over 20 epochs each, and the one-off observation split.  usage: python tools/time_update_arith.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navbot_ppo_amd import nets, ppo

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512 * 4096
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
obs = torch.rand((n, 16), generator=g).to(dev)
acts = torch.stack([torch.rand(n, generator=g), torch.rand(n, generator=g) * 2 - 1], 1).to(dev)
logp = (-1.2 - 2.3 * torch.rand(n, generator=g)).to(dev)
rtg = (torch.randn(n, generator=g) * 60 + 20).to(dev)
adv = torch.randn(n, generator=g).to(dev)
for arith in ("f32", "bf16x3", "f32", "bf16x3"):
    torch.manual_seed(0)
    a, c = nets.make_policy("mlp64x2")
    a.to(dev), c.to(dev)
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="mlp64x2", update_arith=arith), None, dev)
    st = torch.zeros(8, device=dev)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    torch.cuda.synchronize()
    e[0].record()
    if up.bf16x3:
        up.prepare(obs)
    e[1].record()
    for _ in range(5):
        up._fused_epoch(obs, acts, logp, rtg, adv, 0.8, st)
    torch.cuda.synchronize()
    e[1].record()
    for _ in range(20):
        up._fused_epoch(obs, acts, logp, rtg, adv, 0.8, st)
    e[2].record()
    torch.cuda.synchronize()
    print(f"{arith:7s} n={n}: epoch {e[1].elapsed_time(e[2]) / 20 * 1e3:8.1f} us   (actor loss {st[0].item():.6f}, critic loss {st[4].item():.4f})")
