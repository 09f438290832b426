#!/usr/bin/env python3
"""Dev tool: 50 update epochs (navppo_mlp64_update_epoch) launched one by one vs replayed from one hipGraph.
usage: python tools/time_update_graph.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from navbot_ppo_amd import nets, ppo
dev = torch.device("cuda"); torch.manual_seed(0)
a, c = nets.make_policy("mlp64x2"); a.to(dev); c.to(dev)
up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="mlp64x2"), None, dev)
n = 512 * 4096
obs = torch.rand((n, 16), device=dev); acts = torch.rand((n, 2), device=dev); logp = -torch.rand(n, device=dev) - 1
rtg = torch.randn(n, device=dev) * 50; adv = torch.randn(n, device=dev)
hist = torch.zeros((50, 8), device=dev)
def epochs():
    for ep in range(50): up._fused_epoch(obs, acts, logp, rtg, adv, 0.8, hist[ep])
epochs(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); epochs(); epochs(); e1.record(); torch.cuda.synchronize()
print(f"one launch pair per epoch: {e0.elapsed_time(e1) / 100 * 1e3:.1f} us per epoch")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        epochs()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
print(f"hipGraph of 50 epochs:     {e0.elapsed_time(e1) / 100 * 1e3:.1f} us per epoch")
