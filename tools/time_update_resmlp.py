#!/usr/bin/env python3
"""Dev tool: times navppo_resmlp512_update_epoch (both nets, one epoch) on a BASELINE configs[1]-sized batch.
usage: python tools/time_update_resmlp.py [n_samples] [epochs]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from navbot_ppo_amd import nets, ppo
dev = torch.device("cuda"); torch.manual_seed(0)
a, c = nets.make_policy("resmlp512"); a.to(dev); c.to(dev)
up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="resmlp512"), None, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512 * 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
obs = torch.rand((n, 16), device=dev); acts = torch.rand((n, 2), device=dev); logp = -torch.rand(n, device=dev) - 1
rtg = torch.randn(n, device=dev) * 50; adv = torch.randn(n, device=dev)
st = torch.zeros(8, device=dev)
for _ in range(2): up._fused_epoch(obs, acts, logp, rtg, adv, 0.8, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): up._fused_epoch(obs, acts, logp, rtg, adv, 0.8, st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
flop = 2 * 2 * 155648 * n   # both nets, MACs per sample incl. the recomputed hidden layer
print(f"n={n}: {ms:.3f} ms per epoch (both nets) = {flop / ms / 1e9:.1f} TFLOP/s = {flop / ms / 1e9 / 157.3:.3f} of the f32-MFMA peak")
V = up._fused_value(obs); torch.cuda.synchronize()
e0.record()
for _ in range(reps): V = up._fused_value(obs)
e1.record(); torch.cuda.synchronize()
print(f"value: {e0.elapsed_time(e1) / reps:.3f} ms")
