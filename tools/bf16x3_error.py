#!/usr/bin/env python3
"""Dev tool: gradients of the fused 16-64-64 update on both arithmetics (f32-input MFMA / split bf16 "bf16x3") against float64 autograd of
the same losses, per parameter tensor, as a fraction of the tensor's gradient scale -- the table behind tests/test_gpu_bf16x3.py and
DESIGN.md 5e (profiles/r05_bf16x3_error.txt).  usage: python tools/bf16x3_error.py"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from navbot_ppo_amd import ppo
from test_gpu_bf16x3 import _batch, _grad, _nets

dev = torch.device("cuda")
names = ["a.W1", "a.b1", "a.W2", "a.b2", "a.w3", "a.b3", "a.w4", "a.b4", "c.W1", "c.b1", "c.W2", "c.b2", "c.w3", "c.b3"]
print("error against float64 autograd / max |gradient| of the tensor; columns: " + " ".join(names))
for half in (False, True):
    for n in (1000, 128 * 300 + 7, 1 << 17, 512 * 4096):
        for seed in (0, 1):
            a, c = _nets(dev, seed=3 + seed)
            batch = _batch(n, 100 + n + seed, dev, half)
            obs, acts, logp, rtg, adv = batch
            a64, c64 = copy.deepcopy(a).double(), copy.deepcopy(c).double()
            al, cl, _, _, _ = ppo.ppo_losses(a64, c64, obs.double(), acts.double(), logp.double(), rtg.double(), adv.double(),
                                             torch.tensor(0.5, dtype=torch.float64, device=dev), 0.2)
            g64 = torch.cat([t.reshape(-1) for t in torch.autograd.grad(al + cl, list(a64.parameters()) + list(c64.parameters()))])
            # float32 autograd as a third column: what PyTorch's own float32 evaluation is off by
            a32, c32 = copy.deepcopy(a), copy.deepcopy(c)
            al2, cl2, _, _, _ = ppo.ppo_losses(a32, c32, obs.float(), acts, logp, rtg, adv, torch.tensor(0.5, device=dev), 0.2)
            gt = torch.cat([t.reshape(-1) for t in torch.autograd.grad(al2 + cl2, list(a32.parameters()) + list(c32.parameters()))])
            rows = {}
            for arith in ("f32", "bf16x3"):
                up, g, st = _grad(a, c, arith, batch, dev)
                rows[arith] = g
            rows["torch32"] = gt
            offs = np.cumsum([0] + [q.numel() for q in up.fp.params])
            print(f"n = {n} {'float16' if half else 'float32'} rows, seed {seed}")
            for k, g in rows.items():
                mx = [((g64[o:e] - g[o:e].double()).abs().max() / (g64[o:e].abs().max() + 1e-300)).item() for o, e in zip(offs[:-1], offs[1:])]
                rm = [(((g64[o:e] - g[o:e].double()) ** 2).mean().sqrt() / (g64[o:e].abs().max() + 1e-300)).item() for o, e in zip(offs[:-1], offs[1:])]
                print(f"  {k:8s} max " + " ".join(f"{v:.1e}" for v in mx) + f"   | worst {max(mx):.2e}")
                print(f"  {'':8s} rms " + " ".join(f"{v:.1e}" for v in rm) + f"   | all   {np.sqrt(np.mean(np.square(rm))):.2e}")
