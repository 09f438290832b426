#!/usr/bin/env python3
"""Dev tool: the float64 comparison of tools/bf16x3_error.py on KINK-FREE batches (tests/_kinks.py: samples within 2e-5 of a relu or
clip kink replaced), several seeds, pooled per parameter tensor -- the table behind the strict gate of tests/test_gpu_bf16x3.py
(VERDICT round 5, item 2).  usage: python tools/bf16x3_error_kinkfree.py [n] [seeds]"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from navbot_ppo_amd import ppo
from test_gpu_bf16x3 import _batch, _grad, _nets
from _kinks import replace_kink_samples

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128 * 300 + 7
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda")
names = ["a.W1", "a.b1", "a.W2", "a.b2", "a.w3", "a.b3", "a.w4", "a.b4", "c.W1", "c.b1", "c.W2", "c.b2", "c.w3", "c.b3"]
R = {k: [] for k in ("f32", "bf16x3", "torch32")}
M = {k: [] for k in ("f32", "bf16x3", "torch32")}
for seed in range(seeds):
    a, c = _nets(dev, seed=3 + seed)
    batch = _batch(n, 100 + n + seed, dev)
    obs, acts, logp, rtg, adv = batch
    nk = replace_kink_samples(a, c, obs, acts, logp, rtg, adv, 0.5)
    a64, c64 = copy.deepcopy(a).double(), copy.deepcopy(c).double()
    al, cl, _, _, _ = ppo.ppo_losses(a64, c64, obs.double(), acts.double(), logp.double(), rtg.double(), adv.double(),
                                     torch.tensor(0.5, dtype=torch.float64, device=dev), 0.2)
    g64 = torch.cat([t.reshape(-1) for t in torch.autograd.grad(al + cl, list(a64.parameters()) + list(c64.parameters()))])
    a32, c32 = copy.deepcopy(a), copy.deepcopy(c)
    al2, cl2, _, _, _ = ppo.ppo_losses(a32, c32, obs.float(), acts, logp, rtg, adv, torch.tensor(0.5, device=dev), 0.2)
    rows = {"torch32": torch.cat([t.reshape(-1) for t in torch.autograd.grad(al2 + cl2, list(a32.parameters()) + list(c32.parameters()))])}
    for arith in ("f32", "bf16x3"):
        up, g, st = _grad(a, c, arith, batch, dev)
        rows[arith] = g
    offs = np.cumsum([0] + [q.numel() for q in up.fp.params])
    for k, g in rows.items():
        R[k].append([(((g64[o:e] - g[o:e].double()) ** 2).mean().sqrt() / (g64[o:e].abs().max() + 1e-300)).item() for o, e in zip(offs[:-1], offs[1:])])
        M[k].append([((g64[o:e] - g[o:e].double()).abs().max() / (g64[o:e].abs().max() + 1e-300)).item() for o, e in zip(offs[:-1], offs[1:])])
    print(f"seed {seed}: {nk} kink samples replaced;  rms over all tensors  f32 {np.sqrt(np.mean(np.square(R['f32'][-1]))):.2e}  bf16x3 {np.sqrt(np.mean(np.square(R['bf16x3'][-1]))):.2e}  torch32 {np.sqrt(np.mean(np.square(R['torch32'][-1]))):.2e}")
print(f"n = {n}, {seeds} seeds, pooled over seeds (sqrt of the mean squared rms error; worst max error); columns: " + " ".join(names))
for k in R:
    r = np.sqrt(np.mean(np.square(np.array(R[k])), 0))
    m = np.max(np.array(M[k]), 0)
    print(f"  {k:8s} rms " + " ".join(f"{v:.1e}" for v in r) + f"   | all {np.sqrt(np.mean(r ** 2)):.2e}")
    print(f"  {'':8s} max " + " ".join(f"{v:.1e}" for v in m) + f"   | worst {m.max():.2e}")
rx, rf, rt = (np.sqrt(np.mean(np.square(np.array(R[k])), 0)) for k in ("bf16x3", "f32", "torch32"))
print("  bf16x3 / f32     " + " ".join(f"{v:7.2f}" for v in rx / rf) + f"   | all {np.sqrt(np.mean(rx ** 2)) / np.sqrt(np.mean(rf ** 2)):.2f}")
print("  bf16x3 / torch32 " + " ".join(f"{v:7.2f}" for v in rx / rt) + f"   | all {np.sqrt(np.mean(rx ** 2)) / np.sqrt(np.mean(rt ** 2)):.2f}")
if os.environ.get("PER_SEED"):
    for k in R:
        for sd, row in enumerate(R[k]):
            print(f"  {k:8s} seed {sd} rms " + " ".join(f"{v:.1e}" for v in row))
