#!/usr/bin/env python3
"""Dev tool: gradients of the split-bf16 pass against the f32-MFMA pass on the same batch, per parameter tensor (max |diff| / scale,
NaN count, where the worst entries are).  usage: python tools/verify/x3_grad_diff.py [n] [d]"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from navbot_ppo_amd import ppo
from test_gpu_bf16x3 import _batch, _nets, _grad

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda")
a, c = _nets(dev, d=d)
batch = _batch(n, n, dev, d=d)
up, g32, s32 = _grad(a, c, "f32", batch, dev)
_, gx3, sx3 = _grad(a, c, "bf16x3", batch, dev)
print("stats f32   ", s32.cpu().numpy())
print("stats bf16x3", sx3.cpu().numpy())
off = 0
names = [n_ for m in (a, c) for n_, _ in m.named_parameters()]
for name, prm in zip(names, up.fp.params):
    k = prm.numel()
    r, x = g32[off:off + k], gx3[off:off + k]
    bad = torch.isnan(x).sum().item()
    dlt = (r - x).abs()
    dlt[torch.isnan(dlt)] = 0
    sc = r.abs().max().item() + 1e-30
    w = int(dlt.argmax())
    print(f"{name:16s} {tuple(prm.shape)!s:10s} nan {bad:5d}  max|diff|/scale {dlt.max().item() / sc:9.2e}  at {w} (ref {r[w].item():+.4e} got {x[w].item():+.4e})"
          + (f"  first nan at {int(torch.isnan(x).nonzero()[0])}" if bad else ""))
    off += k
if os.environ.get("X3_SHOW"):
    import numpy as np
    np.set_printoptions(linewidth=200, precision=4)
    k0 = 64 * d
    print("actor db1 ref ", g32[k0:k0 + 64].cpu().numpy())
    print("actor db1 got ", gx3[k0:k0 + 64].cpu().numpy())
