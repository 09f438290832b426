#!/usr/bin/env python3
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from navbot_ppo_amd import _native, nets, ppo
_native.LIB_PATH = os.path.abspath("build/libnavsim_ppotiming.so")
dev = torch.device("cuda")
torch.manual_seed(0)
a, c = nets.make_policy("mlp64x2"); a.to(dev); c.to(dev)
up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="mlp64x2"), None, dev)
n = 512 * 4096
obs = torch.rand((n, 16), device=dev); acts = torch.rand((n, 2), device=dev); logp = -torch.rand(n, device=dev) - 1
rtg = torch.randn(n, device=dev) * 50; adv = torch.randn(n, device=dev)
L = _native.lib()
for _ in range(3): up._fused_loss_grad(obs, acts, logp, rtg, adv, 0.8)
torch.cuda.synchronize()
L.navppo_dbg_zero()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); up._fused_loss_grad(obs, acts, logp, rtg, adv, 0.8); e1.record(); torch.cuda.synchronize()
buf = np.zeros(16, dtype=np.int64); L.navppo_dbg_read.argtypes = [C.c_void_p]; L.navppo_dbg_read(buf.ctypes.data_as(C.c_void_p))
names = ["X load", "F1", "F2", "heads", "dH2", "B2", "G2+G1", "tail"]
tot = buf[:8].sum()
print(f"one epoch (both nets) {e0.elapsed_time(e1)*1e3:.0f} us ; workgroup 0 phase totals over both passes (us):")
for k, nm in enumerate(names): print(f"  {nm:8s} {buf[k]/100:8.1f}  {100*buf[k]/tot:5.1f}%")
