#!/usr/bin/env python3
"""Dev tool (needs a library built with -DMLP64_DEBUG_Z: tools/build_mlp64_variant.py dbgz -DMLP64_DEBUG_Z, NAVSIM_LIB=build/libnavsim_dbgz.so):
the pre-activations z3 / z4 of the heads as the f32-MFMA pass and the split-bf16 pass compute them, against float64 PyTorch -- which
forward is closer, per net.  usage: NAVSIM_LIB=build/libnavsim_dbgz.so python tools/verify/x3_forward_error.py [n]"""
import copy, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from test_gpu_bf16x3 import _batch, _nets, _grad

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128 * 300 + 7
dev = torch.device("cuda")
for seed in range(3):
    a, c = _nets(dev, seed=3 + seed)
    batch = _batch(n, 100 + n + seed, dev)
    x = batch[0].double()
    def heads(net, k):
        W1, b1, W2, b2 = [p.detach().double() for p in list(net.parameters())[:4]]
        h2 = torch.relu(torch.relu(x @ W1.T + b1) @ W2.T + b2)
        ps = [p.detach().double() for p in list(net.parameters())[4:]]
        return [h2 @ ps[2 * i].T.squeeze(-1) + ps[2 * i + 1] if ps[2 * i].dim() == 1 else (h2 @ ps[2 * i].T).squeeze(-1) + ps[2 * i + 1] for i in range(k)]
    za = heads(a, 2); zc = heads(c, 1)
    with torch.no_grad():
        x32 = batch[0].float()
        def heads32(net, k):
            ps = list(net.parameters())
            h2 = torch.relu(torch.relu(x32 @ ps[0].T + ps[1]) @ ps[2].T + ps[3])
            return [(h2 @ ps[4 + 2 * i].T).squeeze(-1) + ps[5 + 2 * i] for i in range(k)]
        t = heads32(a, 2)
        f0 = lambda e: f"rms {e.pow(2).mean().sqrt().item():.2e} mean {e.mean().item():+.2e}"
        print(f"seed {seed} torch32 actor z3: {f0(t[0].double() - za[0])}   actor z4: {f0(t[1].double() - za[1])}")
    for arith in ("f32", "bf16x3"):
        up, g, st = _grad(a, c, arith, batch, dev)
        ws = up._ws
        pa, pc = 5378, 5313
        za_k = ws[200 * pa:200 * pa + 2 * n].double()
        zc_k = ws[256 * pa + 200 * pc:256 * pa + 200 * pc + n].double()
        e3, e4, ec = (za_k[:n] - za[0]), (za_k[n:] - za[1]), (zc_k - zc[0])
        # the output gradients g3 = dL/dz3, g4 = dL/dz4 per sample against float64 autograd of the actor loss
        gk = ws[200 * pa + 2 * n:200 * pa + 4 * n].double()
        zz = [z.clone().requires_grad_(True) for z in za]
        from navbot_ppo_amd import ppo
        mean = torch.stack([torch.sigmoid(zz[0]), torch.tanh(zz[1])], 1)
        lp = ppo.gaussian_log_prob(mean, batch[1].double(), torch.tensor(0.5, dtype=torch.float64, device=dev))
        ratio = torch.exp(lp - batch[2].double())
        A = batch[4].double()
        L = (-torch.min(ratio * A, torch.clamp(ratio, 0.8, 1.2) * A)).mean()
        gt = torch.autograd.grad(L, zz)
        for i in range(2):
            eg = gk[i * n:(i + 1) * n] - gt[i]
            print(f"        {arith:7s} g{3 + i}: sum err / |sum| {eg.sum().item() / abs(gt[i].sum().item()):+.2e}   rms err / rms {eg.pow(2).mean().sqrt().item() / gt[i].pow(2).mean().sqrt().item():.2e}   (sum {gt[i].sum().item():+.3e}, sum|g| {gt[i].abs().sum().item():.3e})")
        f = lambda e, z: f"rms {e.pow(2).mean().sqrt().item():.2e} mean {e.mean().item():+.2e} (|z| rms {z.pow(2).mean().sqrt().item():.1f})"
        print(f"seed {seed} {arith:7s} actor z3: {f(e3, za[0])}   actor z4: {f(e4, za[1])}   critic z3: {f(ec, zc[0])}")
