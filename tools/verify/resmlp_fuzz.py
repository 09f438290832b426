#!/usr/bin/env python3
"""Dev tool: the fused loss / gradient entry point of the 512-wide nets on many ragged batch sizes, against float64 autograd of the same
losses -- a sweep over the tile / wave / group boundaries of the hand-placed backward (csrc/ppo_resmlp512_bwd2s.h) and of the forward
kernels' slice pairs that the six sizes of tests/test_gpu_resmlp512.py do not visit.  Prints the worst tensor error relative to the
tensor's scale per size; exit code 1 above 2e-4 (the test suite's bound).
usage: python tools/verify/resmlp_fuzz.py [n_sizes] [f16]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from navbot_ppo_amd import nets, ppo
dev = torch.device("cuda")
torch.manual_seed(0)
a, c = nets.make_policy("resmlp512"); a.to(dev); c.to(dev)
up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="resmlp512"), None, dev)
a64, c64 = nets.make_policy("resmlp512"); a64.to(dev).double(); c64.to(dev).double()
a64.load_state_dict({k: v.double() for k, v in a.state_dict().items()}); c64.load_state_dict({k: v.double() for k, v in c.state_dict().items()})
n_sizes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
f16 = len(sys.argv) > 2 and sys.argv[2] == "f16"
g = torch.Generator().manual_seed(1)
fixed = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8192, 8193, 32 * 1024 - 1, 32 * 1024 + 1]
sizes = fixed + [int(torch.randint(1, 70000, (1,), generator=g)) for _ in range(max(0, n_sizes - len(fixed)))]
def kink_mask(net, x, eps=2e-5):
    """samples (float64 view) with a LeakyReLU pre-activation of either residual block within eps of its kink"""
    bad = torch.zeros(x.shape[0], dtype=torch.bool, device=x.device)
    inp = x
    for rb in (net.rb1, net.rb2):
        z1 = rb.fc1(inp)
        z2 = inp + rb.fc2(rb.act(z1))
        bad |= (z1.abs() < eps).any(1) | (z2.abs() < eps).any(1)
        inp = torch.cat([x, rb.act(z2)], 1) if rb is net.rb1 else None
    return bad


worst_all = 0.0
n_rep_all = 0
for n in sizes:
    obs = torch.rand((n, 16), device=dev) * 2 - 0.5
    if f16:
        obs = obs.half()
    acts = torch.rand((n, 2), device=dev); logp = -torch.rand(n, device=dev) - 1
    rtg = torch.randn(n, device=dev) * 5; adv = torch.randn(n, device=dev)
    with torch.no_grad():   # the samples on a kink of the piecewise-smooth loss are replaced by one that is not (tests/_kinks.py has the why)
        x64 = obs.double()
        v64 = torch.tensor(0.8, device=dev, dtype=torch.float64)
        bad = kink_mask(a64, x64) | kink_mask(c64, x64)
        ratio = torch.exp(ppo.gaussian_log_prob(a64(x64), acts.double(), v64) - logp.double())
        bad |= ((ratio - 0.8).abs() < 2e-5) | ((ratio - 1.2).abs() < 2e-5)
        if bool(bad.any()) and not bool(bad.all()):
            good = int((~bad).nonzero()[0])
            for t in (obs, acts, logp, rtg, adv):
                t[bad] = t[good].clone()
        n_rep_all += int(bad.sum())
    up._fused_loss_grad(obs, acts, logp, rtg, adv, 0.8)
    got = up.fp.grad.clone().double()
    for p in list(a64.parameters()) + list(c64.parameters()):
        p.grad = None
    la, lc, *_ = ppo.ppo_losses(a64, c64, obs.double(), acts.double(), logp.double(), rtg.double(), adv.double(), torch.tensor(0.8, device=dev, dtype=torch.float64), 0.2)
    (la + lc).backward()
    worst, name = 0.0, ""
    off = 0
    for net, mod in (("actor", a64), ("critic", c64)):
        for k, p in mod.named_parameters():
            if ".bn" in "." + k or k.startswith("bn"):
                continue
            ref = p.grad.reshape(-1)
            seg = got[off:off + ref.numel()]
            off += ref.numel()
            err = float((seg - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
            if err > worst:
                worst, name = err, f"{net}.{k}"
    assert off == got.numel(), (off, got.numel())
    worst_all = max(worst_all, worst)
    print(f"n={n:6d}  worst tensor error / scale {worst:.2e}  ({name})")
print(f"worst over {len(sizes)} sizes: {worst_all:.2e}   ({n_rep_all} samples replaced in all)")
sys.exit(1 if worst_all > 2e-4 else 0)
