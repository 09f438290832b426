import ctypes as C, os, sys
sys.path.insert(0, __import__("os").getcwd())
import torch
from navbot_ppo_amd import nets, ppo
dev = torch.device("cuda"); torch.manual_seed(0)
a, c = nets.make_policy("mlp64x2"); a.to(dev); c.to(dev)
up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="mlp64x2"), None, dev)
for n in (256*4096//4, 512*4096//4, 512*4096//2, 512*4096):
    obs = torch.rand((n, 16), device=dev); acts = torch.rand((n, 2), device=dev); logp = -torch.rand(n, device=dev) - 1
    rtg = torch.randn(n, device=dev) * 50; adv = torch.randn(n, device=dev)
    for _ in range(3): up._fused_loss_grad(obs, acts, logp, rtg, adv, 0.8)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): up._fused_loss_grad(obs, acts, logp, rtg, adv, 0.8)
    e1.record(); torch.cuda.synchronize()
    print(n, f"{e0.elapsed_time(e1)/20*1e3:.0f} us per epoch (both nets), tiles per wave {n/32/2048:.1f}")
