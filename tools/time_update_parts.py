import sys, time
sys.path.insert(0, "/root/repo")
import torch
from navbot_ppo_amd import nets, ppo
dev = torch.device("cuda"); torch.manual_seed(0)
a, c = nets.make_policy("mlp64x2"); a.to(dev); c.to(dev)
up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="mlp64x2"), None, dev)
n = 512*4096
obs = torch.rand((n, 16), device=dev); acts = torch.rand((n, 2), device=dev); logp = -torch.rand(n, device=dev) - 1
rtg = torch.randn(n, device=dev) * 50
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
with torch.no_grad():
    print("V0 ms", t(lambda: c(obs).squeeze(-1)))
    V0 = c(obs).squeeze(-1)
    print("normalise ms", t(lambda: ppo.normalise_advantages(rtg - V0, None)))
print("update total ms", t(lambda: up.update(obs, acts, logp, rtg, torch.tensor(0.8, device=dev)), 3))
