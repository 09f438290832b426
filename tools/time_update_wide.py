import os, sys, torch
sys.path.insert(0, '/root/repo') if os.path.exists('/root/repo/navbot_ppo_amd') else sys.path.insert(0, os.getcwd())
from navbot_ppo_amd import nets, ppo
n = 512 * 4096
dev = torch.device("cuda")
for d, half in ((42, False), (42, True), (16, True)):
    g = torch.Generator().manual_seed(1)
    obs = torch.rand((n, d), generator=g).to(dev)
    if half: obs = obs.half()
    acts = torch.stack([torch.rand(n, generator=g), torch.rand(n, generator=g) * 2 - 1], 1).to(dev)
    logp = (-1.2 - 2.3 * torch.rand(n, generator=g)).to(dev); rtg = (torch.randn(n, generator=g) * 60 + 20).to(dev); adv = torch.randn(n, generator=g).to(dev)
    for arith in ("f32", "bf16x3"):
        torch.manual_seed(0)
        a, c = nets.make_policy("mlp64x2", d); a.to(dev), c.to(dev)
        up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="mlp64x2", update_arith=arith), None, dev)
        st = torch.zeros(8, device=dev)
        if up.bf16x3: up.prepare(obs)
        for _ in range(5): up._fused_epoch(obs, acts, logp, rtg, adv, 0.8, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): up._fused_epoch(obs, acts, logp, rtg, adv, 0.8, st)
        e1.record(); torch.cuda.synchronize()
        print(f"D={d} half={half} {arith}: epoch {e0.elapsed_time(e1) / 10 * 1e3:.1f} us   (actor loss {st[0].item():.6f}, critic loss {st[4].item():.4f})")
