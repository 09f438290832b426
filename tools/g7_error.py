#!/usr/bin/env python3
"""How far is the fused resmlp512 update from the reference's own learn() (golden G7: per-epoch losses, weights after 4 Adam
epochs)?  Prints the achieved errors the bounds of tests/test_gpu_resmlp512.py::test_g7_reference_update_through_the_fused_kernels
are set from (round-3 review: 'measure and state the actual error')."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from navbot_ppo_amd import nets, ppo
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g7_update.npz"))
dev = torch.device("cuda")
for fused in (True, False):
    a, c = nets.make_policy("resmlp512")
    for mod, pre in ((a, "ia/"), (c, "ic/")):
        sd = mod.state_dict()
        with torch.no_grad():
            for k in sd:
                if pre + k in d: sd[k].copy_(torch.from_numpy(d[pre + k]))
    a.to(dev), c.to(dev)
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(n_updates_per_iteration=int(d["epochs"]), fused_update=fused), None, dev)
    t = lambda k: torch.from_numpy(d[k]).to(dev)
    up.update(t("obs"), t("acts"), t("logp"), t("rtgs"), torch.tensor(0.8, device=dev))
    h = up.loss_history.cpu().numpy()
    print(f"fused={fused}: n={d['obs'].shape[0]} epochs={int(d['epochs'])}")
    print("  actor loss rel err per epoch ", np.abs(h[:, 0] - d["actor_losses"]) / np.abs(d["actor_losses"]))
    print("  critic loss rel err per epoch", np.abs(h[:, 1] - d["critic_losses"]) / np.abs(d["critic_losses"]))
    sa, sc = a.state_dict(), c.state_dict()
    worst = (0, "")
    for k in d.files:
        if k.startswith("fa/") or k.startswith("fc/"):
            got = (sa if k.startswith("fa/") else sc)[k[3:]].cpu().numpy()
            init = d[("ia/" if k.startswith("fa/") else "ic/") + k[3:]]
            step = np.abs(d[k] - init).max(); err = np.abs(got - d[k]).max(); scale = np.abs(d[k]).max()
            print(f"  {k:28s} max|w|={scale:9.3e} max step={step:9.3e} max err={err:9.3e} err/step={err/step:8.2e} err/scale={err/scale:8.2e}")
            worst = max(worst, (err / step, k))
    print("  worst err/step", worst)
