"""Actor / critic heads for the PPO loop (plain PyTorch-ROCm modules).

``resmlp512``  the reference's active nets: ``NetActor`` / ``NetCritic``
               (project_ppo/src/net_actor.py:16-144, net_critic.py:13-130): two residual blocks
               (16->512->16, then 32->512->32 on cat[obs, h]) with LeakyReLU(0.2); actor heads
               sigmoid (linear velocity) and tanh (angular velocity); critic head linear.
               Parameter names and the unused BatchNorm entries match the reference's state_dict so
               ``actor_iter*_step*.pth`` checkpoints (ppo.py:452-457) load in either direction.
``mlp64x2``    the 16-64-64 MLP BASELINE.json config 2 names.  The reference holds it only as
               dead code (``NetActor_old``, net_actor.py:147-189, crashes on construction) and in the
               vendored tutorial (graph_code/ppo_for_beginners/network.py:11-50); the actor here has
               NetActor_old's two heads, the critic the tutorial's linear output.
"""
import math

import torch
from torch import nn


class _SplitKLinearFn(torch.autograd.Function):
    """y = x W^T + b whose weight gradient dW = dy^T x is computed as a BATCHED GEMM over row chunks.

    With the PPO batch (T*N = 2.1 M rows) the plain wgrad is a [out x rows] x [rows x in] GEMM with a 64x64
    (or 16x64) output: a handful of tiles, i.e. a handful of the 256 CUs busy for milliseconds (measured
    1.7-2.6 ms per call, 65 % of the whole update; profiles/r01_bench_v0_kernel_stats.csv).  Splitting the
    reduction dimension into `chunks` independent GEMMs fills the chip; the partial [chunks,out,in] products
    are summed afterwards.  Same math, different summation order (fp32 round-off only)."""

    @staticmethod
    def forward(ctx, x, weight, bias, chunks):
        ctx.save_for_backward(x, weight)
        ctx.chunks = chunks
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        P = ctx.chunks
        rows = x.shape[0]
        dx = dy @ weight if ctx.needs_input_grad[0] else None
        dyc = dy.reshape(P, rows // P, dy.shape[1])
        xc = x.reshape(P, rows // P, x.shape[1])
        dW = torch.bmm(dyc.transpose(1, 2), xc).sum(0)
        db = dyc.sum(1).sum(0)
        return dx, dW, db, None


class Linear(nn.Linear):
    """nn.Linear (same parameters, same state_dict keys) with the split-K weight gradient for huge batches."""

    SPLIT_ROWS = 1 << 16   # use the batched wgrad from this many rows
    CHUNK_ROWS = 4096      # rows per partial GEMM

    def forward(self, x):
        rows = x.shape[0] if x.dim() == 2 else 0
        if rows >= self.SPLIT_ROWS and rows % self.CHUNK_ROWS == 0 and torch.is_grad_enabled() and self.weight.requires_grad:
            return _SplitKLinearFn.apply(x, self.weight, self.bias, rows // self.CHUNK_ROWS)
        return super().forward(x)


class ResBlock(nn.Module):
    """x -> leaky(x' + fc2(leaky(fc1(x)))), x' = x (or leaky(fc3 x) when widths differ); net_actor.py:16-53.
    bn1/bn2 exist in the reference's state_dict but are commented out of its forward (:44,:48)."""

    def __init__(self, f_in, f_out, n_neurons=512, actor_init=False):
        super().__init__()
        self.f_in, self.f_out = f_in, f_out
        self.fc1 = Linear(f_in, n_neurons)
        self.bn1 = nn.BatchNorm1d(n_neurons)
        self.fc2 = Linear(n_neurons, f_out)
        self.bn2 = nn.BatchNorm1d(f_out)
        if f_in != f_out:
            self.fc3 = Linear(f_in, f_out)
        if actor_init:  # net_actor.py:28,32 -- the critic keeps nn.Linear's default init (net_critic.py:24-28)
            nn.init.uniform_(self.fc1.weight, -1 / math.sqrt(f_in), 1 / math.sqrt(f_in))
            nn.init.uniform_(self.fc2.weight, -1 / math.sqrt(n_neurons), 1 / math.sqrt(n_neurons))
        self.act = nn.LeakyReLU(0.2)

    def forward(self, x):
        skip = x if self.f_in == self.f_out else self.act(self.fc3(x))
        return self.act(skip + self.fc2(self.act(self.fc1(x))))


class _ResTrunk(nn.Module):
    def __init__(self, in_dim, n_neurons, actor_init):
        super().__init__()
        self.bn1 = nn.BatchNorm1d(in_dim)  # unused in forward (net_actor.py:137), kept for checkpoint keys
        self.rb1 = ResBlock(in_dim, in_dim, n_neurons, actor_init)
        self.rb2 = ResBlock(2 * in_dim, 2 * in_dim, n_neurons, actor_init)

    def trunk(self, obs):
        if obs.dim() == 1:
            obs = obs.unsqueeze(0)
        h = self.rb1(obs)
        return self.rb2(torch.cat([obs, h], dim=-1))


class ResMLPActor(_ResTrunk):
    def __init__(self, in_dim=16, out_dim=2, n_neurons=512):
        super().__init__(in_dim, n_neurons, actor_init=True)
        self.out1 = Linear(2 * in_dim, out_dim - 1)
        self.out2 = Linear(2 * in_dim, out_dim - 1)
        nn.init.uniform_(self.out1.weight, -1 / math.sqrt(in_dim), 1 / math.sqrt(in_dim))            # net_actor.py:89
        nn.init.uniform_(self.out2.weight, -1 / math.sqrt(2 * in_dim), 1 / math.sqrt(2 * in_dim))    # net_actor.py:91

    def forward(self, obs):
        h = self.trunk(obs)
        return torch.cat([torch.sigmoid(self.out1(h)), torch.tanh(self.out2(h))], dim=-1)


class ResMLPCritic(_ResTrunk):
    def __init__(self, in_dim=16, out_dim=1, n_neurons=512):
        super().__init__(in_dim, n_neurons, actor_init=False)
        self.out = Linear(2 * in_dim, out_dim)

    def forward(self, obs):
        return self.out(self.trunk(obs))


class MLP64Actor(nn.Module):
    def __init__(self, in_dim=16, out_dim=2, hidden=64):
        super().__init__()
        self.layer1 = Linear(in_dim, hidden)
        self.layer2 = Linear(hidden, hidden)
        self.layer3 = Linear(hidden, out_dim - 1)
        self.layer4 = Linear(hidden, out_dim - 1)

    def forward(self, obs):
        if obs.dim() == 1:
            obs = obs.unsqueeze(0)
        h = torch.relu(self.layer2(torch.relu(self.layer1(obs))))
        return torch.cat([torch.sigmoid(self.layer3(h)), torch.tanh(self.layer4(h))], dim=-1)


class MLP64Critic(nn.Module):
    def __init__(self, in_dim=16, out_dim=1, hidden=64):
        super().__init__()
        self.layer1 = Linear(in_dim, hidden)
        self.layer2 = Linear(hidden, hidden)
        self.layer3 = Linear(hidden, out_dim)

    def forward(self, obs):
        if obs.dim() == 1:
            obs = obs.unsqueeze(0)
        return self.layer3(torch.relu(self.layer2(torch.relu(self.layer1(obs)))))


POLICIES = {"resmlp512": (ResMLPActor, ResMLPCritic), "mlp64x2": (MLP64Actor, MLP64Critic)}


def make_policy(name, obs_dim=16, act_dim=2):
    if name not in POLICIES:
        raise KeyError(f"unknown policy {name!r}; have {sorted(POLICIES)}")
    a, c = POLICIES[name]
    return a(obs_dim, act_dim), c(obs_dim, 1)
