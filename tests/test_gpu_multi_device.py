"""Multi-GPU readiness that runs itself: on a box that shows TWO OR MORE devices these tests put one rank
on each of two devices over torch's `nccl` backend (= RCCL over xGMI) and check SURVEY.md 8(e)'s parity conditions on real hardware:
the sharded update equals the single-rank update, every env's trajectory is independent of the sharding, and `bench.py --gpus 2`
prints the contract line with rccl_ranks = 2 and a `dist` block.  On the 1-GPU boxes of the build rounds they skip cleanly (the gloo
tests of test_gpu_parity.py / test_gpu_ppo.py / test_ppo_cpu.py cover the same code with two ranks on one device or on the CPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
two_devices = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible devices (RCCL refuses two ranks on one)")


def _rccl_two_rank_worker(rank, world, port, path, policy, overlap):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("NAVBOT_DIST_BACKEND", None)
    os.environ.pop("NAVBOT_DIST_FORCE", None)
    from navbot_ppo_amd import nets, ppo
    dev = torch.device(f"cuda:{rank}")
    torch.cuda.set_device(dev)
    ctx = ppo.DistCtx(device=str(dev))
    assert ctx.enabled and ctx.backend == "nccl" and ctx.world == 2 and ctx.rccl_version, (ctx.enabled, ctx.backend, ctx.world)
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "g7_update.npz"))
    torch.manual_seed(100 + rank)   # replicas start different: the broadcast at construction must make them equal
    a, c = nets.make_policy(policy)
    a.to(dev), c.to(dev)
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(n_updates_per_iteration=5, policy=policy, overlap_allreduce=overlap), ctx, dev)
    assert up.fused_mlp64 or up.fused_resmlp512
    lo, hi = ctx.shard(512)
    cu = lambda k: torch.from_numpy(d[k][lo:hi]).to(dev)
    up.update(cu("obs"), cu("acts"), cu("logp"), cu("rtgs"), torch.tensor(0.8, device=dev))
    torch.cuda.synchronize(dev)
    torch.save({"flat": up.fp.flat.cpu(), "rccl": ctx.rccl_version}, f"{path}.{rank}")
    ctx.barrier()
    torch.distributed.destroy_process_group()


@two_devices
@pytest.mark.parametrize("policy,overlap", [("mlp64x2", False), ("mlp64x2", True), ("resmlp512", False)])
def test_two_ranks_over_rccl_equal_the_single_rank_update(tmp_path, policy, overlap):
    """Two ranks on two devices, shards of the G7 batch, fused passes -> all-reduce of the flat gradient over RCCL -> Adam: both replicas
    end bit-identical to each other and within 3e-6 of the single-rank path on the whole batch (reduction order; SURVEY.md 8(e):
    <= 1e-6 relative on the gradients)."""
    from _ranks import spawn_ranks
    from navbot_ppo_amd import nets, ppo
    path = str(tmp_path / "rccl2")
    spawn_ranks(_rccl_two_rank_worker, 2, lambda port: (2, port, path, policy, overlap))
    r0, r1 = torch.load(path + ".0"), torch.load(path + ".1")
    assert r0["rccl"] and torch.equal(r0["flat"], r1["flat"])
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "g7_update.npz"))
    torch.manual_seed(100)
    a, c = nets.make_policy(policy)
    a.cuda(), c.cuda()
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(n_updates_per_iteration=5, policy=policy), None, torch.device("cuda:0"))
    cu = lambda k: torch.from_numpy(d[k]).cuda()
    up.update(cu("obs"), cu("acts"), cu("logp"), cu("rtgs"), torch.tensor(0.8, device="cuda"))
    np.testing.assert_allclose(r0["flat"].numpy(), up.fp.flat.cpu().numpy(), rtol=0, atol=3e-6)


def _rollout_shard_worker(rank, world, port, path, n_total, T):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("NAVBOT_DIST_BACKEND", None)
    from navbot_ppo_amd import ppo
    from navbot_ppo_amd.env import VecEnv
    dev = torch.device(f"cuda:{rank}")
    torch.cuda.set_device(dev)
    ctx = ppo.DistCtx(device=str(dev))
    lo, hi = ctx.shard(n_total)
    env = VecEnv(hi - lo, map="stage_1", max_episode_steps=40, env_id_base=lo, device=str(dev))
    tr = ppo.PPOTrainer(env, ppo.PPOConfig(policy="mlp64x2", rollout_len=T, n_updates_per_iteration=2, seed=5), ctx)
    tr.rollout()
    torch.cuda.synchronize(dev)
    torch.save({k: getattr(tr, k).cpu() for k in ("obs_buf", "act_buf", "rew_buf", "ended_buf", "rtg_buf")}, f"{path}.{rank}")
    ctx.barrier()
    torch.distributed.destroy_process_group()


@two_devices
def test_env_trajectories_do_not_depend_on_the_sharding(tmp_path):
    """the same global seed and env-id keyed Philox streams give every env the same trajectory whether the 1024 envs
    live on one device or as two shards of 512 on two -- observations, actions, rewards, episode ends and returns bit for bit."""
    from _ranks import spawn_ranks
    from navbot_ppo_amd import ppo
    from navbot_ppo_amd.env import VecEnv
    n_total, T = 1024, 96
    path = str(tmp_path / "shards")
    spawn_ranks(_rollout_shard_worker, 2, lambda port: (2, port, path, n_total, T))
    parts = [torch.load(f"{path}.{r}") for r in range(2)]
    env = VecEnv(n_total, map="stage_1", max_episode_steps=40, device="cuda:0")
    tr = ppo.PPOTrainer(env, ppo.PPOConfig(policy="mlp64x2", rollout_len=T, n_updates_per_iteration=2, seed=5))
    tr.rollout()
    for k in ("obs_buf", "act_buf", "rew_buf", "ended_buf", "rtg_buf"):
        whole = getattr(tr, k).cpu()
        both = torch.cat([p[k] for p in parts], 1)
        assert torch.equal(whole, both), k


@two_devices
def test_bench_two_ranks_over_rccl():
    """bench.py launched exactly as the driver launches N = 2: the contract line carries rccl_ranks = 2 and the `dist` block a first
    scaling run is read from."""
    from _ranks import free_port
    env = dict(os.environ)
    env.pop("NAVBOT_DIST_BACKEND", None)
    env.pop("NAVBOT_DIST_FORCE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras"]
    out = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    js = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert js["n_gpus"] == 2 and js["rccl_ranks"] == 2 and js["dist_backend"] == "nccl" and js["scaling"] == "weak"
    assert js["config"]["n_envs_total"] == 2 * js["config"]["n_envs_per_gpu"] and js["value"] > 1e6
    d = js["dist"]
    assert d["ranks"] == 2 and d["backend"] == "nccl" and d["allreduce_us"] > 0 and d["epoch_us_with_allreduce"] > 0


def test_this_file_skips_cleanly_on_one_device():
    """(The 1-GPU boxes run exactly this: the three tests above are skipped, nothing errors at import or collection.)"""
    assert torch.cuda.device_count() >= 1
