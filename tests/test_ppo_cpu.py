"""PPO host logic on CPU: parity of the nets / action log-prob / update against golden vectors recorded
from the reference (G6, G7, G8), and the world_size-2 gloo data-parallel path."""
import json
import math
import os

import numpy as np
import pytest
import torch

from navbot_ppo_amd import nets, ppo

G = os.path.join(os.path.dirname(__file__), "golden")


def _load_init(mod, d, prefix):
    sd = mod.state_dict()
    with torch.no_grad():
        for k in sd:
            if prefix + k in d:
                sd[k].copy_(torch.from_numpy(d[prefix + k]))
            else:
                assert "num_batches_tracked" in k, k


def test_g8_state_dict_keys_shapes_and_param_counts():
    ref = json.load(open(os.path.join(G, "g8_nets.json")))
    a, c = nets.make_policy("resmlp512")
    for mod, key in ((a, "actor"), (c, "critic")):
        got = [[k, list(v.shape), str(v.dtype)] for k, v in mod.state_dict().items()]
        assert got == ref[key]  # same keys, same order, same shapes: checkpoints are interchangeable
    assert sum(p.numel() for p in a.parameters()) == ref["actor_params"] == 52466
    assert sum(p.numel() for p in c.parameters()) == ref["critic_params"] == 52433
    a2, c2 = nets.make_policy("mlp64x2")
    assert sum(p.numel() for p in a2.parameters()) == 5378 and sum(p.numel() for p in c2.parameters()) == 5313


def test_g8_forward_matches_reference_nets():
    d = np.load(os.path.join(G, "g7_update.npz"))
    a, c = nets.make_policy("resmlp512")
    _load_init(a, d, "ia/")
    _load_init(c, d, "ic/")
    obs = torch.from_numpy(d["obs"])
    with torch.no_grad():
        np.testing.assert_allclose(a(obs).numpy(), d["a_out"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(c(obs).numpy(), d["c_out"], rtol=1e-5, atol=1e-6)
        out = a(obs)
    assert out.shape == (len(obs), 2) and (out[:, 0] >= 0).all() and (out[:, 0] <= 1).all() and (out[:, 1].abs() <= 1).all()
    assert a(obs[0]).shape == (1, 2)  # 1-D observation gets a batch axis (net_actor.py:108-112)


def test_actor_init_ranges():
    torch.manual_seed(0)
    a, c = nets.make_policy("resmlp512")
    assert a.rb1.fc1.weight.abs().max() <= 1 / math.sqrt(16) and a.rb1.fc2.weight.abs().max() <= 1 / math.sqrt(512)
    assert a.out1.weight.abs().max() <= 1 / math.sqrt(16) and a.out2.weight.abs().max() <= 1 / math.sqrt(32)
    assert a.rb1.fc1.weight.abs().max() > 0.9 / math.sqrt(16)  # uniform over the whole range, not kaiming's


def test_g6_log_prob_of_clamped_action_and_decay_table():
    d = np.load(os.path.join(G, "g6_action.npz"))
    rec = d["rec"]
    mean = torch.tensor(rec[:, 0:2], dtype=torch.float32)
    act = torch.tensor(rec[:, 3:5], dtype=torch.float32)
    var = torch.tensor(rec[:, 2], dtype=torch.float32)
    lp = ppo.gaussian_log_prob(mean, act, var)
    np.testing.assert_allclose(lp.numpy(), rec[:, 5], rtol=1e-5, atol=1e-5)
    assert (act[:, 0] >= 0).all() and (act[:, 0] <= 1).all() and (act[:, 1].abs() <= 1).all()
    assert ((act[:, 0] == 0) | (act[:, 0] == 1) | (act[:, 1].abs() == 1)).float().mean() > 0.2  # clamping is common
    # ppo.py:694-695: decay only at episode step 0, after 50000 steps, while var >= 0.1; whole matrix scaled
    for t_step, t_so_far, v0, v1, off in d["decay"]:
        expect = v0 * 0.995 if (t_step == 0 and t_so_far > 50000 and np.float32(v0) >= np.float32(0.1)) else v0
        assert abs(v1 - np.float32(expect)) < 1e-6 and off == 0.0


def test_exploration_decay_schedule_in_trainer():
    cfg = ppo.PPOConfig()
    t = ppo.PPOTrainer.__new__(ppo.PPOTrainer)
    t.cfg, t.var = cfg, torch.tensor(0.8)
    t.env = type("E", (), {"N": 4})()
    t.t_so_far, t.episode_starts = 50000, 8
    t._decay_exploration()
    assert float(t.var) == pytest.approx(0.8)          # not yet past 50000 (strict >)
    t.t_so_far, t.episode_starts = 50001, 8           # 8 starts over 4 envs = 2 decays
    t._decay_exploration()
    assert float(t.var) == pytest.approx(0.8 * 0.995 ** 2, rel=1e-6)
    t.var.fill_(0.0999)
    t.episode_starts = 40
    t._decay_exploration()
    assert float(t.var) == pytest.approx(0.0999)       # below the floor: frozen


def test_normalise_advantages_matches_torch():
    torch.manual_seed(1)
    a = torch.randn(1000) * 37 + 5
    want = (a - a.mean()) / (a.std() + 1e-10)
    np.testing.assert_allclose(ppo.normalise_advantages(a).numpy(), want.numpy(), rtol=1e-5, atol=1e-6)


def test_g7_update_matches_reference_learn():
    """Reference PPO.learn run on a fixed batch (rollout stubbed out) vs PPOUpdater on the same batch and
    the same initial weights: evaluate(), per-epoch losses, and the weights after 4 Adam epochs."""
    d = np.load(os.path.join(G, "g7_update.npz"))
    a, c = nets.make_policy("resmlp512")
    _load_init(a, d, "ia/")
    _load_init(c, d, "ic/")
    cfg = ppo.PPOConfig(n_updates_per_iteration=int(d["epochs"]))
    up = ppo.PPOUpdater(a, c, cfg, None, torch.device("cpu"))
    obs, acts = torch.from_numpy(d["obs"]), torch.from_numpy(d["acts"])
    logp, rtgs = torch.from_numpy(d["logp"]), torch.from_numpy(d["rtgs"])
    var = torch.tensor(0.8)
    with torch.no_grad():
        _, _, _, lp0, V0 = ppo.ppo_losses(a, c, obs, acts, logp, rtgs, torch.zeros_like(rtgs), var, 0.2)
    np.testing.assert_allclose(V0.numpy(), d["V0"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lp0.numpy(), d["lp0"], rtol=1e-5, atol=1e-5)
    stats = up.update(obs, acts, logp, rtgs, var)
    hist = up.loss_history.numpy()
    np.testing.assert_allclose(hist[:, 0], d["actor_losses"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(hist[:, 1], d["critic_losses"], rtol=2e-4)
    assert stats["approx_kl"] == pytest.approx(float(d["approx_kl"]), rel=2e-2, abs=1e-5)
    assert stats["clip_frac"] == pytest.approx(float(d["clip_frac"]), abs=5e-3)
    sa, sc = a.state_dict(), c.state_dict()
    n = 0
    for k in d.files:
        if k.startswith("fa/") or k.startswith("fc/"):
            got = (sa if k.startswith("fa/") else sc)[k[3:]].numpy()
            init = d[("ia/" if k.startswith("fa/") else "ic/") + k[3:]]
            step = np.abs(d[k] - init).max()
            # each weight moved by ~epochs*lr; agree to a small fraction of that movement
            np.testing.assert_allclose(got, d[k], rtol=0, atol=max(2e-5, 0.05 * step), err_msg=k)
            assert step > 0 or "fc3" in k
            n += 1
    assert n == 22  # actor: 4 fc + 2 heads (w,b) = 12 ; critic: 4 fc + 1 head = 10


def _dp_worker(rank, world, port, path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1 if world > 2 else 2)
    ctx = ppo.DistCtx(device="cpu")
    d = np.load(os.path.join(G, "g7_update.npz"))
    torch.manual_seed(100 + rank)  # different initial weights per rank: the broadcast must fix that
    a, c = nets.make_policy("mlp64x2")
    cfg = ppo.PPOConfig(n_updates_per_iteration=3, policy="mlp64x2")
    up = ppo.PPOUpdater(a, c, cfg, ctx, torch.device("cpu"))
    lo, hi = ctx.shard(512)
    sl = slice(lo, hi)
    up.update(torch.from_numpy(d["obs"][sl]), torch.from_numpy(d["acts"][sl]), torch.from_numpy(d["logp"][sl]),
              torch.from_numpy(d["rtgs"][sl]), torch.tensor(0.8))
    torch.save({"flat": up.fp.flat.clone(), "hist": up.loss_history.clone(), "shard": (lo, hi)}, f"{path}.{rank}")
    ctx.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_data_parallel_gloo_equals_single_process(tmp_path, world):
    """world_size 2 and 8 (the node size of BASELINE configs[3]/[4]): contiguous shards of the G7 batch, flat-gradient
    all-reduce per epoch, all-reduced advantage moments == the single-process update on the whole batch."""
    from _ranks import spawn_ranks
    path = str(tmp_path / "dp")
    spawn_ranks(_dp_worker, world, lambda port: (world, port, path))
    rs = [torch.load(f"{path}.{k}") for k in range(world)]
    r0, r1 = rs[0], rs[1]
    per = 512 // world
    assert [r["shard"] for r in rs] == [(k * per, (k + 1) * per) for k in range(world)]
    for r in rs[1:]:
        assert torch.equal(r0["flat"], r["flat"])  # replicas stay bit-identical
    # single process on the full batch, starting from rank 0's initial weights
    d = np.load(os.path.join(G, "g7_update.npz"))
    torch.manual_seed(100)
    a, c = nets.make_policy("mlp64x2")
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(n_updates_per_iteration=3, policy="mlp64x2"), None, torch.device("cpu"))
    up.update(torch.from_numpy(d["obs"]), torch.from_numpy(d["acts"]), torch.from_numpy(d["logp"]),
              torch.from_numpy(d["rtgs"]), torch.tensor(0.8))
    np.testing.assert_allclose(r0["flat"].numpy(), up.fp.flat.numpy(), rtol=0, atol=2e-6)
    # global losses = mean of the shard losses (equal shard sizes)
    np.testing.assert_allclose((sum(r["hist"] for r in rs) / world).numpy(), up.loss_history.numpy(), rtol=1e-4, atol=1e-5)


def test_shard_partition():
    os.environ.pop("WORLD_SIZE", None)
    os.environ.pop("RANK", None)
    ctx = ppo.DistCtx(device="cpu")
    assert ctx.shard(4096) == (0, 4096) and not ctx.enabled
    ctx.world, ctx.rank = 8, 3
    assert ctx.shard(32768) == (3 * 4096, 4 * 4096)
    with pytest.raises(ValueError):
        ctx.shard(1001)


def test_checkpoint_roundtrip_reference_naming(tmp_path):
    a, c = nets.make_policy("resmlp512")
    t = ppo.PPOTrainer.__new__(ppo.PPOTrainer)
    t.cfg = ppo.PPOConfig(output_dir=str(tmp_path), method_name="m")
    t.actor, t.critic, t.i_so_far, t.t_so_far = a, c, 4, 19876
    pa, pc = t.save_checkpoint()
    assert os.path.basename(pa) == "actor_iter0004_step00019876.pth" and os.path.basename(pc) == "critic_iter0004_step00019876.pth"
    sd = torch.load(pa)
    assert sd["rb1.fc1.weight"].shape[1] == 16  # the shape probe of main.py:66-75
    a2, c2 = nets.make_policy("resmlp512")
    t.actor, t.critic = a2, c2
    t.load_checkpoint(pa, pc)
    for k, v in a.state_dict().items():
        assert torch.equal(v, a2.state_dict()[k])


def test_split_k_linear_same_gradients_and_keys():
    """nets.Linear switches to the batched (split-K) weight gradient for huge batches: same numbers."""
    torch.manual_seed(0)
    a, _ = nets.make_policy("mlp64x2")
    assert list(a.state_dict().keys())[:2] == ["layer1.weight", "layer1.bias"]
    x = torch.randn(1 << 17, 16)

    def grads(split):
        nets.Linear.SPLIT_ROWS = (1 << 16) if split else (1 << 40)
        for p in a.parameters():
            p.grad = None
        (a(x) ** 2).sum().backward()
        return [p.grad.clone() for p in a.parameters()]

    try:
        g1, g0 = grads(True), grads(False)
    finally:
        nets.Linear.SPLIT_ROWS = 1 << 16
    for u, v in zip(g1, g0):
        np.testing.assert_allclose(u.numpy(), v.numpy(), rtol=1e-4, atol=1e-4 * float(v.abs().max()))


def test_cli_flags_match_reference_arguments():
    """navbot_ppo_amd.main accepts the reference's flags with its defaults (project_ppo/src/arguments.py:22-46,58-65)."""
    from navbot_ppo_amd import main as M
    a = M.get_args([])
    assert (a.mode, a.method_name, a.eval, a.eval_episodes, a.timesteps_per_episode, a.max_timesteps, a.steps_per_iteration,
            a.save_every_iterations, a.resume, a.use_external_sampler) == ("train", "baseline", False, 100, 500, 5000, 5000, 2, False, False)
    t = M.get_args(["--tiny_debug_run", "--method_name", "x"])
    assert (t.timesteps_per_episode, t.steps_per_iteration, t.max_timesteps) == (20, 40, 400)
    assert M.schedule(M.get_args([]))[0] * M.schedule(M.get_args([]))[1] >= 5000
    assert M.schedule(M.get_args(["--steps_per_iteration", "2097152", "--n_envs", "4096"])) == (4096, 512)
    assert M.schedule(t) == (2, 20) and M.schedule(M.get_args([])) == (10, 500)
    with pytest.raises(SystemExit):
        M.get_args(["--method_name", "vision_mobilenet"])


def test_tensorboard_event_writer_roundtrip(tmp_path):
    """navbot_ppo_amd.tb_writer writes the TFRecord / tensorflow.Event format tensorboardX would (ppo.py:13,149,892)."""
    from navbot_ppo_amd import tb_writer as T
    assert T.crc32c(b"123456789") == 0xE3069283          # CRC-32C check value
    w = T.SummaryWriter(str(tmp_path))
    w.add_scalar("train/mean_return", 12.5, 3)
    w.add_scalar("loss/actor", -0.25, 300)
    w.add_scalar("perf/steps_per_sec", 3.5e7, 1 << 40)
    w.close()
    assert os.path.basename(w.path).startswith("events.out.tfevents.")
    assert T.read_scalars(w.path) == [("train/mean_return", 3, 12.5), ("loss/actor", 300, -0.25), ("perf/steps_per_sec", 1 << 40, 3.5e7)]
    raw = open(w.path, "rb").read()
    assert b"brain.Event:2" in raw[:64]
