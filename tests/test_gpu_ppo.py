"""GPU tests of the PPO side: the fused MFMA loss+gradient kernel (csrc/ppo_mlp64.hip) against PyTorch autograd of the
same losses (float32 reference of the same op), the trainer end to end, and the reference-learn golden G7 on the GPU."""
import os

import numpy as np
import pytest
import torch

from navbot_ppo_amd import nets, ppo

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _batch(n, seed, dev):
    g = torch.Generator().manual_seed(seed)
    obs = torch.rand((n, 16), generator=g)
    acts = torch.stack([torch.rand(n, generator=g), torch.rand(n, generator=g) * 2 - 1], 1)
    acts[torch.rand(n, generator=g) < 0.2, 0] = 0.0
    acts[torch.rand(n, generator=g) < 0.1, 1] = 1.0
    logp = -1.2 - 2.3 * torch.rand(n, generator=g)
    rtg = torch.randn(n, generator=g) * 60 + 20
    adv = torch.randn(n, generator=g)
    adv[torch.rand(n, generator=g) < 0.05] = 0.0
    return [t.to(dev).contiguous() for t in (obs, acts, logp, rtg, adv)]


@pytest.mark.parametrize("n", [128, 1000, 128 * 300 + 7, 1 << 17, 512 * 4096])   # the last one is the bench's own batch
def test_fused_mlp64_gradients_match_autograd(n):
    dev = torch.device("cuda")
    torch.manual_seed(3)
    a, c = nets.make_policy("mlp64x2")
    a.to(dev), c.to(dev)
    with torch.no_grad():  # push the heads away from their init so clipping, saturation and relu masks all occur
        for p in list(a.parameters()) + list(c.parameters()):
            p.mul_(3.0)
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="mlp64x2"), None, dev)
    assert up.fused_mlp64
    obs, acts, logp, rtg, adv = _batch(n, n, dev)
    var = torch.tensor(0.5, device=dev)
    # reference: PyTorch autograd, float32
    up.fp.grad.zero_()
    al, cl, ratios, lp, _ = ppo.ppo_losses(a, c, obs, acts, logp, rtg, adv, var, 0.2)
    (al + cl).backward()
    g_ref = up.fp.grad.clone()
    kl_ref = ((ratios - 1) - (lp - logp)).mean().item()
    cf_ref = ((ratios - 1).abs() > 0.2).float().mean().item()
    assert 0.02 < cf_ref < 0.98  # both branches of the clipped surrogate are exercised
    up.fp.grad.fill_(123.0)  # the kernel overwrites, it does not accumulate
    up._fused_loss_grad(obs, acts, logp, rtg, adv, 0.5)
    torch.cuda.synchronize()
    g = up.fp.grad
    st = up._fstats.cpu().numpy()
    # per-parameter-tensor comparison, tolerance relative to that tensor's gradient scale (fp32 summation order differs)
    off = 0
    for prm in up.fp.params:
        k = prm.numel()
        ref, got = g_ref[off:off + k], g[off:off + k]
        scale = ref.abs().max().item() + 1e-12
        err = (ref - got).abs().max().item()
        assert err <= 2e-4 * scale + 1e-7, (tuple(prm.shape), err, scale)
        off += k
    assert off == 5378 + 5313
    assert st[0] == pytest.approx(al.item(), rel=1e-4, abs=1e-6)
    assert st[4] == pytest.approx(cl.item(), rel=1e-4)
    assert st[1] == pytest.approx(kl_ref, rel=1e-3, abs=1e-5)
    assert st[2] == pytest.approx(cf_ref, abs=1e-6)


def test_fused_update_tracks_pytorch_update_over_epochs():
    """10 Adam epochs with the fused kernel vs 10 with PyTorch autograd from the same start."""
    dev = torch.device("cuda")
    obs, acts, logp, rtg, adv = _batch(1 << 15, 5, dev)
    res = []
    for fused in (True, False):
        torch.manual_seed(11)
        a, c = nets.make_policy("mlp64x2")
        a.to(dev), c.to(dev)
        up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="mlp64x2", n_updates_per_iteration=10, fused_update=fused), None, dev)
        assert up.fused_mlp64 == fused
        st = up.update(obs, acts, logp, rtg, torch.tensor(0.8, device=dev))
        res.append((up.fp.flat.clone(), up.loss_history.clone(), st))
    (w1, h1, s1), (w0, h0, s0) = res
    np.testing.assert_allclose(h1.cpu().numpy(), h0.cpu().numpy(), rtol=2e-4, atol=1e-5)
    assert (w1 - w0).abs().max().item() < 3e-5  # 10 steps of lr 3e-4 move weights by ~3e-3
    for k in ("actor_loss", "critic_loss", "approx_kl", "clip_frac"):
        assert s1[k] == pytest.approx(s0[k], rel=2e-3, abs=2e-5), k


def test_g7_reference_update_on_gpu():
    """G7 (reference PPO.learn on a fixed batch) with the resmlp512 nets on the GPU (PyTorch path, split-K wgrad off at
    this batch size): same per-epoch losses as the reference."""
    d = np.load(os.path.join(G, "g7_update.npz"))
    dev = torch.device("cuda")
    a, c = nets.make_policy("resmlp512")
    for mod, pre in ((a, "ia/"), (c, "ic/")):
        sd = mod.state_dict()
        with torch.no_grad():
            for k in sd:
                if pre + k in d:
                    sd[k].copy_(torch.from_numpy(d[pre + k]))
    a.to(dev), c.to(dev)
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(n_updates_per_iteration=int(d["epochs"])), None, dev)
    t = lambda k: torch.from_numpy(d[k]).to(dev)
    up.update(t("obs"), t("acts"), t("logp"), t("rtgs"), torch.tensor(0.8, device=dev))
    h = up.loss_history.cpu().numpy()
    np.testing.assert_allclose(h[:, 0], d["actor_losses"], rtol=5e-4, atol=5e-5)
    np.testing.assert_allclose(h[:, 1], d["critic_losses"], rtol=5e-4)


@pytest.mark.parametrize("policy", ["mlp64x2", "resmlp512"])
def test_trainer_iterations_run_and_learn_signal(policy):
    """Two full iterations (graph rollout + HIP return scan + update) on a small shard; sanity of the bookkeeping."""
    from navbot_ppo_amd.env import VecEnv
    env = VecEnv(256, map="stage_1", max_episode_steps=40, seed=1)
    cfg = ppo.PPOConfig(rollout_len=64, max_episode_steps=40, n_updates_per_iteration=4, policy=policy, seed=2)
    tr = ppo.PPOTrainer(env, cfg)
    lg1 = tr.iteration()
    lg2 = tr.iteration()
    assert lg1["episodes"] >= 256 and lg2["iteration"] == 2            # every env times out at 40 < 64 steps
    assert lg1["episodes"] == lg1["successes"] + lg1["collisions"] + lg1["timeouts"]
    assert lg1["completed_steps"] <= 64 * 256 and tr.env_steps == 2 * 64 * 256
    assert np.isfinite(lg2["actor_loss"]) and np.isfinite(lg2["critic_loss"]) and lg2["critic_loss"] > 0
    # rollout buffers are self-consistent: stored log-probs are those of the stored (clamped) actions under the
    # pre-update policy -> recompute with the current (post-update) policy just checks shapes/finite
    assert tr.obs_buf.shape == (65, 256, 16) and torch.isfinite(tr.logp_buf).all()
    assert (tr.act_buf[..., 0] >= 0).all() and (tr.act_buf[..., 0] <= 1).all() and (tr.act_buf[..., 1].abs() <= 1).all()
    # return scan inside the trainer == oracle
    from oracle import navsim_oracle as O
    from test_gpu_parity import assert_rtg_close   # <= 1 float32 ulp (T-split scan), see there
    assert_rtg_close(tr.rtg_buf.cpu().numpy(),
                     O.compute_rtgs_tn(tr.rew_buf.cpu().numpy(), tr.ended_buf.cpu().numpy(), cfg.gamma))
    env.close()


def test_fused_act_matches_pytorch_policy_step():
    """mlp64_act with explicit noise == PyTorch: mean = actor(obs), clamp(mean + sqrt(var) eps), log-prob of the clamped
    action (ppo.py:696-704); and the in-kernel Philox/Box-Muller noise is standard normal and shard-invariant."""
    import ctypes as C
    from navbot_ppo_amd._native import lib
    dev = torch.device("cuda")
    torch.manual_seed(5)
    a, c = nets.make_policy("mlp64x2")
    a.to(dev), c.to(dev)
    with torch.no_grad():
        for p in a.parameters():
            p.mul_(2.0)
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="mlp64x2"), None, dev)
    n = 128 * 5 + 77
    obs = torch.rand((n, 16), device=dev)
    eps = torch.randn((n, 2), device=dev)
    var = torch.tensor(0.8, device=dev)
    act = torch.empty((n, 2), device=dev)
    lp = torch.empty(n, device=dev)
    mean = torch.empty((n, 2), device=dev)
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    L = lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.navppo_mlp64_act(ptr(up.fp.flat), ptr(obs), 16, 0, ptr(eps), n, ptr(var), 7, 0, None, 0, ptr(act), ptr(lp), ptr(mean), st) == 0
    with torch.no_grad():
        m_ref = a(obs)
        raw = m_ref + torch.sqrt(var) * eps
        a_ref = torch.stack([raw[:, 0].clamp(0, 1), raw[:, 1].clamp(-1, 1)], 1)
        lp_ref = ppo.gaussian_log_prob(m_ref, a_ref, var)
    np.testing.assert_allclose(mean.cpu().numpy(), m_ref.cpu().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(act.cpu().numpy(), a_ref.cpu().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(lp.cpu().numpy(), lp_ref.cpu().numpy(), rtol=1e-4, atol=2e-5)
    # in-kernel noise: recover eps = (act - mean)/sqrt(var) where unclamped; moments of N(0,1); depends on (seed, id, step) only
    N = 1 << 16
    obs = torch.rand((N, 16), device=dev)
    act, lp, mean = torch.empty((N, 2), device=dev), torch.empty(N, device=dev), torch.empty((N, 2), device=dev)
    big = torch.tensor(1e-4, device=dev)  # tiny variance: nothing clamps except at the sigmoid/tanh edges
    sb = torch.tensor(5, dtype=torch.int32, device=dev)
    assert L.navppo_mlp64_act(ptr(up.fp.flat), ptr(obs), 16, 0, None, N, ptr(big), 9, 0, ptr(sb), 2, ptr(act), ptr(lp), ptr(mean), st) == 0
    e = ((act - mean) / 1e-2).cpu().numpy()
    ok = (np.abs(e) < 6).all(axis=1) & (act[:, 0].cpu().numpy() > 0) & (act[:, 0].cpu().numpy() < 1) & (np.abs(act[:, 1].cpu().numpy()) < 1)
    e = e[ok]
    assert ok.mean() > 0.9 and abs(e.mean()) < 0.02 and abs(e.std() - 1) < 0.02 and abs(np.corrcoef(e[:, 0], e[:, 1])[0, 1]) < 0.02
    act2 = torch.empty((N // 2, 2), device=dev)
    sb7 = torch.tensor(7, dtype=torch.int32, device=dev)
    assert L.navppo_mlp64_act(ptr(up.fp.flat), ptr(obs[N // 2:].contiguous()), 16, 0, None, N // 2, ptr(big), 9, N // 2, ptr(sb7), 0,
                              ptr(act2), ptr(lp), None, st) == 0
    assert torch.equal(act2, act[N // 2:])  # same (seed, global env id, step) -> same draw on another shard


def test_train_checkpoint_then_evaluate_like_main(tmp_path):
    """train -> checkpoint files with the reference's names -> evaluation mode of main.py:135-252 on the newest one."""
    import csv
    from navbot_ppo_amd import evaluate as ev
    from navbot_ppo_amd.env import VecEnv
    env = VecEnv(128, map="stage_1", max_episode_steps=30, seed=3)
    cfg = ppo.PPOConfig(rollout_len=32, max_episode_steps=30, n_updates_per_iteration=2, policy="resmlp512", seed=1,
                        output_dir=str(tmp_path), method_name="m1", save_freq=1)
    tr = ppo.PPOTrainer(env, cfg)
    tr.iteration()
    tr.iteration()
    env.close()
    ck = ev.find_latest_checkpoint(str(tmp_path), "m1")
    assert ck and os.path.basename(ck).startswith("actor_iter0002_step")
    logs = os.path.join(str(tmp_path), "m1", "logs")
    head = open(os.path.join(logs, "m1_train_episodes.csv")).readline().strip().split(",")
    assert head == ["episode", "timestep", "success", "collision", "timeout", "length", "return", "path_length", "time"]
    import json
    sc = [json.loads(l) for l in open(os.path.join(logs, "scalars.jsonl"))]
    assert len(sc) == 2 and {"train/mean_return", "loss/actor", "loss/critic", "perf/steps_per_sec", "ppo/approx_kl"} <= set(sc[0])
    # TensorBoard event file under <run>/tb (ppo.py:66,149) with the reference's tags (ppo.py:892-939), CRC-checked on read
    import glob
    from navbot_ppo_amd import tb_writer
    ev_files = glob.glob(os.path.join(str(tmp_path), "m1", "tb", "events.out.tfevents.*"))
    assert len(ev_files) == 1
    recs = tb_writer.read_scalars(ev_files[0])
    tags = {r[0] for r in recs}
    assert {"train/success_rate", "train/collision_rate", "train/timeout_rate", "train/mean_return", "train/mean_ep_length",
            "train/mean_ep_time", "loss/actor", "loss/critic", "train/timesteps", "time/rollout", "time/update", "time/iteration",
            "perf/steps_per_sec", "perf/actor_grad_steps", "perf/critic_grad_steps", "ppo/approx_kl", "ppo/entropy",
            "ppo/clip_frac", "ppo/actor_grad_norm", "ppo/critic_grad_norm", "ppo/actor_param_delta", "ppo/critic_param_delta",
            "Actor_loss/train", "Critic_loss/train", "Episode_Rewards/train", "avg_ep_rews/train"} <= tags
    assert sorted(s for tg, s, _ in recs if tg == "train/mean_return") == [1, 2]
    assert len([1 for tg, _, _ in recs if tg == "Actor_loss/train"]) == 4          # 2 iterations x 2 epochs
    path_col = [float(r.split(",")[7]) for r in open(os.path.join(logs, "m1_train_episodes.csv")).read().strip().split("\n")[1:]]
    assert max(path_col) > 0.0
    actor, policy = ev.load_actor(ck, "cuda")
    assert policy == "resmlp512"
    s = ev.evaluate(actor, num_episodes=40, max_timesteps_per_episode=25, n_parallel=16, output_dir=str(tmp_path), method_name="m1", log=None)
    assert s["episodes"] == 40 and abs(s["success_rate"] + s["collision_rate"] + s["timeout_rate"] - 1.0) < 1e-9
    rows = list(csv.reader(open(s["csv"])))
    assert rows[0] == ["episode", "success", "collision", "timeout", "length", "return", "path_length", "time"] and len(rows) == 41
    assert all(int(r[4]) <= 25 for r in rows[1:])


def test_cli_train_resume_eval(tmp_path):
    """The command-line surface: tiny debug run -> checkpoints -> --resume -> --eval, as `python3 main.py ...` of the reference."""
    from navbot_ppo_amd import main as M
    out = str(tmp_path)
    assert M.main(["--tiny_debug_run", "--method_name", "dbg", "--output_dir", out, "--policy", "mlp64x2", "--save_every_iterations", "1"]) == 0
    ck = sorted(os.listdir(os.path.join(out, "dbg", "checkpoints")))
    assert any(f.startswith("actor_iter") for f in ck) and any(f.startswith("critic_iter") for f in ck)
    assert M.main(["--tiny_debug_run", "--resume", "--method_name", "dbg", "--output_dir", out, "--policy", "mlp64x2",
                   "--use_external_sampler"]) == 0
    assert M.main(["--eval", "--eval_episodes", "12", "--timesteps_per_episode", "15", "--method_name", "dbg", "--output_dir", out]) == 0
    assert os.path.exists(os.path.join(out, "dbg", "logs", "dbg_eval_episodes.csv"))


@pytest.mark.parametrize("sens", [False, True])
@pytest.mark.parametrize("N,T,map_name,per_env,epb", [
    (4096, 64, "stage_1", False, None), (200, 90, "stage_1", False, None), (96, 40, "stage_2", True, None),
    # rollout_big_kernel (64-env workgroups, policy phase in front of the tape kernel's step): the shard sizes that select it
    # with each cast variant (plain / 128-segment passes of per-env maps / tile boxes of a shared 2048-segment map), and forced
    # onto small ragged shards (200 = 3 workgroups + 8 envs; 40 envs: one partial policy tile)
    (16384, 40, "stage_1", False, None), (16384, 36, "stage_2", True, None), (16384, 34, "house", False, None),
    (4160, 40, "stage_1", False, None), (6144, 36, "stage_2", True, None),   # the first shard sizes of the 64-env shape
    (200, 60, "stage_1", False, "64"), (40, 50, "stage_2", True, "64"), (1000, 36, "house", False, "64"),
    # round 5: the 16-env rollout kernel with the tile-box cast (shared 65..4096-segment maps on shards up to 4096 envs; until then such
    # shards took the hipGraph of per-step launches)
    (1000, 36, "house", False, None), (4096, 34, "house", False, None),
    # ... and the 32-env closed-loop shape (tile-box maps on 4097..8192 envs: one workgroup on every CU), natural and forced
    (8192, 34, "house", False, None), (200, 50, "stage_1", False, "32"), (1000, 36, "house", False, "32"), (96, 40, "stage_2", True, "32")])
def test_persistent_rollout_equals_per_step_rollout(N, T, map_name, per_env, epb, sens, monkeypatch):
    """navsim_rollout_mlp64 (all T steps in one launch) against T pairs of navppo_mlp64_act / navsim_step: same device
    functions and Philox keys, so every rollout buffer and the simulator state must come out bit-identical -- over two
    consecutive rollouts (the noise counter and the cached next-episode records carry over)."""
    from navbot_ppo_amd import ppo
    from navbot_ppo_amd.env import VecEnv
    if epb:
        monkeypatch.setenv("NAVSIM_EPB", epb)   # read by NavSim.__init__ (navsim_set_shape, per handle): both paths then run their 64-env shapes
    kw = dict(lidar_noise_sigma=0.01, lidar_below_min="gazebo") if sens else {}   # SENS = true instantiations of both kernels
    outs = []
    for persistent in (True, False):
        env = VecEnv(N, map=map_name, max_episode_steps=30, seed=3, per_env_map=per_env,
                     sampler="small_house" if map_name == "house" else None, **kw)
        cfg = ppo.PPOConfig(rollout_len=T, max_episode_steps=30, n_updates_per_iteration=1, policy="mlp64x2", seed=5,
                            persistent_rollout=persistent, use_graph=False)
        tr = ppo.PPOTrainer(env, cfg)
        assert tr.uses_persistent_rollout is persistent   # the two sides of the comparison do take different paths
        if persistent and map_name == "house":   # the tile-box cast in whichever rollout kernel the shard selects
            inf = env.sim.info()
            assert inf["tile_boxes"] == 1 and inf["rollout_cast"] == 3 and inf["rollout_kind"] == (2 if (N > 4096 or epb in ("64", "32")) else 1), inf
            assert inf["rollout_epb"] == (32 if (epb == "32" or (epb is None and 4096 < N <= 8192)) else 64 if (N > 4096 or epb == "64") else 16), inf
        bufs = []
        for _ in range(2):
            tr.rollout()
            torch.cuda.synchronize()
            bufs.append([b.clone() for b in (tr.obs_buf, tr.act_buf, tr.logp_buf, tr.rew_buf, tr.done_buf, tr.arrive_buf,
                                             tr.ended_buf, tr.rtg_buf)] +
                        [torch.where(tr.ended_buf.bool(), b, torch.zeros_like(b)) for b in (tr.epret_buf, tr.eplen_buf, tr.eppath_buf)])
        outs.append((bufs, env.sim.get_state()))
        env.close()
    (a, sa), (b, sb) = outs
    assert int(a[0][6].sum()) > N // 4          # episodes ended
    # bit patterns: with lidar_below_min="gazebo" a reading below range_min is -inf and the policy's log-prob NaN on both paths
    bits = lambda x: x.view(torch.int32) if x.dtype == torch.float32 else x
    for ra, rb in zip(a, b):
        for x, y in zip(ra, rb):
            assert torch.equal(bits(x), bits(y))
    for k in sa:
        np.testing.assert_array_equal(sa[k], sb[k])


def test_fused_value_matches_pytorch_critic():
    from navbot_ppo_amd import nets, ppo
    dev = torch.device("cuda")
    torch.manual_seed(3)
    a, c = nets.make_policy("mlp64x2")
    a.to(dev), c.to(dev)
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="mlp64x2"), None, dev)
    for n in (1, 31, 1000, 128 * 300 + 7):
        obs = torch.rand((n, 16), device=dev) * 2 - 0.5
        with torch.no_grad():
            want = c(obs).squeeze(-1)
        got = up._fused_value(obs)
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=2e-5)


def _dp_gpu_worker(rank, world, port, path, overlap=False):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      NAVBOT_DIST_BACKEND="gloo")   # RCCL refuses two ranks on one device: gloo carries the all-reduce here
    from navbot_ppo_amd import nets, ppo
    ctx = ppo.DistCtx(device="cuda:0")
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "g7_update.npz"))
    torch.manual_seed(100 + rank)
    a, c = nets.make_policy("mlp64x2")
    a.cuda(), c.cuda()
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(n_updates_per_iteration=5, policy="mlp64x2", overlap_allreduce=overlap), ctx,
                        torch.device("cuda:0"))
    assert up.fused_mlp64
    lo, hi = ctx.shard(512)
    cu = lambda k: torch.from_numpy(d[k][lo:hi]).cuda()
    up.update(cu("obs"), cu("acts"), cu("logp"), cu("rtgs"), torch.tensor(0.8, device="cuda"))
    torch.save({"flat": up.fp.flat.cpu()}, f"{path}.{rank}")
    ctx.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_fused_multi_rank_epoch_equals_single_rank(tmp_path, overlap):
    """The N > 1 update path on the GPU (fused passes -> one all-reduce of the flat gradient -> scale + Adam kernel; overlap: the
    two-stage per-net pipeline, each net's all-reduce under the other net's pass) with two ranks on shards of the G7 batch == the
    single-rank path (fused passes + in-kernel Adam) on the whole batch."""
    from _ranks import spawn_ranks
    from navbot_ppo_amd import nets, ppo
    path = str(tmp_path / "dpg")
    spawn_ranks(_dp_gpu_worker, 2, lambda port: (2, port, path, overlap))
    r0, r1 = torch.load(path + ".0"), torch.load(path + ".1")
    assert torch.equal(r0["flat"], r1["flat"])
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "g7_update.npz"))
    torch.manual_seed(100)
    a, c = nets.make_policy("mlp64x2")
    a.cuda(), c.cuda()
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(n_updates_per_iteration=5, policy="mlp64x2"), None, torch.device("cuda:0"))
    cu = lambda k: torch.from_numpy(d[k]).cuda()
    up.update(cu("obs"), cu("acts"), cu("logp"), cu("rtgs"), torch.tensor(0.8, device="cuda"))
    np.testing.assert_allclose(r0["flat"].numpy(), up.fp.flat.cpu().numpy(), rtol=0, atol=3e-6)


def _rccl_one_rank_worker(rank, port, path, policy, overlap):
    import os
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NAVBOT_DIST_FORCE="1")
    os.environ.pop("NAVBOT_DIST_BACKEND", None)
    from navbot_ppo_amd import nets, ppo
    ctx = ppo.DistCtx(device="cuda:0")
    assert ctx.enabled and ctx.backend == "nccl" and ctx.rccl_version, (ctx.enabled, ctx.backend, ctx.rccl_version)
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "g7_update.npz"))
    torch.manual_seed(100)
    a, c = nets.make_policy(policy)
    a.cuda(), c.cuda()
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(n_updates_per_iteration=5, policy=policy, overlap_allreduce=overlap), ctx,
                        torch.device("cuda:0"))
    assert up.fused_mlp64 or up.fused_resmlp512
    cu = lambda k: torch.from_numpy(d[k]).cuda()
    stats = up.update(cu("obs"), cu("acts"), cu("logp"), cu("rtgs"), torch.tensor(0.8, device="cuda"))
    t = torch.tensor([3.0, 5.0], device="cuda")
    ctx.all_reduce_max(t), ctx.broadcast(t), ctx.barrier()
    torch.cuda.synchronize()
    torch.save({"flat": up.fp.flat.cpu(), "rccl": ctx.rccl_version, "grad_norm": float(stats.get("grad_norm", 0.0))}, path)
    torch.distributed.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("policy,overlap", [("mlp64x2", False), ("mlp64x2", True), ("resmlp512", False)])
def test_update_through_rccl_single_rank(tmp_path, policy, overlap):
    """The multi-GPU update path with torch's `nccl` backend (= RCCL) really executing: one rank (a one-GPU box; RCCL refuses
    two ranks on one device), process group created with device_id, fused passes -> one all-reduce of the flat gradient (with
    overlap: the per-net pipeline, each net's all-reduce under the other net's pass) -> navppo_adam_step, plus max-reduce /
    broadcast / barrier.  Must equal the single-process path (fused passes + in-kernel Adam) on the same batch."""
    from _ranks import spawn_ranks
    from navbot_ppo_amd import nets, ppo
    path = str(tmp_path / "rccl1.pt")
    spawn_ranks(_rccl_one_rank_worker, 1, lambda port: (port, path, policy, overlap))
    r = torch.load(path)
    assert r["rccl"] and r["rccl"] != "unknown"
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "g7_update.npz"))
    torch.manual_seed(100)
    a, c = nets.make_policy(policy)
    a.cuda(), c.cuda()
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(n_updates_per_iteration=5, policy=policy), None, torch.device("cuda:0"))
    cu = lambda k: torch.from_numpy(d[k]).cuda()
    stats = up.update(cu("obs"), cu("acts"), cu("logp"), cu("rtgs"), torch.tensor(0.8, device="cuda"))
    np.testing.assert_allclose(r["flat"].numpy(), up.fp.flat.cpu().numpy(), rtol=0, atol=3e-6)
    if "grad_norm" in stats:
        np.testing.assert_allclose(r["grad_norm"], float(stats["grad_norm"]), rtol=1e-4)


@pytest.mark.gpu
def test_bench_one_rank_over_rccl(tmp_path):
    """bench.py launched the way the driver launches N > 1 (torch.distributed.run), with one rank forced through RCCL: the
    whole N > 1 code path of the bench (DistCtx over nccl, barriers, max-over-ranks timing, per-epoch all-reduces) runs on the
    GPU and prints the contract line with rccl_ranks = 1."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NAVBOT_DIST_FORCE="1")
    env.pop("NAVBOT_DIST_BACKEND", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--no-extras"], capture_output=True, text=True, env=env, timeout=600, cwd=repo)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["dist_backend"] == "nccl" and r["rccl_ranks"] == 1 and r["rccl_version"], r
    assert r["n_gpus"] == 1 and r["value"] > 1e6


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 1), (7, 33), (512, 4096), (100, 37)])
def test_episode_sums_match_torch_reductions(shape):
    """navppo_episode_sums == the reductions behind the iteration's log line (ppo.py:552-560, :833), counts exact, the
    float64 return sum to 1e-12 relative, and identical from call to call (fixed summation order)."""
    import ctypes as C
    from navbot_ppo_amd._native import lib
    g = torch.Generator().manual_seed(5)
    ended = (torch.rand(shape, generator=g) < 0.05).to(torch.uint8)
    arrive = (torch.rand(shape, generator=g) < 0.3).to(torch.uint8)    # also set where no episode ended: must not count
    done = (torch.rand(shape, generator=g) < 0.5).to(torch.uint8)
    eplen = (torch.randint(1, 500, shape, generator=g) * ended).to(torch.int32)
    epret = (torch.randn(shape, generator=g) * 300).float()            # garbage outside ended entries: masked by the kernel
    e, a, d = ended.bool(), arrive.bool() & ended.bool(), done.bool() & ended.bool()
    want = [e.sum().item(), a.sum().item(), (d & ~a).sum().item(), (e & ~d & ~a).sum().item(), eplen.sum().item(),
            (epret.double() * e).sum().item()]
    bufs = [t.cuda().contiguous() for t in (ended, arrive, done, eplen, epret)]
    ws = torch.empty(256 * 6, dtype=torch.float64, device="cuda")
    outs = []
    for _ in range(2):
        out = torch.full((6,), -1.0, dtype=torch.float64, device="cuda")
        rc = lib().navppo_episode_sums(*[C.c_void_p(t.data_ptr()) for t in bufs], ended.numel(), C.c_void_p(out.data_ptr()),
                                       C.c_void_p(ws.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        outs.append(out.cpu().numpy())
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_array_equal(outs[0][:5], np.array(want[:5], dtype=np.float64))
    assert abs(outs[0][5] - want[5]) <= 1e-12 * max(1.0, abs(want[5])) + 1e-9


@pytest.mark.parametrize("N,T,cap,lo,n_s,world", [(4096, 512, 500, 2000, 96, "stage_1"), (16384, 160, 150, 9000, 96, "stage_1"),
                                                    (16384, 160, 150, 16384 - 96, 96, "cfg3")])
def test_persistent_rollout_at_bench_size_against_the_oracle(N, T, cap, lo, n_s, world):
    """navsim_rollout_mlp64 checked DIRECTLY against the oracle, not against the per-step HIP path, at the timed configuration
    (BASELINE configs[1]: T = 512, N = 4096, episode cap 500, shared stage_1 map: rollout_kernel), at a 16384-env shard on the
    shared stage_1 map (rollout_big_kernel, 32-segment passes) and on configs[2]'s OWN workload (`cfg3`: 16384 envs, per-env
    stage_2 maps with their goal rectangles: rollout_big_kernel with the 128-segments-per-pass cast, the instantiation the
    `roofline_closed_loop` leg of bench.py times): the actions the kernel recorded for a block of 96 envs
    are replayed on an OracleSim keyed by the same global env ids (goal stream = Philox(seed, env id)); flags must be
    bit-exact, observations within 1e-6, rewards within 1e-5; the stored log-probs are those of the stored (clamped)
    actions under PyTorch's evaluation of the same actor (ppo.py:696-704)."""
    from navbot_ppo_amd import maps
    from navbot_ppo_amd.env import VecEnv
    from oracle import navsim_oracle as O
    name = "stage_2" if world == "cfg3" else "stage_1"
    env = VecEnv(N, map=name, max_episode_steps=cap, seed=7, per_env_map=(world == "cfg3"), map_seed=0)
    cfg = ppo.PPOConfig(rollout_len=T, max_episode_steps=cap, policy="mlp64x2", seed=3)
    tr = ppo.PPOTrainer(env, cfg)
    with torch.no_grad():   # drive: a forward bias so that collisions / arrivals happen inside 512 steps, not only timeouts
        tr.actor.layer3.bias.add_(2.0)
    tr.rollout()
    torch.cuda.synchronize()
    assert tr.updater.fused_mlp64 and cfg.persistent_rollout
    sl = slice(lo, lo + n_s)
    acts = tr.act_buf[:, sl].cpu().numpy()
    cpu = O.OracleSim(n_s, max_episode_steps=cap, auto_reset=True, seed=7, env_id_base=lo)
    if world == "cfg3":
        assert env.sim.per_env and env.sim.S == 128
        cpu.set_map(env.sim._seg[sl].cpu().numpy(), per_env=True)
    else:
        cpu.set_map(maps.stage_1())
    rr, rs = maps.goal_rects(name)
    cpu.set_goal_rects(0, rr)
    cpu.set_goal_rects(1, rs)
    obs = tr.obs_buf[:, sl].cpu().numpy()
    np.testing.assert_allclose(obs[0], cpu.reset(), rtol=0, atol=1e-6)
    g = {k: getattr(tr, k + "_buf")[:, sl].cpu().numpy() for k in ("rew", "done", "arrive", "ended", "epret", "eplen", "eppath")}
    n_end = 0
    for t in range(T):
        out = cpu.step(acts[t])
        for k in ("done", "arrive", "ended"):
            np.testing.assert_array_equal(g[k][t], out[k], err_msg=f"{k}, step {t}")
        np.testing.assert_allclose(obs[t + 1], out["obs"], rtol=0, atol=1e-6, err_msg=f"obs, step {t}")
        np.testing.assert_allclose(g["rew"][t], out["reward"], rtol=1e-5, atol=1e-5, err_msg=f"reward, step {t}")
        e = out["ended"].astype(bool)
        np.testing.assert_array_equal(g["eplen"][t][e], out["ep_length"][e])
        np.testing.assert_allclose(g["epret"][t][e], out["ep_return"][e], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(g["eppath"][t][e], out["ep_path"][e], rtol=1e-6, atol=1e-7)
        n_end += int(e.sum())
    assert n_end >= n_s and int(g["done"].sum()) > 0          # every env ended at least once; collisions happened
    # policy side over the WHOLE rollout: log-prob of the stored clamped action under the PyTorch actor, action range
    with torch.no_grad():
        o = tr.obs_buf[:T].reshape(T * N, 16)
        lp_ref = ppo.gaussian_log_prob(tr.actor(o), tr.act_buf.reshape(T * N, 2), tr.var)
    np.testing.assert_allclose(tr.logp_buf.reshape(-1).cpu().numpy(), lp_ref.cpu().numpy(), rtol=1e-4, atol=3e-5)
    a = tr.act_buf
    assert (a[..., 0] >= 0).all() and (a[..., 0] <= 1).all() and (a[..., 1].abs() <= 1).all()
    # return scan of the same buffers == the oracle's compute_rtgs (<= 1 f32 ulp, the T-split scan's contract)
    from test_gpu_parity import assert_rtg_close
    assert_rtg_close(tr.rtg_buf[:, sl].cpu().numpy(), O.compute_rtgs_tn(g["rew"], g["ended"], cfg.gamma))
    env.close()


@pytest.mark.parametrize("N", [300, 16384])
def test_vecenv_rollout_mlp64_equals_the_trainers_rollout(N):
    """VecEnv.rollout_mlp64 (the closed-loop rollout without a trainer: flat actor parameters in, [T, N, .] buffers out) against
    PPOTrainer.rollout with the same actor, variance and noise key -- both shapes of the kernel -- and its log-probs against PyTorch's
    evaluation of the same actor on the stored observations / actions (ppo.py:696-704)."""
    from navbot_ppo_amd.env import VecEnv
    T = 40
    env = VecEnv(N, map="stage_1", max_episode_steps=30, seed=11)
    tr = ppo.PPOTrainer(env, ppo.PPOConfig(rollout_len=T, max_episode_steps=30, policy="mlp64x2", seed=4))
    tr.rollout()
    torch.cuda.synchronize()
    ref = {k: getattr(tr, k + "_buf").clone() for k in ("obs", "act", "logp", "rew", "done", "arrive", "ended")}
    var, seed = tr.var.clone(), tr._act_seed
    flat = tr.updater.fp.flat[:5378].clone()
    env.close()
    env2 = VecEnv(N, map="stage_1", max_episode_steps=30, seed=11)
    out = env2.rollout_mlp64(flat, T, var, seed=seed, step_base=0)
    for k, v in (("obs", out.obs), ("act", out.act), ("logp", out.logp), ("rew", out.reward), ("done", out.done),
                 ("arrive", out.arrive), ("ended", out.ended)):
        assert torch.equal(ref[k], v), k
    assert int(out.ended.sum()) >= N
    with torch.no_grad():
        lp = ppo.gaussian_log_prob(tr.actor(out.obs[:T].reshape(T * N, 16)), out.act.reshape(T * N, 2), var)
    np.testing.assert_allclose(out.logp.reshape(-1).cpu().numpy(), lp.cpu().numpy(), rtol=1e-4, atol=3e-5)
    with pytest.raises(Exception):
        env2.rollout_mlp64(flat[:100], T, var)
    env2.close()


@pytest.mark.parametrize("N,epb", [(2048, None), (4160, None), (200, "64")])
def test_persistent_rollout_with_arrival_respawn(N, epb, monkeypatch):
    """The in-step policy derives the observation row a step will leave -- incl. WHICH reset record replaces it -- before the rules
    lane has run (next_obs4).  With respawn_on_arrive the record depends on how the episode ended (record 1 behind an arrival's
    re-spawn draw, environment_new.py:245-267; record 0 otherwise): arrivals (threshold 0.4: ~1 % of the goals lie that close to
    the spawn pose), collisions and time-outs all occur here, in both rollout kernels; every buffer must equal the per-step path."""
    from navbot_ppo_amd import ppo
    from navbot_ppo_amd.env import VecEnv
    if epb:
        monkeypatch.setenv("NAVSIM_EPB", epb)
    outs = []
    for persistent in (True, False):
        env = VecEnv(N, map="stage_1", max_episode_steps=70, seed=4, is_training=False, respawn_on_arrive=True)
        cfg = ppo.PPOConfig(rollout_len=150, max_episode_steps=70, n_updates_per_iteration=1, policy="mlp64x2", seed=6,
                            persistent_rollout=persistent, use_graph=False)
        tr = ppo.PPOTrainer(env, cfg)
        assert tr.uses_persistent_rollout is persistent   # the two sides of the comparison do take different paths
        with torch.no_grad():
            tr.actor.layer3.bias.add_(2.0)   # drive forward: collisions as well as arrivals and time-outs
        tr.rollout()
        torch.cuda.synchronize()
        outs.append(([b.clone() for b in (tr.obs_buf, tr.act_buf, tr.logp_buf, tr.rew_buf, tr.done_buf, tr.arrive_buf, tr.ended_buf)],
                     env.sim.get_state()))
        env.close()
    (a, sa), (b, sb) = outs
    assert int(a[5].sum()) > 0 and int(a[4].sum()) > 0 and int(a[6].sum()) > int(a[4].sum()) + int(a[5].sum())
    for x, y in zip(a, b):
        assert torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x, y.view(torch.int32) if y.dtype == torch.float32 else y)
    for k in sa:
        np.testing.assert_array_equal(sa[k], sb[k])


def test_persistent_rollout_on_a_streamed_map():
    """16384 envs x 1280 segments per env = 335 MB per step: beyond 1.25 x the Infinity Cache, where navsim_set_map selects the
    non-temporal instantiations of the 128-segment pass in the persistent kernels too (rollout_big_kernel<..., PAIR = 2>); the
    rows must equal the per-step path's bit for bit, like everywhere else."""
    from navbot_ppo_amd import maps, ppo
    from navbot_ppo_amd.env import VecEnv
    N, T = 16384, 6
    seg = torch.from_numpy(maps.replicate_per_env(maps.stage_2(sides=312), N, seed=5)).cuda()
    assert seg.shape[1] == 1280 and seg.numel() * 4 >= 320 << 20
    outs = []
    for persistent in (True, False):
        env = VecEnv(N, map=seg, max_episode_steps=4, seed=9)
        cfg = ppo.PPOConfig(rollout_len=T, max_episode_steps=4, n_updates_per_iteration=1, policy="mlp64x2", seed=2,
                            persistent_rollout=persistent, use_graph=False)
        tr = ppo.PPOTrainer(env, cfg)
        assert tr.uses_persistent_rollout is persistent   # the two sides of the comparison do take different paths
        tr.rollout()
        torch.cuda.synchronize()
        outs.append([b.clone() for b in (tr.obs_buf, tr.act_buf, tr.logp_buf, tr.rew_buf, tr.done_buf, tr.arrive_buf, tr.ended_buf)])
        env.close()
        del tr
    assert int(outs[0][6].sum()) >= N
    for x, y in zip(*outs):
        assert torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x, y.view(torch.int32) if y.dtype == torch.float32 else y)


@pytest.mark.parametrize("policy,arith", [("mlp64x2", "bf16x3"), ("mlp64x2", "f32"), ("resmlp512", "bf16x3")])
def test_fused_update_reports_gradient_norms_averaged_over_the_epochs(policy, arith):
    """The reference logs the MEAN over an iteration's epochs of each net's gradient norm (ppo.py:351-352, 389-390, 449-450).  The
    fused epochs deliver those without a norm launch per epoch: reduce_adam / resmlp_reduce leave the squared per-net norms of the
    epoch before in columns 3 / 7 of the statistics row.  Against the PyTorch formulation of the same update (same start, same data):
    grad_norm / actor_grad_norm / critic_grad_norm agree, and they are NOT the last epoch's values."""
    import copy
    dev = torch.device("cuda")
    torch.manual_seed(5)
    a, c = nets.make_policy(policy)
    a.to(dev), c.to(dev)
    n, n_ep = 20000, 6
    obs, acts, logp, rtg, adv = _batch(n, 11, dev)
    var = torch.tensor(0.6, device=dev)
    cfg = dict(n_updates_per_iteration=n_ep, policy=policy, lr=3e-3)
    up_f = ppo.PPOUpdater(copy.deepcopy(a), copy.deepcopy(c), ppo.PPOConfig(update_arith=arith, **cfg), None, dev)
    up_t = ppo.PPOUpdater(copy.deepcopy(a), copy.deepcopy(c), ppo.PPOConfig(fused_update=False, **cfg), None, dev)
    assert up_f.fused and not up_t.fused
    sf = up_f.update(obs, acts, logp, rtg, var)
    st = up_t.update(obs, acts, logp, rtg, var)
    n_a = up_f.fp.module_numel[0]
    last = {"grad_norm": float(up_f.fp.grad.norm()), "actor_grad_norm": float(up_f.fp.grad[:n_a].norm()),
            "critic_grad_norm": float(up_f.fp.grad[n_a:].norm())}
    for k in ("grad_norm", "actor_grad_norm", "critic_grad_norm"):
        assert sf[k] == pytest.approx(st[k], rel=5e-3), (k, sf[k], st[k])
    # the series moves over the epochs at this learning rate, so a last-epoch value would not pass for the mean
    assert abs(last["critic_grad_norm"] - sf["critic_grad_norm"]) > 0.02 * sf["critic_grad_norm"], (last, sf)
