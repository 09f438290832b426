"""GPU tests of the PPO loop on the observation formats of BASELINE configs[3] and configs[4]:
42-D rows (36 beams) and float16 rows through the fused D-64-64 kernels (csrc/ppo_mlp64.hip, csrc/mlp64_policy.h) and the
persistent rollout (csrc/navsim.hip) -- same references and tolerances as the 16-D float32 tests of test_gpu_ppo.py:
PyTorch autograd in float32 for the update (project_ppo/src/ppo.py:305-397), the per-step HIP path (bit for bit) and the
oracle (flags exact, observations 1e-6 / rounded to half) for the rollout (ppo.py:463-641)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from navbot_ppo_amd import nets, ppo

pytestmark = pytest.mark.gpu


def _batch(n, seed, dev, d=16, half=False):
    g = torch.Generator().manual_seed(seed)
    obs = torch.rand((n, d), generator=g)
    acts = torch.stack([torch.rand(n, generator=g), torch.rand(n, generator=g) * 2 - 1], 1)
    acts[torch.rand(n, generator=g) < 0.2, 0] = 0.0
    acts[torch.rand(n, generator=g) < 0.1, 1] = 1.0
    logp = -1.2 - 2.3 * torch.rand(n, generator=g)
    rtg = torch.randn(n, generator=g) * 60 + 20
    adv = torch.randn(n, generator=g)
    adv[torch.rand(n, generator=g) < 0.05] = 0.0
    if half:
        obs = obs.half()
    return [t.to(dev).contiguous() for t in (obs, acts, logp, rtg, adv)]


def _updater(d, dev, scale=3.0, seed=3, **cfg):
    torch.manual_seed(seed)
    a, c = nets.make_policy("mlp64x2", d)
    a.to(dev), c.to(dev)
    with torch.no_grad():  # push the heads away from their init so clipping, saturation and relu masks all occur
        for p in list(a.parameters()) + list(c.parameters()):
            p.mul_(scale)
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="mlp64x2", **cfg), None, dev)
    assert up.fused_mlp64 and up.obs_dim == d
    return a, c, up


@pytest.mark.parametrize("n", [1, 128, 1000, 128 * 300 + 7, 512 * 4096])   # the last one is a configs[3] / configs[4] shard's batch
@pytest.mark.parametrize("d,half", [(42, False), (16, True), (42, True)])
def test_fused_gradients_match_autograd_on_wide_and_half_rows(n, d, half):
    """navppo_mlp64_loss_grad on 42-D rows and on float16 rows against PyTorch autograd (float32) of ppo.py:307-349,386 on the
    same rows (half rows widened, which is exact): the tolerances of test_fused_mlp64_gradients_match_autograd, unchanged."""
    dev = torch.device("cuda")
    a, c, up = _updater(d, dev, scale=3.0 if d == 16 else 2.0)
    obs, acts, logp, rtg, adv = _batch(n, n + d, dev, d, half)
    var = torch.tensor(0.5, device=dev)
    from _kinks import replace_kink_samples
    n_kink = replace_kink_samples(a, c, obs, acts, logp, rtg, adv, 0.5)   # (tests/_kinks.py: the bounds below are about arithmetic)
    assert n_kink <= 2 + n // 100
    up.fp.grad.zero_()
    al, cl, ratios, lp, _ = ppo.ppo_losses(a, c, obs.float(), acts, logp, rtg, adv, var, 0.2)
    (al + cl).backward()
    g_ref = up.fp.grad.clone()
    tol = [2e-4] * len(up.fp.params)
    if n > (1 << 17):
        # At 2.1 M samples two float32 evaluations of the same gradient differ by more than 2e-4 of a small tensor's scale for a
        # reason that is not arithmetic quality: ~2.7e8 relu units, a few dozen of them within float32 round-off of zero, and each
        # flipped mask moves a mean-gradient entry by ~|dH| x / n ~ 2e-7.  The reference there is the SAME losses under float64
        # autograd, and the kernel must be as close to it as PyTorch's own float32 autograd is (x 2), never worse than that.
        import copy
        a64, c64 = copy.deepcopy(a).double(), copy.deepcopy(c).double()
        al64, cl64, _, _, _ = ppo.ppo_losses(a64, c64, obs.double(), acts.double(), logp.double(), rtg.double(), adv.double(),
                                             var.double(), 0.2)
        gs = torch.autograd.grad(al64 + cl64, list(a64.parameters()) + list(c64.parameters()))
        g64 = torch.cat([t.reshape(-1) for t in gs])
        assert g64.numel() == g_ref.numel()
        offs = np.cumsum([0] + [q.numel() for q in up.fp.params])
        f32_err = [((g64[o:e] - g_ref[o:e].double()).abs().max() / (g64[o:e].abs().max() + 1e-12)).item()
                   for o, e in zip(offs[:-1], offs[1:])]
        print(f"float32 autograd vs float64 at n = {n}: " + " ".join(f"{e:.1e}" for e in f32_err))
        tol = [max(2e-4, 2 * e) for e in f32_err]
        g_ref = g64.float()
    kl_ref = ((ratios - 1) - (lp - logp)).mean().item()
    cf_ref = ((ratios - 1).abs() > 0.2).float().mean().item()
    if n >= 128:
        assert 0.02 < cf_ref < 0.98  # both branches of the clipped surrogate are exercised
    up.fp.grad.fill_(123.0)  # the kernel overwrites, it does not accumulate
    up._fused_loss_grad(obs, acts, logp, rtg, adv, 0.5)
    torch.cuda.synchronize()
    g = up.fp.grad
    st = up._fstats.cpu().numpy()
    off = 0
    errs = []
    for prm, tl in zip(up.fp.params, tol):
        k = prm.numel()
        ref, got = g_ref[off:off + k], g[off:off + k]
        scale = ref.abs().max().item() + 1e-12
        err = (ref - got).abs().max().item()
        errs.append(err / scale)
        assert err <= tl * scale + 1e-7, (tuple(prm.shape), err, scale, tl)
        off += k
    if n > (1 << 17):
        print(f"fused kernel     vs float64 at n = {n}: " + " ".join(f"{e:.1e}" for e in errs))
    assert off == (64 * d + 4354) + (64 * d + 4289)
    assert st[0] == pytest.approx(al.item(), rel=1e-4, abs=1e-6)
    assert st[4] == pytest.approx(cl.item(), rel=1e-4)
    assert st[1] == pytest.approx(kl_ref, rel=1e-3, abs=1e-5)
    assert st[2] == pytest.approx(cf_ref, abs=1e-6)
    # the one-net entry points of the multi-GPU pipeline write the same slices
    g_both = g.clone()
    up.fp.grad.fill_(7.0)
    stats = torch.zeros(8, device=dev)
    up._fused_loss_grad_net(0, obs, acts, logp, rtg, adv, 0.5, stats)
    up._fused_loss_grad_net(1, obs, acts, logp, rtg, adv, 0.5, stats)
    torch.cuda.synchronize()
    assert torch.equal(up.fp.grad, g_both)


@pytest.mark.parametrize("d,half", [(42, False), (16, True), (42, True)])
def test_fused_update_tracks_pytorch_update_on_wide_and_half_rows(d, half):
    """10 Adam epochs (navppo_mlp64_update_epoch) vs 10 with PyTorch autograd from the same start, as on 16-D float32 rows."""
    dev = torch.device("cuda")
    obs, acts, logp, rtg, adv = _batch(1 << 15, 5, dev, d, half)
    res = []
    for fused in (True, False):
        torch.manual_seed(11)
        a, c = nets.make_policy("mlp64x2", d)
        a.to(dev), c.to(dev)
        up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="mlp64x2", n_updates_per_iteration=10, fused_update=fused), None, dev)
        assert up.fused_mlp64 == fused
        st = up.update(obs, acts, logp, rtg, torch.tensor(0.8, device=dev))
        res.append((up.fp.flat.clone(), up.loss_history.clone(), st))
    (w1, h1, s1), (w0, h0, s0) = res
    np.testing.assert_allclose(h1.cpu().numpy(), h0.cpu().numpy(), rtol=2e-4, atol=1e-5)
    assert (w1 - w0).abs().max().item() < 3e-5
    for k in ("actor_loss", "critic_loss", "approx_kl", "clip_frac"):
        assert s1[k] == pytest.approx(s0[k], rel=2e-3, abs=2e-5), k


@pytest.mark.parametrize("d,half", [(42, False), (16, True), (42, True)])
def test_fused_act_and_value_on_wide_and_half_rows(d, half):
    """navppo_mlp64_act with explicit noise and navppo_mlp64_value against the PyTorch nets on the widened rows (ppo.py:696-704, :275)."""
    from navbot_ppo_amd._native import lib
    dev = torch.device("cuda")
    a, c, up = _updater(d, dev, scale=2.0, seed=5)
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    L = lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n in (1, 31, 128 * 5 + 77, 128 * 300 + 7):
        obs = torch.rand((n, d), device=dev) * 2 - 0.5
        if half:
            obs = obs.half()
        eps = torch.randn((n, 2), device=dev)
        var = torch.tensor(0.8, device=dev)
        act, lp, mean = torch.empty((n, 2), device=dev), torch.empty(n, device=dev), torch.empty((n, 2), device=dev)
        assert L.navppo_mlp64_act(ptr(up.fp.flat), ptr(obs), d, int(half), ptr(eps), n, ptr(var), 7, 0, None, 0, ptr(act), ptr(lp),
                                  ptr(mean), st) == 0
        with torch.no_grad():
            m_ref = a(obs.float())
            raw = m_ref + torch.sqrt(var) * eps
            a_ref = torch.stack([raw[:, 0].clamp(0, 1), raw[:, 1].clamp(-1, 1)], 1)
            lp_ref = ppo.gaussian_log_prob(m_ref, a_ref, var)
            v_ref = c(obs.float()).squeeze(-1)
        np.testing.assert_allclose(mean.cpu().numpy(), m_ref.cpu().numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(act.cpu().numpy(), a_ref.cpu().numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(lp.cpu().numpy(), lp_ref.cpu().numpy(), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(up._fused_value(obs).cpu().numpy(), v_ref.cpu().numpy(), rtol=2e-5, atol=2e-5)
    # wrong widths / dtypes are refused before any pointer crosses the boundary
    with pytest.raises(ValueError):
        up._fused_value(torch.rand((8, d + 2), device=dev))
    with pytest.raises(ValueError):
        up._fused_value(torch.rand((8, d), device=dev).double())
    assert L.navppo_mlp64_value(ptr(up.fp.flat), ptr(obs), 20, 0, 8, ptr(lp), st) != 0


ROLLOUT_CASES = [
    # N, T, map, per-env, beams, f16, sampler, forced shape
    (4096, 48, "stage_4", False, 36, False, None, None),      # configs[3]'s shard: rollout_kernel<36, 16>
    (200, 60, "stage_4", False, 36, False, None, None),       # ragged: 12 workgroups + 8 envs
    (96, 40, "stage_2", True, 36, False, None, None),         # per-env maps, 36 beams
    (4608, 36, "stage_4", False, 36, False, None, None),      # beyond one round of workgroups
    (300, 60, "stage_1", False, 10, True, None, None),        # f16 rows, 16-env shape
    (8192, 34, "house", False, 10, True, "small_house", None),   # configs[4]'s shard: rollout_big_kernel, tile boxes, tables, f16
    (16384, 30, "stage_2", True, 10, True, None, None),       # f16 on the 128-segment passes
    (200, 50, "stage_1", False, 10, True, None, 64),          # the 64-env shape forced onto a small ragged shard
    (120, 40, "stage_4", False, 36, True, None, None),        # both at once
    (1000, 36, "house", False, 36, False, "small_house", None),   # 36 beams on the house map: the 16-env rollout kernel's tile-box cast
    (2048, 34, "house", False, 10, True, "small_house", None),    # float16 rows + tile boxes on the 16-env shape
]


@pytest.mark.parametrize("sens", [False, True])
@pytest.mark.parametrize("N,T,map_name,per_env,beams,half,sampler,epb", ROLLOUT_CASES)
def test_persistent_rollout_equals_per_step_rollout_on_wide_and_half_rows(N, T, map_name, per_env, beams, half, sampler, epb, sens):
    """navsim_rollout_mlp64 with 36 beams / float16 buffers against T pairs of navppo_mlp64_act / navsim_step: every rollout
    buffer and the simulator state bit-identical over two consecutive rollouts (as test_persistent_rollout_equals_per_step_rollout
    for 10 beams / float32).  With float16 buffers the in-kernel policy reads its observation tile rounded to half -- what the
    per-step policy launch reads from the buffer."""
    from navbot_ppo_amd.env import VecEnv
    kw = dict(lidar_noise_sigma=0.01, lidar_below_min="gazebo") if sens else {}
    outs = []
    for persistent in (True, False):
        env = VecEnv(N, map=map_name, n_beams=beams, max_episode_steps=30, seed=3, per_env_map=per_env, sampler=sampler,
                     obs_f16=half, **kw)
        if epb:
            env.sim.set_shape(epb)
        cfg = ppo.PPOConfig(rollout_len=T, max_episode_steps=30, n_updates_per_iteration=1, policy="mlp64x2", seed=5,
                            persistent_rollout=persistent, use_graph=False)
        tr = ppo.PPOTrainer(env, cfg)
        assert tr.uses_persistent_rollout is persistent   # the two sides of the comparison do take different paths
        assert tr.updater.fused_mlp64 and tr.obs_buf.dtype == (torch.float16 if half else torch.float32)
        if persistent:
            inf = env.sim.info()
            assert inf["rollout_kind"] == (2 if (beams == 10 and (N > 4096 or epb == 64)) else 1), inf
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
    bits = lambda x: x.view(torch.int32) if x.dtype == torch.float32 else x.view(torch.int16) if x.dtype == torch.float16 else x
    for ra, rb in zip(a, b):
        for x, y in zip(ra, rb):
            assert torch.equal(bits(x), bits(y))
    for k in sa:
        np.testing.assert_array_equal(sa[k], sb[k])


@pytest.mark.parametrize("N,T,cap,lo,n_s,world,beams,half", [
    (4096, 256, 200, 2000, 96, "stage_4", 36, False),          # configs[3]'s shard, closed loop
    (8192, 120, 100, 8192 - 96, 96, "house", 10, True),        # configs[4]'s shard, closed loop, float16 buffers
    (1024, 160, 150, 500, 64, "stage_1", 10, True)])           # float16 on the 16-env shape
def test_wide_and_half_rollouts_against_the_oracle(N, T, cap, lo, n_s, world, beams, half):
    """The closed-loop rollout on configs[3]'s / configs[4]'s shards checked DIRECTLY against the oracle: the actions the kernel
    recorded for a block of envs are replayed on an OracleSim keyed by the same global env ids; flags bit-exact, float32
    observation rows within 1e-6, float16 rows equal to the oracle's rows rounded to half (up to 1e-6 before the rounding: one
    half ulp at a rounding boundary), rewards 1e-5; the stored log-probs are those of the stored actions under PyTorch's
    evaluation of the same actor on the stored (widened) rows (ppo.py:696-704)."""
    from navbot_ppo_amd import maps
    from navbot_ppo_amd.env import VecEnv
    from oracle import navsim_oracle as O
    house = world == "house"
    env = VecEnv(N, map=world, n_beams=beams, max_episode_steps=cap, seed=7, obs_f16=half, sampler="small_house" if house else None)
    cfg = ppo.PPOConfig(rollout_len=T, max_episode_steps=cap, policy="mlp64x2", seed=3)
    tr = ppo.PPOTrainer(env, cfg)
    with torch.no_grad():   # drive: a forward bias so that collisions / arrivals happen inside the rollout, not only timeouts
        tr.actor.layer3.bias.add_(2.0)
    tr.rollout()
    torch.cuda.synchronize()
    assert tr.updater.fused_mlp64 and cfg.persistent_rollout and env.sim.info()["rollout_kind"] == (2 if N > 4096 else 1)
    D = beams + 6
    sl = slice(lo, lo + n_s)
    acts = tr.act_buf[:, sl].cpu().numpy()
    cpu = O.OracleSim(n_s, n_beams=beams, max_episode_steps=cap, auto_reset=True, seed=7, env_id_base=lo)
    cpu.set_map(maps.by_name(world))
    rr, rs = maps.goal_rects(world)
    cpu.set_goal_rects(0, rr)
    cpu.set_goal_rects(1, rs)
    if house:
        cpu.set_spawn_sampler(*maps.spawn_tables("small_house"))
    obs = tr.obs_buf[:, sl].float().cpu().numpy()

    def check_obs(got, want, msg):
        if half:   # the oracle's float32 row rounded to half, or its neighbour where 1e-6 moves the row across a rounding boundary
            w16 = want.astype(np.float16).astype(np.float32)
            ok = (got == w16) | (got == (want + 1e-6).astype(np.float16).astype(np.float32)) | \
                 (got == (want - 1e-6).astype(np.float16).astype(np.float32))
            assert ok.all(), (msg, got[~ok][:4], want[~ok][:4])
        else:
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-6, err_msg=msg)

    check_obs(obs[0], cpu.reset(), "reset")
    g = {k: getattr(tr, k + "_buf")[:, sl].cpu().numpy() for k in ("rew", "done", "arrive", "ended", "epret", "eplen", "eppath")}
    n_end = 0
    for t in range(T):
        out = cpu.step(acts[t])
        for k in ("done", "arrive", "ended"):
            np.testing.assert_array_equal(g[k][t], out[k], err_msg=f"{k}, step {t}")
        check_obs(obs[t + 1], out["obs"], f"obs, step {t}")
        np.testing.assert_allclose(g["rew"][t], out["reward"], rtol=1e-5, atol=1e-5, err_msg=f"reward, step {t}")
        e = out["ended"].astype(bool)
        np.testing.assert_array_equal(g["eplen"][t][e], out["ep_length"][e])
        np.testing.assert_allclose(g["epret"][t][e], out["ep_return"][e], rtol=1e-5, atol=1e-4)
        n_end += int(e.sum())
    assert n_end >= n_s and int(g["done"].sum()) > 0
    with torch.no_grad():
        o = tr.obs_buf[:T].reshape(T * N, D).float()
        lp_ref = ppo.gaussian_log_prob(tr.actor(o), tr.act_buf.reshape(T * N, 2), tr.var)
    np.testing.assert_allclose(tr.logp_buf.reshape(-1).cpu().numpy(), lp_ref.cpu().numpy(), rtol=1e-4, atol=3e-5)
    env.close()


@pytest.mark.parametrize("beams,half,world,sampler", [(36, False, "stage_4", None), (10, True, "stage_1", None),
                                                      (10, True, "house", "small_house")])
def test_trainer_learns_on_wide_and_half_rows(beams, half, world, sampler):
    """PPOTrainer end to end on configs[3]'s / configs[4]'s observation format: HIP rollout + fused update, finite statistics,
    the policy moves, returns improve over a few iterations on the small maps."""
    from navbot_ppo_amd.env import VecEnv
    env = VecEnv(1024, map=world, n_beams=beams, max_episode_steps=150, seed=2, obs_f16=half, sampler=sampler)
    cfg = ppo.PPOConfig(rollout_len=160, max_episode_steps=150, n_updates_per_iteration=8, policy="mlp64x2", seed=1)
    tr = ppo.PPOTrainer(env, cfg)
    assert tr.updater.fused == "navppo_mlp64" and tr.updater.obs_dim == beams + 6
    w0 = tr.updater.fp.flat.clone()
    logs = [tr.iteration() for _ in range(6)]
    for lg in logs:
        for k in ("actor_loss", "critic_loss", "approx_kl", "clip_frac", "avg_ep_rews"):
            assert np.isfinite(lg[k]), (k, lg[k])
        assert lg["episodes"] > 0
    assert (tr.updater.fp.flat - w0).abs().max().item() > 1e-4
    if world != "house":
        assert logs[-1]["avg_ep_rews"] > logs[0]["avg_ep_rews"]
    env.close()


def test_buffers_of_the_wrong_type_are_refused():
    """The C ABI takes raw pointers; the Python shim must know what is behind them (PPOTrainer over an f16
    VecEnv used to write half rows into a float32 obs_buf, silently)."""
    from navbot_ppo_amd.env import NavsimError, VecEnv
    dev = torch.device("cuda")
    env = VecEnv(64, map="stage_1", obs_f16=True)
    sim = env.sim
    io = sim.alloc_io()
    act = torch.zeros((64, 2), device=dev)
    sim.reset(io.obs)
    sim.step(act, io.obs, io.reward, io.done, io.arrive)
    f32_obs = torch.zeros((64, 16), device=dev)
    with pytest.raises(NavsimError):
        sim.reset(f32_obs)                                    # float32 rows into an f16 handle
    with pytest.raises(NavsimError):
        sim.step(act, f32_obs, io.reward, io.done, io.arrive)
    with pytest.raises(NavsimError):
        sim.step(act.double(), io.obs, io.reward, io.done, io.arrive)
    with pytest.raises(NavsimError):
        sim.step(act, io.obs[:32], io.reward, io.done, io.arrive)   # too few rows
    with pytest.raises(NavsimError):
        sim.step(act.cpu(), io.obs, io.reward, io.done, io.arrive)  # host tensor
    with pytest.raises(NavsimError):
        sim.step(act, io.obs, io.reward, io.done.int(), io.arrive)
    with pytest.raises(NavsimError):
        sim.step_seq(torch.zeros((4, 64, 2), device=dev), torch.zeros((4, 64, 16), device=dev), torch.zeros((4, 64), device=dev),
                     torch.zeros((4, 64), dtype=torch.uint8, device=dev), torch.zeros((4, 64), dtype=torch.uint8, device=dev))
    tr = ppo.PPOTrainer(env, ppo.PPOConfig(rollout_len=8, policy="mlp64x2"))
    assert tr.obs_buf.dtype == torch.float16                  # the trainer's buffers follow the simulator's row type
    tr5 = ppo.PPOTrainer(env, ppo.PPOConfig(rollout_len=8, policy="resmlp512"))   # round 6: the 512-wide kernels read float16 rows too
    assert tr5.obs_buf.dtype == torch.float16 and tr5.updater.fused_resmlp512 and tr5.uses_persistent_rollout
    env.close()


def test_two_handles_with_different_forced_shapes_in_one_process():
    """navsim_set_shape is per-handle state: two handles with different forced workgroup shapes, alive at
    the same time and stepped alternately, both give the oracle's rows; navsim_get_info reports what each launches."""
    from navbot_ppo_amd import maps
    from navbot_ppo_amd.env import VecEnv
    from oracle import navsim_oracle as O
    dev = torch.device("cuda")
    N, T = 320, 24
    envs = [VecEnv(N, map="stage_2", per_env_map=True, max_episode_steps=20, seed=5, envs_per_workgroup=e, pair_cast=p)
            for e, p in ((8, True), (64, False), (None, None))]
    infos = [e.sim.info() for e in envs]
    assert [(i["step_epb"], i["step_waves"], i["step_cast"]) for i in infos[:2]] == [(8, 8, 1), (64, 16, 0)]
    assert infos[2]["forced_epb"] == 0 and infos[2]["step_epb"] == 16 and infos[0]["n_segments"] == 128 and infos[0]["per_env_map"] == 1
    for i in infos:   # the resources of the selected instantiations, as the loaded code object reports them
        for k in ("step", "seq"):
            assert 32 <= i[k + "_vgprs"] <= 512 and 0 < i[k + "_lds_bytes"] <= 160 * 1024 and 0 <= i[k + "_scratch_bytes"] <= 512, i
    cpu = O.OracleSim(N, max_episode_steps=20, auto_reset=True, seed=5)
    cpu.set_map(envs[0].sim._seg.cpu().numpy(), per_env=True)
    rr, rs = maps.goal_rects("stage_2")
    cpu.set_goal_rects(0, rr)
    cpu.set_goal_rects(1, rs)
    want = cpu.reset()
    for e in envs:
        np.testing.assert_allclose(e.reset().cpu().numpy(), want, rtol=0, atol=1e-6)
    g = torch.Generator().manual_seed(1)
    for t in range(T):
        a = torch.stack([torch.rand(N, generator=g), torch.rand(N, generator=g) * 2 - 1], 1)
        out = cpu.step(a.numpy())
        for e in envs:   # alternately: a later handle's shape must not leak into an earlier one
            obs, rew, done, arrive = e.step(a.to(dev))
            np.testing.assert_array_equal(done.cpu().numpy(), out["done"])
            np.testing.assert_array_equal(arrive.cpu().numpy(), out["arrive"])
            np.testing.assert_allclose(obs.cpu().numpy(), out["obs"], rtol=0, atol=1e-6)
    envs[0].sim.set_shape(32)
    assert envs[0].sim.info()["step_epb"] == 32 and envs[1].sim.info()["step_epb"] == 64
    with pytest.raises(Exception):
        envs[0].sim.set_shape(5)
    for e in envs:
        e.close()


def _format_batch(d, half, n=768, seed=11):
    g = torch.Generator().manual_seed(seed)
    obs = torch.rand((n, d), generator=g) * 1.5 - 0.25
    acts = torch.rand((n, 2), generator=g)
    acts[:, 1] = acts[:, 1] * 2 - 1
    logp = -1.0 - torch.rand(n, generator=g)
    rtg = torch.randn(n, generator=g) * 3
    return (obs.half() if half else obs), acts, logp, rtg


def _dp_format_worker(rank, world, port, path, d, half, overlap):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      NAVBOT_DIST_BACKEND="gloo")   # RCCL refuses two ranks on one device: gloo carries the all-reduce here
    from navbot_ppo_amd import nets, ppo
    ctx = ppo.DistCtx(device="cuda:0")
    torch.manual_seed(100 + rank)   # rank 0's parameters are broadcast
    a, c = nets.make_policy("mlp64x2", d)
    a.cuda(), c.cuda()
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(n_updates_per_iteration=5, policy="mlp64x2", overlap_allreduce=overlap), ctx,
                        torch.device("cuda:0"))
    assert up.fused_mlp64 and up.obs_dim == d
    obs, acts, logp, rtg = _format_batch(d, half)
    lo, hi = ctx.shard(obs.shape[0])
    up.update(obs[lo:hi].cuda(), acts[lo:hi].cuda(), logp[lo:hi].cuda(), rtg[lo:hi].cuda(), torch.tensor(0.8, device="cuda"))
    torch.save({"flat": up.fp.flat.cpu()}, f"{path}.{rank}")
    ctx.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("d,half,overlap", [(42, False, False), (42, False, True), (16, True, True), (42, True, False)])
def test_two_rank_update_on_row_formats_equals_single_rank(tmp_path, d, half, overlap):
    """BASELINE configs[3] (36 beams -> 42-D rows, "RCCL grad all-reduce") and configs[4] (float16 observation buffers) are 8-GPU
    TRAINING configurations (ppo.py:305-397 on each rank's shard + one gradient all-reduce per epoch): the N > 1 update path on these
    row formats -- fused passes on the shard, all-reduce of the flat gradient (overlap: the per-net pipeline), scale + Adam kernel --
    with two ranks on halves of a batch == the single-rank path on the whole batch."""
    from _ranks import spawn_ranks
    from navbot_ppo_amd import nets, ppo
    path = str(tmp_path / "dpf")
    spawn_ranks(_dp_format_worker, 2, lambda port: (2, port, path, d, half, overlap))
    r0, r1 = torch.load(path + ".0"), torch.load(path + ".1")
    assert torch.equal(r0["flat"], r1["flat"])
    torch.manual_seed(100)
    a, c = nets.make_policy("mlp64x2", d)
    a.cuda(), c.cuda()
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(n_updates_per_iteration=5, policy="mlp64x2"), None, torch.device("cuda:0"))
    obs, acts, logp, rtg = _format_batch(d, half)
    before = up.fp.flat.clone()
    up.update(obs.cuda(), acts.cuda(), logp.cuda(), rtg.cuda(), torch.tensor(0.8, device="cuda"))
    assert (up.fp.flat - before).abs().max().item() > 1e-3   # five Adam steps moved the parameters
    np.testing.assert_allclose(r0["flat"].numpy(), up.fp.flat.cpu().numpy(), rtol=0, atol=5e-6)


def test_split_pass_entry_points_refuse_other_widths():
    """navppo_mlp64_bf16x3_* exist for the reference's two observation widths (16, 42): any other width, a missing buffer or a
    misaligned one is an error code with a message, never a launch (include/navppo.h)."""
    from navbot_ppo_amd._native import lib
    L = lib()
    dev = torch.device("cuda")
    assert L.navppo_mlp64_bf16x3_prep_bytes(64, 16) == 2 * 6144 and L.navppo_mlp64_bf16x3_prep_bytes(65, 42) == 3 * 18432
    assert L.navppo_mlp64_bf16x3_prep_bytes(64, 20) == 0 and L.navppo_mlp64_bf16x3_prep_bytes(0, 16) == 0
    obs = torch.zeros((64, 42), device=dev)
    prep = torch.zeros(L.navppo_mlp64_bf16x3_prep_bytes(64, 42) + 16, dtype=torch.uint8, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.navppo_mlp64_bf16x3_prepare(p(obs), 42, 0, 64, p(prep), st) == 0
    assert L.navppo_mlp64_bf16x3_prepare(p(obs), 20, 0, 64, p(prep), st) == -1 and b"obs_dim" in L.navppo_last_error()
    assert L.navppo_mlp64_bf16x3_prepare(p(obs), 42, 0, 64, C.c_void_p(prep.data_ptr() + 8), st) == -1   # prep not 16-byte aligned
    assert L.navppo_mlp64_bf16x3_prepare(p(obs), 42, 0, 0, p(prep), st) == -1
    a, c, up = _updater(42, dev)
    acts, lp, rtg, adv = torch.zeros((64, 2), device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    ws = up._workspace(64)
    stats = torch.zeros(8, device=dev)
    args = lambda d: (p(up.fp.flat), p(prep), d, p(acts), p(lp), p(rtg), p(adv), 64, 0.5, 0.2, p(up.fp.grad), p(stats), p(ws), st)
    assert L.navppo_mlp64_bf16x3_loss_grad(*args(42)) == 0
    assert L.navppo_mlp64_bf16x3_loss_grad(*args(17)) == -1 and b"obs_dim" in L.navppo_last_error()
    assert L.navppo_mlp64_bf16x3_loss_grad_net(2, *args(42)) == -1
    torch.cuda.synchronize()
