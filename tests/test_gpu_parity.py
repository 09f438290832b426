"""GPU parity tests proper: the HIP path (through the C ABI, libnavsim.so) against the CPU oracle
on the same seeded inputs, against golden vectors recorded from the reference, and through
size-independent properties at BASELINE sizes.

Tolerances (BASELINE.json north_star): collision / arrival / ended flags BIT-EXACT; float
observations and rewards within 1e-5 (we assert 1e-6 absolute on observations, which are O(1),
and 1e-5 relative on rewards).
"""
import math
import os

import numpy as np
import pytest
import torch

from navbot_ppo_amd import maps
from oracle import navsim_oracle as O

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
OBS_ATOL = 1e-6
REW_RTOL = 1e-5


def _mk(N, seg, per_env=False, B=10, **kw):
    from navbot_ppo_amd.env import NavSim
    gpu = NavSim(N, n_beams=B, **kw)
    cpu = O.OracleSim(N, n_beams=B, **{k: v for k, v in kw.items() if k != "obs_f16"})
    gpu.set_map(seg, per_env=per_env)
    cpu.set_map(seg, per_env=per_env)
    return gpu, cpu


def _actions(rng, K, N):
    a = np.stack([rng.uniform(0, 1, (K, N)), rng.uniform(-1, 1, (K, N))], 2).astype(np.float32)
    q = max(1, N // 4)  # a quarter of the envs drive fast and nearly straight: collisions and arrivals happen
    a[:, :q, 1] *= 0.05
    a[:, :q, 0] = 0.8 + 0.2 * a[:, :q, 0]
    return a


def _lockstep(gpu, cpu, actions, check_state_every=25):
    N = gpu.N
    io = gpu.alloc_io()
    obs_g = gpu.reset(io.obs).float().cpu().numpy()
    obs_c = cpu.reset()
    np.testing.assert_allclose(obs_g, obs_c, rtol=0, atol=OBS_ATOL)
    stats = dict(done=0, arrive=0, ended=0, exact_obs=0, total=0)
    for k in range(actions.shape[0]):
        a = torch.from_numpy(actions[k]).to(gpu.device)
        gpu.step(a, io.obs, io.reward, io.done, io.arrive, io.ended, io.ep_return, io.ep_length, ep_path=io.ep_path)
        out = cpu.step(actions[k])
        og = io.obs.float().cpu().numpy()
        np.testing.assert_array_equal(io.done.cpu().numpy(), out["done"], err_msg=f"done, step {k}")
        np.testing.assert_array_equal(io.arrive.cpu().numpy(), out["arrive"], err_msg=f"arrive, step {k}")
        np.testing.assert_array_equal(io.ended.cpu().numpy(), out["ended"], err_msg=f"ended, step {k}")
        np.testing.assert_allclose(og, out["obs"], rtol=0, atol=OBS_ATOL, err_msg=f"obs, step {k}")
        np.testing.assert_allclose(io.reward.cpu().numpy(), out["reward"], rtol=REW_RTOL, atol=1e-5, err_msg=f"reward, step {k}")
        e = out["ended"].astype(bool)
        np.testing.assert_array_equal(io.ep_length.cpu().numpy()[e], out["ep_length"][e])
        np.testing.assert_allclose(io.ep_return.cpu().numpy()[e], out["ep_return"][e], rtol=REW_RTOL, atol=1e-4)
        np.testing.assert_allclose(io.ep_path.cpu().numpy()[e], out["ep_path"][e], rtol=1e-6, atol=1e-7)
        stats["done"] += int(out["done"].sum())
        stats["arrive"] += int(out["arrive"].sum())
        stats["ended"] += int(e.sum())
        stats["exact_obs"] += int((og == out["obs"]).all(axis=1).sum())
        stats["total"] += N
        if (k + 1) % check_state_every == 0 or k == actions.shape[0] - 1:
            sg, sc = gpu.get_state(), cpu.get_state()
            np.testing.assert_allclose(sg["pose"], sc["pose"], rtol=0, atol=1e-11)
            np.testing.assert_array_equal(sg["goal"], sc["goal"])  # same Philox stream, same rejection decisions
            np.testing.assert_allclose(sg["past_dist"], sc["past_dist"], rtol=1e-14)
            np.testing.assert_array_equal(sg["ep_step"], sc["ep_step"])
            np.testing.assert_array_equal(sg["rng_ctr"], sc["rng_ctr"])
            np.testing.assert_array_equal(sg["past_action"], sc["past_action"])
    return stats


def test_step_parity_stage1_shared_map_autoreset():
    rng = np.random.default_rng(11)
    gpu, cpu = _mk(512, maps.stage_1(), max_episode_steps=90, auto_reset=True, seed=5)
    st = _lockstep(gpu, cpu, _actions(rng, 200, 512))
    assert st["done"] > 20 and st["ended"] > 512  # collisions and timeouts both exercised
    assert st["exact_obs"] > 0.99 * st["total"]   # nearly every row is bit-identical to the oracle


def test_step_parity_arrivals_and_respawn():
    # goals pulled in close to the spawn so arrivals are frequent; respawn_on_arrive path (Env drop-in mode)
    rng = np.random.default_rng(12)
    N = 256
    gpu, cpu = _mk(N, maps.stage_1(), max_episode_steps=0, auto_reset=False, respawn_on_arrive=True, seed=7,
                   goal_box=(-0.6, 0.6))
    for s in (gpu, cpu):
        s.set_goal_rects(0, np.zeros((0, 4)))
        s.set_goal_rects(1, np.zeros((0, 4)))
    a = _actions(rng, 120, N)
    a[..., 0] = np.maximum(a[..., 0], 0.5)
    st = _lockstep(gpu, cpu, a)
    assert st["arrive"] > 30


def test_step_parity_per_env_maps_stage2():
    rng = np.random.default_rng(13)
    N = 320  # five full blocks
    seg = maps.replicate_per_env(maps.stage_2(), N, seed=3)
    gpu, cpu = _mk(N, seg, per_env=True, max_episode_steps=40, auto_reset=True, seed=9)
    rr, rs = maps.goal_rects("stage_2")
    for s in (gpu, cpu):
        s.set_goal_rects(0, rr)
        s.set_goal_rects(1, rs)
    st = _lockstep(gpu, cpu, _actions(rng, 100, N))
    assert st["ended"] > N


@pytest.mark.parametrize("N", [1, 63, 65, 100])
def test_ragged_env_counts(N):
    rng = np.random.default_rng(14 + N)
    gpu, cpu = _mk(N, maps.stage_1(), max_episode_steps=15, auto_reset=True, seed=N)
    _lockstep(gpu, cpu, _actions(rng, 40, N))
    segp = maps.replicate_per_env(maps.stage_4(), N, seed=1)  # S=64 per env
    gpu, cpu = _mk(N, segp, per_env=True, max_episode_steps=15, auto_reset=True, seed=N)
    _lockstep(gpu, cpu, _actions(rng, 25, N))


def test_36_beams_stage4():
    rng = np.random.default_rng(15)
    gpu, cpu = _mk(192, maps.stage_4(), B=36, max_episode_steps=30, auto_reset=True, seed=2)
    assert gpu.D == 42
    _lockstep(gpu, cpu, _actions(rng, 70, 192))
    segp = maps.replicate_per_env(maps.stage_1(), 128, seed=2)  # S=32 < 64 lanes
    gpu, cpu = _mk(128, segp, per_env=True, B=36, max_episode_steps=30, auto_reset=True, seed=2)
    _lockstep(gpu, cpu, _actions(rng, 40, 128))


def test_large_shared_map_tiles_through_lds():
    # 5000 segments > one 2048-segment LDS tile: a dense random clutter far from the robot + stage_1
    rng = np.random.default_rng(16)
    clutter = rng.uniform(-3.8, 3.8, (5000 - 32, 2))
    d = rng.uniform(-0.05, 0.05, (5000 - 32, 2))
    seg = np.concatenate([maps.stage_1(), np.concatenate([clutter, clutter + d], 1).astype(np.float32)])
    keep = np.hypot(seg[:, 0], seg[:, 1]) > 0.5  # keep the spawn clear
    seg = seg[keep]
    gpu, cpu = _mk(96, seg, max_episode_steps=20, auto_reset=True, seed=3)
    st = _lockstep(gpu, cpu, _actions(rng, 30, 96))
    assert st["done"] > 0


def test_raycast_entry_point_matches_oracle():
    from navbot_ppo_amd.env import NavSim
    rng = np.random.default_rng(17)
    N = 300
    seg = maps.stage_2()
    gpu = NavSim(N)
    gpu.set_map(seg)
    pose = np.stack([rng.uniform(-3.7, 3.7, N), rng.uniform(-3.7, 3.7, N), rng.uniform(-7, 7, N)], 1)
    got = gpu.raycast(torch.from_numpy(pose)).cpu().numpy()
    want = np.stack([O.raycast(seg, *p) for p in pose])
    np.testing.assert_array_equal(np.isinf(got), np.isinf(want))
    fin = np.isfinite(want)
    np.testing.assert_allclose(got[fin], want[fin], rtol=0, atol=1e-6)
    assert (got[fin] == want[fin]).mean() > 0.999


def test_g9_closed_loop_against_reference_outputs():
    """GPU sim replays the G9 action tape; obs/reward/flags must match what the REFERENCE Env.step
    produced for the same trajectory (recorded by tests/golden/generate_golden.py)."""
    from navbot_ppo_amd.env import NavSim
    d = np.load(os.path.join(G, "g9_closed_loop.npz"))
    K, E = d["actions"].shape[:2]
    gpu = NavSim(E, seed=int(d["seed"]))
    gpu.set_map(maps.stage_1())
    io = gpu.alloc_io()
    gpu.reset(io.obs)
    goals = d["goals"]
    gpu.set_state(goal=goals, past_dist=np.hypot(goals[:, 0], goals[:, 1]))
    past = torch.zeros((E, 2), dtype=torch.float32, device=gpu.device)
    n = 0
    for k in range(K):
        a = torch.from_numpy(d["actions"][k]).to(gpu.device)
        gpu.step(a, io.obs, io.reward, io.done, io.arrive, io.ended, past_action=past)
        al = d["alive"][k].astype(bool)
        np.testing.assert_allclose(io.obs.cpu().numpy()[al], d["ref_obs"][k][al].astype(np.float32), rtol=0, atol=OBS_ATOL)
        np.testing.assert_allclose(io.reward.cpu().numpy()[al], d["ref_rew"][k][al], rtol=REW_RTOL, atol=1e-5)
        np.testing.assert_array_equal(io.done.cpu().numpy()[al], d["ref_flags"][k][al, 0])
        np.testing.assert_array_equal(io.arrive.cpu().numpy()[al], d["ref_flags"][k][al, 1])
        n += int(al.sum())
        past = a
    assert n > 1000


def test_g1_goal_angles_through_the_kernel():
    """Reference getOdometry goldens (yaw-only, unit quaternions) pushed through the GPU step with a zero
    action: obs[13:16] = yaw/360, rel_theta/360, diff/180."""
    from navbot_ppo_amd.env import NavSim
    d = np.load(os.path.join(G, "g1_odometry.npz"))
    inp, out = d["inp"], d["out"]
    sel = (inp[:, 2] == 0) & (inp[:, 3] == 0) & (np.abs(inp[:, 4] ** 2 + inp[:, 5] ** 2 - 1) < 1e-12)
    inp, out = inp[sel], out[sel]
    th = 2 * np.arctan2(inp[:, 4], inp[:, 5])
    deg = np.degrees(th)
    not_tie = np.abs(deg - np.round(deg) - 0.5) % 1 > 1e-9  # exact .5-degree ties depend on the last ulp of sin/cos
    tie = np.abs(np.abs(deg - np.floor(deg)) - 0.5) < 1e-9
    N = len(inp)
    gpu = NavSim(N)
    gpu.set_map(np.array([[50, 50, 51, 50]], dtype=np.float32))  # nothing in range
    io = gpu.alloc_io()
    gpu.reset(io.obs)
    pose = np.stack([inp[:, 0], inp[:, 1], th], 1)
    gpu.set_state(pose=pose, goal=inp[:, 6:8], past_dist=np.ones(N))
    gpu.step(torch.zeros((N, 2), device=gpu.device), io.obs, io.reward, io.done, io.arrive, io.ended)
    o = io.obs.cpu().numpy()
    ok = ~tie
    assert ok.sum() > 1500
    np.testing.assert_array_equal(o[ok, 13], (out[ok, 0] / 360).astype(np.float32))
    np.testing.assert_array_equal(o[ok, 14], (out[ok, 1] / 360).astype(np.float32))
    np.testing.assert_array_equal(o[ok, 15], (out[ok, 2] / 180).astype(np.float32))
    assert np.all(o[:, :10] == 1.0)  # +inf -> 3.5 -> 1.0


def test_g1_get_odometry_on_the_device_all_cases():
    """a-3 in full: every reference getOdometry golden (G1: 2307 cases incl. general quaternions, exact half-degree yaw
    ties, quadrant edges, dx == 0 / dy == 0, wrap-around) through navsim_odometry -- the device functions the step kernel
    uses -- bit for bit (environment_new.py:138-181)."""
    from navbot_ppo_amd.env import odometry
    d = np.load(os.path.join(G, "g1_odometry.npz"))
    inp, out = d["inp"], d["out"]
    dev = torch.device("cuda:0")
    got = odometry(torch.from_numpy(inp[:, 0].copy()).to(dev), torch.from_numpy(inp[:, 1].copy()).to(dev),
                   torch.from_numpy(inp[:, 2:6].copy()).to(dev), torch.from_numpy(inp[:, 6:8].copy()).to(dev)).cpu().numpy()
    np.testing.assert_array_equal(got, out)
    assert not np.signbit(got).any() or (got[np.signbit(got)] != 0).all()   # Python's round() never yields -0.0 for the yaw
    assert odometry(torch.zeros(0, device=dev), torch.zeros(0, device=dev), torch.zeros((0, 4), device=dev),
                    torch.zeros((0, 2), device=dev)).shape == (0, 3)


def assert_rtg_close(got, ref):
    """The return scan's contract on the T-split path (N % 16 == 0): every float32 within ONE ulp of ppo.py:660-669, and
    all but a vanishing share identical (a differing carry rounding moves a store with probability ~2e-8 per element)."""
    assert got.shape == ref.shape and got.dtype == ref.dtype == np.float32
    d = np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
    assert d.max(initial=0) <= 1, f"return scan off by {d.max()} ulp"
    assert int((d != 0).sum()) <= 1 + got.size // 100000, f"{int((d != 0).sum())} of {got.size} stores differ"


def test_rtg_scan_parity_and_golden(monkeypatch):
    from navbot_ppo_amd.env import rtg_scan
    rng = np.random.default_rng(18)
    # (T, N, episode-end rate): N % 16 != 0 -> serial kernel (bit-exact); otherwise the T-split kernel, incl. columns
    # without any episode end (every chunk takes its carry), T below / at / above one 512-row super-chunk and ragged T
    for T, N, p_end in [(1, 1, .03), (7, 3, .03), (512, 100, .03), (33, 65, .03), (128, 4096, .03), (512, 4096, .004),
                        (512, 256, 0.), (1, 16, .5), (31, 32, .05), (513, 48, .001), (1300, 640, .002), (1024, 64, 0.)]:
        rew = rng.uniform(-25, 25, (T, N)).astype(np.float32)
        rew[rng.random((T, N)) < 0.02] = 120.0
        ended = (rng.random((T, N)) < p_end).astype(np.uint8)
        ref = O.compute_rtgs_tn(rew, ended, 0.99)
        got = rtg_scan(torch.from_numpy(rew).cuda(), torch.from_numpy(ended).cuda(), 0.99).cpu().numpy()
        if N % 16:
            np.testing.assert_array_equal(got, ref)  # bit-exact: same f64 recurrence
        else:
            assert_rtg_close(got, ref)
            # behind an episode end inside its own 32-row chunk a row never sees a carry: identical bits
            t = np.arange(T)[:, None]
            chunk_end = np.minimum(T, T - ((T - 1 - t) // 32) * 32)   # chunks are cut from the batch end
            cs = np.concatenate([np.zeros((1, N), np.int64), np.cumsum(ended, 0)])
            has_end_ahead = cs[chunk_end[:, 0]] - cs[:-1] > 0   # an end in rows [t, chunk_end)
            np.testing.assert_array_equal(got[has_end_ahead], ref[has_end_ahead])
            ex = rtg_scan(torch.from_numpy(rew).cuda(), torch.from_numpy(ended).cuda(), 0.99, exact=True).cpu().numpy()   # the serial kernel: every bit
            np.testing.assert_array_equal(ex, ref)
    # reference golden (ragged episodes of one env -> one column)
    d = np.load(os.path.join(G, "g5_rtgs.npz"))
    rews, lens, gammas, out = d["rews"], d["lens"], d["gammas"], d["out"]
    ro = oo = k = 0
    case = []
    for L in lens:
        if L >= 0:
            case.append(rews[ro:ro + L])
            ro += L
            continue
        flat = np.concatenate(case) if case else np.zeros(0)
        n = len(flat)
        if n and np.all(flat.astype(np.float32).astype(np.float64) == flat):
            e = np.zeros(n, np.uint8)
            e[np.cumsum([len(c) for c in case if len(c)]) - 1] = 1
            got = rtg_scan(torch.from_numpy(flat.astype(np.float32)[:, None]).cuda(), torch.from_numpy(e[:, None]).cuda(),
                           float(gammas[k])).cpu().numpy()[:, 0]
            np.testing.assert_array_equal(got, out[oo:oo + n])
        oo += n
        case = []
        k += 1
    # empty input is a no-op, not an error
    rtg_scan(torch.zeros((0, 4), device="cuda"), torch.zeros((0, 4), dtype=torch.uint8, device="cuda"), 0.99)


def test_shard_invariance_and_determinism():
    """SURVEY 8e parity condition: env i's trajectory depends on (seed, global env id) only, so two
    shards of 128 (env_id_base 0 / 128) reproduce one handle of 256 exactly."""
    from navbot_ppo_amd.env import NavSim
    rng = np.random.default_rng(19)
    acts = _actions(rng, 60, 256)

    def run(N, base, a):
        s = NavSim(N, max_episode_steps=25, auto_reset=True, seed=77, env_id_base=base)
        s.set_map(maps.stage_1())
        io = s.alloc_io()
        s.reset(io.obs)
        obs, rew = [], []
        for k in range(a.shape[0]):
            s.step(torch.from_numpy(a[k]).to(s.device), io.obs, io.reward, io.done, io.arrive, io.ended)
            obs.append(io.obs.cpu().numpy().copy())
            rew.append(io.reward.cpu().numpy().copy())
        return np.stack(obs), np.stack(rew)

    o_all, r_all = run(256, 0, acts)
    o_a, r_a = run(128, 0, acts[:, :128])
    o_b, r_b = run(128, 128, acts[:, 128:])
    np.testing.assert_array_equal(o_all, np.concatenate([o_a, o_b], 1))
    np.testing.assert_array_equal(r_all, np.concatenate([r_a, r_b], 1))
    o2, r2 = run(256, 0, acts)
    np.testing.assert_array_equal(o_all, o2)


def test_f16_observations():
    rng = np.random.default_rng(20)
    from navbot_ppo_amd.env import NavSim
    N = 128
    g16 = NavSim(N, max_episode_steps=20, auto_reset=True, seed=4, obs_f16=True)
    g32 = NavSim(N, max_episode_steps=20, auto_reset=True, seed=4)
    for s in (g16, g32):
        s.set_map(maps.stage_1())
    i16, i32 = g16.alloc_io(), g32.alloc_io()
    assert i16.obs.dtype == torch.float16
    g16.reset(i16.obs)
    g32.reset(i32.obs)
    acts = _actions(rng, 30, N)
    for k in range(30):
        a = torch.from_numpy(acts[k]).cuda()
        g16.step(a, i16.obs, i16.reward, i16.done, i16.arrive, i16.ended)
        g32.step(a, i32.obs, i32.reward, i32.done, i32.arrive, i32.ended)
        assert torch.equal(i16.obs, i32.obs.half())  # same values, rounded once to half
        assert torch.equal(i16.reward, i32.reward) and torch.equal(i16.ended, i32.ended)


def test_env_dropin_surface():
    """The N=1 Python surface of the reference Env (environment_new.py:27,272,312)."""
    from navbot_ppo_amd.env import Env
    env = Env(is_training=True, seed=3)
    assert env.threshold_arrive == 0.2 and env.use_vision is False
    obs = env.reset()
    assert isinstance(obs, np.ndarray) and obs.shape == (16,) and obs.dtype == np.float64
    assert env.position.x == 0.0 and env.position.y == 0.0
    gx, gy = env.goal_position.position.x, env.goal_position.position.y
    assert abs(env.past_distance - math.hypot(gx, gy)) < 1e-12
    o2, rew, done, arrive = env.step(np.array([1.0, 0.0]), np.array([0.25, -0.5]))
    assert o2.shape == (16,) and isinstance(rew, float) and isinstance(done, bool) and isinstance(arrive, bool)
    assert o2[10] == 0.25 and o2[11] == -0.5                 # past_action argument honoured
    assert abs(env.position.x - 0.05) < 1e-12                 # 0.25 m/s for 0.2 s
    assert abs(rew - 500 * (math.hypot(gx, gy) - math.hypot(gx - 0.05, gy))) < 1e-3
    with pytest.raises(IndexError):
        env.step(np.array([1.0]), np.array([0.0, 0.0]))
    assert Env(is_training=False).threshold_arrive == 0.4


def test_errors_are_reported_not_swallowed():
    from navbot_ppo_amd.env import NavSim, NavsimError
    s = NavSim(8)
    io = s.alloc_io()
    with pytest.raises(NavsimError, match="set_map"):
        s.reset(io.obs)
    with pytest.raises(NavsimError):
        NavSim(8, n_beams=7)
    with pytest.raises(NavsimError):
        s.set_map(np.zeros((3, 5), dtype=np.float32))


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg4", "cfg5", "cfg4_full", "cfg5_full"])
def test_full_size_properties(cfg):
    """BASELINE sizes -- configs[1] 4096 shared stage_1 ; configs[2] 16384 per-env stage_2 ; configs[3] per GPU: 4096 envs,
    stage_4, 36 beams ; configs[4] per GPU: 8192 envs, 2048-segment house map, f16 observations, start/goal tables ; and
    configs[3] / configs[4] at their FULL size on one GPU (32768 x 36 beams: the 32-env / 36-beam instantiation pick_epb selects
    beyond 16384 envs; 65536 x house x f16 x tables: the 64-env shape with tile boxes) --
    through size-independent properties:
    * a sample of 96 envs replayed on the oracle matches (flags exact, obs 1e-6; f16: the oracle's row rounded to half)
    * observation ranges of SURVEY A3#10
    * episode accounting: sum(ep_length at ended) + live ep_step == steps * N
    """
    from navbot_ppo_amd.env import NavSim
    rng = np.random.default_rng(21)
    B, f16, sampler, cap = 10, False, None, 50
    if cfg == "cfg2":
        N, K, seg, per_env = 4096, 64, maps.stage_1(), False
    elif cfg == "cfg3":
        N, K, per_env = 16384, 24, True
        seg = maps.replicate_per_env(maps.stage_2(), N, seed=0)
    elif cfg in ("cfg4", "cfg4_full"):
        N, K, seg, per_env, B = (4096, 60, maps.stage_4(), False, 36) if cfg == "cfg4" else (32768, 56, maps.stage_4(), False, 36)
    else:
        N, K, seg, per_env, f16, cap = (8192 if cfg == "cfg5" else 65536), 30, maps.house(2048), False, True, 20
        st, g, lo, hi = maps.spawn_tables("small_house")
        sampler = maps.open_tables(seg, st, g) + (lo, hi)
    s = NavSim(N, n_beams=B, max_episode_steps=cap, auto_reset=True, seed=1, obs_f16=f16)
    s.set_map(seg, per_env=per_env)
    if sampler:
        s.set_spawn_sampler(*sampler)
    inf = s.info()   # the instantiation the rule picks at this size is the one under test
    want_shape = {"cfg2": (16, 8, 0), "cfg3": (64, 16, 2), "cfg4": (16, 8, 0), "cfg5": (16, 8, 3), "cfg4_full": (32, 8, 0),
                  "cfg5_full": (64, 16, 3)}[cfg]
    assert (inf["step_epb"], inf["step_waves"], inf["step_cast"]) == want_shape, inf
    io = s.alloc_io()
    s.reset(io.obs)
    sample = np.sort(rng.choice(N, 96, replace=False))
    cpu = [O.OracleSim(1, n_beams=B, max_episode_steps=cap, auto_reset=True, seed=1, env_id_base=int(i)) for i in sample]
    for c, i in zip(cpu, sample):
        c.set_map(seg[i] if per_env else seg)
        if sampler:
            c.set_spawn_sampler(*sampler)
        c.reset()
    total_len = 0
    n_end = 0
    for k in range(K):
        a = np.stack([rng.uniform(0, 1, N), rng.uniform(-1, 1, N)], 1).astype(np.float32)
        a[: N // 4, 1] *= 0.05   # a quarter drive nearly straight: collisions / arrivals, not only timeouts
        s.step(torch.from_numpy(a).cuda(), io.obs, io.reward, io.done, io.arrive, io.ended, io.ep_return, io.ep_length)
        o = io.obs.float().cpu().numpy()
        e = io.ended.cpu().numpy().astype(bool)
        total_len += int(io.ep_length.cpu().numpy()[e].sum())
        n_end += int(e.sum())
        tol = 5e-4 if f16 else 1e-7
        assert np.all((o[:, :B] >= 0.12 / 3.5 - tol) & (o[:, :B] <= 1.0))
        assert np.all((o[:, B + 3] >= 0) & (o[:, B + 3] <= 1) & (o[:, B + 4] >= 0) & (o[:, B + 4] <= 1) & (np.abs(o[:, B + 5]) <= 1))
        if not f16:
            assert np.all((o[:, B + 3] < 1) & (o[:, B + 4] < 1))
        for c, i in zip(cpu, sample):
            out = c.step(a[i:i + 1])
            want = out["obs"][0].astype(np.float16).astype(np.float32) if f16 else out["obs"][0]
            np.testing.assert_allclose(o[i], want, rtol=0, atol=1e-3 if f16 else OBS_ATOL)
            assert io.done[i].item() == out["done"][0] and io.arrive[i].item() == out["arrive"][0]
            assert io.ended[i].item() == out["ended"][0]
    assert n_end > 64   # (cfg3 runs 24 steps of a 50-step cap: few ends; the others thousands)
    assert total_len + int(s.get_state()["ep_step"].sum()) == K * N


def test_bench_two_ranks_share_one_gpu_over_gloo():
    """The N>1 launch path of bench.py end to end (torch.distributed.run, env shards, flat-gradient all-reduce,
    max-over-ranks timing, one JSON line from rank 0).  RCCL needs one GPU per rank, so on this 1-GPU box the two
    ranks share cuda:0 and the collectives go over gloo; everything else is the production path."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NAVBOT_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(__import__("_ranks").free_port()), os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--envs-per-gpu", "512", "--rollout", "64", "--epochs", "3"]
    out = subprocess.run(cmd, env=env, cwd=repo, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    js = json.loads(lines[0])
    assert js["n_gpus"] == 2 and js["config"]["n_envs_total"] == 1024 and js["scaling"] == "weak"
    assert js["value"] > 0 and js["metric"] == "env_steps_per_sec"
    assert js["roofline"]["bound"] == "hbm" and js["roofline"]["frac"] > 0   # measured on rank 0 at every N
    assert "step_kernel" in js["roofline"]["kernel"]                          # the like-for-like key: navsim_step, one launch per step
    assert "cpu_baseline" not in js                                            # N == 1 only
    # what a first real multi-GPU run is read from: per-rank extremes of the split, the collective alone, and
    # the collective's cost inside an epoch
    d = js["dist"]
    assert d["ranks"] == 2 and d["backend"] == "gloo" and d["allreduce_bytes"] == 4 * 10691
    assert 0 < d["rollout_ms_min"] <= d["rollout_ms_max"] and 0 < d["update_ms_min"] <= d["update_ms_max"]
    assert d["allreduce_us"] > 0 and d["epoch_us_with_allreduce"] > 0 and d["epoch_us_local"] > 0
    assert d["allreduce_cost_in_epoch_us"] == pytest.approx(d["epoch_us_with_allreduce"] - d["epoch_us_local"], abs=0.02)


def test_bench_refuses_a_multi_rank_run_that_is_not_over_rccl():
    """bench.py --gpus 2 on a box with ONE GPU and no backend override: two ranks cannot both own a device over RCCL, and the run
    must fail loudly instead of printing a scaling number (round-3 review, item 6)."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one visible GPU")
    env = {k: v for k, v in os.environ.items() if k != "NAVBOT_DIST_BACKEND"}
    from _ranks import free_port
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--envs-per-gpu", "256", "--rollout", "16", "--epochs", "1", "--no-extras"]
    out = subprocess.run(cmd, env=env, cwd=repo, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    # ... and for THIS reason: bench.py's own refusal, or RCCL rejecting two ranks on one device before bench.py gets that far --
    # not a busy port or any other start-up failure that would make the test pass for the wrong reason
    err = out.stderr + out.stdout
    assert ("refusing to report a scaling number" in err or "one rank per GPU" in err or "NCCL" in err or "nccl" in err), err[-2000:]
    assert "EADDRINUSE" not in err and "address already in use" not in err.lower(), err[-2000:]


def test_degenerate_small_segments_take_the_ieee_divide_path():
    """Segments shorter than 2^-10 m (and zero-length ones) are outside the range of the kernel's unscaled exact divide;
    tiles that hold one must fall back to the plain IEEE divide and still agree with the oracle bit for bit on flags.
    Covers the shared-map tile flag and the per-env wave-uniform branch."""
    rng = np.random.default_rng(77)
    base = maps.stage_1()
    extra = np.array([[0.50, 0.00, 0.50 + 3e-4, 2e-4],      # tiny, right on the spawn's forward beam
                      [-0.30, 0.20, -0.30, 0.20],           # zero length
                      [0.80, -0.40, 0.80 + 1e-5, -0.40],    # tiny, axis aligned
                      [1.10, 0.60, 1.10, 0.60 + 9e-4]],     # just under the 2^-10 threshold
                     dtype=np.float32)
    seg = np.concatenate([base, extra]).astype(np.float32)
    gpu, cpu = _mk(256, seg, max_episode_steps=40, auto_reset=True, seed=5)
    st = _lockstep(gpu, cpu, _actions(rng, 60, 256))
    assert st["ended"] > 256
    segp = np.repeat(seg[None], 64, axis=0).copy()
    segp[:, :, [0, 2]] += rng.uniform(-0.05, 0.05, (64, 1, 1)).astype(np.float32)   # per-env jitter, tiny segments stay tiny
    gpu, cpu = _mk(64, segp, per_env=True, max_episode_steps=40, auto_reset=True, seed=6)
    _lockstep(gpu, cpu, _actions(rng, 60, 64))


@pytest.mark.parametrize("per_env", [False, True])
def test_soak_all_episode_ends_with_respawn_and_autoreset(per_env):
    """Long lockstep run with every way an episode can end, auto_reset AND respawn_on_arrive together: an arrival draws
    the re-spawn goal and then the next episode's goal from the same Philox stream (the step kernel prepares both
    possible next-episode records ahead of the rules), with the reference's rejection rectangles active so the draw
    loops run more than once.  Thousands of resets; counters, goals and poses must stay identical to the oracle."""
    rng = np.random.default_rng(91 + per_env)
    N, K = (768, 700) if not per_env else (256, 500)
    seg = maps.stage_1()
    if per_env:
        seg = maps.replicate_per_env(seg, N, seed=8)
    gpu, cpu = _mk(N, seg, per_env=per_env, max_episode_steps=70, auto_reset=True, respawn_on_arrive=True, seed=17,
                   goal_box=(-1.5, 1.5), threshold_arrive=0.4)
    rects = np.array([[0.3, 0.9, -0.6, 0.6], [-0.9, -0.3, -0.6, 0.6]])   # rejects ~27 % of the draws
    for s_ in (gpu, cpu):
        s_.set_goal_rects(0, rects)
        s_.set_goal_rects(1, rects * 1.1)
    a = _actions(rng, K, N)
    a[:, N // 2:, 0] = np.maximum(a[:, N // 2:, 0], 0.6)   # half of the envs keep moving: arrivals and collisions
    st = _lockstep(gpu, cpu, a, check_state_every=50)
    assert st["arrive"] > 200 and st["done"] > 10 and st["ended"] > 4 * N


def test_table_sampler_parity_shared_and_per_env():
    """navsim_set_spawn_sampler (GoalSpawnSampler tables) vs the oracle: resets and in-step auto-resets pick the same start
    poses / goals (same Philox stream) and observe the scan of the picked pose."""
    rng = np.random.default_rng(31)
    st, g, lo, hi = maps.spawn_tables("stage1")
    gpu, cpu = _mk(384, maps.stage_1(), max_episode_steps=12, auto_reset=True, seed=21)
    gpu.set_spawn_sampler(st, g, lo, hi)
    cpu.set_spawn_sampler(st, g, lo, hi)
    stt = _lockstep(gpu, cpu, _actions(rng, 40, 384), check_state_every=5)
    assert stt["ended"] >= 384 * 3
    segp = maps.replicate_per_env(maps.stage_1(), 128, seed=4)
    gpu, cpu = _mk(128, segp, per_env=True, max_episode_steps=9, auto_reset=True, seed=22)
    gpu.set_spawn_sampler(st, g, lo, hi)
    cpu.set_spawn_sampler(st, g, lo, hi)
    _lockstep(gpu, cpu, _actions(rng, 30, 128), check_state_every=5)
    # sampler installed before the map is also fine
    from navbot_ppo_amd.env import NavSim
    s = NavSim(8, seed=1)
    s.set_spawn_sampler(st, g, lo, hi)
    s.set_map(maps.stage_1())
    io = s.alloc_io()
    s.reset(io.obs)
    assert np.all(np.isin(s.get_state()["pose"][:, 0], st[:, 0]))


def test_cfg5_house_map_f16_sampler_parity():
    """BASELINE configs[4] in small: the 2048-segment house map (shared), curated start/goal tables filtered to open space,
    half-precision observation buffers; GPU vs oracle (obs compared after the oracle's rows are rounded to half)."""
    rng = np.random.default_rng(41)
    N = 192
    seg = maps.house()
    st, g, lo, hi = maps.spawn_tables("small_house")
    st, g = maps.open_tables(seg, st, g)
    from navbot_ppo_amd.env import NavSim
    gpu = NavSim(N, max_episode_steps=25, auto_reset=True, seed=8, obs_f16=True)
    cpu = O.OracleSim(N, max_episode_steps=25, auto_reset=True, seed=8)
    for s in (gpu, cpu):
        s.set_map(seg)
        s.set_spawn_sampler(st, g, lo, hi)
    io = gpu.alloc_io()
    og = gpu.reset(io.obs)
    oc = cpu.reset()
    assert og.dtype == torch.float16
    assert torch.equal(og.cpu(), torch.from_numpy(oc).half())
    acts = _actions(rng, 60, N)
    n_end = 0
    for k in range(60):
        gpu.step(torch.from_numpy(acts[k]).cuda(), io.obs, io.reward, io.done, io.arrive, io.ended, io.ep_return, io.ep_length)
        out = cpu.step(acts[k])
        diff = (io.obs.float().cpu() - torch.from_numpy(out["obs"]).half().float()).abs().max().item()
        assert diff <= 1e-3  # one half ulp at 1.0 is 4.9e-4; rows are equal except where f32 differs in its last bit
        for name in ("done", "arrive", "ended"):
            np.testing.assert_array_equal(getattr(io, name).cpu().numpy(), out[name])
        np.testing.assert_allclose(io.reward.cpu().numpy(), out["reward"], rtol=REW_RTOL, atol=1e-5)
        n_end += int(out["ended"].sum())
    assert n_end > N
    sg, sc = gpu.get_state(), cpu.get_state()
    np.testing.assert_array_equal(sg["goal"], sc["goal"])
    np.testing.assert_allclose(sg["pose"], sc["pose"], atol=1e-11)


def test_cfg1_single_env_scripted_tape():
    """BASELINE configs[0] / SURVEY 8d cfg 1: ONE env on stage_1, a 1000-step scripted action tape from
    np.random.default_rng(0), goals drawn from default_rng(1) through the reset-rejection rule and injected; the GPU
    path (N=1 is a single lane of a single workgroup) against the CPU oracle step by step, flags bit-exact."""
    from navbot_ppo_amd.env import NavSim
    tape = np.random.default_rng(0)
    goals = np.random.default_rng(1)

    def next_goal():
        while True:
            g = goals.uniform(-3.6, 3.6, 2)
            if not O.goal_rejected(0, g[0], g[1]):
                return g

    gpu = NavSim(1, max_episode_steps=500, auto_reset=False, seed=0)
    cpu = O.OracleSim(1, max_episode_steps=500, auto_reset=False, seed=0)
    for s in (gpu, cpu):
        s.set_map(maps.stage_1())
    io = gpu.alloc_io()

    def reset_both():
        gpu.reset(io.obs)
        cpu.reset()
        g = next_goal()[None, :]
        pd = np.hypot(g[:, 0], g[:, 1])
        gpu.set_state(goal=g, past_dist=pd)
        cpu.set_state(goal=g, past_dist=pd)

    reset_both()
    n_eps = n_exact = 0
    for k in range(1000):
        a = np.array([[tape.uniform(0, 1), tape.uniform(-1, 1)]], dtype=np.float32)
        gpu.step(torch.from_numpy(a).cuda(), io.obs, io.reward, io.done, io.arrive, io.ended)
        out = cpu.step(a)
        og = io.obs.cpu().numpy()
        np.testing.assert_allclose(og, out["obs"], rtol=0, atol=OBS_ATOL)
        assert abs(io.reward.item() - out["reward"][0]) <= 1e-5 * max(1.0, abs(out["reward"][0]))
        assert (io.done.item(), io.arrive.item(), io.ended.item()) == (out["done"][0], out["arrive"][0], out["ended"][0])
        n_exact += int(np.array_equal(og, out["obs"]))
        if out["ended"][0]:
            n_eps += 1
            reset_both()
    assert n_eps >= 3 and n_exact >= 990


def test_sensor_options_parity_noise_and_gazebo_below_min():
    """lidar_noise_sigma / lidar_below_min on the GPU vs the oracle.  The noise goes through float32 logf/sinf/cosf whose
    device and glibc versions may differ in the last ulp, so observations are compared at 1e-6; flags stay exact."""
    rng = np.random.default_rng(51)
    gpu, cpu = _mk(256, maps.stage_1(), max_episode_steps=30, auto_reset=True, seed=31, lidar_noise_sigma=0.01)
    st = _lockstep(gpu, cpu, _actions(rng, 80, 256))
    assert st["ended"] > 256
    segp = maps.replicate_per_env(maps.stage_2(), 64, seed=6)
    gpu, cpu = _mk(64, segp, per_env=True, max_episode_steps=20, auto_reset=True, seed=32, lidar_noise_sigma=0.02,
                   lidar_below_min="gazebo")
    _lockstep(gpu, cpu, _actions(rng, 50, 64))
    # -inf readings: drive envs nose-first into the inner box; both sides must report -inf lidar and done == 0
    gpu, cpu = _mk(8, maps.stage_1(), lidar_below_min="gazebo", seed=3)
    io = gpu.alloc_io()
    gpu.reset(io.obs)
    cpu.reset()
    pose = np.tile(np.array([[1.9 - 0.06 + 0.032, 0.0, 0.0]]), (8, 1))
    pose[:, 1] = np.linspace(-0.5, 0.5, 8)
    for s in (gpu, cpu):
        s.set_state(pose=pose, goal=np.full((8, 2), 3.0), past_dist=np.full(8, 3.0))
    a = np.zeros((8, 2), np.float32)
    gpu.step(torch.from_numpy(a).cuda(), io.obs, io.reward, io.done, io.arrive, io.ended)
    out = cpu.step(a)
    og = io.obs.cpu().numpy()
    assert np.isneginf(og[:, :10]).any() and not io.done.any()
    np.testing.assert_array_equal(np.isneginf(og), np.isneginf(out["obs"]))
    fin = np.isfinite(out["obs"])
    np.testing.assert_allclose(og[fin], out["obs"][fin], atol=OBS_ATOL)
    np.testing.assert_array_equal(io.done.cpu().numpy(), out["done"])


def test_cull_is_conservative_on_segment_soups():
    """The cast only tests segments that survive a range / behind-the-fan cull; whatever the geometry, the scan must keep
    the oracle's bits.  Random soups: long, short and degenerate segments, some through or next to the robots' paths (the
    sim is kinematic: nothing stops a robot from crossing a segment the beams miss)."""
    rng = np.random.default_rng(77)
    for S, B in ((48, 10), (200, 10), (90, 36), (7, 10)):
        N = 160
        c = rng.uniform(-3.0, 3.0, (N, S, 2))
        ln = rng.choice([1e-4, 0.02, 0.4, 3.0, 9.0], (N, S, 1))
        d = rng.normal(size=(N, S, 2))
        seg = np.concatenate([c - d * ln / 2, c + d * ln / 2], 2).astype(np.float32)
        seg[:, 0] = [0.05, -1.0, 0.05, 1.0]          # a wall 5 cm ahead of the spawn pose: robots drive through it
        seg[:, 1 % S] = [-0.032, 0.0, 2.0, 0.0]      # starts exactly at the sensor origin of the spawn pose
        seg[:, 2 % S] = [1.0, 1.0, 1.0, 1.0]         # zero length
        gpu, cpu = _mk(N, seg, per_env=True, B=B, max_episode_steps=25, auto_reset=True, seed=3)
        for s in (gpu, cpu):
            s.set_goal_rects(0, np.zeros((0, 4)))
            s.set_goal_rects(1, np.zeros((0, 4)))
        a = _actions(rng, 60, N)
        a[..., 0] = np.maximum(a[..., 0], 0.6)
        st = _lockstep(gpu, cpu, a)
        assert st["exact_obs"] > 0.98 * st["total"]
        # shared-map form of the same soup (env 0's)
        gpu, cpu = _mk(N, seg[0], per_env=False, B=B, max_episode_steps=25, auto_reset=True, seed=4)
        _lockstep(gpu, cpu, a[:30])


def test_g10_reference_rollout_through_the_kernel():
    """a-8: the reference's PPO.rollout + compute_rtgs over its own Env (G10) replayed on NavSim(1, auto_reset,
    respawn_on_arrive) + navsim_rtg_scan: store-obs-before-act order, past_action rule, termination, reset obs, trailing
    partial episode, batch_lens / t_so_far accounting, returns (ppo.py:463-671)."""
    from navbot_ppo_amd.env import NavSim, rtg_scan
    from test_rollout_golden_cpu import check_against_g10, replay_on
    d = np.load(os.path.join(G, "g10_rollout.npz"))
    sim = NavSim(1, max_episode_steps=int(d["cap"]), auto_reset=True, respawn_on_arrive=True, seed=int(d["seed"]))
    sim.set_map(maps.stage_1())
    io = sim.alloc_io()

    class Adapter:
        def reset(self):
            return sim.reset(io.obs).cpu().numpy()

        def step(self, a):
            sim.step(torch.from_numpy(a).cuda(), io.obs, io.reward, io.done, io.arrive, io.ended, io.ep_return, io.ep_length,
                     ep_path=io.ep_path)
            return {k: getattr(io, k).cpu().numpy() for k in ("obs", "reward", "done", "arrive", "ended", "ep_return", "ep_length", "ep_path")}

    obs, rew, ended, flags, eplen, epret, eppath = replay_on(Adapter(), d)
    rtg = rtg_scan(torch.from_numpy(rew[:, None].copy()).cuda(), torch.from_numpy(ended[:, None].copy()).cuda(),
                   float(d["gamma"])).cpu().numpy()[:, 0]
    check_against_g10(d, obs, rew, ended, flags, eplen, epret, eppath, rtg)
    assert int(sim.get_state()["rng_ctr"][0]) == int(d["rng_ctr_final"])


@pytest.mark.parametrize("epb", [8, 16, 32])
def test_workgroup_shapes_keep_parity_36_beams(epb):
    """The 36-beam instantiations of every workgroup shape (8 / 16 / 32 envs: the rule picks them by shard size -- 32 only beyond
    16384 envs, configs[3] at full size) forced onto a small ragged shard and run in lock step with the oracle, one launch per
    step and as a tape (navsim_step_seq rows == the per-step rows bit for bit): shared stage_4 map, per-env maps, the
    2048-segment house map with tile boxes and start / goal tables, sensor options."""
    from navbot_ppo_amd.env import NavSim
    rng = np.random.default_rng(150 + epb)
    N = 200   # ragged against every shape
    cases = [(maps.stage_4(), False, {}), (maps.replicate_per_env(maps.stage_2(), N, seed=5), True, {}),
             (maps.house(1000), False, {}), (maps.stage_4(), False, dict(lidar_noise_sigma=0.01, lidar_below_min="gazebo"))]
    for k, (seg, per_env, kw) in enumerate(cases):
        gpu, cpu = _mk(N, seg, per_env=per_env, B=36, max_episode_steps=25, auto_reset=True, seed=6 + k, **kw)
        gpu.set_shape(epb)
        samp = None
        if k == 2:
            tb = maps.spawn_tables("small_house")
            samp = maps.open_tables(maps.house(1000), tb[0], tb[1]) + tuple(tb[2:])
            for s in (gpu, cpu):
                s.set_spawn_sampler(*samp)
        inf = gpu.info()
        assert inf["step_epb"] == epb and inf["seq_epb"] == epb and inf["n_beams"] == 36, inf
        acts = _actions(rng, 50, N)
        stats = _lockstep(gpu, cpu, acts)
        assert stats["ended"] > N // 2
        # the same tape in one launch on a second handle of the same shape: rows bit-identical to the per-step launches
        a, b = (NavSim(N, n_beams=36, max_episode_steps=25, auto_reset=True, seed=6 + k, envs_per_workgroup=epb, **kw) for _ in range(2))
        for s in (a, b):
            s.set_map(seg, per_env=per_env)
            if samp:
                s.set_spawn_sampler(*samp)
        T = 30
        at = torch.from_numpy(acts[:T]).cuda()
        ioa, iob = a.alloc_io(), b.alloc_io()
        a.reset(ioa.obs)
        b.reset(iob.obs)
        out = dict(obs=torch.empty((T, N, 42), device="cuda"), reward=torch.empty((T, N), device="cuda"),
                   done=torch.empty((T, N), dtype=torch.uint8, device="cuda"), arrive=torch.empty((T, N), dtype=torch.uint8, device="cuda"),
                   ended=torch.empty((T, N), dtype=torch.uint8, device="cuda"))
        a.step_seq(at, out["obs"], out["reward"], out["done"], out["arrive"], out["ended"])
        for t in range(T):
            b.step(at[t], iob.obs, iob.reward, iob.done, iob.arrive, iob.ended)
            assert torch.equal(out["obs"][t].view(torch.int32), iob.obs.view(torch.int32)), (k, t)
            assert torch.equal(out["reward"][t].view(torch.int32), iob.reward.view(torch.int32)) and torch.equal(out["ended"][t], iob.ended)


@pytest.mark.parametrize("epb", [8, 32, 64])
def test_workgroup_shapes_keep_parity(epb, monkeypatch):
    """The step kernel picks 16 / 32 / 64 envs per workgroup (4 / 8 / 16 waves) by shard size; NAVSIM_EPB forces one.  Every
    shape must reproduce the oracle: auto-reset with the cached next-episode records, per-env maps, and the
    respawn-on-arrive mode (two records per env: with 64 envs per workgroup they live on two spec waves)."""
    monkeypatch.setenv("NAVSIM_EPB", str(epb))
    rng = np.random.default_rng(50 + epb)
    N = 200   # ragged against every shape
    gpu, cpu = _mk(N, maps.stage_1(), max_episode_steps=30, auto_reset=True, seed=4)
    st = _lockstep(gpu, cpu, _actions(rng, 80, N))
    assert st["ended"] > N
    seg = maps.replicate_per_env(maps.stage_2(), N, seed=5)
    gpu, cpu = _mk(N, seg, per_env=True, max_episode_steps=25, auto_reset=True, seed=6)
    _lockstep(gpu, cpu, _actions(rng, 60, N))
    # shared map big enough for the Morton-ordered copy with tile bounding boxes (whole tiles are skipped)
    gpu, cpu = _mk(N, maps.house(1000), max_episode_steps=25, auto_reset=True, seed=8)
    st, g, lo, hi = maps.spawn_tables("small_house")
    for s in (gpu, cpu):
        s.set_spawn_sampler(*(maps.open_tables(maps.house(1000), st, g) + (lo, hi)))
    _lockstep(gpu, cpu, _actions(rng, 50, N))
    for auto in (False, True):
        gpu, cpu = _mk(N, maps.stage_1(), max_episode_steps=0 if not auto else 40, auto_reset=auto, respawn_on_arrive=True,
                       seed=7, goal_box=(-0.6, 0.6))
        for s in (gpu, cpu):
            s.set_goal_rects(0, np.zeros((0, 4)))
            s.set_goal_rects(1, np.zeros((0, 4)))
        a = _actions(rng, 90, N)
        a[..., 0] = np.maximum(a[..., 0], 0.5)
        st = _lockstep(gpu, cpu, a)
        assert st["arrive"] > 20


def test_actions_outside_the_policy_range():
    """The simulator does not clamp: PPO clamps a0 to [0, 1] and a1 to [-1, 1] before Env.step (ppo.py:700-703), but the ABI
    accepts any action (reverse driving, fast spins: up to 0.6 m and 0.6 rad per step here).  Same results as the oracle."""
    rng = np.random.default_rng(91)
    N = 160
    gpu, cpu = _mk(N, maps.stage_1(), max_episode_steps=20, auto_reset=True, seed=12)
    a = rng.uniform(-3.0, 3.0, (70, N, 2)).astype(np.float32)
    st = _lockstep(gpu, cpu, a)
    assert st["ended"] > N
    seg = maps.replicate_per_env(maps.stage_2(), N, seed=2)
    gpu, cpu = _mk(N, seg, per_env=True, max_episode_steps=20, auto_reset=True, seed=13)
    _lockstep(gpu, cpu, a[:40])


def test_g10_reference_rollout_through_the_env_class():
    """a-8 / b: G10 replayed through the N = 1 drop-in class `Env`, driven EXACTLY as the reference's rollout drives its Env
    (ppo.py:486 reset at batch start; :535 position read for the path length; :541 step(action, past_action);
    :543 past_action <- action; :552-553 termination; :591-593 past_action <- [0, 0] and a caller-side reset()).
    `Env.step` never resets and re-spawns the goal on arrival like setReward (environment_new.py:245-267), so the goal
    stream must advance exactly as the reference's did (rng_ctr_final)."""
    from navbot_ppo_amd.env import Env, rtg_scan
    from test_rollout_golden_cpu import check_against_g10, replay_on
    d = np.load(os.path.join(G, "g10_rollout.npz"))
    cap = int(d["cap"])
    env = Env(is_training=True, seed=int(d["seed"]))

    class Caller:   # the caller side of ppo.py:486-594, returning what replay_on() stores
        def reset(self):
            self.past_action, self.one_round, self.ep_ret, self.path, self.prev = [0, 0], 0, 0.0, 0.0, None
            return env.reset()[None, :]

        def step(self, a):
            cur = np.array([env.position.x, env.position.y])               # ppo.py:535
            if self.prev is not None:
                self.path += float(np.linalg.norm(cur - self.prev))
            self.prev = cur
            obs, rew, done, arrive = env.step(a[0], self.past_action)       # ppo.py:541
            assert obs.shape == (16,) and obs.dtype == np.float64 and isinstance(rew, float)
            assert isinstance(done, bool) and isinstance(arrive, bool)
            self.past_action = a[0]                                         # ppo.py:543
            self.ep_ret += rew
            self.one_round += 1
            ended = done or arrive or self.one_round >= cap                 # ppo.py:552-553
            out = dict(reward=np.float32([rew]), done=np.uint8([done]), arrive=np.uint8([arrive]), ended=np.uint8([ended]),
                       ep_return=np.float32([self.ep_ret if ended else 0]), ep_length=np.int32([self.one_round if ended else 0]),
                       ep_path=np.float32([self.path if ended else 0]))
            if ended:
                obs = self.reset()[0]                                       # ppo.py:591-593
            out["obs"] = obs[None, :]
            return out

    obs, rew, ended, flags, eplen, epret, eppath = replay_on(Caller(), d)
    rtg = rtg_scan(torch.from_numpy(rew[:, None].copy()).cuda(), torch.from_numpy(ended[:, None].copy()).cuda(),
                   float(d["gamma"])).cpu().numpy()[:, 0]
    check_against_g10(d, obs, rew, ended, flags, eplen, epret, eppath, rtg)
    assert int(env._sim.get_state()["rng_ctr"][0]) == int(d["rng_ctr_final"])
    env.close()


def test_bench_strong_scaling_two_ranks_over_gloo():
    """`bench.py --scaling strong`: the env total stays fixed (SURVEY.md 8d(i) read literally: 4096 envs on 1 / 2 / 4 / 8 GPUs),
    every rank takes envs_total / N, and the line says so.  Two ranks on one GPU over gloo, as above; --no-extras skips the
    roofline legs."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NAVBOT_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29633", os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--scaling", "strong", "--envs-total", "1024", "--rollout", "64", "--epochs", "3", "--no-extras"]
    out = subprocess.run(cmd, env=env, cwd=repo, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    js = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert js["n_gpus"] == 2 and js["scaling"] == "strong"
    assert js["config"]["n_envs_total"] == 1024 and js["config"]["n_envs_per_gpu"] == 512
    assert js["value"] == pytest.approx(64 * 1024 / (js["ms_per_step"] * 1e-3), rel=1e-3)


def _gae_ref(rew, ended, V, gamma, lam, last=None):
    """float64 statement of GAE with the reference's terminal rule (episode ends and, without `last`, the batch end)."""
    T, N = rew.shape
    R = np.zeros((T, N))
    nxt_R = np.zeros(N) if last is None else last.astype(np.float64)
    nxt_V = np.zeros(N) if last is None else last.astype(np.float64)
    for t in range(T - 1, -1, -1):
        e = ended[t].astype(bool)
        R[t] = rew[t].astype(np.float64) + np.where(e, 0.0, gamma * (1 - lam) * nxt_V + gamma * lam * nxt_R)
        nxt_R, nxt_V = R[t], V[t].astype(np.float64)
    return R


@pytest.mark.parametrize("T,N", [(512, 4096), (700, 64), (33, 48), (90, 37)])
def test_gae_scan(T, N, monkeypatch):
    """navsim_gae_scan (extension; north_star "GAE / return scan"): lambda = 1 without bootstrap IS the reference's estimator --
    returns bit-identical to navsim_rtg_scan (ppo.py:643-671), advantages bit-identical to rtgs - V (ppo.py:277); lambda < 1
    and a bootstrapped batch end match a float64 oracle."""
    from navbot_ppo_amd.env import gae_scan, rtg_scan
    rng = np.random.default_rng(T * 1000 + N)
    rew = (rng.standard_normal((T, N)) * 20).astype(np.float32)
    ended = (rng.random((T, N)) < 0.02).astype(np.uint8)
    V = (rng.standard_normal((T, N)) * 50).astype(np.float32)
    last = (rng.standard_normal(N) * 50).astype(np.float32)
    cu = lambda a: torch.from_numpy(a).cuda()
    for exact in (False, True):
        rtg = rtg_scan(cu(rew), cu(ended), 0.99, exact=exact)
        adv, ret = gae_scan(cu(rew), cu(ended), cu(V), 0.99, 1.0, exact=exact)
        assert torch.equal(ret, rtg)
        assert torch.equal(adv, rtg - cu(V))
        for lam, lv in ((0.95, None), (0.9, last), (0.0, None), (1.0, last)):
            adv, ret = gae_scan(cu(rew), cu(ended), cu(V), 0.99, lam, last_value=None if lv is None else cu(lv), exact=exact)
            want = _gae_ref(rew, ended, V, 0.99, lam, lv)
            np.testing.assert_allclose(ret.cpu().numpy(), want, rtol=2e-6, atol=2e-5)
            np.testing.assert_allclose(adv.cpu().numpy(), want.astype(np.float32) - V, rtol=0, atol=3e-4)
            adv2, none = gae_scan(cu(rew), cu(ended), cu(V), 0.99, lam, last_value=None if lv is None else cu(lv), want_returns=False,
                                  exact=exact)
            assert none is None and torch.equal(adv2, adv)


def _seq_vs_steps(N, seg, per_env, T, B=10, sampler=None, rects=None, **kw):
    """navsim_step_seq against T navsim_step launches from the same start: every output row and the final state bit for bit."""
    from navbot_ppo_amd.env import NavSim
    rng = np.random.default_rng(5)
    acts = torch.from_numpy(_actions(rng, T, N)).cuda()
    outs = []
    for mode in ("steps", "seq"):
        s = NavSim(N, n_beams=B, seed=3, **kw)
        if rects:
            rr, rs = maps.goal_rects(rects)
            s.set_goal_rects(0, rr)
            s.set_goal_rects(1, rs)
        s.set_map(seg, per_env=per_env)
        if sampler:
            s.set_spawn_sampler(*sampler)
        io = s.alloc_io()
        s.reset(io.obs)
        dev = s.device
        buf = dict(obs=torch.zeros((T, N, s.D), dtype=s.obs_dtype, device=dev), reward=torch.zeros((T, N), device=dev),
                   done=torch.zeros((T, N), dtype=torch.uint8, device=dev), arrive=torch.zeros((T, N), dtype=torch.uint8, device=dev),
                   ended=torch.zeros((T, N), dtype=torch.uint8, device=dev), ep_return=torch.zeros((T, N), device=dev),
                   ep_length=torch.zeros((T, N), dtype=torch.int32, device=dev), ep_path=torch.zeros((T, N), device=dev))
        if mode == "steps":
            for t in range(T):
                s.step(acts[t], buf["obs"][t], buf["reward"][t], buf["done"][t], buf["arrive"][t], buf["ended"][t],
                       buf["ep_return"][t], buf["ep_length"][t], ep_path=buf["ep_path"][t])
        else:
            s.step_seq(acts, buf["obs"], buf["reward"], buf["done"], buf["arrive"], buf["ended"], buf["ep_return"],
                       buf["ep_length"], buf["ep_path"])
        torch.cuda.synchronize()
        outs.append(({k: v.cpu() for k, v in buf.items()}, s.get_state()))
        s.close()
    (a, sa), (b, sb) = outs
    for k in a:
        assert torch.equal(a[k].view(torch.uint8) if a[k].dtype == torch.float16 else a[k], b[k].view(torch.uint8) if b[k].dtype == torch.float16 else b[k]), k
    for k in sa:
        np.testing.assert_array_equal(sa[k], sb[k], err_msg=k)
    return a


@pytest.mark.parametrize("case", ["cfg2", "cfg3", "cfg4", "cfg5", "house_4096", "small", "ragged", "sens", "no_reset"])
def test_step_seq_equals_step_launches(case):
    """One launch for a whole action tape (navsim_step_seq: env state on chip between the steps) == one navsim_step launch per
    step, bit for bit, in every workgroup shape and cast variant: configs[1] (32-env shape, shared 32 segments), configs[2]
    (64-env shape, per-env 128 segments, two-segments-per-lane passes), configs[3]'s shard (36 beams), configs[4]'s shard (tile
    boxes, f16 observations, start / goal tables), the house map on a 4000-env shard (eight-wave 8-env shape), a 16-env-shape case, ragged N, the sensor options, and no auto-reset."""
    if case == "cfg2":
        a = _seq_vs_steps(4096, maps.stage_1(), False, 40, max_episode_steps=25, auto_reset=True, respawn_on_arrive=True)
    elif case == "cfg3":
        a = _seq_vs_steps(16384, maps.replicate_per_env(maps.stage_2(), 16384, seed=0), True, 30, rects="stage_2", max_episode_steps=20,
                          auto_reset=True)
    elif case == "cfg4":
        a = _seq_vs_steps(4096, maps.stage_4(), False, 30, B=36, max_episode_steps=20, auto_reset=True)
    elif case == "cfg5":
        seg = maps.house(2048)
        st, g, lo, hi = maps.spawn_tables("small_house")
        a = _seq_vs_steps(8192, seg, False, 24, sampler=maps.open_tables(seg, st, g) + (lo, hi), max_episode_steps=15, auto_reset=True,
                          obs_f16=True)
    elif case == "house_4096":   # tile boxes on the eight-wave 8-env shape, both forms, with the sensor options
        seg = maps.house(2048)
        st, g, lo, hi = maps.spawn_tables("small_house")
        a = _seq_vs_steps(4000, seg, False, 24, sampler=maps.open_tables(seg, st, g) + (lo, hi), max_episode_steps=15, auto_reset=True,
                          lidar_noise_sigma=0.01, lidar_below_min="gazebo")
    elif case == "small":
        a = _seq_vs_steps(96, maps.replicate_per_env(maps.stage_2(), 96, seed=1), True, 60, max_episode_steps=30, auto_reset=True,
                          respawn_on_arrive=True)
    elif case == "ragged":
        a = _seq_vs_steps(1000, maps.stage_1(), False, 33, max_episode_steps=16, auto_reset=True)
    elif case == "sens":
        a = _seq_vs_steps(512, maps.stage_1(), False, 40, max_episode_steps=25, auto_reset=True, lidar_noise_sigma=0.01,
                          lidar_below_min="gazebo")
    else:
        a = _seq_vs_steps(256, maps.stage_1(), False, 50, respawn_on_arrive=True)
    assert int(a["ended"].sum()) > 0 or case == "no_reset"


def _seq_vs_oracle(N, seg, per_env, T, blocks, rects=None, sampler=None, acts_seed=11, **kw):
    """navsim_step_seq on all N envs (the shard size picks the workgroup shape and the cast variant), then the envs of `blocks`
    (a list of (first env, count)) replayed on the CPU oracle keyed by the same global env ids: flags exact, observations 1e-6,
    rewards, episode statistics, final state."""
    from navbot_ppo_amd.env import NavSim
    gpu = NavSim(N, **kw)
    if rects:
        rr, rs = maps.goal_rects(rects)
        gpu.set_goal_rects(0, rr)
        gpu.set_goal_rects(1, rs)
    gpu.set_map(seg, per_env=per_env)
    if sampler:
        gpu.set_spawn_sampler(*sampler)
    acts = _actions(np.random.default_rng(acts_seed), T, N)
    io = gpu.alloc_io()
    obs0 = gpu.reset(io.obs).cpu().numpy()
    dev = gpu.device
    obs = torch.zeros((T, N, gpu.D), device=dev)
    rew, epr, epp = (torch.zeros((T, N), device=dev) for _ in range(3))
    done, arrive, ended = (torch.zeros((T, N), dtype=torch.uint8, device=dev) for _ in range(3))
    epl = torch.zeros((T, N), dtype=torch.int32, device=dev)
    gpu.step_seq(torch.from_numpy(acts).to(dev), obs, rew, done, arrive, ended, epr, epl, epp)
    torch.cuda.synchronize()
    sg = gpu.get_state()
    g = {k: v.cpu().numpy() for k, v in dict(obs=obs, rew=rew, done=done, arrive=arrive, ended=ended, epr=epr, epl=epl, epp=epp).items()}
    exact = total = n_end = 0
    for lo, n in blocks:
        sl = slice(lo, lo + n)
        cpu = O.OracleSim(n, **{k: v for k, v in kw.items() if k not in ("obs_f16", "env_id_base")},
                          env_id_base=kw.get("env_id_base", 0) + lo)
        if rects:
            cpu.set_goal_rects(0, rr)
            cpu.set_goal_rects(1, rs)
        cpu.set_map(np.ascontiguousarray(seg[sl]) if per_env else seg, per_env=per_env)
        if sampler:
            cpu.set_spawn_sampler(*sampler)
        np.testing.assert_allclose(obs0[sl], cpu.reset(), rtol=0, atol=OBS_ATOL)
        for t in range(T):
            out = cpu.step(acts[t, sl])
            for k in ("done", "arrive", "ended"):
                np.testing.assert_array_equal(g[k][t, sl], out[k], err_msg=f"{k}, step {t}, envs {lo}..")
            og = g["obs"][t, sl]
            np.testing.assert_allclose(og, out["obs"], rtol=0, atol=OBS_ATOL, err_msg=f"obs, step {t}, envs {lo}..")
            np.testing.assert_allclose(g["rew"][t, sl], out["reward"], rtol=REW_RTOL, atol=1e-5)
            e = out["ended"].astype(bool)
            np.testing.assert_array_equal(g["epl"][t, sl][e], out["ep_length"][e])
            np.testing.assert_allclose(g["epr"][t, sl][e], out["ep_return"][e], rtol=REW_RTOL, atol=1e-4)
            np.testing.assert_allclose(g["epp"][t, sl][e], out["ep_path"][e], rtol=1e-6, atol=1e-7)
            exact += int((og == out["obs"]).all(axis=1).sum())
            total += n
            n_end += int(e.sum())
        sc = cpu.get_state()
        np.testing.assert_allclose(sg["pose"][sl], sc["pose"], rtol=0, atol=1e-11)
        np.testing.assert_array_equal(sg["goal"][sl], sc["goal"])
        np.testing.assert_array_equal(sg["ep_step"][sl], sc["ep_step"])
        np.testing.assert_array_equal(sg["rng_ctr"][sl], sc["rng_ctr"])
    gpu.close()
    assert exact > 0.99 * total and n_end > 0
    return n_end


@pytest.mark.parametrize("case", ["small", "cfg3", "cfg5_house", "house_4096", "cfg4_36beams", "hbm_stream"])
def test_step_seq_against_the_oracle(case):
    """navsim_step_seq checked DIRECTLY against the CPU oracle (not only against the per-step launches), in the instantiations the
    BASELINE shards run: `small` 512 envs on per-env stage_2 maps with auto-reset and arrival re-spawn (16-env shape); `cfg3`
    configs[2]'s own workload -- 16384 envs, per-env stage_2 maps with their goal rectangles: the 64-env workgroup and the
    128-segments-per-pass cast -- with three blocks of envs (first, middle, last workgroups) replayed on the oracle; `cfg5_house` an
    8192-env shard on the shared 2048-segment house map with the start / goal tables (tile boxes); `cfg4_36beams` a 4096-env shard
    of configs[3] (stage_4, 36 beams: stage B on (segment, beam-group) entries); `hbm_stream` per-env maps large enough that one
    step's segment stream exceeds 1.25 x the Infinity Cache (navsim_set_map switches the persistent kernels' segment loads to
    non-temporal; the one-launch-per-step kernel streams per-env maps that way from 32 MiB on its 32 / 64-env 10-beam shapes -- navsim_get_info: step_cast == 2)."""
    if case == "small":
        seg = maps.replicate_per_env(maps.stage_2(), 512, seed=2)
        _seq_vs_oracle(512, seg, True, 48, [(0, 512)], max_episode_steps=20, auto_reset=True, respawn_on_arrive=True, seed=9)
    elif case == "cfg3":
        seg = maps.replicate_per_env(maps.stage_2(), 16384, seed=0)
        _seq_vs_oracle(16384, seg, True, 40, [(0, 96), (8000, 160), (16384 - 80, 80)], rects="stage_2", max_episode_steps=25,
                       auto_reset=True, seed=5)
    elif case == "cfg5_house":
        seg = maps.house(2048)
        st, g, lo, hi = maps.spawn_tables("small_house")
        _seq_vs_oracle(8192, seg, False, 30, [(0, 64), (4090, 100), (8192 - 40, 40)], sampler=maps.open_tables(seg, st, g) + (lo, hi),
                       max_episode_steps=18, auto_reset=True, seed=6)
    elif case == "house_4096":   # the same map on a 4096-env shard: 8-env workgroups on EIGHT waves (cfg5_house: on four), tile boxes
        seg = maps.house(2048)
        st, g, lo, hi = maps.spawn_tables("small_house")
        _seq_vs_oracle(4096, seg, False, 30, [(0, 72), (2044, 90), (4096 - 24, 24)], sampler=maps.open_tables(seg, st, g) + (lo, hi),
                       max_episode_steps=18, auto_reset=True, seed=12)
    elif case == "cfg4_36beams":
        _seq_vs_oracle(4096, maps.stage_4(), False, 40, [(0, 64), (2040, 80), (4096 - 32, 32)], rects="stage_4", n_beams=36,
                       max_episode_steps=25, auto_reset=True, seed=7)
    else:   # 16384 envs x 1536 segments x 16 B = 402 MB per step > 1.25 x the Infinity Cache: non-temporal stream, persistent form too
        seg = maps.replicate_per_env(maps.stage_2(sides=376), 16384, seed=3)
        assert seg.shape[1] == 1536
        _seq_vs_oracle(16384, seg, True, 8, [(0, 48), (16384 - 48, 48)], rects="stage_2", max_episode_steps=5, auto_reset=True, seed=8)


def test_g10_reference_rollout_in_one_launch():
    """a-8 through navsim_step_seq: the action tape the reference's PPO.rollout produced over its own Env (G10, 600 steps, episode
    cap 45) replayed in ONE launch; the stored observations (obs BEFORE acting, ppo.py:508: row 0 = the reset observation, row
    t = the observation after step t - 1), rewards, terminations, episode statistics and returns must match the recording."""
    from navbot_ppo_amd.env import NavSim, rtg_scan
    from test_rollout_golden_cpu import check_against_g10
    d = np.load(os.path.join(G, "g10_rollout.npz"))
    sim = NavSim(1, max_episode_steps=int(d["cap"]), auto_reset=True, respawn_on_arrive=True, seed=int(d["seed"]))
    sim.set_map(maps.stage_1())
    io = sim.alloc_io()
    acts = torch.from_numpy(np.ascontiguousarray(d["acts_tape"], dtype=np.float32)).cuda().view(-1, 1, 2)
    T = acts.shape[0]
    obs = torch.zeros((T + 1, 1, 16), device="cuda")
    sim.reset(obs[0])
    rew, epr, epp = (torch.zeros((T, 1), device="cuda") for _ in range(3))
    done, arrive, ended = (torch.zeros((T, 1), dtype=torch.uint8, device="cuda") for _ in range(3))
    epl = torch.zeros((T, 1), dtype=torch.int32, device="cuda")
    sim.step_seq(acts, obs[1:], rew, done, arrive, ended, epr, epl, epp)
    rtg = rtg_scan(rew, ended, float(d["gamma"])).cpu().numpy()[:, 0]
    flags = torch.cat([done, arrive], 1).cpu().numpy()
    check_against_g10(d, obs[:T, 0].cpu().numpy(), rew[:, 0].cpu().numpy(), ended[:, 0].cpu().numpy(), flags, epl[:, 0].cpu().numpy(),
                      epr[:, 0].cpu().numpy(), epp[:, 0].cpu().numpy(), rtg)
    assert int(sim.get_state()["rng_ctr"][0]) == int(d["rng_ctr_final"])


def test_vecenv_step_seq_surface():
    """VecEnv.step_seq (the host-side mirror of navsim_step_seq) returns what T VecEnv.step calls return."""
    from navbot_ppo_amd.env import VecEnv
    T, N = 12, 64
    acts = torch.from_numpy(_actions(np.random.default_rng(2), T, N)).cuda()
    a, b = VecEnv(N, map="stage_1", max_episode_steps=6, seed=4), VecEnv(N, map="stage_1", max_episode_steps=6, seed=4)
    a.reset(), b.reset()
    out = a.step_seq(acts)
    for t in range(T):
        obs, rew, done, arrive = b.step(acts[t])
        assert torch.equal(out.obs[t], obs) and torch.equal(out.reward[t], rew) and torch.equal(out.done[t], done)
        assert torch.equal(out.arrive[t], arrive) and torch.equal(out.ended[t], b.io.ended)
    assert int(out.ended.sum()) >= N


def test_integration_md_binding_runs_as_documented(monkeypatch):
    """INTEGRATION.md section 3 shows the binding a reference maintainer would add next to environment_new.py (main.py:433 imports
    `Env` from it; PPO.rollout drives reset() / step(action, past_action), ppo.py:486,541,593).  The python block is extracted from
    the document and executed AS WRITTEN against the built libnavsim.so -- raw ctypes, no argtypes, the ABI's argument order -- and
    60 steps (with the caller-side episode handling of ppo.py:552-593) must equal navbot_ppo_amd.env.Env bit for bit."""
    import sys
    import types
    from navbot_ppo_amd import _native
    from navbot_ppo_amd.env import Env
    md = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    start = md.index("# project_ppo/src/navsim_env.py")
    code = md[start:md.index("```", start)]
    assert "navsim_step(" in code and "navsim_reset(" in code and "class Env" in code
    stage1 = types.ModuleType("navsim_stage1")
    stage1.STAGE1_SEGMENTS = maps.stage_1().tolist()
    monkeypatch.setitem(sys.modules, "navsim_stage1", stage1)
    monkeypatch.setenv("NAVSIM_LIB", _native.LIB_PATH)
    ns = {"__name__": "navsim_env"}
    exec(compile(code, "INTEGRATION.md#navsim_env", "exec"), ns)
    doc, ours = ns["Env"](True), Env(True)
    rng = np.random.default_rng(3)
    o1, o2 = doc.reset(), ours.reset()
    assert o1.shape == (16,) and o1.dtype == np.float64
    np.testing.assert_array_equal(o1, o2)
    past = np.zeros(2)
    n_end = 0
    for k in range(60):
        a = np.array([0.7 + 0.3 * rng.uniform(), 0.2 * rng.uniform(-1, 1)])
        r1, r2 = doc.step(a, past), ours.step(a, past)
        np.testing.assert_array_equal(r1[0], r2[0])
        assert r1[1:] == r2[1:] and isinstance(r1[1], float) and isinstance(r1[2], bool) and isinstance(r1[3], bool)
        past = a
        if r1[2] or r1[3] or (k % 25 == 24):   # ppo.py:552-553,591-593
            np.testing.assert_array_equal(doc.reset(), ours.reset())
            past = np.zeros(2)
            n_end += 1
    assert n_end >= 2
    ours.close()
    ns["_lib"].navsim_destroy(doc.h)
