#!/usr/bin/env python3
"""Record golden vectors from the REFERENCE itself (run in the build container only).

    python3 -B tests/golden/generate_golden.py

Imports /root/reference/project_ppo/src/{environment_new,ppo,net_actor,net_critic}.py at
run time with stubbed ROS/gym/tensorboard modules (tests/golden/_ref_stubs.py), drives the
reference's own functions on seeded / crafted inputs and stores inputs + outputs as small
.npz / .json fixtures next to this script.  No reference source is copied: the fixtures are
data.  The GPU box never sees /root/reference; tests there read only the fixtures.

Sets:
  g1_odometry.npz    Env.getOdometry      (environment_new.py:138-181)
  g2_state.npz       Env.getState         (:183-207)
  g3_step.npz        Env.step, no arrival (:272-310) incl. setReward (:209-222)
  g3b_arrive.npz     Env.step arrival branch: reward, respawn goal, past_distance (:220-268)
  g4_reset.npz       Env.reset (:312-382) + goal-rejection rectangles of reset and respawn
  g5_rtgs.npz        PPO.compute_rtgs     (ppo.py:643-671)
  g6_action.npz      PPO.get_action clamp/log_prob + covariance decay (ppo.py:673-706)
  g7_update.npz      PPO.learn on a fixed batch: V, log-probs, losses, weights after k epochs (ppo.py:275-397)
  g8_nets.npz/.json  NetActor/NetCritic forward + state_dict keys (net_actor.py, net_critic.py)
  g9_closed_loop.npz K-step closed loop: oracle sim poses/scans -> reference Env.step outputs
  g10_rollout.npz    PPO.rollout (ppo.py:463-641) + compute_rtgs over the reference Env: batch_obs/acts/log_probs/rtgs/lens,
                     per-episode metrics (length, return, path_length), for a replayed action tape
"""
import json
import math
import os
import random
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

import _ref_stubs  # noqa: E402

rospy = _ref_stubs.install()
sys.path.insert(0, "/root/reference/project_ppo/src")
import environment_new as REF_ENV  # noqa: E402

NS = types.SimpleNamespace


def mk_odom(x, y, qx, qy, qz, qw):
    return NS(pose=NS(pose=NS(position=NS(x=x, y=y, z=0.0), orientation=NS(x=qx, y=qy, z=qz, w=qw))))


def mk_env(threshold_training=True):
    return REF_ENV.Env(threshold_training)


def set_scan(ranges):
    scan = NS(ranges=list(ranges))
    rospy.wait_for_message = lambda *a, **k: scan
    REF_ENV.rospy.wait_for_message = rospy.wait_for_message


# ------------------------------------------------------------------ G1
def gen_g1():
    rng = np.random.default_rng(101)
    rows = []
    # random poses / goals
    for _ in range(1500):
        yaw = rng.uniform(-math.pi, math.pi)
        rows.append([rng.uniform(-3.9, 3.9), rng.uniform(-3.9, 3.9), 0.0, 0.0, math.sin(yaw / 2), math.cos(yaw / 2),
                     rng.uniform(-3.6, 3.6), rng.uniform(-3.6, 3.6)])
    # unnormalised / tilted quaternions
    for _ in range(100):
        q = rng.normal(size=4)
        rows.append([rng.uniform(-3, 3), rng.uniform(-3, 3), q[0] * 0.05, q[1] * 0.05, q[2], q[3],
                     rng.uniform(-3.6, 3.6), rng.uniform(-3.6, 3.6)])
    # yaw rounding ties (x.5 degrees) and wrap
    for deg in [0.5, 1.5, 2.5, -0.5, -1.5, 179.5, -179.5, 180.0, -180.0, 359.5 - 360, 90.0, -90.0, 44.5, 45.5, -10.0, 0.0]:
        yaw = math.radians(deg)
        rows.append([0.0, 0.0, 0.0, 0.0, math.sin(yaw / 2), math.cos(yaw / 2), 1.0, 1.0])
    # decimal rounding ties of dx, dy (1 decimal) incl. values whose binary expansion is below/above the tie
    for dx in [0.05, 0.15, 0.25, 0.35, 0.45, -0.05, -0.15, -0.25, 1.85, 2.675, -2.675, 0.04, -0.04, 0.0, 1e-9, -1e-9]:
        for dy in [0.05, -0.15, 0.25, 1.0, -1.0, 0.0]:
            rows.append([0.0, 0.0, 0.0, 0.0, 0.0, 1.0, dx, dy])
            rows.append([0.3, -0.7, 0.0, 0.0, math.sin(0.4), math.cos(0.4), 0.3 + dx, -0.7 + dy])
    # quadrant edges / axis cases / diff_angle wrap branches
    for gx, gy in [(1, 0), (-1, 0), (0, 1), (0, -1), (0, 0), (1, 1), (-1, 1), (-1, -1), (1, -1), (2, 0.04), (2, -0.04)]:
        for deg in [0, 10, 90, 170, 180, 190, 270, 350, 359]:
            yaw = math.radians(deg)
            rows.append([0.0, 0.0, 0.0, 0.0, math.sin(yaw / 2), math.cos(yaw / 2), float(gx), float(gy)])
    # rel_theta 2-decimal ties are not constructible exactly; dense sweep instead
    for k in range(400):
        a = k * (2 * math.pi / 400)
        rows.append([0.0, 0.0, 0.0, 0.0, math.sin(1.0), math.cos(1.0), 3.0 * math.cos(a), 3.0 * math.sin(a)])
    inp = np.array(rows, dtype=np.float64)
    out = np.zeros((len(rows), 3), dtype=np.float64)
    env = mk_env()
    for i, r in enumerate(inp):
        env.goal_position.position.x = float(r[6])
        env.goal_position.position.y = float(r[7])
        env.getOdometry(mk_odom(float(r[0]), float(r[1]), float(r[2]), float(r[3]), float(r[4]), float(r[5])))
        assert isinstance(env.yaw, int)
        out[i] = [env.yaw, env.rel_theta, env.diff_angle]
    np.savez_compressed(os.path.join(HERE, "g1_odometry.npz"), inp=inp, out=out)
    print("g1", inp.shape)


# ------------------------------------------------------------------ G2
def gen_g2():
    rng = np.random.default_rng(202)
    inf, nan = float("inf"), float("nan")
    scans = []
    for _ in range(300):
        s = rng.uniform(0.12, 3.5, size=10).astype(np.float32).astype(np.float64)
        k = rng.integers(0, 4)
        for _ in range(k):
            s[rng.integers(0, 10)] = rng.choice([inf, nan, -inf, 0.19, 0.2, 0.15, 0.0, 3.5, 0.12, float(np.float32(0.2)),
                                                 float(np.float32(0.19999999))])
        scans.append(s)
    scans.append(np.array([1, 2, inf, .5, .19, 3, nan, 1.2, 1.3, 1.4], dtype=np.float64))
    scans.append(np.array([0.2] * 10))
    scans.append(np.array([0.15] + [1.0] * 9))
    scans.append(np.array([-inf] + [0.15] * 9))
    scans.append(np.array([inf] * 10))
    scans = np.array(scans)
    M = len(scans)
    pos = rng.uniform(-3.5, 3.5, size=(M, 2))
    goal = pos + rng.uniform(-0.5, 0.5, size=(M, 2)) * rng.choice([0.3, 1.0, 6.0], size=(M, 1))
    # exact-threshold cases
    pos[0] = [0.0, 0.0]; goal[0] = [0.2, 0.0]
    pos[1] = [0.0, 0.0]; goal[1] = [0.4, 0.0]
    pos[2] = [1.0, 1.0]; goal[2] = [1.0 + 0.12, 1.0 + 0.16]  # hypot 0.2 (+-ulp)
    thr = rng.choice([0.2, 0.4], size=M)
    out_scan = np.zeros_like(scans)
    out = np.zeros((M, 3))
    for i in range(M):
        env = mk_env(thr[i] == 0.2)
        assert env.threshold_arrive == thr[i]
        env.position = NS(x=float(pos[i, 0]), y=float(pos[i, 1]))
        env.goal_position.position.x = float(goal[i, 0])
        env.goal_position.position.y = float(goal[i, 1])
        env.yaw, env.rel_theta, env.diff_angle = 0, 0.0, 0.0
        sc, dist, _, _, _, done, arrive = env.getState(NS(ranges=[float(v) for v in scans[i]]))
        out_scan[i] = sc
        out[i] = [dist, float(done), float(arrive)]
    np.savez_compressed(os.path.join(HERE, "g2_state.npz"), scans=scans, pos=pos, goal=goal, thr=thr,
                        out_scan=out_scan, out=out)
    print("g2", scans.shape)


# ------------------------------------------------------------------ G3 / G3b
def _rand_case(rng, collide=False, arrive=False):
    yaw = rng.uniform(-math.pi, math.pi)
    pos = rng.uniform(-3.5, 3.5, size=2)
    if arrive:
        goal = pos + rng.uniform(-0.13, 0.13, size=2)
    else:
        goal = rng.uniform(-3.6, 3.6, size=2)
    scan = rng.uniform(0.25, 3.4, size=10).astype(np.float32).astype(np.float64)
    if rng.random() < 0.5:
        scan[rng.integers(0, 10)] = float("inf")
    if collide:
        scan[rng.integers(0, 10)] = float(np.float32(rng.uniform(0.12, 0.1999)))
    action = np.array([rng.uniform(0, 1), rng.uniform(-1, 1)], dtype=np.float32).astype(np.float64)
    past = np.array([rng.uniform(0, 1), rng.uniform(-1, 1)], dtype=np.float32).astype(np.float64)
    past_dist = float(np.hypot(*(goal - pos)) + rng.uniform(-0.05, 0.05))
    return pos, yaw, goal, scan, action, past, past_dist


def gen_g3():
    rng = np.random.default_rng(303)
    M = 400
    inp = np.zeros((M, 2 + 1 + 2 + 10 + 2 + 2 + 1))
    out = np.zeros((M, 16 + 4))  # obs16, reward, done, arrive, new past_distance
    env = mk_env()
    for i in range(M):
        pos, yaw, goal, scan, action, past, pd = _rand_case(rng, collide=(i % 5 == 0))
        if i == 0:  # the SURVEY section-4 known answer
            pos, yaw, goal = np.array([0.5, -0.25]), 0.7, np.array([2.0, 1.0])
            scan = np.array([1, 2, float("inf"), .5, .19, 3, float("nan"), 1.2, 1.3, 1.4])
            action, past, pd = np.array([0.5, 0.1]), np.array([0.0, 0.0]), 2.0
        env.goal_position.position.x, env.goal_position.position.y = float(goal[0]), float(goal[1])
        env.getOdometry(mk_odom(float(pos[0]), float(pos[1]), 0.0, 0.0, math.sin(yaw / 2), math.cos(yaw / 2)))
        env.past_distance = pd
        set_scan([float(v) for v in scan])
        obs, rew, done, arrive = env.step([float(action[0]), float(action[1])], [float(past[0]), float(past[1])])
        if arrive:  # keep G3 arrival-free (arrival consumes the global RNG): re-draw far goal
            continue
        inp[i] = np.concatenate([pos, [yaw], goal, scan, action, past, [pd]])
        out[i] = np.concatenate([obs, [rew, float(done), float(arrive), env.past_distance]])
    keep = np.any(inp != 0, axis=1)
    np.savez_compressed(os.path.join(HERE, "g3_step.npz"), inp=inp[keep], out=out[keep])
    print("g3", int(keep.sum()))


def gen_g3b():
    rng = np.random.default_rng(313)
    M = 120
    inp = np.zeros((M, 2 + 1 + 2 + 10 + 2 + 2 + 1 + 1))
    out = np.zeros((M, 16 + 4 + 2))  # obs16, reward, done, arrive, new past_dist, new goal xy
    env = mk_env()
    for i in range(M):
        pos, yaw, goal, scan, action, past, pd = _rand_case(rng, collide=(i % 7 == 0), arrive=True)
        env.goal_position.position.x, env.goal_position.position.y = float(goal[0]), float(goal[1])
        env.getOdometry(mk_odom(float(pos[0]), float(pos[1]), 0.0, 0.0, math.sin(yaw / 2), math.cos(yaw / 2)))
        env.past_distance = pd
        set_scan([float(v) for v in scan])
        random.seed(1000 + i)
        obs, rew, done, arrive = env.step([float(action[0]), float(action[1])], [float(past[0]), float(past[1])])
        assert arrive
        inp[i] = np.concatenate([pos, [yaw], goal, scan, action, past, [pd], [1000 + i]])
        out[i] = np.concatenate([obs, [rew, float(done), float(arrive), env.past_distance,
                                       env.goal_position.position.x, env.goal_position.position.y]])
    np.savez_compressed(os.path.join(HERE, "g3b_arrive.npz"), inp=inp, out=out)
    print("g3b", M)


# ------------------------------------------------------------------ G4
def _accepts(env, which, x, y):
    """Does the reference's rejection loop accept candidate (x, y)?  which=0 reset, 1 respawn.
    random.uniform is scripted: first the candidate, then a point that is always accepted."""
    seq = [x, y, 0.123, 0.456, 0.123, 0.456]
    it = iter(seq)
    orig = REF_ENV.random.uniform
    REF_ENV.random.uniform = lambda a, b: next(it)
    try:
        if which == 0:
            set_scan([1.0] * 10)
            env.reset()
        else:
            env.position = NS(x=0.0, y=0.0)
            env.goal_position.position.x, env.goal_position.position.y = 0.05, 0.05
            env.yaw, env.rel_theta, env.diff_angle = 0, 0.0, 0.0
            env.past_distance = 0.1
            set_scan([1.0] * 10)
            env.step([0.0, 0.0], [0.0, 0.0])
    finally:
        REF_ENV.random.uniform = orig
    return env.goal_position.position.x == x and env.goal_position.position.y == y


def gen_g4():
    rng = np.random.default_rng(404)
    env = mk_env()
    env.getOdometry(mk_odom(0.0, 0.0, 0.0, 0.0, 0.0, 1.0))  # first odom callback after /gazebo/reset_world
    # (a) seeded resets: obs, goal, past_distance with robot at spawn and a canned scan
    M = 100
    res = np.zeros((M, 1 + 10 + 2 + 1 + 16))
    for i in range(M):
        scan = rng.uniform(0.3, 3.4, size=10).astype(np.float32).astype(np.float64)
        if i % 3 == 0:
            scan[rng.integers(0, 10)] = float("inf")
        set_scan([float(v) for v in scan])
        random.seed(2000 + i)
        env.goal_position.position.x = env.goal_position.position.y = 0.0
        # /gazebo/reset_world puts the robot at the spawn pose and the odom callback fires:
        obs0 = None
        # reference order: reset() samples goal, then waits for scan; odom callback runs asynchronously.
        # We mimic "odom already delivered for the new goal" by calling reset twice with the same seed:
        random.seed(2000 + i)
        env.reset()
        env.getOdometry(mk_odom(0.0, 0.0, 0.0, 0.0, 0.0, 1.0))  # odom for spawn pose vs the new goal
        gx, gy = env.goal_position.position.x, env.goal_position.position.y
        random.seed(2000 + i)
        obs0 = env.reset()
        assert (env.goal_position.position.x, env.goal_position.position.y) == (gx, gy)
        res[i] = np.concatenate([[2000 + i], scan, [gx, gy], [env.past_distance], obs0])
    # (b) acceptance table of the two rectangle sets
    pts = []
    for v in [1.2, 1.4, 1.6, 1.7, 2.3, 2.4, 0.0]:
        for e in [-1e-9, 0.0, 1e-9]:
            for u in [1.2, 1.4, 1.6, 1.7, 2.0, 2.3, 2.4, 0.0, 1.0]:
                for sx in (1, -1):
                    for sy in (1, -1):
                        pts.append((sx * (v + e), sy * u))
                        pts.append((sx * u, sy * (v + e)))
    pts = np.array(sorted(set(pts)))
    extra = rng.uniform(-3.6, 3.6, size=(1500, 2))
    pts = np.concatenate([pts, extra])
    acc = np.zeros((len(pts), 2), dtype=np.uint8)
    for i, (x, y) in enumerate(pts):
        acc[i, 0] = _accepts(env, 0, float(x), float(y))
        acc[i, 1] = _accepts(env, 1, float(x), float(y))
    np.savez_compressed(os.path.join(HERE, "g4_reset.npz"), resets=res, pts=pts, accepted=acc)
    print("g4", M, len(pts), acc.mean(axis=0))


# ------------------------------------------------------------------ PPO side
def import_ppo():
    import torch  # noqa: F401
    import ppo as REF_PPO
    import net_actor as REF_ACTOR
    import net_critic as REF_CRITIC
    return REF_PPO, REF_ACTOR, REF_CRITIC


def gen_g5():
    REF_PPO, _, _ = import_ppo()
    rng = np.random.default_rng(505)
    cases = [[[1, 2, 3], [10, -100]], [[5.0]], [[1.0, 1.0], []], [[], [2.0]]]
    for _ in range(20):
        n_eps = rng.integers(1, 6)
        eps = []
        for _ in range(n_eps):
            L = int(rng.integers(0, 60))
            r = rng.uniform(-25, 25, size=L)
            if L and rng.random() < 0.5:
                r[-1] = rng.choice([-100.0, 120.0])
            eps.append([float(v) for v in r])
        cases.append(eps)
    flat, lens, offs, outs, gammas = [], [], [0], [], []
    for k, eps in enumerate(cases):
        gamma = 0.99 if k % 3 else [0.99, 0.95, 1.0][k // 3 % 3]
        rt = REF_PPO.PPO.compute_rtgs(NS(gamma=gamma), eps).numpy()
        assert rt.dtype == np.float32
        flat += [v for e in eps for v in e]
        lens += [len(e) for e in eps] + [-1]  # -1 terminates a case
        outs.append(rt)
        gammas.append(gamma)
    np.savez_compressed(os.path.join(HERE, "g5_rtgs.npz"), rews=np.array(flat, dtype=np.float64),
                        lens=np.array(lens, dtype=np.int32), gammas=np.array(gammas), out=np.concatenate(outs))
    print("g5", len(cases))


def gen_g6():
    import torch
    REF_PPO, _, _ = import_ppo()
    dev = REF_PPO.device
    rng = np.random.default_rng(606)
    M = 200
    means = np.stack([rng.uniform(0, 1, M), rng.uniform(-1, 1, M)], 1).astype(np.float32)
    rec = np.zeros((M, 2 + 1 + 2 + 1), dtype=np.float64)  # mean, var, action, logp
    for i in range(M):
        var = float(np.float32(rng.choice([0.8, 0.5, 0.1, 0.0999, 0.3])))
        mean_t = torch.tensor(means[i:i + 1], device=dev)
        self_ = NS(actor=lambda obs, m=mean_t: m, use_vision=False,
                   cov_mat=torch.diag(torch.full((2,), var)).to(dev))
        torch.manual_seed(6000 + i)
        act, lp = REF_PPO.PPO.get_action(self_, np.zeros(16), 0, 5)
        rec[i] = [means[i, 0], means[i, 1], var, act[0], act[1], lp]
    # covariance decay trigger table (ppo.py:694-695): (t_step, t_so_far, var) -> var'
    tab = []
    for t_step in (0, 1):
        for t_so_far in (0, 50000, 50001, 10 ** 6):
            for var in (0.8, 0.1, float(np.float32(0.1)), 0.0999, 0.100001):
                cov = torch.diag(torch.full((2,), var)).to(dev)
                self_ = NS(actor=lambda obs: torch.tensor([[0.5, 0.0]], device=dev), use_vision=False, cov_mat=cov)
                REF_PPO.PPO.get_action(self_, np.zeros(16), t_so_far, t_step)
                tab.append([t_step, t_so_far, float(torch.diag(torch.full((2,), var))[0, 0]), float(self_.cov_mat[0, 0]),
                            float(self_.cov_mat[0, 1])])
    np.savez_compressed(os.path.join(HERE, "g6_action.npz"), rec=rec, decay=np.array(tab, dtype=np.float64))
    print("g6", M, len(tab))


def _batch(rng, n):
    obs = np.concatenate([rng.uniform(0.03, 1, (n, 10)), rng.uniform(0, 1, (n, 1)), rng.uniform(-1, 1, (n, 1)),
                          rng.uniform(0, 0.7, (n, 1)), rng.integers(0, 360, (n, 1)) / 360, rng.uniform(0, 1, (n, 1)),
                          rng.uniform(-1, 1, (n, 1))], 1).astype(np.float32)
    acts = np.stack([rng.uniform(0, 1, n), rng.uniform(-1, 1, n)], 1).astype(np.float32)
    acts[rng.random(n) < 0.2, 0] = 0.0  # clamped actions occur often
    acts[rng.random(n) < 0.1, 1] = 1.0
    logp = rng.uniform(-3.5, -1.2, n).astype(np.float32)
    rtgs = rng.normal(20, 60, n).astype(np.float32)
    return obs, acts, logp, rtgs


def gen_g7_g8():
    import torch
    REF_PPO, REF_ACTOR, REF_CRITIC = import_ppo()
    assert str(REF_PPO.device) == "cpu"
    rng = np.random.default_rng(707)
    n = 512
    obs, acts, logp, rtgs = _batch(rng, n)
    tmp = tempfile.mkdtemp(prefix="g7_")
    epochs = 4
    torch.manual_seed(7)
    agent = REF_PPO.PPO(policy_class=REF_ACTOR.NetActor, value_func=REF_CRITIC.NetCritic, env=NS(use_vision=False),
                        state_dim=16, action_dim=2, timesteps_per_batch=n, max_timesteps_per_episode=100,
                        gamma=0.99, n_updates_per_iteration=epochs, lr=3e-4, clip=0.2, save_freq=10 ** 6,
                        method_name="g7", output_dir=tmp)
    init_actor = {k: v.detach().clone().numpy() for k, v in agent.actor.state_dict().items()}
    init_critic = {k: v.detach().clone().numpy() for k, v in agent.critic.state_dict().items()}

    # G8: forward of the seeded nets + key/shape list
    with torch.no_grad():
        a_out = agent.actor(torch.from_numpy(obs)).numpy()
        c_out = agent.critic(torch.from_numpy(obs)).numpy()
    keys = {"actor": [[k, list(v.shape), str(v.dtype)] for k, v in agent.actor.state_dict().items()],
            "critic": [[k, list(v.shape), str(v.dtype)] for k, v in agent.critic.state_dict().items()],
            "actor_params": int(sum(p.numel() for p in agent.actor.parameters())),
            "critic_params": int(sum(p.numel() for p in agent.critic.parameters()))}
    with open(os.path.join(HERE, "g8_nets.json"), "w") as f:
        json.dump(keys, f, indent=0)

    # G7: evaluate() before the update
    V0, lp0 = agent.evaluate(torch.from_numpy(obs), torch.from_numpy(acts), None)
    V0, lp0 = V0.detach().numpy().copy(), lp0.detach().numpy().copy()

    # learn() on the fixed batch: rollout is replaced by the canned batch, nothing else is touched
    def fake_rollout(past_action, t_so_far):
        return (torch.from_numpy(obs), torch.from_numpy(acts), torch.from_numpy(logp), torch.from_numpy(rtgs),
                [n], {"successes": 0, "collisions": 0, "timeouts": 0, "ep_times": [0.0], "ep_count": 1}, None)

    agent.rollout = fake_rollout
    agent.logger["batch_rews"] = [[0.0]]
    agent.logger["batch_lens"] = [n]
    agent._log_summary = lambda: None
    agent.learn(total_timesteps=1, past_action=np.array([0.0, 0.0]))
    actor_losses = np.array([float(x) for x in agent.logger["actor_losses"]])
    critic_losses = np.array([float(x) for x in agent.logger["critic_losses"]])
    fin_actor = {k: v.detach().numpy() for k, v in agent.actor.state_dict().items()}
    fin_critic = {k: v.detach().numpy() for k, v in agent.critic.state_dict().items()}
    trainable_a = [k for k, _ in agent.actor.named_parameters()]
    trainable_c = [k for k, _ in agent.critic.named_parameters()]
    save = dict(obs=obs, acts=acts, logp=logp, rtgs=rtgs, V0=V0, lp0=lp0, a_out=a_out, c_out=c_out,
                actor_losses=actor_losses, critic_losses=critic_losses, epochs=np.array(epochs),
                approx_kl=np.array(agent.logger["approx_kl"]), clip_frac=np.array(agent.logger["clip_frac"]),
                actor_grad_norm=np.array(agent.logger["actor_grad_norm"]),
                critic_grad_norm=np.array(agent.logger["critic_grad_norm"]))
    for k, v in init_actor.items():
        if "bn" in k and "num_batches" in k:
            continue
        save["ia/" + k] = v
    for k, v in init_critic.items():
        if "bn" in k and "num_batches" in k:
            continue
        save["ic/" + k] = v
    # final weights: only the tensors that train, as float32
    for k in trainable_a:
        if "bn" not in k:
            save["fa/" + k] = fin_actor[k]
    for k in trainable_c:
        if "bn" not in k:
            save["fc/" + k] = fin_critic[k]
    np.savez_compressed(os.path.join(HERE, "g7_update.npz"), **save)
    print("g7/g8", n, epochs, actor_losses, critic_losses)


# ------------------------------------------------------------------ G9
def gen_g9():
    """Closed loop: the ORACLE's simulator supplies pose and scan each step (the part Gazebo
    plays for the reference); the REFERENCE Env.step/reset turn them into obs/reward/flags."""
    from oracle import navsim_oracle as O
    from navbot_ppo_amd import maps
    seg = maps.stage_1()
    rng = np.random.default_rng(909)
    E, K = 24, 220
    sim = O.OracleSim(E, n_beams=10, max_episode_steps=0, auto_reset=False, respawn_on_arrive=False, seed=9)
    sim.set_map(seg)
    sim.reset()
    goals = np.stack([rng.uniform(-3.6, 3.6, E), rng.uniform(-3.6, 3.6, E)], 1)
    # a few goals straight ahead so arrivals happen, some envs drive into walls
    goals[:6] = [[0.8, 0.0], [1.2, 0.3], [0.6, -0.4], [1.0, 0.0], [0.5, 0.5], [1.4, -0.1]]
    st = sim.get_state()
    pd0 = np.hypot(goals[:, 0] - st["pose"][:, 0], goals[:, 1] - st["pose"][:, 1])
    sim.set_state(goal=goals, past_dist=pd0)
    actions = np.stack([rng.uniform(0.3, 1, (K, E)), rng.uniform(-1, 1, (K, E))], 2).astype(np.float32)
    actions[:, :6, 1] *= 0.15
    actions[:, 6:10, 1] *= 0.05  # nearly straight -> wall collisions
    envs = [mk_env() for _ in range(E)]
    for e in range(E):
        envs[e].goal_position.position.x, envs[e].goal_position.position.y = float(goals[e, 0]), float(goals[e, 1])
        envs[e].past_distance = float(pd0[e])
    ref_obs = np.zeros((K, E, 16))
    ref_rew = np.zeros((K, E))
    ref_flags = np.zeros((K, E, 2), dtype=np.uint8)
    poses = np.zeros((K, E, 3))
    scans = np.zeros((K, E, 10), dtype=np.float32)
    alive = np.ones(E, dtype=bool)
    alive_rec = np.zeros((K, E), dtype=np.uint8)
    past = np.zeros((E, 2), dtype=np.float32)
    for k in range(K):
        alive_rec[k] = alive
        out = sim.step(actions[k], past_action=past)
        st = sim.get_state()
        poses[k] = st["pose"]
        for e in range(E):
            x, y, th = st["pose"][e]
            sc = O.raycast(seg, x, y, th, 10)
            scans[k, e] = sc
            envs[e].getOdometry(mk_odom(float(x), float(y), 0.0, 0.0, math.sin(th / 2), math.cos(th / 2)))
            set_scan([float(v) for v in sc])
            random.seed(1)
            o, r, d, a = envs[e].step([float(actions[k, e, 0]), float(actions[k, e, 1])],
                                      [float(past[e, 0]), float(past[e, 1])])
            ref_obs[k, e], ref_rew[k, e], ref_flags[k, e] = o, r, (d, a)
            if alive[e]:
                # generation-time cross-check of the oracle against the reference
                assert np.allclose(out["obs"][e], o.astype(np.float32), atol=1e-6), (k, e, out["obs"][e], o)
                assert abs(out["reward"][e] - r) <= 1e-4 * max(1, abs(r)), (k, e, out["reward"][e], r)
                assert (bool(out["done"][e]), bool(out["arrive"][e])) == (d, a), (k, e)
            if d or a:
                alive[e] = False  # episode over: later rows of this env are not compared
        past = actions[k].copy()
    np.savez_compressed(os.path.join(HERE, "g9_closed_loop.npz"), goals=goals, actions=actions, poses=poses,
                        scans=scans, ref_obs=ref_obs, ref_rew=ref_rew, ref_flags=ref_flags, alive=alive_rec,
                        seed=np.array(9))
    print("g9", K, E, "episodes ended:", int((~alive).sum()), "collisions", int(ref_flags[..., 0].sum()),
          "arrivals", int(ref_flags[..., 1].sum()))

# ------------------------------------------------------------------ G10
class _PhiloxRandom:
    """Stands in for the `random` module inside environment_new (the reference draws goals from the UNSEEDED global
    `random`): uniform(a, b) = a + (b - a) * u like CPython's, with u taken from the simulator's documented
    goal stream -- Philox4x32-10 keyed by (seed, env id), counter = draws so far, one call per (x, y) pair."""

    def __init__(self, seed, gid):
        self.key = [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF]
        self.gid, self.ctr, self.pending = gid, 0, None

    def uniform(self, a, b):
        from oracle import navsim_oracle as O
        if self.pending is None:
            r = [int(v) for v in O.philox4x32_10([self.gid & 0xFFFFFFFF, self.gid >> 32, self.ctr, 0x6e617673], self.key)]
            self.ctr += 1
            ux = float(((r[0] << 32) | r[1]) >> 11) * 2.0 ** -53
            self.pending = float(((r[2] << 32) | r[3]) >> 11) * 2.0 ** -53
            return a + (b - a) * ux
        u, self.pending = self.pending, None
        return a + (b - a) * u


def gen_g10():
    """The reference's own PPO.rollout + compute_rtgs drive the reference's own Env; the oracle simulator only plays
    Gazebo's part (pose after 0.2 s of motion, the scan at that pose) and `random` is the injected Philox goal stream, so
    a NavSim(1, auto_reset, respawn_on_arrive) handle with the same seed must reproduce every output."""
    import torch
    from oracle import navsim_oracle as O
    from navbot_ppo_amd import maps
    REF_PPO, _, _ = import_ppo()
    seg = maps.stage_1()
    SEED, T, CAP, GAMMA = 10, 600, 45, 0.99
    rng = np.random.default_rng(1010)
    acts = np.stack([rng.uniform(0.55, 1.0, T), rng.uniform(-1, 1, T)], 1).astype(np.float32)
    # stretches of nearly straight full-speed driving so that collisions and arrivals occur, not only timeouts
    for s0 in range(0, T, 90):
        acts[s0:s0 + 45, 0] = 1.0
        acts[s0:s0 + 45, 1] *= 0.1
    logps = rng.normal(-1.5, 0.3, T).astype(np.float32)

    sim = O.OracleSim(1, n_beams=10, max_episode_steps=0, auto_reset=False, respawn_on_arrive=False, seed=SEED)
    sim.set_map(seg)
    ref = mk_env(True)
    fake_random = _PhiloxRandom(SEED, 0)
    REF_ENV.random = fake_random
    goals, tape = [], {"k": 0}

    def feed(x, y, th):
        # Gazebo's part: while Env.step / Env.reset block on the 5 Hz scan (environment_new.py:282-286, :352-357) the 30 Hz
        # odometry callback (:33,:138) delivers the new pose -- after reset() has drawn its goal
        odom = mk_odom(float(x), float(y), 0.0, 0.0, math.sin(th / 2), math.cos(th / 2))
        scan = NS(ranges=[float(v) for v in O.raycast(seg, x, y, th, 10)])

        def wait_for_message(*a, **k):
            ref.getOdometry(odom)
            return scan
        REF_ENV.rospy.wait_for_message = wait_for_message

    class BackedEnv:
        use_vision = False

        @property
        def position(self):
            return ref.position

        def reset(self):
            sim.set_state(pose=np.zeros((1, 3)))   # /gazebo/reset_world: spawn pose (turtlebot3_stage_1.launch:3-5)
            feed(0.0, 0.0, 0.0)
            o = ref.reset()
            goals.append([ref.goal_position.position.x, ref.goal_position.position.y])
            return o

        def step(self, action, past_action):
            sim.step(np.asarray(action, dtype=np.float32).reshape(1, 2))
            x, y, th = sim.get_state()["pose"][0]
            feed(x, y, th)
            return ref.step(action, past_action)

    episodes = []

    def get_action(obs, t_so_far, one_round, vision_feat=None):
        # every other episode is driven at its goal by a bang-bang controller on diff_angle (obs[15] * 180) so that
        # arrivals occur; the emitted actions ARE the tape the GPU test replays
        k = tape["k"]
        tape["k"] += 1
        if one_round == 0:
            tape["ep"] = tape.get("ep", -1) + 1
        if tape["ep"] % 2 == 0:
            diff = float(obs[15]) * 180.0
            acts[k] = [1.0 if abs(diff) < 30 else 0.3, min(1.0, max(-1.0, -diff / 30.0))]
        return acts[k].copy(), logps[k]

    tmp = tempfile.mkdtemp(prefix="g10_")
    me = NS(use_vision=False, env=BackedEnv(), timesteps_per_batch=T, max_timesteps_per_episode=CAP, gamma=GAMMA,
            get_action=get_action, episode_count=0, logger={"Episode_Rewards": []}, log_dir_path=tmp,
            _log_episode_metrics=lambda **kw: episodes.append(kw))
    me.compute_rtgs = types.MethodType(REF_PPO.PPO.compute_rtgs, me)
    b_obs, b_acts, b_logp, b_rtgs, b_lens, metrics, _ = REF_PPO.PPO.rollout(me, np.array([0.0, 0.0]), 0)
    rews = [r for e in me.logger["batch_rews"] for r in e]
    ep = {k: np.array([e[k] for e in episodes], dtype=np.float64)
          for k in ("timestep", "success", "collision", "timeout", "length", "ep_return", "path_length")}
    np.savez_compressed(
        os.path.join(HERE, "g10_rollout.npz"), seed=np.array(SEED), cap=np.array(CAP), gamma=np.array(GAMMA), acts_tape=acts,
        logp_tape=logps, batch_obs=b_obs.numpy(), batch_acts=b_acts.numpy(), batch_log_probs=b_logp.numpy(),
        batch_rtgs=b_rtgs.numpy(), batch_lens=np.array(b_lens, dtype=np.int32), rews=np.array(rews, dtype=np.float64),
        goals=np.array(goals), rng_ctr_final=np.array(fake_random.ctr),
        trailing_len=np.array(len(me.logger["batch_rews"][-1])),
        episode_rewards_log=np.array(me.logger["Episode_Rewards"], dtype=np.float64),
        iter_counts=np.array([metrics["successes"], metrics["collisions"], metrics["timeouts"], metrics["ep_count"]]),
        **{"ep_" + k: v for k, v in ep.items()})
    print("g10", T, "episodes", len(b_lens), "lens", b_lens, "succ/coll/timeout",
          metrics["successes"], metrics["collisions"], metrics["timeouts"], "trailing", len(me.logger["batch_rews"][-1]))


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g3b", "g4", "g5", "g6", "g7", "g9", "g10"]
    fns = dict(g1=gen_g1, g2=gen_g2, g3=gen_g3, g3b=gen_g3b, g4=gen_g4, g5=gen_g5, g6=gen_g6, g7=gen_g7_g8, g9=gen_g9, g10=gen_g10)
    for w in which:
        fns[w]()
