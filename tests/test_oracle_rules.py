"""Pins layer (A) of the oracle -- the restated reference rules -- against golden vectors
recorded from the reference itself (tests/golden/generate_golden.py).  CPU only."""
import math
import os

import numpy as np

from oracle import navsim_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_py_round_matches_python_round():
    rng = np.random.default_rng(0)
    xs = list(rng.uniform(-8, 8, 20000)) + [0.05, 0.15, 0.25, 0.35, 0.45, 2.675, -2.675, 1.005, -0.04, 0.0, -0.0,
                                             1e-9, 0.5, 1.5, 2.5, -0.5, 359.995, 179.995, -180.005, 0.125, 0.375]
    xs += [k / 20 for k in range(-200, 200)] + [k / 200 for k in range(-2000, 2000)]
    for x in xs:
        for nd in (1, 2):
            got, want = O.py_round_nd(x, nd), round(x, nd)
            assert got == want and math.copysign(1, got) == math.copysign(1, want), (x, nd, got, want)


def test_g1_get_odometry_exact():
    d = load("g1_odometry.npz")
    for r, o in zip(d["inp"], d["out"]):
        yaw, rt, da = O.get_odometry(*[float(v) for v in r])
        assert yaw == int(o[0]), (r, yaw, o)
        assert rt == o[1], (r, rt, o)
        assert da == o[2], (r, da, o)


def test_g2_get_state():
    d = load("g2_state.npz")
    for i in range(len(d["scans"])):
        scan, dist, done, arrive = O.get_state(d["scans"][i], d["pos"][i, 0], d["pos"][i, 1], d["goal"][i, 0],
                                               d["goal"][i, 1], float(d["thr"][i]))
        np.testing.assert_array_equal(scan, d["out_scan"][i])
        assert abs(dist - d["out"][i, 0]) <= 2e-16 * max(1.0, dist)
        # flags: bit-exact unless the distance sits within 1 ulp of the threshold (hypot implementations differ)
        if abs(d["out"][i, 0] - d["thr"][i]) > 1e-15:
            assert arrive == bool(d["out"][i, 2]), i
        assert done == bool(d["out"][i, 1]), i


def _step_via_rules(pos, yaw, goal, scan, past, past_dist, thr=0.2):
    yw, rt, da = O.get_odometry(pos[0], pos[1], 0.0, 0.0, math.sin(yaw / 2), math.cos(yaw / 2), goal[0], goal[1])
    sc, dist, done, arrive = O.get_state(scan, pos[0], pos[1], goal[0], goal[1], thr)
    obs = O.assemble_obs(sc, past, dist, yw, rt, da)
    rew, new_pd = O.set_reward(past_dist, dist, done, arrive)
    return obs, rew, done, arrive, new_pd, dist


def test_g3_step_composition():
    d = load("g3_step.npz")
    assert len(d["inp"]) > 300
    for r, o in zip(d["inp"], d["out"]):
        pos, yaw, goal, scan, action, past, pd = r[0:2], r[2], r[3:5], r[5:15], r[15:17], r[17:19], r[19]
        obs, rew, done, arrive, new_pd, _ = _step_via_rules(pos, yaw, goal, scan, past, pd)
        np.testing.assert_allclose(obs, o[:16], rtol=0, atol=1e-15)
        assert abs(rew - o[16]) <= 1e-12 * max(1.0, abs(o[16]))
        assert (done, arrive) == (bool(o[17]), bool(o[18]))
        assert abs(new_pd - o[19]) <= 1e-15 * max(1.0, o[19])


def test_g3_known_answer_from_survey():
    d = load("g3_step.npz")
    o = d["out"][0]
    np.testing.assert_allclose(o[:16], [.28571429, .57142857, 1., .14285714, .05428571, .85714286, 0., .34285714,
                                        .37142857, .4, 0, 0, .18166712, .11111111, .10738889, .00744444], atol=5e-9)
    assert abs(o[16] - 23.718790511668253) < 1e-12 and o[17] == 0 and o[18] == 0


def test_g3b_arrival_branch():
    d = load("g3b_arrive.npz")
    for r, o in zip(d["inp"], d["out"]):
        pos, yaw, goal, scan, action, past, pd = r[0:2], r[2], r[3:5], r[5:15], r[15:17], r[17:19], r[19]
        obs, rew, done, arrive, _, _ = _step_via_rules(pos, yaw, goal, scan, past, pd)
        np.testing.assert_allclose(obs, o[:16], rtol=0, atol=1e-15)
        assert arrive and rew == 120.0 == o[16] and done == bool(o[17]) and o[18] == 1.0
        # the reference respawned a goal outside the respawn rectangles and re-based past_distance on it
        ngx, ngy = o[20], o[21]
        assert not O.goal_rejected(1, ngx, ngy) and -3.6 <= ngx <= 3.6 and -3.6 <= ngy <= 3.6
        assert abs(o[19] - math.hypot(ngx - pos[0], ngy - pos[1])) < 1e-14


def test_g4_reset_and_rejection_rectangles():
    d = load("g4_reset.npz")
    for row in d["resets"]:
        scan, gx, gy, pd, obs = row[1:11], row[11], row[12], row[13], row[14:30]
        assert not O.goal_rejected(0, gx, gy)
        yw, rt, da = O.get_odometry(0.0, 0.0, 0.0, 0.0, 0.0, 1.0, gx, gy)
        sc, dist, _, _ = O.get_state(scan, 0.0, 0.0, gx, gy, 0.2)
        got = O.assemble_obs(sc, [0.0, 0.0], dist, yw, rt, da)
        np.testing.assert_allclose(got, obs, rtol=0, atol=1e-15)
        assert abs(pd - dist) < 1e-15
    for (x, y), acc in zip(d["pts"], d["accepted"]):
        assert O.goal_rejected(0, x, y) == (not acc[0]), (x, y)
        assert O.goal_rejected(1, x, y) == (not acc[1]), (x, y)


def test_g5_compute_rtgs():
    d = load("g5_rtgs.npz")
    rews, lens, gammas, out = d["rews"], d["lens"], d["gammas"], d["out"]
    ro = oo = 0
    case = []
    k = 0
    for L in lens:
        if L >= 0:
            case.append([float(v) for v in rews[ro:ro + L]])
            ro += L
            continue
        got = O.compute_rtgs_ragged(case, float(gammas[k]))
        np.testing.assert_array_equal(got, out[oo:oo + len(got)])
        # the [T,N] form of the same recurrence: one env column, `ended` at each episode's last step
        flat = np.array([v for e in case for v in e], dtype=np.float64)
        if len(flat) and np.all(flat.astype(np.float32).astype(np.float64) == flat):
            ended = np.zeros(len(flat), np.uint8)
            pos = np.cumsum([len(e) for e in case if len(e)]) - 1
            ended[pos] = 1
            tn = O.compute_rtgs_tn(flat.astype(np.float32)[:, None], ended[:, None], float(gammas[k]))[:, 0]
            np.testing.assert_array_equal(tn, got)
        oo += len(got)
        case = []
        k += 1
    assert k == len(gammas) and oo == len(out)


def test_rtgs_tn_equals_ragged_on_f32_rewards():
    rng = np.random.default_rng(3)
    T, N = 97, 13
    rew = rng.uniform(-25, 25, (T, N)).astype(np.float32)
    ended = (rng.random((T, N)) < 0.06).astype(np.uint8)
    tn = O.compute_rtgs_tn(rew, ended, 0.99)
    for n in range(N):
        eps, cur = [], []
        for t in range(T):
            cur.append(float(rew[t, n]))
            if ended[t, n]:
                eps.append(cur)
                cur = []
        eps.append(cur)
        np.testing.assert_array_equal(O.compute_rtgs_ragged(eps, 0.99), tn[:, n])
