"""G10: the reference's own PPO.rollout + compute_rtgs (project_ppo/src/ppo.py:463-671) over the reference's Env, recorded by
tests/golden/generate_golden.py, against the oracle's episode logic (the checker the GPU rollout is compared with).  CPU only."""
import os

import numpy as np

from navbot_ppo_amd import maps
from oracle import navsim_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def replay_on(sim, d):
    """Drives an auto-resetting 1-env simulator (oracle or GPU adapter with .reset() -> obs[1,16] and
    .step(a[1,2]) -> dict) with the recorded action tape, storing what PPO.rollout stores (ppo.py:508,541-546)."""
    acts = d["acts_tape"]
    T = len(acts)
    obs = np.zeros((T, 16), np.float32)
    rew = np.zeros(T, np.float32)
    ended = np.zeros(T, np.uint8)
    flags = np.zeros((T, 2), np.uint8)
    eplen = np.zeros(T, np.int32)
    epret = np.zeros(T, np.float32)
    eppath = np.zeros(T, np.float32)
    o = sim.reset()
    for t in range(T):
        obs[t] = o[0]                                  # ppo.py:508: the obs is stored BEFORE acting
        out = sim.step(acts[t:t + 1])
        o = out["obs"]                                 # post-reset obs when the episode ended (ppo.py:593)
        rew[t], ended[t] = out["reward"][0], out["ended"][0]
        flags[t] = out["done"][0], out["arrive"][0]
        eplen[t], epret[t], eppath[t] = out["ep_length"][0], out["ep_return"][0], out["ep_path"][0]
    return obs, rew, ended, flags, eplen, epret, eppath


def check_against_g10(d, obs, rew, ended, flags, eplen, epret, eppath, rtg):
    np.testing.assert_allclose(obs, d["batch_obs"], atol=1e-6, rtol=0)          # store-before-act order, reset obs, past_action rule
    np.testing.assert_allclose(rew, d["rews"].astype(np.float32), rtol=1e-5, atol=1e-5)
    ends = np.nonzero(ended)[0]
    lens = eplen[ends]
    np.testing.assert_array_equal(lens, d["batch_lens"])                         # completed episodes only (ppo.py:582)
    assert len(ended) - (ends[-1] + 1) == int(d["trailing_len"])                 # the trailing partial episode (ppo.py:601)
    assert int(lens.sum()) == int(d["batch_lens"].sum())                         # what t_so_far advances by (ppo.py:258)
    succ = flags[ends, 1].astype(bool)
    coll = flags[ends, 0].astype(bool) & ~succ
    tmo = ~flags[ends, 0].astype(bool) & ~succ                                   # ppo.py:558-560
    np.testing.assert_array_equal(succ, d["ep_success"].astype(bool))
    np.testing.assert_array_equal(coll, d["ep_collision"].astype(bool))
    np.testing.assert_array_equal(tmo, d["ep_timeout"].astype(bool))
    np.testing.assert_array_equal([succ.sum(), coll.sum(), tmo.sum(), len(ends)], d["iter_counts"])
    np.testing.assert_allclose(epret[ends], d["ep_ep_return"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(eppath[ends], d["ep_path_length"], rtol=1e-5, atol=1e-6)             # ppo.py:533-537
    np.testing.assert_allclose(epret[ends] / lens, d["episode_rewards_log"], rtol=1e-5, atol=1e-5)   # ppo.py:586
    np.testing.assert_array_equal(ends + 1, d["ep_timestep"].astype(np.int64))   # t_so_far + sum(batch_lens) + one_round, :564
    # returns: every episode end AND the batch end restart the scan at 0 (ppo.py:658-666); f64 accumulate, f32 store
    np.testing.assert_allclose(rtg, d["batch_rtgs"], rtol=1e-6, atol=1e-5)


def test_g10_reference_rollout_vs_oracle():
    d = np.load(os.path.join(G, "g10_rollout.npz"))
    assert d["ep_success"].sum() >= 1 and d["ep_collision"].sum() >= 1 and d["ep_timeout"].sum() >= 1
    sim = O.OracleSim(1, max_episode_steps=int(d["cap"]), auto_reset=True, respawn_on_arrive=True, seed=int(d["seed"]))
    sim.set_map(maps.stage_1())
    obs, rew, ended, flags, eplen, epret, eppath = replay_on(sim, d)
    rtg = O.compute_rtgs_tn(rew[:, None], ended[:, None], float(d["gamma"]))[:, 0]
    check_against_g10(d, obs, rew, ended, flags, eplen, epret, eppath, rtg)
    assert int(sim.get_state()["rng_ctr"][0]) == int(d["rng_ctr_final"])         # same number of goal draws as the reference made
    # stored actions / log-probs are the tape itself (ppo.py:546-547)
    np.testing.assert_array_equal(d["batch_acts"], d["acts_tape"])
    np.testing.assert_array_equal(d["batch_log_probs"], d["logp_tape"])
