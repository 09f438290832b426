import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
from navbot_ppo_amd import maps
from navbot_ppo_amd.env import NavSim
from oracle import navsim_oracle as O
_S0 = int(os.environ.get("SOAK_SEED", "0"))   # another goal stream, action stream and policy for every configuration
def soak(N, seg, per_env, K, B=10, cap=60, seed=_S0, sampler=None, amax=1.0):
    gpu = NavSim(N, n_beams=B, max_episode_steps=cap, auto_reset=True, seed=seed)
    cpu = O.OracleSim(N, n_beams=B, max_episode_steps=cap, auto_reset=True, seed=seed)
    for s in (gpu, cpu):
        s.set_map(seg, per_env=per_env)
        if sampler: s.set_spawn_sampler(*sampler)
    io = gpu.alloc_io(); og = gpu.reset(io.obs).cpu().numpy(); oc = cpu.reset()
    assert np.abs(og - oc).max() <= 1e-6
    rng = np.random.default_rng(seed); bad = 0; exact = 0; tot = 0; ends = 0; mx = 0.0
    for k in range(K):
        a = np.stack([rng.uniform(0, amax, N), rng.uniform(-amax, amax, N)], 1).astype(np.float32)
        a[: N // 3, 1] *= 0.05
        gpu.step(torch.from_numpy(a).cuda(), io.obs, io.reward, io.done, io.arrive, io.ended)
        out = cpu.step(a)
        o = io.obs.cpu().numpy()
        d = np.abs(o - out["obs"]).max(); mx = max(mx, d)
        fl = (io.done.cpu().numpy() != out["done"]).sum() + (io.arrive.cpu().numpy() != out["arrive"]).sum() + (io.ended.cpu().numpy() != out["ended"]).sum()
        bad += int(fl) + int(d > 1e-6)
        exact += int((o == out["obs"]).all(1).sum()); tot += N; ends += int(out["ended"].sum())
        if fl or d > 1e-6:
            print("MISMATCH step", k, "maxdiff", d, "flags", fl); break
    print(f"N={N} S={seg.shape[-2]} per_env={per_env} B={B} steps={K}: bad={bad} max|dobs|={mx:.2e} exact rows {exact/tot:.5f} episode ends {ends}")
st, g, lo, hi = maps.spawn_tables("small_house"); seg = maps.house(2048)
if "--rollout" not in sys.argv:   # (--rollout: only the closed-loop part below)
    soak(4096, maps.replicate_per_env(maps.stage_2(), 4096, seed=1), True, 600)
    soak(16384, maps.replicate_per_env(maps.stage_2(), 16384, seed=2), True, 80)
    soak(4096, maps.stage_1(), False, 800)
    soak(2048, seg, False, 400, cap=40, sampler=maps.open_tables(seg, st, g) + (lo, hi))
    soak(8192, seg, False, 60, cap=40, sampler=maps.open_tables(seg, st, g) + (lo, hi))
    soak(2048, maps.stage_4(), False, 300, B=36)
    soak(2048, maps.replicate_per_env(maps.stage_2(sides=56), 2048, seed=3), True, 300, amax=3.0)
    # round 5 (ADVICE: the 36-beam stage B tests only the beams of a segment's float32 extent +- 0.02 beams): 36 beams on per-env maps, on
    # the house map (near-sensor and grazing segments, tile boxes) and with fast spins
    soak(2048, maps.replicate_per_env(maps.stage_2(), 2048, seed=4), True, 300, B=36)
    soak(2048, seg, False, 250, B=36, cap=40, sampler=maps.open_tables(seg, st, g) + (lo, hi))
    soak(1024, maps.replicate_per_env(maps.stage_2(sides=56), 1024, seed=5), True, 300, B=36, amax=3.0)


def soak_rollout(N, seg, per_env, T, iters=2, cap=60, seed=_S0, sampler=None, B=10, half=False, policy="mlp64x2"):
    """The persistent rollout CLOSED-LOOP (navsim_rollout_mlp64: rollout_big_kernel beyond 4096 envs): the actions the in-kernel policy
    chose are replayed on the oracle for EVERY env; every observation row, flag and reward of every step is compared."""
    from navbot_ppo_amd import ppo
    from navbot_ppo_amd.env import VecEnv
    env = VecEnv(N, map=seg, n_beams=B, max_episode_steps=cap, seed=seed, per_env_map=False, sampler=sampler, obs_f16=half)
    tr = ppo.PPOTrainer(env, ppo.PPOConfig(rollout_len=T, max_episode_steps=cap, policy=policy, seed=seed + 1))
    assert tr.uses_persistent_rollout
    with torch.no_grad():
        (tr.actor.layer3 if policy == "mlp64x2" else tr.actor.out1).bias.add_(2.0)   # drive forward: collisions and arrivals, not only timeouts
    cpu = O.OracleSim(N, n_beams=B, max_episode_steps=cap, auto_reset=True, seed=seed)
    cpu.set_map(seg, per_env=per_env)
    if sampler: cpu.set_spawn_sampler(*sampler)
    # float16 buffers: a row must be the oracle's row rounded to half (compared in half: 1e-6 before the rounding = at most one half ulp)
    rnd = (lambda x: x.astype(np.float16).astype(np.float32)) if half else (lambda x: x)
    tol = 1e-3 if half else 1e-6
    bad = 0; exact = 0; tot = 0; ends = 0; mx = 0.0; mr = 0.0
    for it in range(iters):
        tr.rollout(); torch.cuda.synchronize()
        obs, acts = tr.obs_buf.float().cpu().numpy(), tr.act_buf.cpu().numpy()
        fl = {k: getattr(tr, k + "_buf").cpu().numpy() for k in ("done", "arrive", "ended", "rew")}
        o0 = cpu.reset()   # ppo.py:486: every batch starts from a reset
        bad += int(np.abs(obs[0] - rnd(o0)).max() > tol)
        for t in range(T):
            out = cpu.step(acts[t])
            out["obs"] = rnd(out["obs"])
            d = np.abs(obs[t + 1] - out["obs"]).max(); mx = max(mx, d)
            mr = max(mr, float(np.abs(fl["rew"][t] - out["reward"]).max()))
            f = sum(int((fl[k][t] != out[k]).sum()) for k in ("done", "arrive", "ended"))
            bad += f + int(d > tol)
            exact += int((obs[t + 1] == out["obs"]).all(1).sum()); tot += N; ends += int(out["ended"].sum())
            if f or d > tol:
                print("MISMATCH iteration", it, "step", t, "maxdiff", d, "flags", f); break
    print(f"closed-loop rollout {policy} N={N} S={seg.shape[-2]} per_env={per_env} B={B} {'f16' if half else 'f32'} rows T={T} x {iters}: bad={bad} max|dobs|={mx:.2e} max|dreward|={mr:.2e} "
          f"exact rows {exact/tot:.5f} episode ends {ends}")
    env.close()


if True:
    soak_rollout(16384, maps.replicate_per_env(maps.stage_2(), 16384, seed=2), True, 200)
    soak_rollout(16384, maps.stage_1(), False, 200)
    soak_rollout(16384, seg, False, 60, cap=40, sampler=maps.open_tables(seg, st, g) + (lo, hi))
    soak_rollout(4096, maps.stage_1(), False, 512, cap=500)
    # round 5: the 36-beam and float16 instantiations of the rollout kernels (configs[3]'s / configs[4]'s shards and around them)
    soak_rollout(4096, maps.stage_4(), False, 400, cap=200, B=36)
    soak_rollout(8192, seg, False, 80, cap=40, sampler=maps.open_tables(seg, st, g) + (lo, hi), half=True)
    soak_rollout(4096, maps.stage_1(), False, 300, cap=150, half=True)
    soak_rollout(4608, maps.stage_4(), False, 120, B=36, half=True)
    # round 5: the persistent rollout of the reference's 512-wide actor (navsim_rollout_resmlp512)
    soak_rollout(4096, maps.stage_1(), False, 512, cap=500, policy="resmlp512")
    soak_rollout(3000, seg, False, 100, cap=40, sampler=maps.open_tables(seg, st, g) + (lo, hi), policy="resmlp512")
    soak_rollout(1024, maps.replicate_per_env(maps.stage_2(), 1024, seed=3), True, 300, cap=120, policy="resmlp512")
