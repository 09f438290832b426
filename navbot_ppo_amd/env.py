"""Host-side mirror of the reference environment interface over libnavsim.so.

``Env`` is the N=1 drop-in for ``project_ppo/src/environment_new.py:26`` (same constructor
arguments, ``reset()``/``step(action, past_action)`` return types, and the attributes its callers
read: ``position.x/.y`` (ppo.py:535, main.py:202), ``goal_position.position.x/.y``,
``threshold_arrive``, ``past_distance``, ``use_vision``).  ``VecEnv`` is the batched form the
rollout of ``ppo.py:463-641`` becomes on a GPU: N envs per call, device tensors in and out,
auto-reset inside the step kernel.

PyTorch is used for device memory and streams only; all simulation happens in the HIP kernels.
"""
import ctypes as C
import math
import os
import types

import numpy as np
import torch

from . import maps as _maps
from ._native import NavsimCfg, NavsimError, NavsimInfo, check, lib

__all__ = ["NavSim", "VecEnv", "Env", "NavsimError", "rtg_scan", "gae_scan"]


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class NavSim:
    """One libnavsim handle (= one GPU's shard of envs) with torch tensors as buffers."""

    def __init__(self, n_envs, n_beams=10, max_episode_steps=0, auto_reset=False, respawn_on_arrive=False,
                 seed=0, env_id_base=0, threshold_arrive=0.2, spawn=(0.0, 0.0, 0.0), goal_box=(-3.6, 3.6),
                 obs_f16=False, device=None, lidar_below_min="clamp", lidar_noise_sigma=0.0, envs_per_workgroup=None,
                 pair_cast=None):
        """envs_per_workgroup / pair_cast: kernel-shape overrides of THIS handle (navsim_set_shape; tests and A/B timing --
        results never depend on them).  None = the dev tools' NAVSIM_EPB / NAVSIM_PAIR_CAST environment variables if set
        (read here, per handle; libnavsim itself reads nothing from the environment), else the built-in rule."""
        if not torch.cuda.is_available():
            raise NavsimError("navbot_ppo_amd needs a HIP device (MI355X); there is no CPU path")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise NavsimError(f"navbot_ppo_amd needs a HIP device, got device={device!r}; there is no CPU path")
        if self.device.index is None:   # "cuda" names the current device: tensors report "cuda:N", and _chk compares devices
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.N, self.B, self.D = int(n_envs), int(n_beams), int(n_beams) + 6
        self.obs_dtype = torch.float16 if obs_f16 else torch.float32
        self.cfg = NavsimCfg(self.N, self.B, int(max_episode_steps), int(bool(auto_reset)), int(bool(respawn_on_arrive)),
                             int(bool(obs_f16)), {"clamp": 0, "gazebo": 1}[lidar_below_min], float(lidar_noise_sigma),
                             int(seed), int(env_id_base), float(threshold_arrive),
                             float(spawn[0]), float(spawn[1]), float(spawn[2]), float(goal_box[0]), float(goal_box[1]))
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().navsim_create(C.byref(self.cfg), C.byref(self._h)), "navsim_create")
        self._seg = None  # keeps the map tensor alive: the handle only borrows the pointer
        self.generation = 0  # bumped whenever device pointers / scalars a captured hipGraph would have frozen change
        def knob(name):   # a dev tool's environment variable; anything that is not an integer is ignored, as the library used to
            try:
                return int(os.environ[name])
            except (KeyError, ValueError):
                return None
        if envs_per_workgroup is None and knob("NAVSIM_EPB") is not None:
            envs_per_workgroup = knob("NAVSIM_EPB")
        if pair_cast is None and knob("NAVSIM_PAIR_CAST") is not None:
            pair_cast = knob("NAVSIM_PAIR_CAST") != 0
        if envs_per_workgroup is not None or pair_cast is not None:
            self.set_shape(envs_per_workgroup or 0, pair_cast)

    def set_shape(self, envs_per_workgroup=0, pair_cast=None):
        """navsim_set_shape: force the workgroup shape (0 = rule; 4 | 8 | 16 | 32 | 64) / the 128-segment passes of this handle."""
        check(lib().navsim_set_shape(self._h, int(envs_per_workgroup), -1 if pair_cast is None else int(bool(pair_cast))),
              "navsim_set_shape")
        self.generation += 1   # a captured hipGraph froze the old instantiation

    def info(self):
        """navsim_get_info as a dict: handle facts + the (envs, waves, cast variant) each entry point launches right now."""
        inf = NavsimInfo()
        check(lib().navsim_get_info(self._h, C.byref(inf)), "navsim_get_info")
        return {n: getattr(inf, n) for n, _ in NavsimInfo._fields_ if n != "reserved"}

    def close(self):
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            lib().navsim_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- world
    def set_map(self, seg, per_env=None):
        """seg [S,4] (shared) or [N,S,4] (per env) float32 segments (ax, ay, bx, by).  The handle BORROWS the device tensor it
        is given (kept alive here) -- except for shared maps of 65..4096 segments, of which navsim_set_map takes its own
        Morton-ordered SNAPSHOT with tile bounding boxes (a blocking copy + host sort).  So: after editing a map tensor in
        place (domain randomisation), call set_map again; never call it inside a stream capture or a hot loop."""
        seg = torch.as_tensor(np.asarray(seg, dtype=np.float32) if not torch.is_tensor(seg) else seg)
        seg = seg.to(device=self.device, dtype=torch.float32).contiguous()
        if per_env is None:
            per_env = seg.dim() == 3
        if seg.shape[-1] != 4 or seg.dim() != (3 if per_env else 2) or (per_env and seg.shape[0] != self.N):
            raise NavsimError(f"bad map shape {tuple(seg.shape)} for per_env={per_env}, N={self.N}")
        with torch.cuda.device(self.device):
            check(lib().navsim_set_map(self._h, _ptr(seg), int(seg.shape[-2]), int(bool(per_env)), _stream()), "navsim_set_map")
        self._seg = seg
        self.S, self.per_env = int(seg.shape[-2]), bool(per_env)
        self.generation += 1

    def set_goal_rects(self, which, rects):
        r = np.ascontiguousarray(rects, dtype=np.float64).reshape(-1, 4)
        with torch.cuda.device(self.device):   # the call synchronises the CURRENT device before touching the rectangles
            check(lib().navsim_set_goal_rects(self._h, int(which), _np_ptr(r), r.shape[0]), "navsim_set_goal_rects")
        self.generation += 1

    def set_spawn_sampler(self, starts, goals=None, min_dist=1.5, max_dist=6.0):
        """GoalSpawnSampler tables (spawn_goal_sampler.py:37-62): start poses [K,3], goal points [G,2] or None."""
        st = np.ascontiguousarray(starts, dtype=np.float64).reshape(-1, 3)
        g = None if goals is None else np.ascontiguousarray(goals, dtype=np.float64).reshape(-1, 2)
        with torch.cuda.device(self.device):
            check(lib().navsim_set_spawn_sampler(self._h, _np_ptr(st), st.shape[0], _np_ptr(g), 0 if g is None else g.shape[0],
                                                 float(min_dist), float(max_dist), _stream()), "navsim_set_spawn_sampler")
        self.generation += 1

    # -- buffers
    def alloc_io(self):
        dev, N = self.device, self.N
        return types.SimpleNamespace(
            obs=torch.zeros((N, self.D), dtype=self.obs_dtype, device=dev),
            reward=torch.zeros(N, dtype=torch.float32, device=dev),
            done=torch.zeros(N, dtype=torch.uint8, device=dev),
            arrive=torch.zeros(N, dtype=torch.uint8, device=dev),
            ended=torch.zeros(N, dtype=torch.uint8, device=dev),
            ep_return=torch.zeros(N, dtype=torch.float32, device=dev),
            ep_length=torch.zeros(N, dtype=torch.int32, device=dev),
            ep_path=torch.zeros(N, dtype=torch.float32, device=dev))

    # -- calls (all asynchronous on torch's current stream)
    # The C ABI takes raw device pointers and cannot know what is behind them: every tensor is checked HERE -- device, dtype,
    # element count, contiguity -- before its data_ptr() crosses the boundary (a float32 buffer handed to an f16 handle, or rows of
    # the wrong width, would otherwise be overwritten or overrun silently).
    def _chk(self, name, t, dtype, numel, optional=False):
        if t is None:
            if optional:
                return
            raise NavsimError(f"{name}: required")
        if not torch.is_tensor(t) or t.device != self.device:
            raise NavsimError(f"{name}: expected a tensor on {self.device}, got {getattr(t, 'device', type(t))}")
        if t.dtype != dtype:
            raise NavsimError(f"{name}: expected {dtype}, got {t.dtype}"
                              + (" (this handle writes float16 observations: obs_f16=True)" if dtype == torch.float16 else ""))
        if t.numel() != numel or not t.is_contiguous():
            raise NavsimError(f"{name}: expected {numel} contiguous elements, got shape {tuple(t.shape)}"
                              f"{'' if t.is_contiguous() else ' (not contiguous)'}")

    def _chk_step_io(self, rows, obs, reward, done, arrive, ended, ep_return, ep_length, ep_path, what):
        n = rows * self.N
        self._chk(f"{what}: obs", obs, self.obs_dtype, n * self.D)
        self._chk(f"{what}: reward", reward, torch.float32, n)
        self._chk(f"{what}: done", done, torch.uint8, n)
        self._chk(f"{what}: arrive", arrive, torch.uint8, n)
        self._chk(f"{what}: ended", ended, torch.uint8, n, optional=True)
        self._chk(f"{what}: ep_return", ep_return, torch.float32, n, optional=True)
        self._chk(f"{what}: ep_length", ep_length, torch.int32, n, optional=True)
        self._chk(f"{what}: ep_path", ep_path, torch.float32, n, optional=True)

    def reset(self, obs, mask=None):
        self._chk("reset: obs", obs, self.obs_dtype, self.N * self.D)
        self._chk("reset: mask", mask, torch.uint8, self.N, optional=True)
        with torch.cuda.device(self.device):
            check(lib().navsim_reset(self._h, _ptr(mask), _ptr(obs), _stream()), "navsim_reset")
        return obs

    def step(self, action, obs, reward, done, arrive, ended=None, ep_return=None, ep_length=None, past_action=None,
             ep_path=None):
        self._chk("step: action", action, torch.float32, 2 * self.N)
        self._chk("step: past_action", past_action, torch.float32, 2 * self.N, optional=True)
        self._chk_step_io(1, obs, reward, done, arrive, ended, ep_return, ep_length, ep_path, "step")
        with torch.cuda.device(self.device):
            check(lib().navsim_step(self._h, _ptr(action), _ptr(past_action), _ptr(obs), _ptr(reward), _ptr(done),
                                    _ptr(arrive), _ptr(ended), _ptr(ep_return), _ptr(ep_length), _ptr(ep_path), _stream()),
                  "navsim_step")

    def step_seq(self, actions, obs, reward, done, arrive, ended=None, ep_return=None, ep_length=None, ep_path=None):
        """All steps of an action tape in ONE launch (navsim_step_seq): actions [T, N, 2] float32; every output is [T, N, ...]
        and row t is what ``step(actions[t], ...)`` would have written.  For loops whose actions do not depend on the
        observations they produce (recorded tapes, scripted / random policies): the workgroups keep their envs on chip
        between the steps, so the launch ramp and the kernel boundaries of T launches are paid once."""
        T = int(actions.shape[0])
        if tuple(actions.shape) != (T, self.N, 2):
            raise NavsimError(f"step_seq: actions must be [T, {self.N}, 2], got {tuple(actions.shape)}")
        self._chk("step_seq: actions", actions, torch.float32, 2 * T * self.N)
        self._chk_step_io(T, obs, reward, done, arrive, ended, ep_return, ep_length, ep_path, "step_seq")
        with torch.cuda.device(self.device):
            check(lib().navsim_step_seq(self._h, _ptr(actions), T, _ptr(obs), _ptr(reward), _ptr(done), _ptr(arrive), _ptr(ended),
                                        _ptr(ep_return), _ptr(ep_length), _ptr(ep_path), _stream()), "navsim_step_seq")

    def raycast(self, pose):
        pose = pose.to(device=self.device, dtype=torch.float64).contiguous()
        out = torch.empty((self.N, self.B), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().navsim_raycast(self._h, _ptr(pose), _ptr(out), _stream()), "navsim_raycast")
        return out

    def get_state(self):
        N = self.N
        st = dict(pose=np.empty((N, 3)), goal=np.empty((N, 2)), past_dist=np.empty(N),
                  past_action=np.empty((N, 2), np.float32), ep_step=np.empty(N, np.int32), rng_ctr=np.empty(N, np.uint32))
        with torch.cuda.device(self.device):
            check(lib().navsim_get_state(self._h, *[_np_ptr(st[k]) for k in
                                                    ("pose", "goal", "past_dist", "past_action", "ep_step", "rng_ctr")],
                                         _stream()), "navsim_get_state")
        return st

    def set_state(self, pose=None, goal=None, past_dist=None, past_action=None, ep_step=None, rng_ctr=None):
        def cv(a, dt, shape):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dt)
            if a.shape != shape:
                raise NavsimError(f"set_state: expected shape {shape}, got {a.shape}")
            return a
        N = self.N
        arrs = [cv(pose, np.float64, (N, 3)), cv(goal, np.float64, (N, 2)), cv(past_dist, np.float64, (N,)),
                cv(past_action, np.float32, (N, 2)), cv(ep_step, np.int32, (N,)), cv(rng_ctr, np.uint32, (N,))]
        with torch.cuda.device(self.device):
            check(lib().navsim_set_state(self._h, *[_np_ptr(a) for a in arrs], _stream()), "navsim_set_state")


def rtg_scan(rew, ended, gamma, out=None, exact=False):
    """PPO.compute_rtgs (ppo.py:643-671) on [T,N] device tensors; see navsim_rtg_scan.  exact: the serial recurrence (every
    bit the reference's) instead of the T-split scan (<= 1 float32 ulp)."""
    if not rew.is_cuda:
        raise NavsimError("rtg_scan needs device tensors; there is no CPU path")
    T, N = rew.shape
    rew = rew.contiguous()
    ended = ended.contiguous()
    assert rew.dtype == torch.float32 and ended.dtype == torch.uint8 and ended.shape == rew.shape
    if out is None:
        out = torch.empty_like(rew)
    with torch.cuda.device(rew.device):
        check(lib().navsim_rtg_scan(_ptr(rew), _ptr(ended), T, N, float(gamma), _ptr(out), int(bool(exact)), _stream()), "navsim_rtg_scan")
    return out


def odometry(x, y, quat, goal):
    """Env.getOdometry (environment_new.py:138-181) for n samples on the device: x, y [n], quat [n,4] = (qx, qy, qz, qw),
    goal [n,2], all float64 -> [n,3] float64 (yaw, rel_theta, diff_angle); see navsim_odometry."""
    if not x.is_cuda:
        raise NavsimError("odometry needs device tensors; there is no CPU path")
    x, y, quat, goal = (t.to(torch.float64).contiguous() for t in (x, y, quat, goal))
    n = x.shape[0]
    assert y.shape == (n,) and quat.shape == (n, 4) and goal.shape == (n, 2)
    out = torch.empty((n, 3), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().navsim_odometry(n, _ptr(x), _ptr(y), _ptr(quat), _ptr(goal), _ptr(out), _stream()), "navsim_odometry")
    return out


def gae_scan(rew, ended, value, gamma, lam, last_value=None, want_returns=True, exact=False):
    """GAE(lambda) on [T,N] device tensors (navsim_gae_scan): returns (adv, lambda_returns).  lam = 1 without `last_value`
    is PPO.compute_rtgs followed by A = rtgs - V (ppo.py:277, 643-671), bit for bit."""
    if not rew.is_cuda:
        raise NavsimError("gae_scan needs device tensors; there is no CPU path")
    T, N = rew.shape
    rew, ended, value = rew.contiguous(), ended.contiguous(), value.contiguous()
    assert rew.dtype == torch.float32 and ended.dtype == torch.uint8 and value.dtype == torch.float32
    assert ended.shape == rew.shape == value.shape and (last_value is None or last_value.shape == (N,))
    if last_value is not None:
        last_value = last_value.to(torch.float32).contiguous()
    adv = torch.empty_like(rew)
    ret = torch.empty_like(rew) if want_returns else None
    with torch.cuda.device(rew.device):
        check(lib().navsim_gae_scan(_ptr(rew), _ptr(ended), _ptr(value), _ptr(last_value), T, N, float(gamma), float(lam),
                                    _ptr(adv), _ptr(ret), int(bool(exact)), _stream()), "navsim_gae_scan")
    return adv, ret


class VecEnv:
    """N environments stepped by one kernel launch; tensors stay on the GPU.

    ``step(action)`` mirrors ``Env.step`` per env and the episode bookkeeping of
    ``PPO.rollout`` (ppo.py:543-593): past_action tracking, timeout after
    ``max_episode_steps``, and -- with ``auto_reset`` -- the reset, in which case the returned
    observation is the post-reset one (what ``rollout`` stores next, ppo.py:508,593).
    """

    def __init__(self, n_envs, map="stage_1", n_beams=10, max_episode_steps=500, auto_reset=True, is_training=True,
                 seed=0, env_id_base=0, per_env_map=False, map_seed=0, obs_f16=False, device=None, sampler=None,
                 lidar_below_min="clamp", lidar_noise_sigma=0.0, respawn_on_arrive=False, envs_per_workgroup=None, pair_cast=None):
        thr = 0.2 if is_training else 0.4  # environment_new.py:44-47
        self.sim = NavSim(n_envs, n_beams=n_beams, max_episode_steps=max_episode_steps, auto_reset=auto_reset,
                          respawn_on_arrive=respawn_on_arrive, seed=seed, env_id_base=env_id_base, threshold_arrive=thr,
                          obs_f16=obs_f16, device=device, lidar_below_min=lidar_below_min,
                          lidar_noise_sigma=lidar_noise_sigma, envs_per_workgroup=envs_per_workgroup, pair_cast=pair_cast)
        self.N, self.B, self.D, self.device = self.sim.N, self.sim.B, self.sim.D, self.sim.device
        self.threshold_arrive = thr
        self.use_vision = False
        if isinstance(map, str):
            seg = _maps.by_name(map)
            reset_rects, respawn_rects = _maps.goal_rects(map)
            self.sim.set_goal_rects(0, reset_rects)
            self.sim.set_goal_rects(1, respawn_rects)
        else:
            seg = map
        if per_env_map and not (torch.is_tensor(seg) and seg.dim() == 3) and np.ndim(seg) == 2:
            seg = _maps.replicate_per_env(seg, self.N, seed=map_seed)
        self.sim.set_map(seg)
        if sampler is not None:  # e.g. "stage1" / "small_house" (spawn_goal_sampler.py) or a (starts, goals, min, max) tuple
            if isinstance(sampler, str):
                sampler = _maps.spawn_tables(sampler)
            self.sim.set_spawn_sampler(*sampler)
        self.io = self.sim.alloc_io()

    def close(self):
        self.sim.close()

    def reset(self, mask=None):
        return self.sim.reset(self.io.obs, mask)

    def step(self, action, io=None):
        """action: [N,2] float32 device tensor.  Returns (obs, reward, done, arrive); `ended`,
        `ep_return`, `ep_length` are in ``self.io`` (or the `io` passed in)."""
        io = io or self.io
        action = action.to(device=self.device, dtype=torch.float32).contiguous()
        self.sim.step(action, io.obs, io.reward, io.done, io.arrive, io.ended, io.ep_return, io.ep_length, ep_path=io.ep_path)
        return io.obs, io.reward, io.done, io.arrive


    def step_seq(self, actions):
        """A whole action tape in ONE launch (navsim_step_seq): actions [T, N, 2] float32 on the device.  Returns a namespace of
        [T, N, ...] tensors (obs after each step, reward, done, arrive, ended, ep_return, ep_length, ep_path); row t is what
        ``step(actions[t])`` would have produced.  For loops whose actions do not depend on the observations they produce."""
        actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        T, N, dev = int(actions.shape[0]), self.N, self.device
        out = types.SimpleNamespace(
            obs=torch.empty((T, N, self.sim.D), dtype=self.sim.obs_dtype, device=dev), reward=torch.empty((T, N), device=dev),
            done=torch.empty((T, N), dtype=torch.uint8, device=dev), arrive=torch.empty((T, N), dtype=torch.uint8, device=dev),
            ended=torch.empty((T, N), dtype=torch.uint8, device=dev), ep_return=torch.zeros((T, N), device=dev),
            ep_length=torch.zeros((T, N), dtype=torch.int32, device=dev), ep_path=torch.zeros((T, N), device=dev))
        self.sim.step_seq(actions, out.obs, out.reward, out.done, out.arrive, out.ended, out.ep_return, out.ep_length, out.ep_path)
        return out


    def rollout_mlp64(self, actor_params, n_steps, var, seed=0, step_base=0, obs0=None):
        """PPO.rollout's hot loop (ppo.py:505-594) for the (B + 6)-64-64 policy in ONE launch (navsim_rollout_mlp64): per step
        PPO.get_action (ppo.py:673-706) on the observation the previous step left on chip, then the env step.
        actor_params: flat float32 device tensor [5378] (10 beams) / [7042] (36 beams) in nn.Module.named_parameters order
        (include/navppo.h); var: exploration variance (float or device scalar); action noise = Philox(seed, global env id,
        step_base + t).  obs0 [N, B + 6]: the observations to start from (default: a fresh reset).  Returns a namespace: obs
        [T + 1, N, B + 6] (row 0 = obs0; float16 on an obs_f16 env), act [T, N, 2], logp / reward / done / arrive / ended /
        ep_return / ep_length / ep_path [T, N] -- bit-identical to T pairs of navppo_mlp64_act / step calls."""
        import ctypes as C
        from ._native import check, lib
        if self.B not in (10, 36):
            raise NavsimError("rollout_mlp64 needs 10 or 36 beams")
        T, N, D, dev = int(n_steps), self.N, self.D, self.device
        n_actor = 64 * D + 64 + 64 * 64 + 64 + 2 * (64 + 1)
        prm = actor_params.to(device=dev, dtype=torch.float32).contiguous()
        if prm.numel() != n_actor or prm.data_ptr() % 16:
            raise NavsimError(f"actor_params: {n_actor} float32 values, 16-byte aligned")
        out = types.SimpleNamespace(
            obs=torch.empty((T + 1, N, D), dtype=self.sim.obs_dtype, device=dev), act=torch.empty((T, N, 2), device=dev),
            logp=torch.empty((T, N), device=dev),
            reward=torch.empty((T, N), device=dev), done=torch.empty((T, N), dtype=torch.uint8, device=dev),
            arrive=torch.empty((T, N), dtype=torch.uint8, device=dev), ended=torch.empty((T, N), dtype=torch.uint8, device=dev),
            ep_return=torch.zeros((T, N), device=dev), ep_length=torch.zeros((T, N), dtype=torch.int32, device=dev),
            ep_path=torch.zeros((T, N), device=dev))
        if obs0 is None:
            self.sim.reset(out.obs[0])
        else:
            out.obs[0].copy_(obs0)
        var_t = var if torch.is_tensor(var) else torch.tensor(float(var), device=dev)
        var_t = var_t.to(device=dev, dtype=torch.float32).reshape(())
        base_t = torch.tensor(int(step_base), dtype=torch.int32, device=dev)
        p = lambda x: C.c_void_p(x.data_ptr())
        with torch.cuda.device(dev):
            check(lib().navsim_rollout_mlp64(self.sim._h, p(prm), p(out.obs), p(out.act), p(out.logp), p(out.reward), p(out.done),
                                             p(out.arrive), p(out.ended), p(out.ep_return), p(out.ep_length), p(out.ep_path), p(var_t),
                                             int(seed) & 0xFFFFFFFFFFFFFFFF, p(base_t), T,
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "navsim_rollout_mlp64")
            torch.cuda.current_stream().synchronize()   # var_t / base_t / prm may be temporaries of this call
        return out


class _XY:
    __slots__ = ("x", "y", "z")

    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = x, y, z


class _Pose:
    def __init__(self):
        self.position = _XY()


class Env:
    """Single-env drop-in for the reference ``Env`` (environment_new.py:26-382).

    >>> env = Env(is_training=True)
    >>> obs = env.reset()                              # np.ndarray (16,) float64
    >>> obs, reward, done, arrive = env.step(action, past_action)

    Differences from the reference, all deliberate: the world is simulated on the GPU instead of
    Gazebo (so ``step`` does not block on a 5 Hz scan), goal sampling uses a seeded counter-based
    RNG instead of the unseeded global ``random``, and service failures raise instead of being
    swallowed.  The episode loop stays with the caller exactly as in ``PPO.rollout``: ``step`` never
    resets, and on arrival it re-spawns the goal like ``setReward`` does (environment_new.py:245-267).
    """

    def __init__(self, is_training, use_vision=False, vision_dim=64, map="stage_1", seed=0, device=None):
        if use_vision:
            raise NotImplementedError("camera modality is outside the LiDAR hot path (SURVEY.md 2a)")
        self.use_vision = False
        self.vision_dim = vision_dim
        self.threshold_arrive = 0.2 if is_training else 0.4
        self._sim = NavSim(1, n_beams=10, max_episode_steps=0, auto_reset=False, respawn_on_arrive=True, seed=seed,
                           threshold_arrive=self.threshold_arrive, device=device)
        if isinstance(map, str):
            seg = _maps.by_name(map)
            rr, rs = _maps.goal_rects(map)
            self._sim.set_goal_rects(0, rr)
            self._sim.set_goal_rects(1, rs)
        else:
            seg = map
        self._sim.set_map(seg)
        # One pinned (host-mapped, device-visible) block carries a step's inputs and outputs: the kernel reads the action pair
        # from it and stores observation, reward and flags into it, so a step is one launch + a poll of the block (_wait_results), no copies.
        self._pin = torch.zeros(48, dtype=torch.float32).pin_memory()
        self._pin_u8 = torch.zeros(16, dtype=torch.uint8).pin_memory()
        self._pin_np, self._pin_u8_np = self._pin.numpy(), self._pin_u8.numpy()
        self._pin_i32 = self._pin_np.view(np.int32)   # the same block as bit patterns (the "not delivered yet" marker below)
        self._act_t, self._past_t = self._pin[0:2].view(1, 2), self._pin[2:4].view(1, 2)
        self._obs_t, self._rew_t = self._pin[16:32].view(1, 16), self._pin[32:33]
        self._done_t, self._arrive_t, self._ended_t = self._pin_u8[0:1], self._pin_u8[1:2], self._pin_u8[2:3]
        self._state = None   # host copy of pose / goal / past_distance, fetched when an attribute is read
        # One env step from Python is launch-latency bound (kernel 6 us, the rest is the way there and back): the argument list of
        # navsim_step / navsim_reset is built ONCE -- ctypes pointers of the pinned block, the stream the env was created on -- so
        # that a step is one foreign call and one wait (building twelve c_void_p objects, looking up the current stream and
        # entering a device context per step were 5 of the 25 us).  The stream is therefore FROZEN at construction: an Env built
        # under torch.cuda.stream(s) launches on s for its whole life (as do its attribute reads, which wait on that stream).
        dev = self._sim.device
        self._dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        self._stream_obj = torch.cuda.current_stream(dev)
        st = C.c_void_p(self._stream_obj.cuda_stream)
        self._lib = lib()
        self._step_args = (self._sim._h, _ptr(self._act_t), _ptr(self._past_t), _ptr(self._obs_t), _ptr(self._rew_t), _ptr(self._done_t),
                           _ptr(self._arrive_t), _ptr(self._ended_t), None, None, None, st)
        self._reset_args = (self._sim._h, None, _ptr(self._obs_t), st)

    # -- attributes the reference's callers read (ppo.py:535, main.py:202; environment_new.py:29-41): fetched lazily, one
    #    navsim_get_state per step at most, and none at all for callers that never look
    def _st(self):
        if self._state is None:
            with torch.cuda.stream(self._stream_obj):   # the stream the steps were launched on (frozen at construction)
                self._state = self._sim.get_state()
        return self._state

    @property
    def position(self):
        p = self._st()["pose"][0]
        return _XY(float(p[0]), float(p[1]))

    @property
    def yaw_rad(self):
        return float(self._st()["pose"][0, 2])

    @property
    def goal_position(self):
        g = self._st()["goal"][0]
        pose = _Pose()
        pose.position.x, pose.position.y = float(g[0]), float(g[1])
        return pose

    @property
    def past_distance(self):
        return float(self._st()["past_dist"][0])

    def _wait(self):
        self._stream_obj.synchronize()

    # A step's results are waited for where they land.  The pinned block is host-coherent memory: the kernel's stores reach it while
    # the kernel runs, so the caller plants a bit pattern the kernel never writes in every output slot -- _PLANT, a quiet NaN with a
    # payload no arithmetic produces (a diverged policy's NaN observation or reward is the canonical 0x7fc00000 / a propagated
    # input payload, and is returned to the caller like the reference's Env.step returns it), 0xFF in the flags -- and polls until
    # all of them are gone, each slot checked for itself, no assumption on the order the stores arrive in.
    # Measured (tools/time_env_n1_parts.py): launch + stream.synchronize() 23.5 us, launch + this 14.1 us.
    # A step that has not delivered after _POLLS polls (a stalled queue, a fault) falls back to the stream wait, which reports it.
    # The poll returns when the RESULTS have landed, which can be before the kernel has retired: later readers of the handle's
    # device state are ordered behind it by the stream (get_state and every other call run on the stream the Env was built on).
    _POLLS = 1 << 20
    _PLANT = np.int32(0x7FC0DEAD)

    def _plant(self):
        self._pin_i32[16:33] = self._PLANT
        self._pin_u8_np[0:3] = 255

    def _wait_results(self, obs_only=False):
        pin, u8, plant = self._pin_i32, self._pin_u8_np, self._PLANT
        for _ in range(self._POLLS):
            if pin[31] != plant and (obs_only or (pin[32] != plant and u8[0] != 255 and u8[1] != 255 and u8[2] != 255)) \
                    and not (pin[16:32] == plant).any():
                return
        self._wait()
        if (pin[16:32] == plant).any() or (not obs_only and (pin[32] == plant or (u8[0:3] == 255).any())):
            raise RuntimeError("the launch finished without delivering its results to the pinned block")

    def _call(self, fn, args, what):
        if args is None:
            raise NavsimError(f"{what}: the env is closed")
        if torch.cuda.current_device() == self._dev_index:
            check(fn(*args), what)
        else:
            with torch.cuda.device(self._dev_index):
                check(fn(*args), what)

    def reset(self):
        self._pin_i32[16:32] = self._PLANT
        self._call(self._lib.navsim_reset, self._reset_args, "navsim_reset")
        self._wait_results(obs_only=True)
        self._state = None
        return self._pin_np[16:32].astype(np.float64)

    def step(self, action, past_action):
        a = np.asarray(action, dtype=np.float32).reshape(-1)
        p = np.asarray(past_action, dtype=np.float32).reshape(-1)
        if a.shape[0] < 2 or p.shape[0] < 2:
            raise IndexError("action and past_action need two components")  # as action[1] would in the reference
        self._pin_np[0:2] = a[:2]
        self._pin_np[2:4] = p[:2]
        self._plant()
        self._call(self._lib.navsim_step, self._step_args, "navsim_step")
        self._wait_results()
        self._state = None
        return (self._pin_np[16:32].astype(np.float64), float(self._pin_np[32]), bool(self._pin_u8_np[0]),
                bool(self._pin_u8_np[1]))

    def getLatestImage(self):
        return None

    def close(self):
        self._step_args = self._reset_args = None   # they hold the native handle: a step after close() raises instead of using it
        self._sim.close()
