"""ctypes binding of oracle/libnavsim_oracle.so (see navsim_oracle.c header).

TEST INFRASTRUCTURE ONLY: the checker for the HIP path, never the thing that is
shipped or measured (except as bench.py's reported `cpu_baseline`).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnavsim_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "navsim_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libnavsim_oracle.so"] + (["-B"] if force else []))
    return _SO


class OrcCfg(C.Structure):
    _fields_ = [
        ("n_envs", C.c_int32),
        ("n_beams", C.c_int32),
        ("max_episode_steps", C.c_int32),
        ("auto_reset", C.c_int32),
        ("respawn_on_arrive", C.c_int32),
        ("obs_f16", C.c_int32),
        ("lidar_below_min", C.c_int32),
        ("lidar_noise_sigma", C.c_float),
        ("seed", C.c_uint64),
        ("env_id_base", C.c_uint64),
        ("threshold_arrive", C.c_double),
        ("spawn_x", C.c_double),
        ("spawn_y", C.c_double),
        ("spawn_yaw", C.c_double),
        ("goal_lo", C.c_double),
        ("goal_hi", C.c_double),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        d, i32, vp = C.c_double, C.c_int32, C.c_void_p
        L.orc_py_round_nd.restype = d
        L.orc_py_round_nd.argtypes = [d, C.c_int]
        L.orc_get_odometry.restype = None
        L.orc_get_odometry.argtypes = [d] * 8 + [C.POINTER(i32), C.POINTER(d), C.POINTER(d)]
        L.orc_get_state.restype = None
        L.orc_get_state.argtypes = [vp, C.c_int, d, d, d, d, d, vp, C.POINTER(d), C.POINTER(i32), C.POINTER(i32)]
        L.orc_assemble_obs.restype = None
        L.orc_assemble_obs.argtypes = [vp, C.c_int, C.c_int, vp, d, d, d, d, vp]
        L.orc_set_reward.restype = d
        L.orc_set_reward.argtypes = [C.POINTER(d), d, C.c_int, C.c_int]
        L.orc_goal_rejected.restype = C.c_int
        L.orc_goal_rejected.argtypes = [C.c_int, d, d]
        L.orc_compute_rtgs_ragged.restype = None
        L.orc_compute_rtgs_ragged.argtypes = [vp, vp, C.c_int, d, vp]
        L.orc_compute_rtgs_tn.restype = None
        L.orc_compute_rtgs_tn.argtypes = [vp, vp, C.c_int, C.c_int, d, vp]
        L.orc_philox4x32_10.restype = None
        L.orc_philox4x32_10.argtypes = [vp, vp, vp]
        L.orc_raycast.restype = None
        L.orc_raycast.argtypes = [vp, C.c_int, d, d, d, vp, vp, C.c_int, vp]
        L.orc_sim_create.restype = vp
        L.orc_sim_create.argtypes = [C.POINTER(OrcCfg)]
        L.orc_sim_destroy.restype = None
        L.orc_sim_destroy.argtypes = [vp]
        L.orc_sim_set_map.restype = C.c_int
        L.orc_sim_set_map.argtypes = [vp, vp, C.c_int, C.c_int]
        L.orc_sim_set_goal_rects.restype = C.c_int
        L.orc_sim_set_goal_rects.argtypes = [vp, C.c_int, vp, C.c_int]
        L.orc_sim_set_spawn_sampler.restype = C.c_int
        L.orc_sim_set_spawn_sampler.argtypes = [vp, vp, C.c_int, vp, C.c_int, d, d]
        L.orc_sim_get_state.restype = None
        L.orc_sim_get_state.argtypes = [vp] * 7
        L.orc_sim_set_state.restype = None
        L.orc_sim_set_state.argtypes = [vp] * 7
        L.orc_sim_reset.restype = None
        L.orc_sim_reset.argtypes = [vp, vp, vp]
        L.orc_sim_step.restype = None
        L.orc_sim_step.argtypes = [vp] * 11
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------- rules (A)
def py_round_nd(x, nd):
    return lib().orc_py_round_nd(float(x), int(nd))


def get_odometry(px, py, qx, qy, qz, qw, gx, gy):
    yaw, rt, da = C.c_int32(), C.c_double(), C.c_double()
    lib().orc_get_odometry(px, py, qx, qy, qz, qw, gx, gy, C.byref(yaw), C.byref(rt), C.byref(da))
    return yaw.value, rt.value, da.value


def get_state(ranges, px, py, gx, gy, threshold):
    r = np.ascontiguousarray(ranges, dtype=np.float64)
    scan = np.empty_like(r)
    dist, done, arrive = C.c_double(), C.c_int32(), C.c_int32()
    lib().orc_get_state(_p(r), len(r), px, py, gx, gy, threshold, _p(scan), C.byref(dist), C.byref(done), C.byref(arrive))
    return scan, dist.value, bool(done.value), bool(arrive.value)


def assemble_obs(scan, past_action, dist, yaw, rel_theta, diff_angle, n_feat=10):
    s = np.ascontiguousarray(scan, dtype=np.float64)
    pa = np.ascontiguousarray(past_action, dtype=np.float64)
    obs = np.empty(n_feat + 6, dtype=np.float64)
    lib().orc_assemble_obs(_p(s), len(s), n_feat, _p(pa), dist, float(yaw), rel_theta, diff_angle, _p(obs))
    return obs


def set_reward(past_distance, current_distance, done, arrive):
    pd = C.c_double(past_distance)
    r = lib().orc_set_reward(C.byref(pd), current_distance, int(done), int(arrive))
    return r, pd.value


def goal_rejected(which, x, y):
    return bool(lib().orc_goal_rejected(int(which), float(x), float(y)))


def compute_rtgs_ragged(batch_rews, gamma):
    lens = np.array([len(e) for e in batch_rews], dtype=np.int32)
    flat = np.array([r for e in batch_rews for r in e], dtype=np.float64)
    out = np.empty(len(flat), dtype=np.float32)
    lib().orc_compute_rtgs_ragged(_p(flat), _p(lens), len(lens), gamma, _p(out))
    return out


def compute_rtgs_tn(rew, ended, gamma):
    rew = np.ascontiguousarray(rew, dtype=np.float32)
    ended = np.ascontiguousarray(ended, dtype=np.uint8)
    T, N = rew.shape
    out = np.empty((T, N), dtype=np.float32)
    lib().orc_compute_rtgs_tn(_p(rew), _p(ended), T, N, gamma, _p(out))
    return out


def philox4x32_10(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.empty(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_p(c), _p(k), _p(o))
    return o


def beam_tables(n_beams):
    a0, a1 = -1.5707975, 1.5707975
    phi = np.array([a0 + i * ((a1 - a0) / (n_beams - 1)) for i in range(n_beams)]) if n_beams > 1 else np.zeros(1)
    return np.cos(phi), np.sin(phi)


def raycast(seg, x, y, th, n_beams=10):
    seg = np.ascontiguousarray(seg, dtype=np.float32).reshape(-1, 4)
    bc, bs = beam_tables(n_beams)
    out = np.empty(n_beams, dtype=np.float32)
    lib().orc_raycast(_p(seg), seg.shape[0], x, y, th, _p(bc), _p(bs), n_beams, _p(out))
    return out


# ---------------------------------------------------------------- sim (B)
class OracleSim:
    """Batched CPU simulator with the same call surface as the C-ABI (navsim.h)."""

    def __init__(self, n_envs, n_beams=10, max_episode_steps=0, auto_reset=False, respawn_on_arrive=False,
                 seed=0, env_id_base=0, threshold_arrive=0.2, spawn=(0.0, 0.0, 0.0), goal_box=(-3.6, 3.6),
                 lidar_below_min="clamp", lidar_noise_sigma=0.0):
        self.cfg = OrcCfg(n_envs, n_beams, max_episode_steps, int(auto_reset), int(respawn_on_arrive), 0,
                          {"clamp": 0, "gazebo": 1}[lidar_below_min], float(lidar_noise_sigma), seed, env_id_base, threshold_arrive, spawn[0], spawn[1], spawn[2], goal_box[0], goal_box[1])
        self.N, self.B, self.D = n_envs, n_beams, n_beams + 6
        self._h = lib().orc_sim_create(C.byref(self.cfg))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_sim_destroy(self._h)
            self._h = None

    def set_map(self, seg, per_env=False):
        seg = np.ascontiguousarray(seg, dtype=np.float32)
        S = seg.shape[-2]
        assert seg.shape[-1] == 4 and (seg.ndim == 3) == bool(per_env)
        lib().orc_sim_set_map(self._h, _p(seg), S, int(per_env))

    def set_goal_rects(self, which, rects):
        r = np.ascontiguousarray(rects, dtype=np.float64).reshape(-1, 4)
        assert lib().orc_sim_set_goal_rects(self._h, which, _p(r), r.shape[0]) == 0

    def set_spawn_sampler(self, starts, goals=None, min_dist=1.5, max_dist=6.0):
        st = np.ascontiguousarray(starts, dtype=np.float64).reshape(-1, 3)
        g = None if goals is None else np.ascontiguousarray(goals, dtype=np.float64).reshape(-1, 2)
        assert lib().orc_sim_set_spawn_sampler(self._h, _p(st), st.shape[0], _p(g), 0 if g is None else g.shape[0],
                                               float(min_dist), float(max_dist)) == 0

    def get_state(self):
        N = self.N
        st = dict(pose=np.empty((N, 3)), goal=np.empty((N, 2)), past_dist=np.empty(N),
                  past_action=np.empty((N, 2), np.float32), ep_step=np.empty(N, np.int32),
                  rng_ctr=np.empty(N, np.uint32))
        lib().orc_sim_get_state(self._h, *[_p(st[k]) for k in ("pose", "goal", "past_dist", "past_action", "ep_step", "rng_ctr")])
        return st

    def set_state(self, pose=None, goal=None, past_dist=None, past_action=None, ep_step=None, rng_ctr=None):
        def cv(a, dt):
            return None if a is None else np.ascontiguousarray(a, dtype=dt)
        a = [cv(pose, np.float64), cv(goal, np.float64), cv(past_dist, np.float64), cv(past_action, np.float32),
             cv(ep_step, np.int32), cv(rng_ctr, np.uint32)]
        lib().orc_sim_set_state(self._h, *[_p(x) for x in a])

    def reset(self, mask=None, obs=None):
        if obs is None:
            obs = np.zeros((self.N, self.D), dtype=np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().orc_sim_reset(self._h, _p(m), _p(obs))
        return obs

    def step(self, action, past_action=None):
        a = np.ascontiguousarray(action, dtype=np.float32).reshape(self.N, 2)
        pa = None if past_action is None else np.ascontiguousarray(past_action, dtype=np.float32).reshape(self.N, 2)
        obs = np.zeros((self.N, self.D), dtype=np.float32)
        rew = np.zeros(self.N, np.float32)
        done = np.zeros(self.N, np.uint8)
        arrive = np.zeros(self.N, np.uint8)
        ended = np.zeros(self.N, np.uint8)
        ep_ret = np.zeros(self.N, np.float32)
        ep_len = np.zeros(self.N, np.int32)
        ep_path = np.zeros(self.N, np.float32)
        lib().orc_sim_step(self._h, _p(a), _p(pa), _p(obs), _p(rew), _p(done), _p(arrive), _p(ended), _p(ep_ret), _p(ep_len),
                           _p(ep_path))
        return dict(obs=obs, reward=rew, done=done, arrive=arrive, ended=ended, ep_return=ep_ret, ep_length=ep_len,
                    ep_path=ep_path)
