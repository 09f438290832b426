"""CPU-only checks of the drop-in boundary: libnavsim.so loads without a GPU and exports exactly the entry points the
headers declare; the ctypes table in navbot_ppo_amd/_native.py covers all of them; the product has no CPU fallback."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(REPO, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nav(?:sim|ppo)_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from navbot_ppo_amd import _native
    assert os.path.exists(_native.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_native.LIB_PATH)
    declared = _declared("navsim.h") + _declared("navppo.h")
    assert len(declared) >= 17
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    bound = {n for n, _, _ in _native.SYMBOLS}
    assert set(declared) == bound, (set(declared) ^ bound)
    assert lib.navsim_version() == 5


def test_default_cfg_matches_reference_constants():
    from navbot_ppo_amd import _native
    L = _native.lib()
    cfg = _native.NavsimCfg()
    L.navsim_default_cfg(ctypes.byref(cfg))
    assert (cfg.n_envs, cfg.n_beams, cfg.threshold_arrive, cfg.goal_lo, cfg.goal_hi) == (1, 10, 0.2, -3.6, 3.6)
    assert (cfg.spawn_x, cfg.spawn_y, cfg.spawn_yaw) == (0.0, 0.0, 0.0)


def test_no_cpu_fallback_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from navbot_ppo_amd import _native
    from navbot_ppo_amd.env import Env, NavSim, NavsimError, VecEnv, rtg_scan
    for ctor in (lambda: NavSim(4), lambda: VecEnv(4), lambda: Env(True)):
        with pytest.raises(NavsimError):
            ctor()
    with pytest.raises(NavsimError):
        rtg_scan(torch.zeros((4, 4)), torch.zeros((4, 4), dtype=torch.uint8), 0.99)
    # the C ABI itself reports the missing device instead of computing on the host
    L = _native.lib()
    cfg = _native.NavsimCfg()
    L.navsim_default_cfg(ctypes.byref(cfg))
    h = ctypes.c_void_p()
    assert L.navsim_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"HIP" in L.navsim_last_error() or b"hip" in L.navsim_last_error() or b"device" in L.navsim_last_error()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under navbot_ppo_amd/ (or bench.py's timed path) may import or link it."""
    pkg = os.path.join(REPO, "navbot_ppo_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            txt = open(os.path.join(root, f), errors="ignore").read() if f.endswith((".py", ".hip", ".h")) else ""
            assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), (root, f)
            assert not re.search(r"#include.*oracle|libnavsim_oracle", txt), (root, f)


def test_library_reads_no_shape_knobs_from_the_environment():
    """The workgroup-shape overrides are per-handle state (navsim_set_shape, reported by navsim_get_info): nothing in
    navsim_create -- or anywhere else in the library -- may read NAVSIM_EPB / NAVSIM_PAIR_CAST from the process environment
    (a later handle's environment used to change the kernel shape of every earlier handle)."""
    src = open(os.path.join(REPO, "navbot_ppo_amd", "csrc", "navsim.hip")).read()
    code = re.sub(r"//[^\n]*", "", src)
    assert "NAVSIM_EPB" not in code and "NAVSIM_PAIR_CAST" not in code
    assert not re.search(r"\bg_epb\b|\bg_pair_cast\b", code)
    from navbot_ppo_amd import _native
    L = _native.lib()
    # argument checks run without a device
    assert L.navsim_set_shape(None, 16, -1) != 0
    assert L.navsim_get_info(None, None) != 0
