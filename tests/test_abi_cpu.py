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
    assert lib.navsim_version() == _native.NAVSIM_ABI_VERSION == 6


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
    # ... and nothing else either: no translation unit of the library calls getenv (NAVSIM_RTG_EXACT became the `exact` argument
    # of navsim_rtg_scan / navsim_gae_scan in ABI v6), and the shared object does not import the symbol
    for f in os.listdir(os.path.join(REPO, "navbot_ppo_amd", "csrc")):
        txt = re.sub(r"//[^\n]*", "", open(os.path.join(REPO, "navbot_ppo_amd", "csrc", f)).read())
        assert "getenv" not in txt, f
    import subprocess
    from navbot_ppo_amd import _native as _n
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", _n.LIB_PATH], text=True)
    assert "getenv" not in undefined
    from navbot_ppo_amd import _native
    L = _native.lib()
    # argument checks run without a device
    assert L.navsim_set_shape(None, 16, -1) != 0
    assert L.navsim_get_info(None, None) != 0


def test_ctypes_structs_match_the_c_headers(tmp_path):
    """navsim_cfg / navsim_info as the C compiler lays them out (gcc on include/navsim.h) against the ctypes mirrors of
    navbot_ppo_amd/_native.py: size and every field offset."""
    import subprocess
    from navbot_ppo_amd import _native
    fields = {"navsim_cfg": [n for n, _ in _native.NavsimCfg._fields_], "navsim_info": [n for n, _ in _native.NavsimInfo._fields_]}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "navsim.h"', '#include "navppo.h"', 'int main(void) {']
    for st, fs in fields.items():
        src.append(f'  printf("{st} %zu\\n", sizeof({st}));')
        for f in fs:
            src.append(f'  printf("{st}.{f} %zu\\n", offsetof({st}, {f}));')
    src += ['  printf("actor16 %d critic16 %d actor42 %d critic42 %d\\n", NAVPPO_MLP64_ACTOR_PARAMS, NAVPPO_MLP64_CRITIC_PARAMS,',
            '         NAVPPO_MLP64_ACTOR_PARAMS_D(42), NAVPPO_MLP64_CRITIC_PARAMS_D(42));', '  return 0; }']
    c = tmp_path / "abi.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), str(c), "-o", str(exe)])
    got = dict(line.rsplit(" ", 1) for line in subprocess.check_output([str(exe)], text=True).splitlines() if not line.startswith("actor16"))
    for st, cls in (("navsim_cfg", _native.NavsimCfg), ("navsim_info", _native.NavsimInfo)):
        assert int(got[st]) == ctypes.sizeof(cls), st
        for f in fields[st]:
            assert int(got[f"{st}.{f}"]) == getattr(cls, f).offset, (st, f)
    tail = subprocess.check_output([str(exe)], text=True).splitlines()[-1]
    assert tail == "actor16 5378 critic16 5313 actor42 7042 critic42 6977"
    # the flat parameter counts the Python side derives from the net shapes are the header's
    import torch
    from navbot_ppo_amd import nets
    for d, (pa, pc) in ((16, (5378, 5313)), (42, (7042, 6977))):
        a, c_ = nets.make_policy("mlp64x2", d)
        assert (sum(p.numel() for p in a.parameters()), sum(p.numel() for p in c_.parameters())) == (pa, pc)


def test_update_arith_is_validated_on_the_cpu_too():
    import torch
    from navbot_ppo_amd import nets, ppo
    a, c = nets.make_policy("mlp64x2")
    with pytest.raises(ValueError):
        ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="mlp64x2", update_arith="bf16"), None, torch.device("cpu"))
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="mlp64x2"), None, torch.device("cpu"))
    assert up.fused is None and up.bf16x3 is False and up.obs_dim == 16   # no HIP device: PyTorch formulation, no split pass
