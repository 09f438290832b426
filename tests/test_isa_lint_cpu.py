"""The hand-placed instruction stream of csrc/ppo_mlp64_x3s.h (every MFMA an `asm volatile` statement) is outside the compiler's
hazard recogniser: it may place a register copy right in front of an asm MFMA that reads it (gfx950 does not interlock that pair:
tools/ubench/valu_mfma_hazard.hip), or read an MFMA result too early.  This test compiles the file to a listing with the product's
flags and checks the listing of every kernel that carries such statements (tools/verify/mfma_hazard_lint.py) -- so a different hipcc,
or an edit that moves a copy, fails HERE and not as a wrong gradient on the GPU.  No GPU needed (hipcc cross-compiles)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools", "verify"))


def test_asm_mfma_streams_have_no_unguarded_hazards(tmp_path):
    from mfma_hazard_lint import lint
    from navbot_ppo_amd import build
    out = tmp_path / "ppo_mlp64.s"
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call([build.hipcc()] + flags + ["-I", build.INC, "-I", os.path.join(build.HERE, "csrc"), "-S", "--cuda-device-only",
                                                    os.path.join(build.HERE, "csrc", "ppo_mlp64.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    n, bad = lint(str(out), "mlp64_pass_both_x3sE")
    assert n >= 400, n            # both nets' streams are in the listing (168 + 36 MFMAs per tile and net, + the tails)
    assert not bad, bad[:5]
    # the checker itself: the compiler-scheduled kernels of the same file (real MFMA instructions, hazards handled by LLVM) are clean too
    n2, bad2 = lint(str(out), "mlp64_pass_both_x3ILi42")
    assert n2 > 100 and not bad2, bad2[:5]


def test_asm_mfma_stream_of_the_512_wide_backward_has_no_unguarded_hazards(tmp_path):
    """csrc/ppo_resmlp512_bwd2s.h: the same kind of stream for rb2's backward (both instantiations: float32 / float16 observation rows).
    Compiled with the product's per-source flags -- the register-allocation flag of build.EXTRA_FLAGS is what keeps a copy of a pinned
    accumulator tile from landing right behind the MFMA that writes it."""
    from mfma_hazard_lint import lint
    from navbot_ppo_amd import build
    out = tmp_path / "ppo_resmlp512.s"
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    extra = build.EXTRA_FLAGS.get("ppo_resmlp512.hip", [])
    if extra and not build.flags_accepted(extra):
        import pytest
        pytest.skip("hipcc rejects the allocation flag: build.FALLBACK_FLAGS builds the file without resmlp_bwd2s (-DRESMLP_BWD2S=0)")
    subprocess.check_call([build.hipcc()] + flags + list(extra) + ["-I", build.INC, "-I", os.path.join(build.HERE, "csrc"), "-S", "--cuda-device-only",
                                                                  os.path.join(build.HERE, "csrc", "ppo_resmlp512.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    for key in ("resmlp_bwd2sILb0", "resmlp_bwd2sILb1"):
        n, bad = lint(str(out), key)
        assert n >= 450, (key, n)      # 24 (prologue) + 432 per tile
        assert not bad, (key, bad[:5])


def test_lint_flags_the_measured_hazards(tmp_path):
    from mfma_hazard_lint import lint
    src = tmp_path / "k.s"
    src.write_text("\n".join([
        "_Zk:",
        "\tv_mov_b32_e32 v7, v9",
        "\tv_mfma_f32_32x32x16_bf16 v[16:31], a[0:3], v[4:7], v[16:31]",      # srcB written one instruction earlier
        "\tv_accvgpr_write_b32 a40, v1",
        "\tv_mfma_f32_16x16x32_bf16 a[40:43], a[0:3], a[4:7], a[40:43]",      # srcC written one instruction earlier
        "\ts_nop 3",
        "\tv_add_f32_e32 v1, v1, v16",                                         # result of the first MFMA read after 5 wait states
        "\tv_mov_b32_e32 v5, v9",
        "\ts_nop 0",
        "\tv_mfma_f32_32x32x16_bf16 v[16:31], a[0:3], v[4:7], v[16:31]",      # fine: one wait state between
        "\t.amdhsa_kernel _Zk",
    ]))
    n, bad = lint(str(src), "_Zk")
    assert n == 3 and len(bad) == 3, bad


def test_resmlp512_falls_back_to_the_compiler_scheduled_backward_without_its_allocation_flag(monkeypatch):
    """build.EXTRA_FLAGS gives ppo_resmlp512.hip the register-allocation flag under which resmlp_bwd2s's listing is hazard-free; a hipcc
    that rejects it (or NAVSIM_NO_EXTRA_FLAGS=1) must not build that stream at all: build.FALLBACK_FLAGS then selects resmlp_bwd<32, 2, 4>."""
    from navbot_ppo_amd import build
    monkeypatch.setattr(build, "flags_accepted", lambda flags: False)
    assert build.per_source_flags("ppo_resmlp512.hip") == ["-DRESMLP_BWD2S=0"]
    assert build.per_source_flags("navsim.hip") == []          # a flag that only buys speed: built without it
    assert build.per_source_flags("ppo_mlp64.hip") == []
    monkeypatch.setattr(build, "flags_accepted", lambda flags: True)
    assert build.per_source_flags("ppo_resmlp512.hip") == build.EXTRA_FLAGS["ppo_resmlp512.hip"]
    monkeypatch.setenv("NAVSIM_NO_EXTRA_FLAGS", "1")
    assert build.per_source_flags("ppo_resmlp512.hip") == ["-DRESMLP_BWD2S=0"]
