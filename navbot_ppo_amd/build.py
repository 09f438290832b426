"""Builds navbot_ppo_amd/libnavsim.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRCS = [os.path.join(HERE, "csrc", f) for f in ("navsim.hip", "ppo_mlp64.hip", "ppo_resmlp512.hip")]
HDRS = ["navsim.h", "navppo.h"]
INC = os.path.join(REPO, "include")
LIB = os.path.join(HERE, "libnavsim.so")

# -ffp-contract=off: the arithmetic contract writes every fused multiply-add explicitly
# -fno-slp-vectorize: packed f32 ops (v_pk_fma_f32 ...) issue at half rate on gfx950, so SLP-packing the ray tests buys
# nothing and costs register-pairing moves (measured: step kernel 28.8 -> 27.1 us)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
               "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libnavsim.so cannot be built (there is no CPU path)")
    return exe


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = SRCS + [os.path.join(INC, h) for h in HDRS] + [os.path.join(HERE, "csrc", h) for h in ("mlp64_policy.h", "navppo_internal.h", "resmlp_policy.h", "bf16x3.h", "ppo_mlp64_x3s.h", "ppo_resmlp512_bwd2s.h")]
    return any(os.path.getmtime(p) > t for p in deps)


# per-source flags.  navsim.hip: its persistent kernels (rollout_kernel, steps_kernel) inline the whole step body into a loop over
# steps, and MachineLICM then hoists every 64-bit constant the body materialises (the float64 polynomial coefficients of sincos,
# atan, ...) out of that loop and keeps them in registers across it: 200-256 VGPRs and scratch spills in the 16-wave shape,
# against 82-101 VGPRs (two workgroups per CU) with the pass off.  The single-step kernels have no such loop and do not change.
# ppo_resmlp512.hip: resmlp_bwd2s pins its 32 accumulator tiles to a[0:127] with physical-register asm constraints; with the default
# priority the greedy allocator still splits one of them through VGPRs right behind an MFMA (a hazard the compiler cannot see inside the
# asm: tools/verify/mfma_hazard_lint.py, tests/test_isa_lint_cpu.py) and copies 8 more registers per tile; with register-class priority
# first it does neither.  (The other kernels of the file: same registers, no spills either way.)
EXTRA_FLAGS = {"navsim.hip": ["-mllvm", "-disable-machine-licm"],
               "ppo_resmlp512.hip": ["-mllvm", "-greedy-regclass-priority-trumps-globalness=1"]}
# what a source is built with INSTEAD when hipcc rejects (or NAVSIM_NO_EXTRA_FLAGS=1 drops) its EXTRA_FLAGS: a flag that only buys speed
# has no entry; resmlp_bwd2s without its allocation flag carries a hazard the compiler cannot see, so the file then launches the
# compiler-scheduled resmlp_bwd<32, 2, 4> (same arithmetic, 4.3 instead of 3.2 ms per epoch)
FALLBACK_FLAGS = {"ppo_resmlp512.hip": ["-DRESMLP_BWD2S=0"]}

_flag_ok = {}


def flags_accepted(flags):
    """-mllvm options are internal to LLVM and unversioned: probe them once on an empty translation unit, so that a hipcc that
    no longer knows one builds without it (slower persistent kernels, same results) instead of failing the build."""
    key = tuple(flags)
    if key not in _flag_ok:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "probe.hip")
            open(src, "w").write("#include <hip/hip_runtime.h>\n__global__ void probe() {}\n")
            r = subprocess.run([hipcc(), "--offload-arch=gfx950", "-c", src, "-o", os.path.join(d, "probe.o")] + list(flags),
                               capture_output=True, text=True)
        _flag_ok[key] = r.returncode == 0
        if not _flag_ok[key]:
            print(f"navbot_ppo_amd.build: hipcc rejects {' '.join(flags)} -- building without it", flush=True)
    return _flag_ok[key]


def per_source_flags(base):
    """the flags one source is compiled with besides HIPCC_FLAGS: its EXTRA_FLAGS if hipcc takes them, else its FALLBACK_FLAGS"""
    per_src = [] if os.environ.get("NAVSIM_NO_EXTRA_FLAGS") == "1" else EXTRA_FLAGS.get(base, [])   # (A/B builds)
    if per_src and not flags_accepted(per_src):
        per_src = []
    if not per_src and base in EXTRA_FLAGS:
        per_src = list(FALLBACK_FLAGS.get(base, []))
    return list(per_src)


def build_native(force=False, verbose=False, navsim_src=None, out=None, extra=()):
    """navsim_src / out / extra: dev tools build variants of csrc/navsim.hip (patched copies, instrumented builds) as another
    library with exactly the product's flags."""
    if navsim_src is not None and out is None:
        raise ValueError("build_native: a variant source (navsim_src) needs its own output library (out)")
    if navsim_src is None and not force and not needs_build():
        return LIB
    tag = "" if navsim_src is None else "_" + os.path.splitext(os.path.basename(out))[0]
    objdir = os.path.join(REPO, "build", "obj" + tag)
    os.makedirs(objdir, exist_ok=True)
    compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    procs, objs = [], []
    srcs = SRCS if navsim_src is None else [navsim_src] + SRCS[1:]
    lib_out = LIB if out is None else out
    hdrs = [os.path.join(INC, h) for h in HDRS] + [os.path.join(HERE, "csrc", h) for h in ("mlp64_policy.h", "navppo_internal.h", "resmlp_policy.h", "bf16x3.h", "ppo_mlp64_x3s.h", "ppo_resmlp512_bwd2s.h")]
    for k, src in enumerate(srcs):   # the sources compile side by side
        obj = os.path.join(objdir, os.path.basename(SRCS[k]) + ".o")
        per_src = per_source_flags(os.path.basename(SRCS[k]))
        cmd = ([hipcc()] + compile_flags + per_src + (list(extra) if k == 0 else []) +
               ["-I", INC, "-I", os.path.join(HERE, "csrc"), "-c", src, "-o", obj])
        objs.append(obj)
        # incremental: an object newer than its source, the headers and this file, built by the same command line, is kept
        stamp = obj + ".cmd"
        fresh = (not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == " ".join(cmd) and
                 all(os.path.getmtime(d) < os.path.getmtime(obj) for d in [src, os.path.abspath(__file__)] + hdrs))
        if fresh:
            continue
        if verbose:
            print(" ".join(cmd))
        if os.path.exists(stamp):
            os.remove(stamp)
        procs.append((cmd, subprocess.Popen(cmd), stamp))
    for cmd, p, stamp in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
        open(stamp, "w").write(" ".join(cmd))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden"] + objs + ["-o", lib_out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib_out


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
