#!/usr/bin/env python3
"""Dev tool: build/libnavsim_ppotiming.so = ppo_mlp64.hip with wall_clock64 stamps (10 ns ticks) at the phase boundaries of
mlp64_pass for workgroup 0, accumulated over its tiles (read back with navppo_dbg_read)."""
import os, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = open(os.path.join(R, "navbot_ppo_amd/csrc/ppo_mlp64.hip")).read()
def rep(a, b, n=1):
    global t
    assert a in t, a
    t = t.replace(a, b, n)
rep("template <bool ACTOR, int PT>\n__global__", "__device__ long long g_ph[16];\n#define PH(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { long long now_ = wall_clock64(); g_ph[k] += now_ - last_; last_ = now_; } } while (0)\ntemplate <bool ACTOR, int PT>\n__global__")
rep("    const long long n_tiles = (M + TM - 1) / TM;\n", "    long long last_ = wall_clock64();\n    const long long n_tiles = (M + TM - 1) / TM;\n")
rep("        // ---- F1: H1 = relu(X W1^T + b1), K = 16", "        PH(0);\n        // ---- F1: H1 = relu(X W1^T + b1), K = 16")
rep("        // ---- F2: H2 = relu(H1 W2^T + b2), K = 64", "        PH(1);\n        // ---- F2: H2 = relu(H1 W2^T + b2), K = 64")
rep("        // ---- output units + loss: 4 threads per sample", "        PH(2);\n        // ---- output units + loss: 4 threads per sample")
rep("        // ---- dH2 = (g3 w3 + g4 w4) . [H2 > 0] ; column sums for db2, dW3, dW4\n        {\n            const int k = tid & 63;", "        PH(3);\n        // ---- dH2 = (g3 w3 + g4 w4) . [H2 > 0] ; column sums for db2, dW3, dW4\n        {\n            const int k = tid & 63;")
rep("        // ---- B2: dH1 = (dH2 W2) . [H1 > 0] -> stored over H2 ; db1 partial sums\n        {\n            f32x16 c = zero16();", "        PH(4);\n        // ---- B2: dH1 = (dH2 W2) . [H1 > 0] -> stored over H2 ; db1 partial sums\n        {\n            f32x16 c = zero16();")
rep("        // ---- G2: dW2[n][k] += sum_m dH2[m][n] H1[m][k]; quadrant", "        PH(5);\n        // ---- G2: dW2[n][k] += sum_m dH2[m][n] H1[m][k]; quadrant")
rep("    // ---- workgroup reduction of the partial gradient in LDS, then one coalesced row of `partial`\n    __syncthreads();", "    // ---- workgroup reduction of the partial gradient in LDS, then one coalesced row of `partial`\n    __syncthreads();\n    PH(7);")
# G2+G1 end: stamp at loop end -> before closing brace of tile loop: find "accW1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ptr[m0 * LDH], b, accW1, 0, 0, 0);\n            }\n        }\n    }"
rep("                accW1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ptr[m0 * LDH], b, accW1, 0, 0, 0);\n            }\n        }\n    }", "                accW1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ptr[m0 * LDH], b, accW1, 0, 0, 0);\n            }\n        }\n        PH(6);\n    }")
rep('const char* navppo_last_error(void) { return g_err.c_str(); }', 'const char* navppo_last_error(void) { return g_err.c_str(); }\nint navppo_dbg_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ph), sizeof(long long) * 16); }\nint navppo_dbg_zero(void) { long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ph), z, sizeof(z)); }')
open("/tmp/ppo_timing.hip", "w").write(t)
out = os.path.join(R, "build", "libnavsim_ppotiming.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                       "-fvisibility=hidden", "-I", os.path.join(R, "include"), os.path.join(R, "navbot_ppo_amd/csrc/navsim.hip"),
                       "/tmp/ppo_timing.hip", "-o", out])
print(out)
