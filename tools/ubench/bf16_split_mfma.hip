// Device half of tools/design/bf16_split_error.py: a float32 product sum C = A B (32 x K times K x 32, K = 64) evaluated on ONE wave
//   (a) as a chain of v_mfma_f32_32x32x2_f32 (what the update kernels do today), and
//   (b) as six v_mfma_f32_32x32x16_bf16 terms of operands split into three bf16 pieces (small terms first / big terms first),
// both against a float64 reference on the host; then the issue rate of the two instructions on all CUs (4 independent accumulators
// per wave, one wave per SIMD).  Dev tool, gfx950 only:  hipcc --offload-arch=gfx950 -O3 tools/ubench/bf16_split_mfma.hip -o build/bf16_split_mfma
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int K = 64;

// operand layouts: f32 32x32x2: lane l holds A[l & 31][k = l >> 5], B[k = l >> 5][l & 31].
// bf16 32x32x16, hypothesis `alt` = 0: lane l holds the 8 consecutive k = 8 (l >> 5) + e; alt = 1: k = 4 (l >> 5) + (e & 3) + 8 (e >> 2)
__device__ __forceinline__ int kidx(int lane, int e, int alt) { return alt ? 4 * (lane >> 5) + (e & 3) + 8 * (e >> 2) : 8 * (lane >> 5) + e; }

__global__ void accuracy(const float* A, const float* B, const uint16_t* Ap, const uint16_t* Bp, int alt, float* out) {
    const int lane = threadIdx.x, i = lane & 31, hi = lane >> 5;
    f32x16 c = {0};
    for (int k0 = 0; k0 < K; k0 += 2) c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k0 + hi], B[(k0 + hi) * 32 + i], c, 0, 0, 0);
    f32x16 s[2] = {{0}, {0}};
    const int order[6][2] = {{2, 0}, {1, 1}, {0, 2}, {1, 0}, {0, 1}, {0, 0}};   // small terms first
    for (int v = 0; v < 2; ++v)
        for (int t = 0; t < 6; ++t) {
            const int pa = order[v ? 5 - t : t][0], pb = order[v ? 5 - t : t][1];
            for (int k0 = 0; k0 < K; k0 += 16) {
                bf16x8 a, b;
                for (int e = 0; e < 8; ++e) {
                    const int k = k0 + kidx(lane, e, alt);
                    uint16_t ua = Ap[(pa * 32 + i) * K + k], ub = Bp[(pb * K + k) * 32 + i];
                    __bf16 xa, xb;
                    memcpy(&xa, &ua, 2);
                    memcpy(&xb, &ub, 2);
                    a[e] = xa;
                    b[e] = xb;
                }
                s[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s[v], 0, 0, 0);
            }
        }
    for (int r = 0; r < 16; ++r) {   // accumulator register r of lane (j = l & 31, hi): row 8 (r >> 2) + 4 hi + (r & 3), column j
        const int row = 8 * (r >> 2) + 4 * hi + (r & 3);
        out[0 * 1024 + row * 32 + i] = c[r];
        out[1 * 1024 + row * 32 + i] = s[0][r];
        out[2 * 1024 + row * 32 + i] = s[1][r];
    }
}

template <bool BF>
__global__ __launch_bounds__(256) void rate(int iters, float* sink) {
    f32x16 c[4] = {{0}, {0}, {0}, {0}};
    const float x = (float)threadIdx.x * 1e-3f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(x + e); b[e] = (__bf16)(x - e); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (BF) c[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[u], 0, 0, 0);
            else c[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x + 1.f, c[u], 0, 0, 0);
        }
    float t = 0.f;
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) t += c[u][r];
    if (t == 123.456f) sink[0] = t;
}

static uint16_t bf16_rne(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    return (uint16_t)u;
}
static float bf16_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> uw(-0.125f, 0.125f);
    std::normal_distribution<float> nx(0.3f, 0.5f);
    std::vector<float> A(32 * K), B(K * 32);
    for (auto& v : A) v = uw(rng);                      // a weight tile (nn.Linear init at K = 64)
    for (auto& v : B) v = std::fmax(nx(rng), 0.f);      // relu activations
    std::vector<uint16_t> Ap(3 * 32 * K), Bp(3 * K * 32);
    auto split = [](const std::vector<float>& src, std::vector<uint16_t>& dst) {
        const size_t n = src.size();
        for (size_t q = 0; q < n; ++q) {
            float rest = src[q];
            for (int p = 0; p < 3; ++p) { const uint16_t h = bf16_rne(rest); dst[p * n + q] = h; rest -= bf16_f(h); }
        }
    };
    split(A, Ap); split(B, Bp);
    float *dA, *dB, *dout; uint16_t *dAp, *dBp;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dAp, Ap.size() * 2); hipMalloc(&dBp, Bp.size() * 2);
    hipMalloc(&dout, 3 * 1024 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dAp, Ap.data(), Ap.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dBp, Bp.data(), Bp.size() * 2, hipMemcpyHostToDevice);
    std::vector<double> ref(1024), scale(1024);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double r = 0, s = 0;
        for (int k = 0; k < K; ++k) { r += (double)A[i * K + k] * B[k * 32 + j]; s += std::fabs((double)A[i * K + k] * B[k * 32 + j]); }
        ref[i * 32 + j] = r; scale[i * 32 + j] = s;
    }
    for (int alt = 0; alt < 2; ++alt) {
        hipLaunchKernelGGL(accuracy, dim3(1), dim3(64), 0, 0, dA, dB, dAp, dBp, alt, dout);
        std::vector<float> out(3 * 1024);
        hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
        const char* name[3] = {"f32 chain (32 x v_mfma_f32_32x32x2_f32)", "split 3 pieces / 6 terms, small first (24 x v_mfma_f32_32x32x16_bf16)", "split 3 pieces / 6 terms, big first"};
        printf("bf16 operand layout hypothesis %d (%s):\n", alt, alt ? "k = 4 (l >> 5) + (e & 3) + 8 (e >> 2)" : "k = 8 (l >> 5) + e");
        for (int v = 0; v < 3; ++v) {
            double mx = 0, ss = 0;
            for (int q = 0; q < 1024; ++q) { const double e = std::fabs((double)out[v * 1024 + q] - ref[q]) / scale[q]; mx = std::fmax(mx, e); ss += e * e; }
            printf("  %-75s max |err| / sum|a||b| = %.3e   rms = %.3e\n", name[v], mx, std::sqrt(ss / 1024));
        }
    }
    // issue rate: 256 workgroups x 4 waves (one wave per SIMD), 4 independent accumulators per wave
    float* sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int bf = 0; bf < 2; ++bf) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            if (bf) hipLaunchKernelGGL(rate<true>, dim3(256), dim3(256), 0, 0, iters, sink);
            else hipLaunchKernelGGL(rate<false>, dim3(256), dim3(256), 0, 0, iters, sink);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) {
                const double per = ms * 1e6 / (4.0 * iters);   // ns per MFMA per SIMD
                const double flop = bf ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2;
                printf("%s: %.2f ns per instruction per SIMD, %.1f TFLOP/s on 1024 SIMDs, k per ns per SIMD %.3f\n",
                       bf ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32  ", per, flop * 1024 / per / 1e3, (bf ? 16 : 2) / per);
            }
        }
    }
    return 0;
}
