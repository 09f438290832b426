// How does v_mfma_f32_32x32x16_bf16 round?  D = C + sum_k A[k] B[k] with C = 1.0 (or -1.0, or 2^20) and products that sum to a known
// fraction of C's ulp -- as ONE product, or spread over the 16 k-slots -- against the exactly rounded (RNE) result.  The split-bf16
// update pass accumulates its big terms at the magnitude of the result: a truncating accumulator would bias them.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/bf16_mfma_rounding.hip -o build/bf16_mfma_rounding
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// every lane: A row = a[0..7] (its k half), B col = b[0..7]; all 32 x 32 outputs are the same dot product over the 16 k-slots
__global__ void k(const float* __restrict__ a16, const float* __restrict__ b16, float c, float* __restrict__ out) {
    const int hi = threadIdx.x >> 5;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)a16[8 * hi + e]; b[e] = (__bf16)b16[8 * hi + e]; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}

int main() {
    float *da, *db, *dout;
    hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 4);
    auto run = [&](const std::vector<float>& a, const std::vector<float>& b, float c) {
        hipMemcpy(da, a.data(), 64, hipMemcpyHostToDevice);
        hipMemcpy(db, b.data(), 64, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, c, dout);
        float r;
        hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost);
        return r;
    };
    const float fr[] = {0.25f, 0.5f, 0.75f, 1.0f, 1.25f, 1.5f, 1.75f, 2.5f};
    for (float c : {1.0f, -1.0f, 1048576.0f}) {
        const float ulp = std::ldexp(1.0f, (int)std::floor(std::log2(std::fabs(c))) - 23);
        printf("C = %g (ulp %g): result - C in ulps, device | round-to-nearest-even of the exact sum\n", c, ulp);
        for (int mode = 0; mode < 3; ++mode) {   // 0: one product; 1: 16 equal products; 2: 15 products of +x and one that cancels part
            for (int sgn = 1; sgn >= -1; sgn -= 2)
                for (float f : fr) {
                    std::vector<float> a(16, 0.f), b(16, 0.f);
                    double exact = c;
                    if (mode == 0) { a[3] = sgn * f; b[3] = ulp; exact += (double)a[3] * b[3]; }
                    if (mode == 1) for (int q = 0; q < 16; ++q) { a[q] = sgn * f / 16; b[q] = ulp; exact += (double)a[q] * b[q]; }
                    if (mode == 2) for (int q = 0; q < 16; ++q) { a[q] = (q == 7 ? -sgn * 14.f : sgn) * f; b[q] = ulp; exact += (double)a[q] * b[q]; }
                    const float got = run(a, b, c), want = (float)exact;
                    printf("  %s %+5.2f ulp: device %+g  rne %+g%s\n", mode == 0 ? "one product     " : mode == 1 ? "16 equal parts  " : "15 x - 14 x     ",
                           (exact - c) / ulp, (got - c) / ulp, (want - c) / ulp, got == want ? "" : "   <-- differs");
                }
        }
    }
    return 0;
}
