// How many vector instructions hide in the shadow of a bf16 MFMA when they sit in the SAME wave's instruction stream, between the
// MFMAs (sched_group_barrier pins the interleave: 1 MFMA, F fillers, 1 MFMA, ...)?  One and two waves per SIMD.  Fillers: v_fma_f32
// (independent chains) or the bf16 split sequence (cvt_pk / shift / and / sub).  Design input of the split-bf16 update kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/bf16_mfma_fillers.hip -o build/bf16_fillers
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 h = __builtin_convertvector(v, bf16x2);
    unsigned u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}

// F fillers per MFMA; KIND 0: v_fma_f32, 1: split mix (per pair 11 ops: F is rounded to pairs: F = 11 -> one pair per MFMA)
template <int F, int KIND, int THREADS>
__global__ __launch_bounds__(THREADS) void k(int iters, int with_mfma, float* sink) {
    f32x16 c[4] = {{0}, {0}, {0}, {0}};
    const float a = threadIdx.x * 0.001f;
    bf16x8 va, vb;
    for (int e = 0; e < 8; ++e) { va[e] = (__bf16)(a + e); vb[e] = (__bf16)(1.0f - e); }
    float x[32];
    for (int j = 0; j < 32; ++j) x[j] = threadIdx.x + j;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (with_mfma) c[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, c[m & 3], 0, 0, 0);
            if (KIND == 0) {
#pragma unroll
                for (int f = 0; f < F; ++f) x[(m * F + f) & 31] = fmaf(x[(m * F + f) & 31], 0.999f, 0.5f);
            } else {
#pragma unroll
                for (int p = 0; p < F / 11; ++p) {
                    const int j = (2 * (m * (F / 11) + p)) & 31;
                    const unsigned p0 = pk(x[j], x[j + 1]);
                    const float ra = x[j] - __uint_as_float(p0 << 16), rb = x[j + 1] - __uint_as_float(p0 & 0xffff0000u);
                    const unsigned p1 = pk(ra, rb);
                    const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);
                    const unsigned p2 = pk(sa, sb);
                    acc ^= p0 ^ p1 ^ p2;   // (+3 ops)
                    x[j] += 1.0f;          // (+2 ops)
                    x[j + 1] += 0.5f;
                }
            }
            if (with_mfma) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (F > 0) __builtin_amdgcn_sched_group_barrier(0x002, KIND == 0 ? F : (F / 11) * 16, 0);
            }
        }
    }
    float s = __uint_as_float(acc);
    for (int j = 0; j < 32; ++j) s += x[j];
    if (c[0][0] + c[1][3] + c[2][5] + c[3][7] + s == 123.f) sink[0] = s;
}

template <int F, int KIND, int THREADS>
void run(float* sink) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float t[3];
    for (int mode = 0; mode < 3; ++mode) {   // 0: MFMA only (F = 0 kernel), 1: fillers only, 2: both
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL((k<0, 0, THREADS>), dim3(256), dim3(THREADS), 0, 0, iters, 1, sink);
            else hipLaunchKernelGGL((k<F, KIND, THREADS>), dim3(256), dim3(THREADS), 0, 0, iters, mode == 2, sink);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&t[mode], e0, e1);
        }
    }
    const double per = 1e6 / (8.0 * iters) / (THREADS / 256);   // ns per MFMA slot and wave
    const int ops = KIND == 0 ? F : (F / 11) * 16;
    printf("  %d wave(s)/SIMD  %-6s %2d ops per MFMA: mfma only %6.1f  fillers only %6.1f  interleaved %6.1f ns per MFMA slot and wave (sum %6.1f) -> hidden %4.1f of %4.1f ns\n",
           THREADS / 256, KIND ? "split" : "fma", ops, t[0] * per, t[1] * per, t[2] * per, (t[0] + t[1]) * per, (t[0] + t[1] - t[2]) * per,
           (t[0] < t[1] ? t[0] : t[1]) * per);
}

int main() {
    float* sink;
    hipMalloc(&sink, 16);
    run<2, 0, 256>(sink); run<4, 0, 256>(sink); run<6, 0, 256>(sink); run<8, 0, 256>(sink); run<12, 0, 256>(sink);
    run<11, 1, 256>(sink); run<22, 1, 256>(sink);
    run<2, 0, 512>(sink); run<4, 0, 512>(sink); run<6, 0, 512>(sink); run<8, 0, 512>(sink); run<12, 0, 512>(sink);
    run<11, 1, 512>(sink); run<22, 1, 512>(sink);
    return 0;
}
