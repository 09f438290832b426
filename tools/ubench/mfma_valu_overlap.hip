// Do f32 MFMA (v_mfma_f32_32x32x2_f32) and f32 VALU work from two different waves of one SIMD overlap on gfx950?  (dev tool)
// One 512-thread block per CU: waves 0-3 run an MFMA loop, waves 4-7 (same SIMDs) run a VALU loop; times for
// MFMA-only, VALU-only and both.  hipcc --offload-arch=gfx950 -O2 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>  // VALU flavour: 0 = v_fma_f32, 1 = v_cndmask/v_cmp mix, 2 = ds_read_b128
__global__ __launch_bounds__(512) void k(int n_mfma, int n_valu, float* sink) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i;
    __syncthreads();
    if (wave < 4) {
        f32x16 c0 = {}, c1 = {};
        float a = threadIdx.x * 0.001f, b = 1.0f;
        for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
            }
        }
        if (c0[0] + c1[3] == 123.f) sink[0] = c0[1];
    } else {
        float x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = threadIdx.x + j;
        if (KIND == 0) {
            for (int i = 0; i < n_valu; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 16; ++j) x[j] = fmaf(x[j], 0.999f, 0.5f);
            }
        } else if (KIND == 1) {
            for (int i = 0; i < n_valu; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 16; ++j) x[j] = fmaxf(x[j] - 1.0f, 0.25f);
            }
        } else {
            const float4* p = reinterpret_cast<const float4*>(lds) + (threadIdx.x & 63);
            for (int i = 0; i < n_valu; ++i) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float4 v = p[64 * ((j + i) & 15)];
                    x[j] += v.x;
                }
            }
        }
        float s = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += x[j];
        if (s == 123.f) sink[1] = s;
    }
}

template <int KIND>
float run(int n_mfma, int n_valu, float* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, n_mfma, n_valu, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, n_mfma, n_valu, sink);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main() {
    float* sink; hipMalloc(&sink, 16);
    const int NM = 2000, NV = 8000;   // 32000 MFMAs x 64 cyc = 2.05 M cycles ; 512 k VALU ops
    printf("fma:      mfma only %8.1f us | valu only %8.1f us | both %8.1f us\n", run<0>(NM, 0, sink), run<0>(0, NV, sink), run<0>(NM, NV, sink));
    printf("max/sub:  mfma only %8.1f us | valu only %8.1f us | both %8.1f us\n", run<1>(NM, 0, sink), run<1>(0, NV / 2, sink), run<1>(NM, NV / 2, sink));
    printf("ds_b128:  mfma only %8.1f us | lds  only %8.1f us | both %8.1f us\n", run<2>(NM, 0, sink), run<2>(0, NV, sink), run<2>(NM, NV, sink));
    return 0;
}
