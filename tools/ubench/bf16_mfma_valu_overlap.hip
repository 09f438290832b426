// Do bf16 MFMA (v_mfma_f32_32x32x16_bf16) and vector / LDS work overlap on gfx950 -- from two waves of one SIMD, and inside one wave?
// The f32-input MFMA does not (mfma_valu_overlap.hip: the times add); the split-bf16 update kernel is sized on
// the answer for the bf16 instruction.  Dev tool:  hipcc --offload-arch=gfx950 -O3 tools/ubench/bf16_mfma_valu_overlap.hip -o build/bf16_overlap
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

// the split of two floats into three bf16 pieces each (11 vector instructions): what the update kernel does per operand pair
__device__ __forceinline__ void split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = pk(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = pk(ra, rb);
    const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);
    p2 = pk(sa, sb);
}

// MF: 0 = f32 32x32x2, 1 = bf16 32x32x16.  KIND: 0 = v_fma_f32, 1 = split2, 2 = ds_read_b128, 3 = ds_write_b32
template <int MF, int KIND>
__global__ __launch_bounds__(512) void two_waves(int n_mfma, int n_valu, float* sink) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i;
    __syncthreads();
    if (wave < 4) {
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        const float a = threadIdx.x * 0.001f, b = 1.0f;
        bf16x8 va, vb;
        for (int e = 0; e < 8; ++e) { va[e] = (__bf16)(a + e); vb[e] = (__bf16)(b - e); }
        for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MF) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, va, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, va, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, vb, c3, 0, 0, 0);
                } else {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
                }
            }
        }
        if (c0[0] + c1[3] + c2[5] + c3[7] == 123.f) sink[0] = c0[1];
    } else {
        float x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = threadIdx.x + j;
        if (KIND == 0) {
            for (int i = 0; i < n_valu; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 16; ++j) x[j] = fmaf(x[j], 0.999f, 0.5f);
        } else if (KIND == 1) {
            unsigned acc = 0;
            for (int i = 0; i < n_valu; ++i)
#pragma unroll
                for (int j = 0; j < 16; j += 2) {   // 8 pairs x 11 instructions + the feedback
                    unsigned p0, p1, p2;
                    split2(x[j], x[j + 1], p0, p1, p2);
                    acc ^= p0 ^ p1 ^ p2;
                    x[j] += 1.0f;
                    x[j + 1] += 0.5f;
                }
            x[0] = __uint_as_float(acc);
        } else if (KIND == 2) {
            const float4* p = reinterpret_cast<const float4*>(lds) + (threadIdx.x & 63);
            for (int i = 0; i < n_valu; ++i)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float4 v = p[64 * ((j + i) & 15)];
                    x[j] += v.x;
                }
        } else {
            float* p = lds + (threadIdx.x & 63) + 64 * (wave - 4) * 16;
            for (int i = 0; i < n_valu; ++i)
#pragma unroll
                for (int j = 0; j < 16; ++j) p[64 * j] = x[j] + (float)i;
            x[1] = p[5];
        }
        float s = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += x[j];
        if (s == 123.f) sink[1] = s;
    }
}

// one wave per SIMD (256-thread blocks): per iteration 4 independent MFMAs and NV split2 pairs whose inputs do not depend on them
template <int MF, int NV>
__global__ __launch_bounds__(256) void one_wave(int iters, int with_mfma, float* sink) {
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const float a = threadIdx.x * 0.001f, b = 1.0f;
    bf16x8 va, vb;
    for (int e = 0; e < 8; ++e) { va[e] = (__bf16)(a + e); vb[e] = (__bf16)(b - e); }
    float x[2 * (NV > 0 ? NV : 1)];
    for (int j = 0; j < 2 * NV; ++j) x[j] = threadIdx.x + j;
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        if (with_mfma) {
            if (MF) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, va, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, va, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, vb, c3, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            unsigned p0, p1, p2;
            split2(x[2 * j], x[2 * j + 1], p0, p1, p2);
            acc ^= p0 ^ p1 ^ p2;
            x[2 * j] += 1.0f;
            x[2 * j + 1] += 0.5f;
        }
    }
    if (c0[0] + c1[3] + c2[5] + c3[7] + __uint_as_float(acc) == 123.f) sink[0] = c0[1];
}

template <class F>
float timed(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    launch();
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

template <int MF, int KIND>
void two(const char* name, int nm, int nv, float* sink) {
    const float m = timed([&] { hipLaunchKernelGGL((two_waves<MF, KIND>), dim3(256), dim3(512), 0, 0, nm, 0, sink); });
    const float v = timed([&] { hipLaunchKernelGGL((two_waves<MF, KIND>), dim3(256), dim3(512), 0, 0, 0, nv, sink); });
    const float b = timed([&] { hipLaunchKernelGGL((two_waves<MF, KIND>), dim3(256), dim3(512), 0, 0, nm, nv, sink); });
    printf("  %-34s mfma only %8.1f us | other only %8.1f us | both %8.1f us   (sum %8.1f, max %8.1f)\n", name, m, v, b, m + v, m > v ? m : v);
}

template <int MF, int NV>
void one(int iters, float* sink) {
    const float m = timed([&] { hipLaunchKernelGGL((one_wave<MF, 0>), dim3(256), dim3(256), 0, 0, iters, 1, sink); });
    const float v = timed([&] { hipLaunchKernelGGL((one_wave<MF, NV>), dim3(256), dim3(256), 0, 0, iters, 0, sink); });
    const float b = timed([&] { hipLaunchKernelGGL((one_wave<MF, NV>), dim3(256), dim3(256), 0, 0, iters, 1, sink); });
    printf("  4 mfma + %2d split pairs (%3d valu)  mfma only %8.1f us | valu only %8.1f us | both %8.1f us   (sum %8.1f, max %8.1f)\n", NV, 11 * NV + 3 * NV,
           m, v, b, m + v, m > v ? m : v);
}

int main() {
    float* sink;
    hipMalloc(&sink, 16);
    for (int mf = 0; mf < 2; ++mf) {
        printf("%s, two waves per SIMD (one runs the MFMA loop, the other the vector / LDS loop):\n", mf ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32");
        const int NM = mf ? 4000 : 2000;
        if (mf) {
            two<1, 0>("v_fma_f32", NM, 8000, sink);
            two<1, 1>("split into 3 bf16 pieces", NM, 4000, sink);
            two<1, 2>("ds_read_b128", NM, 8000, sink);
            two<1, 3>("ds_write_b32", NM, 8000, sink);
        } else {
            two<0, 0>("v_fma_f32", NM, 8000, sink);
            two<0, 1>("split into 3 bf16 pieces", NM, 4000, sink);
            two<0, 2>("ds_read_b128", NM, 8000, sink);
            two<0, 3>("ds_write_b32", NM, 8000, sink);
        }
    }
    printf("one wave per SIMD, MFMAs and independent vector work in the same instruction stream (20000 iterations):\n");
    printf(" bf16:\n");
    one<1, 2>(20000, sink);
    one<1, 4>(20000, sink);
    one<1, 8>(20000, sink);
    printf(" f32:\n");
    one<0, 4>(20000, sink);
    one<0, 8>(20000, sink);
    one<0, 16>(20000, sink);
    return 0;
}
