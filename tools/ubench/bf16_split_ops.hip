// Splitting float32 values into three bf16 pieces (a = p0 + p1 + p2 exactly): two sequences, their exactness and their issue cost on gfx950.
//   A  cvt_pk, unpack (lshl / and), subtract, ...            11 vector instructions per pair of values
//   B  cvt_pk, v_dot2c_f32_bf16 with a (-1, 0) / (0, -1) selector as the residual (acc + piece * -1), ...   7 per pair
// plus the issue rates of the single instructions.  Dev tool (design input of the split-bf16 update kernel):
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/bf16_split_ops.hip -o build/bf16_split_ops
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 h = __builtin_convertvector(v, bf16x2);
    unsigned u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}
__device__ __forceinline__ bf16x2 as_bf(unsigned u) {
    bf16x2 h;
    __builtin_memcpy(&h, &u, 4);
    return h;
}
__device__ __forceinline__ void splitA(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = pk(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = pk(ra, rb);
    const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);
    p2 = pk(sa, sb);
}
__device__ __forceinline__ void splitB(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    const bf16x2 lo = as_bf(0x0000BF80u), hi = as_bf(0xBF800000u);   // (-1, 0), (0, -1)
    p0 = pk(a, b);
    const float ra = __builtin_amdgcn_fdot2_f32_bf16(as_bf(p0), lo, a, false), rb = __builtin_amdgcn_fdot2_f32_bf16(as_bf(p0), hi, b, false);
    p1 = pk(ra, rb);
    const float sa = __builtin_amdgcn_fdot2_f32_bf16(as_bf(p1), lo, ra, false), sb = __builtin_amdgcn_fdot2_f32_bf16(as_bf(p1), hi, rb, false);
    p2 = pk(sa, sb);
}

__global__ void exact(const float* x, int n, unsigned* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned a0, a1, a2, b0, b1, b2;
    splitA(x[2 * i], x[2 * i + 1], a0, a1, a2);
    splitB(x[2 * i], x[2 * i + 1], b0, b1, b2);
    out[6 * i] = a0; out[6 * i + 1] = a1; out[6 * i + 2] = a2; out[6 * i + 3] = b0; out[6 * i + 4] = b1; out[6 * i + 5] = b2;
}

// KIND: 0 fma, 1 cvt_pk, 2 dot2c, 3 and, 4 lshl, 5 sub, 6 perm, 7 splitA, 8 splitB
template <int KIND>
__global__ __launch_bounds__(256) void rate(int iters, float* sink) {
    float x[16];
    unsigned u[16];
    for (int j = 0; j < 16; ++j) { x[j] = threadIdx.x + j + 0.37f; u[j] = threadIdx.x * 77 + j; }
    const bf16x2 lo = as_bf(0x0000BF80u);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (KIND == 0) x[j] = fmaf(x[j], 0.999f, 0.5f);
            if (KIND == 1) u[j] = pk(__uint_as_float(u[j]), x[j]);
            if (KIND == 2) x[j] = __builtin_amdgcn_fdot2_f32_bf16(as_bf(u[j]), lo, x[j], false);
            if (KIND == 3) u[j] = (u[j] & 0xffff0f0fu) + 1;   // (and + add: halve)
            if (KIND == 4) u[j] = (u[j] << 3) ^ u[(j + 1) & 15];
            if (KIND == 5) x[j] = x[j] - 0.25f;
            if (KIND == 6) u[j] = __builtin_amdgcn_perm(u[j], u[(j + 1) & 15], 0x07060302u);
        }
        if (KIND == 7 || KIND == 8) {
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                unsigned p0, p1, p2;
                if (KIND == 7) splitA(x[j], x[j + 1], p0, p1, p2);
                else splitB(x[j], x[j + 1], p0, p1, p2);
                u[j] ^= p0 ^ p1;
                u[j + 1] ^= p2;
                x[j] += 1.0f;
                x[j + 1] += 0.5f;
            }
        }
    }
    float s = 0;
    unsigned t = 0;
    for (int j = 0; j < 16; ++j) { s += x[j]; t ^= u[j]; }
    if (s == 123.f && t == 77u) sink[0] = s;
}

static float bf(uint32_t h16) { uint32_t u = h16 << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int n = 1 << 20;
    std::mt19937 rng(3);
    std::vector<float> x(n);
    for (int i = 0; i < n; ++i) {
        std::uniform_real_distribution<float> m(-1.f, 1.f);
        std::uniform_int_distribution<int> e(-40, 20);
        x[i] = std::ldexp(m(rng), e(rng));
        if (i % 1000 == 0) x[i] = 0.f;
        if (i % 1000 == 1) x[i] = 1.0f;
        if (i % 1000 == 2) x[i] = std::ldexp(m(rng), -120);   // near the bottom of the normal range: pieces go subnormal
    }
    float* dx; unsigned* dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, (size_t)n * 3 * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(exact, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, dout);
    std::vector<unsigned> out((size_t)n * 3);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    long badA = 0, badB = 0, diff = 0, badA_n = 0, badB_n = 0;
    for (int i = 0; i < n / 2; ++i)
        for (int h = 0; h < 2; ++h) {
            const float v = x[2 * i + h];
            auto piece = [&](unsigned w) { return bf(h ? (w >> 16) : (w & 0xffffu)); };
            const double sa = (double)piece(out[6 * i]) + piece(out[6 * i + 1]) + piece(out[6 * i + 2]);
            const double sb = (double)piece(out[6 * i + 3]) + piece(out[6 * i + 4]) + piece(out[6 * i + 5]);
            const bool tiny = std::fabs(v) < 1e-30f;
            if (sa != (double)v) { ++badA; if (!tiny) ++badA_n; }
            if (sb != (double)v) { ++badB; if (!tiny) ++badB_n; }
            if (out[6 * i] != out[6 * i + 3] || out[6 * i + 1] != out[6 * i + 4] || out[6 * i + 2] != out[6 * i + 5]) ++diff;
        }
    printf("exactness over %d values (|x| from 2^-40 to 2^20, zeros, ones, ~0.1 %% near 2^-120):\n", n);
    printf("  A (unpack + subtract): %ld values with p0 + p1 + p2 != x (%ld of them with |x| >= 1e-30)\n", badA, badA_n);
    printf("  B (v_dot2c residual) : %ld values with p0 + p1 + p2 != x (%ld of them with |x| >= 1e-30); pieces differ from A's in %ld pair slots\n", badB, badB_n, diff);
    float* sink; hipMalloc(&sink, 16);
    const char* names[9] = {"v_fma_f32", "v_cvt_pk_bf16_f32", "v_dot2c_f32_bf16", "v_and + v_add (2 ops)", "v_lshl + v_xor (2 ops)", "v_sub_f32", "v_perm_b32",
                            "split A, 8 pairs (88 ops)", "split B, 8 pairs (56 ops)"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int wv = 1; wv <= 2; ++wv)   // waves per SIMD: blocks of 256 threads (one wave per SIMD), or 512
        for (int kk = 0; kk < 9; ++kk) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
#define GO(K) case K: hipLaunchKernelGGL(rate<K>, dim3(256 * wv), dim3(256), 0, 0, iters, sink); break;
                switch (kk) { GO(0) GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) GO(8) }
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double per_iter_ns = ms * 1e6 / iters / wv;   // per wave and iteration (16 slots)
            printf("  %d wave(s)/SIMD  %-28s %7.2f ns per 16-slot iteration per wave = %5.2f cycles per slot at 2.4 GHz\n", wv, names[kk], per_iter_ns,
                   per_iter_ns / 16 * 2.4);
        }
    return 0;
}
