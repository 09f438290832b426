// How fast can gfx950 run the ray/segment test of navsim.hip when NOTHING else is in the kernel?  (dev tool)
// Each wave repeats: one 64-segment tile (float4 per lane, L2-resident) x NB beams of ray_seg_bits(); 1024 workgroups
// of 4 waves (4 waves per SIMD, like the step kernel at configs[2]) or 8/2 waves per SIMD.  Prints ns per tile-wave and
// the implied VALU issue rate.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize ray_tile_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float div_pos(float K, float Dn) {
    float r = __builtin_amdgcn_rcpf(Dn);
    const float f0 = fmaf(-Dn, r, 1.0f);
    r = fmaf(f0, r, r);
    float q = K * r;
    float e = fmaf(-Dn, q, K);
    q = fmaf(e, r, q);
    e = fmaf(-Dn, q, K);
    return fmaf(e, r, q);
}
__device__ __forceinline__ unsigned ray_seg_bits(float rx, float ry, float ex, float ey, float k, float c, float s) {
    const float den = fmaf(c, ey, -(s * ex));
    const float un = fmaf(rx, s, -(ry * c));
    const float p1 = fmaf(k, den, 0.0f);
    const float p2 = fmaf(un, den, 0.0f);
    const float w = fabsf(den) - fabsf(un);
    const unsigned miss = (unsigned)((int)(__float_as_uint(p1) | __float_as_uint(p2) | __float_as_uint(w)) >> 31);
    return __float_as_uint(div_pos(fabsf(k), fabsf(den))) | miss;
}

template <int NB>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ seg, int iters, unsigned* __restrict__ out) {
    const int lane = threadIdx.x & 63, gw = (blockIdx.x * 256 + threadIdx.x) >> 6;
    float dc[NB], ds[NB];
    unsigned best[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        dc[b] = __cosf(0.6283f * b + gw * 0.001f);
        ds[b] = __sinf(0.6283f * b + gw * 0.001f);
        best[b] = 0x7f800000u;
    }
    const float ox = 0.01f * (gw & 63), oy = -0.02f * (gw & 31);
    float4 nxt = seg[(gw & 1023) * 64 + lane];
    for (int it = 0; it < iters; ++it) {
        const float4 g = nxt;
        nxt = seg[((gw + it + 1) & 1023) * 64 + lane];   // next tile in flight while this one is tested
        const float rx = g.x - ox, ry = g.y - oy, ex = g.z - g.x, ey = g.w - g.y;
        const float kk = fmaf(rx, ey, -(ry * ex));
#pragma unroll
        for (int b = 0; b < NB; ++b) best[b] = min(best[b], ray_seg_bits(rx, ry, ex, ey, kk, dc[b], ds[b]));
    }
    unsigned m = 0xffffffffu;
#pragma unroll
    for (int b = 0; b < NB; ++b) m = min(m, best[b]);
    if (m == 12345u) out[0] = m;
}

int main() {
    std::vector<float4> h(1024 * 64);
    for (size_t i = 0; i < h.size(); ++i) {
        const float a = 0.37f * i, r = 2.0f + 0.001f * (i % 977);
        h[i] = make_float4(r * cosf(a), r * sinf(a), r * cosf(a + 0.3f), r * sinf(a + 0.3f));
    }
    float4* d; unsigned* out;
    hipMalloc(&d, h.size() * sizeof(float4)); hipMalloc(&out, 4);
    hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int blocks : {512, 1024, 1536, 2048}) {
        hipLaunchKernelGGL(k<10>, dim3(blocks), dim3(256), 0, 0, d, iters, out);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<10>, dim3(blocks), dim3(256), 0, 0, d, iters, out);
        hipEventRecord(e1, 0); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double tiles_per_simd = (double)blocks * 4 * iters / 1024.0;
        const double ns_per_tile = ms * 1e6 / tiles_per_simd;
        printf("waves/SIMD %.1f: %8.1f us total, %6.1f ns per tile per SIMD (190 VALU -> %.2f G VALU/s/SIMD), 16 B x 64 lanes / tile -> %.2f TB/s chip-wide\n",
               blocks * 4 / 1024.0, ms * 1e3, ns_per_tile, 190.0 / ns_per_tile, 1024.0 * 1024.0 / ns_per_tile / 1e3);
    }
    return 0;
}
