// How long does gfx950 take to START all workgroups of a grid?  (dev tool)
// Each block records wall_clock64 at entry; we report max(start) - min(start) and the kernel duration for several
// geometries / register footprints.  hipcc --offload-arch=gfx950 -O2 dispatch_rate.hip -o dispatch_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int NREG, int LDSF>
__global__ __launch_bounds__(1024) void k(long long* start, float* sink) {
    __shared__ float lds[LDSF > 0 ? LDSF : 1];
    if (threadIdx.x == 0) start[blockIdx.x] = wall_clock64();
    float r[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = threadIdx.x * 0.5f + i;
    lds[threadIdx.x % (LDSF > 0 ? LDSF : 1)] = r[0];
    __syncthreads();
    float s = lds[0];
#pragma unroll
    for (int i = 0; i < NREG; ++i) s = fmaf(s, r[i], r[(i * 7) % NREG]);
    if (s == 123.456f) sink[0] = s;
}

template <int NREG, int LDSF>
void run(const char* name, int blocks, int threads) {
    long long* d; float* sink;
    hipMalloc(&d, sizeof(long long) * blocks); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<NREG, LDSF>), dim3(blocks), dim3(threads), 0, 0, d, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((k<NREG, LDSF>), dim3(blocks), dim3(threads), 0, 0, d, sink);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks); hipMemcpy(h.data(), d, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    long long mn = *std::min_element(h.begin(), h.end()), mx = *std::max_element(h.begin(), h.end());
    printf("%-34s blocks=%5d threads=%4d  start spread %6.2f us  kernel+gap %6.2f us\n", name, blocks, threads, (mx - mn) * 0.01, ms / 20 * 1e3);
    hipFree(d); hipFree(sink);
}

int main() {
    run<8, 0>("tiny regs, no LDS", 1024, 256);
    run<8, 0>("tiny regs, no LDS", 4096, 64);
    run<8, 0>("tiny regs, no LDS", 512, 512);
    run<8, 0>("tiny regs, no LDS", 256, 1024);
    run<100, 0>("~110 VGPR, no LDS", 1024, 256);
    run<100, 1800>("~110 VGPR, 7 KB LDS", 1024, 256);
    run<100, 1800>("~110 VGPR, 7 KB LDS", 512, 256);
    run<100, 1800>("~110 VGPR, 7 KB LDS", 2048, 256);
    run<40, 1800>("~50 VGPR, 7 KB LDS", 1024, 256);
    return 0;
}
