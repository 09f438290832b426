// Does the command processor preload kernel arguments into SGPRs on this box (-mllvm -amdgpu-kernarg-preload-count)?
// Same one-wave kernel built twice (this file with -DPRELOAD_BUILD=0 / 1 and the flag only in the second build); the time of
// back-to-back launches differs by the kernarg fetch a wave otherwise waits for before its first instruction that needs an
// argument.  Dev tool:  hipcc --offload-arch=gfx950 -O2 kernarg_preload.hip -o kp0
//                       hipcc --offload-arch=gfx950 -O2 -mllvm -amdgpu-kernarg-preload-count=8 kernarg_preload.hip -o kp1
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out, int v, int w) {
    if (threadIdx.x == 0) out[blockIdx.x] = v + w;
}
int main() {
    int* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        const int n = 20000;
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, i, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, i, 1);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%.3f us per launch\n", ms * 1e3 / n);
    }
    // the same through a graph of 64 kernel nodes
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 64; ++i) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, out, i, 1);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < 200; ++i) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("graph: %.3f us per kernel node\n", ms * 1e3 / (200 * 64));
    return 0;
}
