// Dev tool: largest absolute error of the hardware v_sin_f32 / v_cos_f32 (argument in turns, |x| <= 1/2) against float64 sin / cos --
// the scout pose of the step kernel (stage A / A2 culls only) is built on them.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/hw_sincos_error.hip -o build/hw_sincos_error
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(int n, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = ((double)i + 0.5) / n - 0.5;   // turns
    const float xf = (float)x;
    const double es = fabs((double)__builtin_amdgcn_sinf(xf) - sin(6.283185307179586476925 * (double)xf));
    const double ec = fabs((double)__builtin_amdgcn_cosf(xf) - cos(6.283185307179586476925 * (double)xf));
    out[i] = fmax(es, ec);
}
int main() {
    const int n = 1 << 26;
    double* d;
    hipMalloc(&d, sizeof(double) * n);
    k<<<n / 256, 256>>>(n, d);
    double* h = new double[n];
    hipMemcpy(h, d, sizeof(double) * n, hipMemcpyDeviceToHost);
    double m = 0, msmall = 0;
    for (int i = 0; i < n; ++i) {
        m = fmax(m, h[i]);
        const double x = ((double)i + 0.5) / n - 0.5;
        if (fabs(x) < 0.01) msmall = fmax(msmall, h[i]);
    }
    printf("v_sin_f32 / v_cos_f32, %d arguments in [-1/2, 1/2] turns: max abs error %.3e ; |x| < 0.01 turn: %.3e\n", n, m, msmall);
    return 0;
}
