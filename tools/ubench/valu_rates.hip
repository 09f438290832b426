// Per-instruction issue cost on gfx950 (cycles per wave64 instruction per SIMD), one wave per SIMD
// and 4 waves per SIMD.  Dev tool: hipcc --offload-arch=gfx950 -O2 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define BODY(NAME, ASM)                                                                          \
    __global__ void k_##NAME(float* out, long long* cyc, int iters) {                            \
        float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, \
              a6 = a0 + 6, a7 = a0 + 7;                                                          \
        float b = 1.0001f, c = 0.5f;                                                              \
        double d0 = a0, d1 = a1, d2 = a2, d3 = a3;                                                \
        unsigned long long m = 0;                                                                 \
        long long t0 = __builtin_readcyclecounter();                                              \
        for (int i = 0; i < iters; ++i) { REP8(ASM) }                                             \
        long long t1 = __builtin_readcyclecounter();                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + (float)m; \
        if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                  \
    }

// each ASM body = 8 independent instructions; REP8 -> 64 per loop iteration
BODY(fma, asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
BODY(mul, asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
BODY(min, asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
BODY(xor, asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
BODY(rcp, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
BODY(cmp_vcc, asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
BODY(cmp_sgpr, asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[24:25], %2, %8\n v_cmp_lt_f32 s[26:27], %3, %8\n v_cmp_lt_f32 s[28:29], %4, %8\n v_cmp_lt_f32 s[30:31], %5, %8\n v_cmp_lt_f32 s[32:33], %6, %8\n v_cmp_lt_f32 s[34:35], %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35");)
BODY(cmp_cnd, asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
BODY(pk_fma, asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0), "v"(d1));)
BODY(pk_mul, asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0), "v"(d1));)
BODY(fma64, asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0), "v"(d1));)
BODY(divscale, asm volatile("v_div_scale_f32 %0, vcc, %0, %8, %0\n v_div_scale_f32 %1, vcc, %1, %8, %1\n v_div_scale_f32 %2, vcc, %2, %8, %2\n v_div_scale_f32 %3, vcc, %3, %8, %3\n v_div_scale_f32 %4, vcc, %4, %8, %4\n v_div_scale_f32 %5, vcc, %5, %8, %5\n v_div_scale_f32 %6, vcc, %6, %8, %6\n v_div_scale_f32 %7, vcc, %7, %8, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
BODY(divfmas, asm volatile("v_div_fmas_f32 %0, %0, %8, %9\n v_div_fmas_f32 %1, %1, %8, %9\n v_div_fmas_f32 %2, %2, %8, %9\n v_div_fmas_f32 %3, %3, %8, %9\n v_div_fmas_f32 %4, %4, %8, %9\n v_div_fmas_f32 %5, %5, %8, %9\n v_div_fmas_f32 %6, %6, %8, %9\n v_div_fmas_f32 %7, %7, %8, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
BODY(divfixup, asm volatile("v_div_fixup_f32 %0, %0, %8, %9\n v_div_fixup_f32 %1, %1, %8, %9\n v_div_fixup_f32 %2, %2, %8, %9\n v_div_fixup_f32 %3, %3, %8, %9\n v_div_fixup_f32 %4, %4, %8, %9\n v_div_fixup_f32 %5, %5, %8, %9\n v_div_fixup_f32 %6, %6, %8, %9\n v_div_fixup_f32 %7, %7, %8, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
BODY(sand, asm volatile("s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[24:25], s[24:25], s[22:23]\n s_and_b64 s[26:27], s[26:27], s[22:23]\n s_and_b64 s[28:29], s[28:29], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[24:25], s[24:25], s[22:23]\n s_and_b64 s[26:27], s[26:27], s[22:23]\n s_and_b64 s[28:29], s[28:29], s[22:23]\n" ::: "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","scc");)
BODY(mix_cmp_sand, asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n s_and_b64 s[20:21], s[20:21], s[22:23]\n v_cndmask_b32 %2, %2, %9, s[20:21]\n v_cmp_lt_f32 s[24:25], %3, %8\n v_cmp_lt_f32 s[26:27], %4, %8\n s_and_b64 s[24:25], s[24:25], s[26:27]\n v_cndmask_b32 %5, %5, %9, s[24:25]\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s20","s21","s22","s23","s24","s25","s26","s27","scc");)

typedef void (*kern_t)(float*, long long*, int);
struct K { const char* name; kern_t k; };

int main() {
    K ks[] = {{"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_min_f32", k_min}, {"v_xor_b32", k_xor}, {"v_rcp_f32", k_rcp},
              {"v_cmp->vcc", k_cmp_vcc}, {"v_cmp->sgpr", k_cmp_sgpr}, {"cmp+cndmask(x4 pairs)", k_cmp_cnd}, {"v_pk_fma_f32", k_pk_fma},
              {"v_pk_mul_f32", k_pk_mul}, {"v_fma_f64", k_fma64}, {"v_div_scale_f32", k_divscale}, {"v_div_fmas_f32", k_divfmas},
              {"v_div_fixup_f32", k_divfixup}, {"s_and_b64", k_sand}, {"2cmp+s_and+cndmask x2", k_mix_cmp_sand}};
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * 256 * 8 * 1024);
    hipMalloc(&cyc, sizeof(long long));
    const int iters = 2000;
    printf("%-28s %10s %10s %10s %10s  (ticks per instruction per SIMD)\n", "instruction", "1w/SIMD", "4w/SIMD", "8w(1024x2)", "8w(512x4)");
    for (auto& k : ks) {
        double r[4], gips[4];
        // 1 block of 256 (1 wave/SIMD), 1 block of 1024 (4 waves/SIMD), then 512 / 1024 blocks of 1024 threads over
        // 256 CUs (2 / 4 blocks per CU -> 8 / 16 waves per SIMD if registers allow; 16 is over the 8-wave cap)
        int thr[4] = {256, 1024, 1024, 512};
        int grd[4] = {1, 1, 512, 1024};
        double wps[4] = {1, 4, 8, 8};
        for (int c = 0; c < 4; ++c) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k.k, dim3(grd[c]), dim3(thr[c]), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k.k, dim3(grd[c]), dim3(thr[c]), 0, 0, out, cyc, iters);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            // wave-instructions per second per SIMD (grid spreads over 256 CUs x 4 SIMDs when grd >= 256)
            double waves = (double)grd[c] * thr[c] / 64.0;
            double simds = grd[c] >= 256 ? 1024.0 : (thr[c] >= 256 ? 4.0 : 1.0);
            gips[c] = waves * iters * 64.0 / (ms * 1e-3) / simds / 1e9;
            long long h; hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            r[c] = (double)h / (iters * 64.0) / wps[c];
        }
        printf("%-28s %6.2f %6.2f %6.2f %6.2f | G wave-inst/s/SIMD: %6.3f %6.3f %6.3f %6.3f\n", k.name, r[0], r[1], r[2], r[3], gips[0], gips[1], gips[2], gips[3]);
    }
    return 0;
}
