// Does gfx950 interlock a vector-ALU write of a VGPR against an MFMA that reads it as srcB in the NEXT instruction?  (LLVM inserts no
// wait states for that pair; the hand-placed stream of csrc/ppo_mlp64_x3s.h puts asm MFMAs behind compiler-issued moves.)
// Each variant fills v[100:103] with junk, then writes the true B operand with the named instruction(s) right in front of the MFMA and
// compares the product with one computed from operands written long before.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_mfma_hazard.hip -o build/valu_mfma_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define JUNK "v_mov_b32 v100, %[j]\n\tv_mov_b32 v101, %[j]\n\tv_mov_b32 v102, %[j]\n\tv_mov_b32 v103, %[j]\n\ts_nop 7\n\ts_nop 7\n\t"
#define MF16 "v_mfma_f32_16x16x32_bf16 %[acc], %[a], v[100:103], %[acc]\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
#define MF32 "v_mfma_f32_32x32x16_bf16 %[acc], %[a], v[100:103], %[acc]\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
#define OPS(T) : [acc] "+v"(got) : [a] "v"(a), [b0] "v"(b.x), [b1] "v"(b.y), [b2] "v"(b.z), [b3] "v"(b.w), [b01] "v"(b01), [b23] "v"(b23), [j] "v"(junk) : "v100", "v101", "v102", "v103"
#define MOV4 "v_mov_b32 v100, %[b0]\n\tv_mov_b32 v101, %[b1]\n\tv_mov_b32 v102, %[b2]\n\tv_mov_b32 v103, %[b3]\n\t"
#define MOV64 "v_mov_b64 v[100:101], %[b01]\n\tv_mov_b64 v[102:103], %[b23]\n\t"

template <int V, class ACC>
__device__ void run(ACC& got, u32x4 a, u32x4 b, unsigned junk) {
    const u32x2 b01 = {b.x, b.y}, b23 = {b.z, b.w};
    constexpr bool W = sizeof(ACC) == 64;
    if (V == 0) { if (W) asm volatile(JUNK MOV4 "s_nop 7\n\ts_nop 7\n\t" MF32 OPS()); else asm volatile(JUNK MOV4 "s_nop 7\n\ts_nop 7\n\t" MF16 OPS()); }   // reference: far apart
    if (V == 1) { if (W) asm volatile(JUNK MOV4 MF32 OPS()); else asm volatile(JUNK MOV4 MF16 OPS()); }
    if (V == 2) { if (W) asm volatile(JUNK MOV64 MF32 OPS()); else asm volatile(JUNK MOV64 MF16 OPS()); }
    if (V == 3) { if (W) asm volatile(JUNK MOV64 "s_nop 0\n\t" MF32 OPS()); else asm volatile(JUNK MOV64 "s_nop 0\n\t" MF16 OPS()); }
    if (V == 4) { if (W) asm volatile(JUNK MOV64 "s_nop 1\n\t" MF32 OPS()); else asm volatile(JUNK MOV64 "s_nop 1\n\t" MF16 OPS()); }
    if (V == 5) { if (W) asm volatile(JUNK MOV4 "s_nop 0\n\t" MF32 OPS()); else asm volatile(JUNK MOV4 "s_nop 0\n\t" MF16 OPS()); }
    if (V == 6) {   // cvt_pk / sub results (the split's instructions) straight into the operand
        if (W) asm volatile(JUNK "v_mov_b32 v100, %[b0]\n\tv_mov_b32 v101, %[b1]\n\tv_mov_b32 v102, %[b2]\n\ts_nop 7\n\tv_add_u32 v103, %[b3], 0\n\t" MF32 OPS());
        else asm volatile(JUNK "v_mov_b32 v100, %[b0]\n\tv_mov_b32 v101, %[b1]\n\tv_mov_b32 v102, %[b2]\n\ts_nop 7\n\tv_add_u32 v103, %[b3], 0\n\t" MF16 OPS());
    }
    if (V == 7) {   // through an AGPR: v_accvgpr_read in front of the MFMA
        if (W) asm volatile(JUNK "v_mov_b32 v100, %[b0]\n\tv_mov_b32 v101, %[b1]\n\tv_mov_b32 v102, %[b2]\n\tv_accvgpr_write_b32 a200, %[b3]\n\ts_nop 7\n\tv_accvgpr_read_b32 v103, a200\n\t" MF32 OPS());
        else asm volatile(JUNK "v_mov_b32 v100, %[b0]\n\tv_mov_b32 v101, %[b1]\n\tv_mov_b32 v102, %[b2]\n\tv_accvgpr_write_b32 a200, %[b3]\n\ts_nop 7\n\tv_accvgpr_read_b32 v103, a200\n\t" MF16 OPS());
    }
}

template <int V>
__global__ void k(const u32x4* __restrict__ in, float* __restrict__ out) {
    const int lane = threadIdx.x;
    const u32x4 a = in[lane], b = in[64 + lane];
    const unsigned junk = in[128 + lane].x;
    f32x4 g16 = {0, 0, 0, 0};
    f32x16 g32 = {0};
    run<V>(g16, a, b, junk);
    run<V>(g32, a, b, junk);
    for (int r = 0; r < 4; ++r) out[lane * 20 + r] = g16[r];
    for (int r = 0; r < 16; ++r) out[lane * 20 + 4 + r] = g32[r];
}

template <int V>
void go(const u32x4* in, float* out, std::vector<float>& h) {
    hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, in, out);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
}
int main() {
    std::vector<unsigned> h(192 * 4);
    unsigned s = 12345;
    for (auto& w : h) {   // pairs of finite bf16 values of moderate size
        s = s * 1664525u + 1013904223u;
        const unsigned lo = 0x3f00u + ((s >> 8) & 0xff) + ((s >> 3) & 0x8000), hi = 0x3f00u + ((s >> 16) & 0xff) + ((s >> 1) & 0x8000);
        w = lo | (hi << 16);
    }
    for (int i = 128 * 4; i < 192 * 4; ++i) h[i] = 0x7f807f80u;   // junk: +inf pairs
    u32x4* in;
    float* out;
    hipMalloc(&in, h.size() * 4);
    hipMalloc(&out, 64 * 20 * 4);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> ref(64 * 20), got(64 * 20);
    go<0>(in, out, ref);
    const char* names[] = {"reference (16 wait states)", "4 x v_mov_b32, MFMA next", "2 x v_mov_b64, MFMA next", "2 x v_mov_b64, s_nop 0", "2 x v_mov_b64, s_nop 1",
                           "4 x v_mov_b32, s_nop 0", "v_add_u32 of the last dword, MFMA next", "v_accvgpr_read of the last dword, MFMA next"};
    auto cmp = [&](int v) {
        int bad16 = 0, bad32 = 0;
        for (int l = 0; l < 64; ++l) {
            for (int r = 0; r < 4; ++r) bad16 += memcmp(&ref[l * 20 + r], &got[l * 20 + r], 4) != 0;
            for (int r = 0; r < 16; ++r) bad32 += memcmp(&ref[l * 20 + 4 + r], &got[l * 20 + 4 + r], 4) != 0;
        }
        printf("  %-45s 16x16x32: %3d of 256 outputs differ   32x32x16: %4d of 1024 differ\n", names[v], bad16, bad32);
    };
    go<1>(in, out, got); cmp(1);
    go<2>(in, out, got); cmp(2);
    go<3>(in, out, got); cmp(3);
    go<4>(in, out, got); cmp(4);
    go<5>(in, out, got); cmp(5);
    go<6>(in, out, got); cmp(6);
    go<7>(in, out, got); cmp(7);
    printf("  (reference sample: %g %g)\n", ref[0], ref[4]);
    return 0;
}
