// bf16x3.h -- float32 values as three bf16 pieces (a = a0 + a1 + a2 exactly: 8 + 8 + 8 significand bits), the operand form of the
// "bf16x3" products: a b ~ a2 b0 + a1 b1 + a0 b2 + a1 b0 + a0 b1 + a0 b0 on the bf16 MFMA, each piece product exact in float32,
// float32 accumulation, small terms first (csrc/ppo_mlp64.hip has the scheme, the order and the measurements behind it).
// Shared by the update kernels of the 64-wide heads (ppo_mlp64.hip) and of the 512-wide nets (ppo_resmlp512.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace bf16x3 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t cvt_pk_bf16(const float lo, const float hi) {   // v_cvt_pk_bf16_f32: round to nearest even
    const f32x2 v = {lo, hi};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    uint32_t u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}
// (a, b) -> three packed pieces (a in the low half): 11 vector instructions
__device__ __forceinline__ void split_pair(const float a, const float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
#ifdef X3_EXPERIMENT_NO_SPLIT   // (timing experiment: what the kernel costs without the splitting arithmetic; results are wrong)
    p0 = __float_as_uint(a); p1 = __float_as_uint(b); p2 = p0 ^ p1;
    return;
#endif
    p0 = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);   // exact
    p1 = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);   // exact, <= 8 bits left
    p2 = cvt_pk_bf16(sa, sb);
}
struct Pieces {
    uint4 p[3];   // eight values: piece i, elements (0, 1) (2, 3) (4, 5) (6, 7) as packed pairs
};
__device__ __forceinline__ Pieces split8(const float (&v)[8]) {
    Pieces P;
    split_pair(v[0], v[1], P.p[0].x, P.p[1].x, P.p[2].x);
    split_pair(v[2], v[3], P.p[0].y, P.p[1].y, P.p[2].y);
    split_pair(v[4], v[5], P.p[0].z, P.p[1].z, P.p[2].z);
    split_pair(v[6], v[7], P.p[0].w, P.p[1].w, P.p[2].w);
    return P;
}
__device__ __forceinline__ bf16x8 as_bf(const uint4 u) {
    bf16x8 v;
    __builtin_memcpy(&v, &u, 16);
    return v;
}

}  // namespace bf16x3
