// ppo_mlp64_x3s.h -- round 6: the split-bf16 update pass of the 16-64-64 heads as ONE hand-placed instruction stream per wave.
// Included by ppo_mlp64.hip inside its anonymous namespace (uses Layout, XPad, kpos, wslot, split_pair, the pre-split observation
// buffer of mlp64_split_obs and the partial-row / reduce_adam contract of the other passes).  Same arithmetic as pass_body_x3 -- float32
// products from three exact bf16 pieces per operand, six piece products, small terms first, float32 accumulate -- reorganised
// around what round 5's counters and tools/ubench/gen_mfma_stream.py (profiles/r06_mfma_stream.txt) say about gfx950:
//   * ONE wave per SIMD hides up to ~6 vector instructions behind each v_mfma_f32_32x32x16_bf16 of its OWN stream (33.5 -> 37
//     cycles per MFMA slot), every further one costs ~4 cycles; two waves per SIMD do no better per SIMD, and the compiler
//     clusters MFMAs and vector work instead of interleaving them.  So: 4 waves x 512 registers (no spills; round 5's 8 x 256
//     spilled 130), every MFMA and every block of the splits an `asm volatile` statement -- the compiler allocates registers, inserts
//     the s_waitcnt of the LDS / global loads it still issues itself and loads MFMA operands straight into AGPRs, but cannot
//     re-order the stream -- with 5-6 vector instructions written behind each MFMA.
//   * H1 and dH2 are split ONCE (round 5: twice, 352 of 880 split instructions per tile): the bf16 pieces feed F2 / B2 from
//     registers (k = hidden unit), are stored to a wave-private LDS image [piece][sample][unit] and come back transposed for the
//     weight-gradient products (k = sample) through ds_read_b64_tr_b16; dH1 (G1's operand) takes the same way.  No float32 tile,
//     no second split, no wave_lds_fence round trips.
//   * the sample sums db2, dW3, dW4 are per-lane partial sums over the wave's tiles (a lane owns the same sample slot in every
//     tile), reduced across lanes once per launch; db1 rides on G1's operand (one MFMA with a ones operand per piece).
// MFMA -> vector hazards: the compiler does not see inside the asm statements, so every chain whose accumulators the vector unit
// reads next ends in `settle` (12 wait states; gfx950 needs 11 after an 8-pass XDL op).
//
// LDS image of a 32-sample x 64-unit matrix of pieces: row m = 128 bytes, 8-byte slots ("quads": 4 units) XOR-swizzled by
//   f(m) = (m & 3) | ((m >> 2 ^ m >> 3) & 1) << 2 | ((m >> 1 ^ m >> 3) & 1) << 3
// -- conflict-free for the ds_write_b64 of 16 consecutive sample lanes, for the 4-row x 64-byte gathers of the 32x32x16 operands
// and for the 8-row x 32-byte gathers of the 16x16x32 operands.
#pragma once

namespace x3s {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int SW = 4;                    // waves per workgroup: one per SIMD, 512 registers each
constexpr int kPieceB = 32 * 128;        // one piece of an image: [32 samples][64 units] bf16
constexpr int kImgB = 3 * kPieceB;

struct SmemS {
    uint16_t W2p[3][H][H];               // as SmemX: rows = layer-2 units, k = layer-1 units (k-permuted, slots swizzled)
    uint16_t W2Tp[3][H][H];
    uint16_t W1p[3][H][16];
    float b1[H], b2[H], w3[H], w4[H];
    unsigned char img[SW][2][kImgB];     // per wave: [0] H1 pieces, [1] dH2 pieces, then dH1 pieces; at the end the reduction rows
};
static_assert(sizeof(SmemS) <= 160 * 1024, "LDS");
static_assert(offsetof(SmemS, img) % 128 == 0 && kImgB % 128 == 0, "image bases keep the low 7 address bits of a lane offset (the XOR swizzles)");
static_assert(sizeof(SmemS::img) >= 4 * ((Layout<16>::P_ACTOR + 6) & ~3) * sizeof(float), "reduction rows fit the images");
static_assert(sizeof(SmemS::img) / SW >= 96 * 64 * sizeof(float), "lane partials fit a wave's images");

// ---- the stream's building blocks.  Constraint letters: v = VGPR, a = AGPR (the compiler loads LDS / global operands straight into AGPRs)
// gfx950 does not interlock a vector-ALU write of a register against an MFMA that reads it as srcA / srcB in the NEXT instruction
// (tools/ubench/valu_mfma_hazard.hip: wrong products; one wait state cures it), and the compiler, which cannot see the MFMA inside
// the asm statement, may put a copy (v_mov, v_accvgpr_write / read) right in front of it.  A wait state in front of EVERY MFMA
// costs 17 % of the epoch (770 against 658 us: an s_nop takes an issue slot of the one wave), so: the MFMAs that open a phase --
// where operands arrive from compiler-issued loads and copies -- carry it (the ..G variants), inside a phase every operand is
// written by the stream itself at least one MFMA earlier, and tools/verify/mfma_hazard_lint.py checks the LISTING for both hazard
// kinds at every build of the test suite (tests/test_isa_lint_cpu.py): a compiler that places a copy elsewhere fails the test.
#ifndef X3S_GUARD
#define X3S_GUARD ""   // ("s_nop 0\n\t": the guard everywhere, for A/B runs)
#endif
#define X3S_G "s_nop 0\n\t"
#define X3S_MFMA32_V_AV(acc, A, B) asm volatile(X3S_GUARD "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(A), "v"(B))
#define X3S_MFMA32_V_AA(acc, A, B) asm volatile(X3S_GUARD "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(A), "a"(B))
#define X3S_MFMA32G_V_AV(acc, A, B) asm volatile(X3S_G "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(A), "v"(B))
#define X3S_MFMA32G_V_AA(acc, A, B) asm volatile(X3S_G "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(A), "a"(B))
#define X3S_MFMA32_VZ_AV(acc, A, B) asm volatile(X3S_G "v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(A), "v"(B))
#define X3S_MFMA32_VZ_AA(acc, A, B) asm volatile(X3S_G "v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(A), "a"(B))
#define X3S_MFMA32_A_AV(acc, A, B) asm volatile(X3S_GUARD "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(A), "v"(B))
#define X3S_MFMA32_AZ_AV(acc, A, B) asm volatile(X3S_G "v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(acc) : "a"(A), "v"(B))
#define X3S_MFMA32_A_AA(acc, A, B) asm volatile(X3S_GUARD "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(A), "a"(B))
#define X3S_MFMA16_A_AA(acc, A, B) asm volatile(X3S_GUARD "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(A), "a"(B))
#define X3S_MFMA16G_A_AA(acc, A, B) asm volatile(X3S_G "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(A), "a"(B))
#define X3S_MFMA16_A_AV(acc, A, B) asm volatile(X3S_GUARD "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(A), "v"(B))

// the vector unit may read / overwrite these accumulators from here on (8-pass XDL write -> VALU: 11 wait states)
__device__ __forceinline__ void settle(f32x16& a, f32x16& b) { asm volatile("s_nop 7\n\ts_nop 3" : "+v"(a), "+v"(b)); }

// (a, b) -> p0 and the residuals: 5 instructions
__device__ __forceinline__ void split_a(unsigned& p0, float& ra, float& rb, const float a, const float b) {
    asm volatile("v_cvt_pk_bf16_f32 %0, %3, %4\n\t"
                 "v_lshlrev_b32 %1, 16, %0\n\t"
                 "v_and_b32 %2, 0xffff0000, %0\n\t"
                 "v_sub_f32 %1, %3, %1\n\t"
                 "v_sub_f32 %2, %4, %2"
                 : "=&v"(p0), "=&v"(ra), "=&v"(rb) : "v"(a), "v"(b));
}
// residuals -> p1, p2: 6 instructions
__device__ __forceinline__ void split_b(unsigned& p1, unsigned& p2, const float ra, const float rb) {
    float t0, t1;
    asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n\t"
                 "v_lshlrev_b32 %2, 16, %0\n\t"
                 "v_and_b32 %3, 0xffff0000, %0\n\t"
                 "v_sub_f32 %2, %4, %2\n\t"
                 "v_sub_f32 %3, %5, %3\n\t"
                 "v_cvt_pk_bf16_f32 %1, %2, %3"
                 : "=&v"(p1), "=&v"(p2), "=&v"(t0), "=&v"(t1) : "v"(ra), "v"(rb));
}
// four accumulator registers: x = relu(x + b) -- 8 instructions (v_max_i32 on the bit pattern: relu_bits)
// four accumulator registers: h = relu(c) -- v_max_i32 on the bit pattern (relu_bits).  The bias is already in c: the accumulators of F1 /
// F2 START as the bias (as in the f32 pass), so sum and bias are rounded ONCE.  Adding the bias to the rounded sum afterwards -- rounds
// 1-5 did -- leaves a systematic error of ~2e-8 in the heads' pre-activations (the mean over samples does not average out; measured and
// modelled: tools/verify/x3_forward_error.py), which the actor's gradient sums amplify by their cancellation.
// (Accumulator values live on as SCALARS behind the MFMA chain: writing an asm result back into an element of the 16-register tuple
// costs a v_mov per value.)
__device__ __forceinline__ void relu4(float (&h)[16], const f32x16& c, const int g) {
    float y0, y1, y2, y3;
    asm volatile("v_max_i32 %0, %4, 0\n\tv_max_i32 %1, %5, 0\n\tv_max_i32 %2, %6, 0\n\tv_max_i32 %3, %7, 0"
                 : "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3) : "v"(c[4 * g]), "v"(c[4 * g + 1]), "v"(c[4 * g + 2]), "v"(c[4 * g + 3]));
    h[4 * g] = y0; h[4 * g + 1] = y1; h[4 * g + 2] = y2; h[4 * g + 3] = y3;
}
// one value of the head's backward: h = H2 (>= 0) -> dH2 in place; the lane's partial sums  (7 / 5 instructions)
//   pw3 += h g3 ; pw4 += h g4 ; d = g3 w3 + g4 w4 (fma(g3, w3, g4 w4)) ; h = h > 0 ? d : 0 ; pb2 += h
template <bool ACTOR>
__device__ __forceinline__ void dh2_value(float& h, float& pw3, float& pw4, float& pb2, const float g3, const float g4, const float w3,
                                          const float w4) {
    float d;
    if constexpr (ACTOR)
        asm volatile("v_fmac_f32 %1, %0, %5\n\t"
                     "v_fmac_f32 %2, %0, %6\n\t"
                     "v_mul_f32 %4, %6, %8\n\t"
                     "v_fmac_f32 %4, %5, %7\n\t"
                     "v_cmp_lt_f32 vcc, 0, %0\n\t"
                     "v_cndmask_b32 %0, 0, %4, vcc\n\t"
                     "v_add_f32 %3, %3, %0"
                     : "+v"(h), "+v"(pw3), "+v"(pw4), "+v"(pb2), "=&v"(d) : "v"(g3), "v"(g4), "v"(w3), "v"(w4) : "vcc");
    else
        asm volatile("v_fmac_f32 %1, %0, %4\n\t"
                     "v_mul_f32 %3, %4, %5\n\t"
                     "v_cmp_lt_f32 vcc, 0, %0\n\t"
                     "v_cndmask_b32 %0, 0, %3, vcc\n\t"
                     "v_add_f32 %2, %2, %0"
                     : "+v"(h), "+v"(pw3), "+v"(pb2), "=&v"(d) : "v"(g3), "v"(w3) : "vcc");
}

// y_k = (bf16 half k of the words (w0, w1) != 0) ? x_k : 0 -- the relu mask from packed leading pieces: 8 instructions
__device__ __forceinline__ void mask4(float& y0, float& y1, float& y2, float& y3, const float x0, const float x1, const float x2,
                                      const float x3, const unsigned w0, const unsigned w1) {
    asm volatile("v_cmp_ne_u16 vcc, 0, %8\n\t"
                 "v_cndmask_b32 %0, 0, %4, vcc\n\t"
                 "v_cmp_lt_u32 vcc, 0xffff, %8\n\t"
                 "v_cndmask_b32 %1, 0, %5, vcc\n\t"
                 "v_cmp_ne_u16 vcc, 0, %9\n\t"
                 "v_cndmask_b32 %2, 0, %6, vcc\n\t"
                 "v_cmp_lt_u32 vcc, 0xffff, %9\n\t"
                 "v_cndmask_b32 %3, 0, %7, vcc"
                 : "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(w0), "v"(w1) : "vcc");
}

struct P3 {   // one k-step's operand: eight values as three packed pieces
    u32x4 p[3];
};

__device__ __forceinline__ int fswz(const int m) { return (m & 3) | ((((m >> 2) ^ (m >> 3)) & 1) << 2) | ((((m >> 1) ^ (m >> 3)) & 1) << 3); }

typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
__device__ __forceinline__ u32x2 tr_read(const unsigned char* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)p);
    u32x2 r;
    __builtin_memcpy(&r, &v, 8);
    return r;
}
__device__ __forceinline__ u32x4 cat(const u32x2 lo, const u32x2 hi) { return u32x4{lo.x, lo.y, hi.x, hi.y}; }

template <bool ACTOR>
__device__ __forceinline__ void pass_body(SmemS& sm, const float* __restrict__ params, const unsigned char* __restrict__ prep,
                                          const float* __restrict__ act, const float* __restrict__ logp_old,
                                          const float* __restrict__ rtg, const float* __restrict__ adv, long long M, float var,
                                          float clip, float inv_n, float* __restrict__ partial, float* __restrict__ stats_partial) {
    using L = Layout<16>;
    constexpr int OFF_W1 = L::OFF_W1, OFF_B1 = L::OFF_B1, OFF_W2 = L::OFF_W2, OFF_B2 = L::OFF_B2, OFF_W3 = L::OFF_W3, OFF_B3 = L::OFF_B3,
                  OFF_W4 = L::OFF_W4, OFF_B4 = L::OFF_B4;
    constexpr int P = ACTOR ? L::P_ACTOR : L::P_CRITIC;
    constexpr int NT = 64 * SW, kTileBytes = XPad<16>::kTileBytes, kRowsBytes = XPad<16>::kRowsBytes;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5, l15 = lane & 15, kk = lane >> 4;

    // ---- weights -> bf16 pieces in LDS (once per launch), as pass_body_x3
    for (int k = tid; k < H * H / 2; k += NT) {
        const int r = (2 * k) / H, c = (2 * k) % H;
        uint32_t p0, p1, p2;
        bf16x3::split_pair(params[OFF_W2 + r * H + c], params[OFF_W2 + r * H + c + 1], p0, p1, p2);
        const uint32_t pc[3] = {p0, p1, p2};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            *reinterpret_cast<uint32_t*>(&sm.W2p[i][0][0] + wslot(r, kpos(c))) = pc[i];
            (&sm.W2Tp[i][0][0])[wslot(c, kpos(r))] = (uint16_t)(pc[i] & 0xffffu);
            (&sm.W2Tp[i][0][0])[wslot(c + 1, kpos(r))] = (uint16_t)(pc[i] >> 16);
        }
    }
    for (int k = tid; k < H * 16 / 2; k += NT) {
        const int r = (2 * k) / 16, c = (2 * k) % 16;
        uint32_t p0, p1, p2;
        bf16x3::split_pair(params[OFF_W1 + r * 16 + c], params[OFF_W1 + r * 16 + c + 1], p0, p1, p2);
        *reinterpret_cast<uint32_t*>(&sm.W1p[0][r][c]) = p0;
        *reinterpret_cast<uint32_t*>(&sm.W1p[1][r][c]) = p1;
        *reinterpret_cast<uint32_t*>(&sm.W1p[2][r][c]) = p2;
    }
    if (tid < H) {
        sm.b1[tid] = params[OFF_B1 + tid];
        sm.b2[tid] = params[OFF_B2 + tid];
        sm.w3[tid] = params[OFF_W3 + tid];
        sm.w4[tid] = ACTOR ? params[OFF_W4 + tid] : 0.f;
    }
    const float b3 = params[OFF_B3];
    const float b4 = ACTOR ? params[OFF_B4] : 0.f;
    __syncthreads();

    // the wave's images as byte offsets into sm: H1 pieces at img0, dH2 (then dH1) pieces at img0 + kImgB -- the image base is folded
    // into every lane address below (it is a multiple of 128: the XOR swizzles of the low 7 bits pass through), the second image,
    // the piece and the k-step are immediate offsets of the DS instructions
    unsigned char* const smb = reinterpret_cast<unsigned char*>(&sm);
    const int img0 = (int)offsetof(SmemS, img) + wave * 2 * kImgB;
    constexpr int IMG_H = 0, IMG_D = kImgB;
    // ---- lane addresses (everything else is an immediate offset)
    // piece stores: lane (m = l31, lhi) owns the quads 8 t + 4 j + 2 b + lhi of row m
    const int st_base = img0 + l31 * 128 + ((fswz(l31) ^ lhi) << 3);
    auto st_off = [&](const int t, const int j, const int b) { return st_base ^ ((8 * t + 4 * j + 2 * b) << 3); };
    // 32x32x16 operand gathers: rows 16 s + 8 lhi + 4 rd + (l15 >> 2), quads 8 t + 4 ((lane >> 4) & 1) + (lane & 3)
    const int g_row = 8 * lhi + (l15 >> 2), g_quad = 4 * ((lane >> 4) & 1) + (lane & 3);
    const int g_base0 = img0 + g_row * 128 + ((g_quad ^ fswz(g_row)) << 3), g_base1 = img0 + (g_row + 4) * 128 + ((g_quad ^ fswz(g_row + 4)) << 3);
    auto g_off = [&](const int t, const int s, const int rd) { return ((rd ? g_base1 : g_base0) ^ (t << 6)) + s * 2048; };   // (two of the four are formed at the use)
    // 16x16x32 operand gathers: rows 8 kk + 4 rd + (l15 >> 2), quads 4 u + (lane & 3)
    const int h_row = 8 * kk + (l15 >> 2), h_quad = lane & 3;
    const int h_base0 = img0 + h_row * 128 + ((h_quad ^ fswz(h_row)) << 3), h_base1 = img0 + (h_row + 4) * 128 + ((h_quad ^ fswz(h_row + 4)) << 3);
    // weight rows: A operand of (row tile tt, k-step s, piece i)
    auto ldw = [&](const uint16_t* base, const int tt, const int s, const int i) {
        return *reinterpret_cast<const u32x4*>(base + i * H * H + wslot(32 * tt + l31, 16 * s + 8 * lhi));
    };
    const int vec_off = 4 * lhi;
    auto ldv = [&](const float* v, const int t, const int g) { return *reinterpret_cast<const v4f*>(v + 32 * t + 8 * g + vec_off); };

    // ---- accumulators that persist over this wave's tiles
    f32x16 aW2[2][2];
    v4f aW1[4], aB1[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) aW2[a][b] = zero16();
#pragma unroll
    for (int u = 0; u < 4; ++u) aW1[u] = aB1[u] = v4f{0.f, 0.f, 0.f, 0.f};
    float pb2[2][16], pw3[2][16], pw4[2][16];   // per-lane partial sums over the tiles: db2, dW3, dW4 of the lane's 32 units
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) pb2[t][r] = pw3[t][r] = pw4[t][r] = 0.f;
    float adb3 = 0.f, adb4 = 0.f, st0 = 0.f, st1 = 0.f, st2 = 0.f, st3 = 0.f;
    u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};   // bf16 1.0 in every k-slot
    asm volatile("" : "+a"(ones));   // lives in AGPRs for the whole launch

    const long long n_tiles = (M + 31) / 32;
    const long long gw = (long long)blockIdx.x * SW + wave, stride = (long long)gridDim.x * SW;
    // Operands that come from global memory -- the tile's observation rows (F1's B operand) and columns (G1's B operand), pre-split by
    // mlp64_split_obs -- are requested by asm loads straight into AGPRs (a compiler-issued load lands in vector registers, which are full,
    // and is copied over right in front of the MFMA: the hazard above) long before their use; `landed` is their s_waitcnt, placed a
    // few thousand cycles later and tied to the registers so that nothing reads or copies them before.  (Hidden loads are safe for the
    // compiler's own vmcnt bookkeeping: loads return in order, extra outstanding ones only make its waits longer.)
    u32x4 xp[3], xbn[3];   // rows of the NEXT tile ; columns of THIS tile (G1 runs one tile late)
    float pre_a0 = 0.f, pre_a1 = 0.f, pre_lp = 0.f, pre_t = 0.f;
    auto request_tile = [&](const long long next, const long long cur) {   // both clamped by the caller: always valid tiles
        const unsigned char* r = prep + (size_t)next * kTileBytes + ((l31 * 3) * 16 + 8 * lhi) * 2;   // rows past the batch are zero in `prep`
        const unsigned char* c = prep + (size_t)cur * kTileBytes + kRowsBytes + (l15 * 32 + 8 * kk) * 2;   // lane (f = l15, kk): samples 8 kk .. + 7
        asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:32\n\tglobal_load_dwordx4 %2, %3, off offset:64"
                     : "=a"(xp[0]), "=a"(xp[1]), "=a"(xp[2]) : "v"(r));
        asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:1024\n\tglobal_load_dwordx4 %2, %3, off offset:2048"
                     : "=a"(xbn[0]), "=a"(xbn[1]), "=a"(xbn[2]) : "v"(c));
        const long long m = next * 32 + l31;
        if (m < M) {
            if (ACTOR) {
                const float2 a = reinterpret_cast<const float2*>(act)[m];
                pre_a0 = a.x;
                pre_a1 = a.y;
                pre_lp = logp_old[m];
                pre_t = adv[m];
            } else {
                pre_t = rtg[m];
            }
        }
    };
    auto landed = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" : "+a"(xp[0]), "+a"(xp[1]), "+a"(xp[2]), "+a"(xbn[0]), "+a"(xbn[1]), "+a"(xbn[2]));
    };
    static_assert(kTileBytes == 6144 && kRowsBytes == 3072, "offsets of the asm loads: pieces 32 bytes apart in a row, 1024 in the columns");

    // G1 of the tile whose dH1 pieces are in imgD: dW1 += dH1^T X (16x16x32, one k-step), db1 += dH1^T 1 -- as a stream of 36 MFMAs
    // (step k of 0..35: unit pair k / 18) that the caller places one by one behind its vector work; operands gathered up front
    struct G1Ops {
        P3 da[4];
    };
    // pair q of 0..11: (unit tile u = q / 3, piece i = q % 3) -- two 8-byte gathers; the caller spreads the twelve over MFMA slots
    auto g1_gather_pair = [&](G1Ops& o, const int q, const int hb0, const int hb1) {
        const int u = q / 3, i = q % 3;
        o.da[u].p[i] = cat(tr_read(smb + (hb0 ^ (u << 5)) + IMG_D + i * kPieceB), tr_read(smb + (hb1 ^ (u << 5)) + IMG_D + i * kPieceB));
    };
    auto g1_step = [&](const int k, const G1Ops& o, const u32x4 (&xb)[3], const bool guard = false) {
        const int t1 = k / 18, q = k % 18, u = 2 * t1 + (q & 1), e = q >> 1;   // e of 0..8: the pair's e-th product
        // products in the order of mma6_acc16_2 (small terms first), the ones-products of db1 between them
        constexpr int kind[9] = {0, 1, 0, 0, 1, 0, 0, 1, 0};    // 1: ones product
        constexpr int ai[9] = {2, 2, 1, 0, 1, 1, 0, 0, 0};      // piece of dH1
        constexpr int bi[9] = {0, 0, 1, 2, 0, 0, 1, 0, 0};      // piece of X (kind 0)
        if (guard) {   // (the tail after the tile loop: the compiler moves the accumulators around there -- every MFMA waits one state)
            if (kind[e]) { X3S_MFMA16G_A_AA(aB1[u], o.da[u].p[ai[e]], ones); }
            else { X3S_MFMA16G_A_AA(aW1[u], o.da[u].p[ai[e]], xb[bi[e]]); }
        } else {
            if (kind[e]) { X3S_MFMA16_A_AA(aB1[u], o.da[u].p[ai[e]], ones); }
            else { X3S_MFMA16_A_AA(aW1[u], o.da[u].p[ai[e]], xb[bi[e]]); }
        }
    };
    // one k-step's pieces into both 8-byte slots of its two quads (6 stores)
    auto store_pieces = [&](const int img, const int t, const int j, const P3& q) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            *reinterpret_cast<u32x2*>(smb + st_off(t, j, 0) + img + i * kPieceB) = u32x2{q.p[i].x, q.p[i].y};
            *reinterpret_cast<u32x2*>(smb + st_off(t, j, 1) + img + i * kPieceB) = u32x2{q.p[i].z, q.p[i].w};
        }
    };
    // the eight values c[8 j .. 8 j + 7] -> pieces, as 8 blocks (block b of 0..7: pair b >> 1, first / second half)
    struct SplitState {
        float ra[4], rb[4];
    };
    auto split_block = [&](const float (&c)[16], const int j, const int b, P3& q, SplitState& ss) {
        const int pr = b >> 1;
        if ((b & 1) == 0) {
            unsigned p0;
            split_a(p0, ss.ra[pr], ss.rb[pr], c[8 * j + 2 * pr], c[8 * j + 2 * pr + 1]);
            q.p[0][pr] = p0;
        } else {
            unsigned p1, p2;
            split_b(p1, p2, ss.ra[pr], ss.rb[pr]);
            q.p[1][pr] = p1;
            q.p[2][pr] = p2;
        }
    };

    // imgD starts as zeros: the first tile's "previous G1" then adds nothing (the stream below has no special first iteration)
    for (int k = lane; k < kImgB / 16; k += 64) reinterpret_cast<u32x4*>(smb + img0 + IMG_D)[k] = u32x4{0u, 0u, 0u, 0u};
    bool have_prev = false;
    if (gw < n_tiles) {
        request_tile(gw, gw);   // (the first tile's "previous" columns: any finite values -- they meet zero pieces)
        landed();
    }
    for (long long tile = gw; tile < n_tiles; tile += stride) {
        const bool valid = tile * 32 + l31 < M;
        const u32x4 xr[3] = {xp[0], xp[1], xp[2]}, xb[3] = {xbn[0], xbn[1], xbn[2]};
        const float cur_a0 = pre_a0, cur_a1 = pre_a1, cur_lp = pre_lp, cur_t = pre_t;

        // ================================================================ F1: H1^T = relu(b1 + W1 X^T), one k-step
        f32x16 c1[2];
#pragma unroll
        for (int q = 0; q < 8; ++q) {   // the accumulators start as the bias (one rounding of sum + bias)
            const v4f b = ldv(sm.b1, q >> 2, q & 3);
            c1[q >> 2][4 * (q & 3)] = b.x; c1[q >> 2][4 * (q & 3) + 1] = b.y; c1[q >> 2][4 * (q & 3) + 2] = b.z; c1[q >> 2][4 * (q & 3) + 3] = b.w;
        }
        {
            u32x4 wa[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 3; ++i) wa[t][i] = *reinterpret_cast<const u32x4*>(&sm.W1p[i][32 * t + l31][8 * lhi]);
            X3S_MFMA32G_V_AA(c1[0], wa[0][2], xr[0]); X3S_MFMA32G_V_AA(c1[1], wa[1][2], xr[0]);
            X3S_MFMA32_V_AA(c1[0], wa[0][1], xr[1]);  X3S_MFMA32_V_AA(c1[1], wa[1][1], xr[1]);
            X3S_MFMA32_V_AA(c1[0], wa[0][0], xr[2]);  X3S_MFMA32_V_AA(c1[1], wa[1][0], xr[2]);
            X3S_MFMA32_V_AA(c1[0], wa[0][1], xr[0]);  X3S_MFMA32_V_AA(c1[1], wa[1][1], xr[0]);
            X3S_MFMA32_V_AA(c1[0], wa[0][0], xr[1]);  X3S_MFMA32_V_AA(c1[1], wa[1][0], xr[1]);
            X3S_MFMA32_V_AA(c1[0], wa[0][0], xr[0]);  X3S_MFMA32_V_AA(c1[1], wa[1][0], xr[0]);
        }
        // the next tile's rows and this tile's columns: requested here (F1 has read the registers they land in), waited for at the end
        // of the tile
        request_tile(tile + stride < n_tiles ? tile + stride : tile, tile);
        // first F2 operands while F1 drains
        u32x4 wn[2][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            wn[0][i] = ldw(&sm.W2p[0][0][0], 0, 0, i);
            wn[1][i] = ldw(&sm.W2p[0][0][0], 1, 0, i);
        }
        settle(c1[0], c1[1]);
        float h1[2][16];   // H1 = relu(b1 + ..): the values the split reads
#pragma unroll
        for (int q = 0; q < 8; ++q) relu4(h1[q >> 2], c1[q >> 2], q & 3);
        f32x16 c2[2];      // F2's accumulators start as b2: requested here, landed long before the first MFMA
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const v4f b = ldv(sm.b2, q >> 2, q & 3);
            c2[q >> 2][4 * (q & 3)] = b.x; c2[q >> 2][4 * (q & 3) + 1] = b.y; c2[q >> 2][4 * (q & 3) + 2] = b.z; c2[q >> 2][4 * (q & 3) + 3] = b.w;
        }

        // ================================================================ F2: H2^T = relu(b2 + W2 H1^T); H1 split once, k-step s = (t1, j)
        G1Ops g1o;
        {
            P3 hb[4];
            u32x4 wk[4][2];   // the leading weight pieces again, for the big terms: requested behind k-step 3 (its slots carry no split)
            SplitState ss;
#pragma unroll
            for (int b = 0; b < 8; ++b) split_block(h1[0], 0, b, hb[0], ss);   // k-step 0: nothing to hide it behind
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                u32x4 w[2][3];
#pragma unroll
                for (int i = 0; i < 3; ++i) { w[0][i] = wn[0][i]; w[1][i] = wn[1][i]; }
                // the stream of k-step s: 10 MFMAs; behind them the split of k-step s + 1 (8 blocks), this step's piece stores and the
                // next step's weight rows
                auto filler = [&](const int slot) {
                    if (s < 3 && slot >= 1 && slot <= 8) split_block(h1[(s + 1) >> 1], (s + 1) & 1, slot - 1, hb[s + 1], ss);
                    if (s == 3 && slot < 8) wk[slot >> 1][slot & 1] = ldw(&sm.W2p[0][0][0], slot & 1, slot >> 1, 0);
                    if (s < 3 && (slot == 0 || slot == 4 || slot == 9)) {
                        const int i = slot == 0 ? 0 : slot == 4 ? 1 : 2;
                        wn[0][i] = ldw(&sm.W2p[0][0][0], 0, s + 1, i);
                        wn[1][i] = ldw(&sm.W2p[0][0][0], 1, s + 1, i);
                    }
                    if (slot == 9) store_pieces(IMG_H, s >> 1, s & 1, hb[s]);
                };
                if (s == 0) { X3S_MFMA32G_V_AV(c2[0], w[0][2], hb[s].p[0]); } else { X3S_MFMA32_V_AV(c2[0], w[0][2], hb[s].p[0]); }
                filler(0);
                if (s == 0) { X3S_MFMA32G_V_AV(c2[1], w[1][2], hb[s].p[0]); } else { X3S_MFMA32_V_AV(c2[1], w[1][2], hb[s].p[0]); }
                filler(1);
                X3S_MFMA32_V_AV(c2[0], w[0][1], hb[s].p[1]); filler(2);
                X3S_MFMA32_V_AV(c2[1], w[1][1], hb[s].p[1]); filler(3);
                X3S_MFMA32_V_AV(c2[0], w[0][0], hb[s].p[2]); filler(4);
                X3S_MFMA32_V_AV(c2[1], w[1][0], hb[s].p[2]); filler(5);
                X3S_MFMA32_V_AV(c2[0], w[0][1], hb[s].p[0]); filler(6);
                X3S_MFMA32_V_AV(c2[1], w[1][1], hb[s].p[0]); filler(7);
                X3S_MFMA32_V_AV(c2[0], w[0][0], hb[s].p[1]); filler(8);
                X3S_MFMA32_V_AV(c2[1], w[1][0], hb[s].p[1]); filler(9);
            }
            // the big terms; behind them the operand gathers of the PREVIOUS tile's G1 (its dH1 pieces are still in imgD)
            int hb0 = h_base0, hb1 = h_base1;
            asm volatile("" : "+v"(hb0), "+v"(hb1));   // the eight gather addresses are formed here, per tile: not hoisted into eight registers
            int gq = 0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                X3S_MFMA32_V_AV(c2[0], wk[s][0], hb[s].p[0]);
                g1_gather_pair(g1o, gq++, hb0, hb1); g1_gather_pair(g1o, gq++, hb0, hb1);
                X3S_MFMA32_V_AV(c2[1], wk[s][1], hb[s].p[0]);
                g1_gather_pair(g1o, gq++, hb0, hb1);
            }
        }
        settle(c2[0], c2[1]);

        // ================================================================ heads, loss, dH2 (in place), the lane's share of db2 / dW3 / dW4;
        // behind the vector work: G1 of the PREVIOUS tile, 36 MFMAs placed one by one
        int g1k = 0;
        float h2[2][16];   // H2, then dH2 in place
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            relu4(h2[q >> 2], c2[q >> 2], q & 3);
            g1_step(g1k++, g1o, xb);
            g1_step(g1k++, g1o, xb);
        }
        float g3 = 0.f, g4 = 0.f;
        {
            float z3 = 0.f, z4 = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const v4f w = ldv(sm.w3, t, g);
                    z3 = fmaf(h2[t][4 * g], w.x, z3); z3 = fmaf(h2[t][4 * g + 1], w.y, z3);
                    z3 = fmaf(h2[t][4 * g + 2], w.z, z3); z3 = fmaf(h2[t][4 * g + 3], w.w, z3);
                    if (ACTOR) {
                        const v4f v = ldv(sm.w4, t, g);
                        z4 = fmaf(h2[t][4 * g], v.x, z4); z4 = fmaf(h2[t][4 * g + 1], v.y, z4);
                        z4 = fmaf(h2[t][4 * g + 2], v.z, z4); z4 = fmaf(h2[t][4 * g + 3], v.w, z4);
                    }
                }
            z3 += __shfl_xor(z3, 32, 64);
            if (ACTOR) z4 += __shfl_xor(z4, 32, 64);
            const float own = (lhi == 0) ? 1.f : 0.f;   // statistics are counted once per sample
            if (valid) {
                z3 += b3;
                z4 += b4;
#ifdef MLP64_DEBUG_Z   // (dev builds only, tools/verify/x3_forward_error.py: the heads' pre-activations into unused rows of the workspace)
                if (lhi == 0) {
                    float* const dbg = partial + (size_t)200 * P;
                    dbg[tile * 32 + l31] = z3;
                    if (ACTOR) dbg[M + tile * 32 + l31] = z4;
                }
#endif
                if (ACTOR) {
                    const float mu0 = 1.0f / (1.0f + expf(-z3));   // torch.sigmoid, net_actor.py:185
                    const float mu1 = tanhf(z4);                    // net_actor.py:186
                    const float d0 = cur_a0 - mu0, d1 = cur_a1 - mu1;
                    const float lp = -0.5f * ((d0 * d0 + d1 * d1) / var) - 1.8378770664093453f - logf(var);   // ppo.py:734-735
                    const float lr = lp - cur_lp;
                    const float ratio = expf(lr);                  // ppo.py:316
                    const float A = cur_t;
                    const float s1 = ratio * A;                     // ppo.py:319
                    const float rc = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
                    const float s2 = rc * A;                        // ppo.py:320
                    st0 += own * -fminf(s1, s2);                    // ppo.py:342
                    st2 += own * ((ratio - 1.0f) - lr);             // approx KL, ppo.py:326
                    st3 += (fabsf(ratio - 1.0f) > clip) ? own : 0.f;  // clip fraction, ppo.py:335
                    const bool inside = (ratio >= 1.0f - clip) && (ratio <= 1.0f + clip);
                    const float dL_dratio = (inside || s1 < s2) ? -A : 0.f;
                    const float dL_dlp = dL_dratio * ratio * inv_n;
                    g3 = dL_dlp * (d0 / var) * (mu0 * (1.0f - mu0));
                    g4 = dL_dlp * (d1 / var) * (1.0f - mu1 * mu1);
                } else {
                    const float e = z3 - cur_t;                     // critic(obs).squeeze(), ppo.py:724
                    st1 += own * (e * e);                           // MSELoss, ppo.py:343
                    g3 = 2.0f * e * inv_n;
                }
            }
            adb3 += own * g3;
            adb4 += own * g4;
#ifdef MLP64_DEBUG_Z
            if (valid && lhi == 0) {
                float* const dbg = partial + (size_t)200 * P;
                dbg[2 * M + tile * 32 + l31] = g3;
                if (ACTOR) dbg[3 * M + tile * 32 + l31] = g4;
            }
#endif
        }
        // dH2[m][u] = [H2 > 0] (g3 w3[u] + g4 w4[u]) in place; the lane's dW3 / dW4 / db2 sums; the rest of G1 behind it
        {
            v4f wn3 = ldv(sm.w3, 0, 0), wn4 = ACTOR ? ldv(sm.w4, 0, 0) : v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int t = q >> 2, g = q & 3;
                const v4f w = wn3, v = wn4;
                if (q < 7) {
                    wn3 = ldv(sm.w3, (q + 1) >> 2, (q + 1) & 3);
                    if (ACTOR) wn4 = ldv(sm.w4, (q + 1) >> 2, (q + 1) & 3);
                }
                const float wv3[4] = {w.x, w.y, w.z, w.w}, wv4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g + j;
                    dh2_value<ACTOR>(h2[t][r], pw3[t][r], pw4[t][r], pb2[t][r], g3, g4, wv3[j], wv4[j]);
                    if (g1k < 36 && (j & 1)) g1_step(g1k++, g1o, xb);
                }
                if (g1k < 36 && (g & 1)) g1_step(g1k++, g1o, xb);
            }
        }
        while (g1k < 36) g1_step(g1k++, g1o, xb);   // (none left: 16 + 16 + 4)

        // ================================================================ B2: dH1^T = (W2^T dH2^T) . [H1 > 0]; dH2 split once, k-step s = (t2, j)
        f32x16 c3[2];
        P3 hB[2][2], dA[2][2];   // G2's operands, [unit tile][sample k-step]: gathered behind B2's stream
        u32x2 mw[8];           // the leading piece of H1 at the lane's own slots: the relu mask of layer 1
        {
            P3 db[4];
            u32x4 wk[4][2];
            SplitState ss;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                wn[0][i] = ldw(&sm.W2Tp[0][0][0], 0, 0, i);
                wn[1][i] = ldw(&sm.W2Tp[0][0][0], 1, 0, i);
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) split_block(h2[0], 0, b, db[0], ss);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                u32x4 w[2][3];
#pragma unroll
                for (int i = 0; i < 3; ++i) { w[0][i] = wn[0][i]; w[1][i] = wn[1][i]; }
                auto filler = [&](const int slot) {
                    if (s < 3 && slot >= 1 && slot <= 8) split_block(h2[(s + 1) >> 1], (s + 1) & 1, slot - 1, db[s + 1], ss);
                    if (s == 3 && slot < 8) wk[slot >> 1][slot & 1] = ldw(&sm.W2Tp[0][0][0], slot & 1, slot >> 1, 0);
                    // H1's side of G2 (imgH is complete since F2): pair q = (t, sk, i) of 0..11, three per k-step
                    if (slot == 2 || slot == 5 || slot == 8) {
                        const int q = 3 * s + (slot - 2) / 3, t = q / 6, sk = (q / 3) & 1, i = q % 3;
                        hB[t][sk].p[i] = cat(tr_read(smb + g_off(t, sk, 0) + IMG_H + i * kPieceB), tr_read(smb + g_off(t, sk, 1) + IMG_H + i * kPieceB));
                    }
                    if (s == 3 && (slot == 3 || slot == 4 || slot == 6 || slot == 7)) {   // the mask words: two 8-byte reads per slot
                        const int k0 = 2 * (slot - 3 - (slot > 4));
#pragma unroll
                        for (int k = k0; k < k0 + 2; ++k)
                            mw[k] = *reinterpret_cast<const u32x2*>(smb + st_off(k >> 2, (k >> 1) & 1, k & 1) + IMG_H);
                    }
                    if (s < 3 && (slot == 0 || slot == 4 || slot == 9)) {
                        const int i = slot == 0 ? 0 : slot == 4 ? 1 : 2;
                        wn[0][i] = ldw(&sm.W2Tp[0][0][0], 0, s + 1, i);
                        wn[1][i] = ldw(&sm.W2Tp[0][0][0], 1, s + 1, i);
                    }
                    if (slot == 9) store_pieces(IMG_D, s >> 1, s & 1, db[s]);
                };
                if (s == 0) { X3S_MFMA32_VZ_AV(c3[0], w[0][2], db[s].p[0]); } else { X3S_MFMA32_V_AV(c3[0], w[0][2], db[s].p[0]); }
                filler(0);
                if (s == 0) { X3S_MFMA32_VZ_AV(c3[1], w[1][2], db[s].p[0]); } else { X3S_MFMA32_V_AV(c3[1], w[1][2], db[s].p[0]); }
                filler(1);
                X3S_MFMA32_V_AV(c3[0], w[0][1], db[s].p[1]); filler(2);
                X3S_MFMA32_V_AV(c3[1], w[1][1], db[s].p[1]); filler(3);
                X3S_MFMA32_V_AV(c3[0], w[0][0], db[s].p[2]); filler(4);
                X3S_MFMA32_V_AV(c3[1], w[1][0], db[s].p[2]); filler(5);
                X3S_MFMA32_V_AV(c3[0], w[0][1], db[s].p[0]); filler(6);
                X3S_MFMA32_V_AV(c3[1], w[1][1], db[s].p[0]); filler(7);
                X3S_MFMA32_V_AV(c3[0], w[0][0], db[s].p[1]); filler(8);
                X3S_MFMA32_V_AV(c3[1], w[1][0], db[s].p[1]); filler(9);
            }
            int dq = 0;   // dH2's side of G2 behind the big terms (every piece store of imgD is issued): pair q = (t, sk, i)
            auto gather_d = [&]() {
                const int t = dq / 6, sk = (dq / 3) & 1, i = dq % 3;
                dA[t][sk].p[i] = cat(tr_read(smb + g_off(t, sk, 0) + IMG_D + i * kPieceB), tr_read(smb + g_off(t, sk, 1) + IMG_D + i * kPieceB));
                ++dq;
            };
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                X3S_MFMA32_V_AV(c3[0], wk[s][0], db[s].p[0]);
                gather_d(); gather_d();
                X3S_MFMA32_V_AV(c3[1], wk[s][1], db[s].p[0]);
                gather_d();
            }
        }

        // ================================================================ G2: dW2 += dH2^T H1 over the tile's samples, both operands gathered
        // transposed from the piece images; behind it: the relu mask of layer 1 on dH1, its split, its piece stores (into imgD: every
        // gather of dH2 is issued before the first of them)
        {
            settle(c3[0], c3[1]);
            // [H1 > 0] from the leading piece of H1, read back from the lane's own slots of the image (keeping H1 itself would hold 32
            // vector registers through F2 / B2, where the file is full): bf16(H1) != 0.  Differs from H1 > 0 only for
            // 0 < H1 < 2^-126 (the bf16 conversion flushes / rounds such values to zero) -- not a value a sum of O(1) terms takes.
            float d1[2][16];   // dH1 = (W2^T dH2^T) . [H1 > 0]
#pragma unroll
            for (int k = 0; k < 8; ++k) {   // k = (t, j, b): registers 8 j + 4 b .. + 3 of c3[t]
                const int t = k >> 2, r0 = 8 * ((k >> 1) & 1) + 4 * (k & 1);
                mask4(d1[t][r0], d1[t][r0 + 1], d1[t][r0 + 2], d1[t][r0 + 3], c3[t][r0], c3[t][r0 + 1], c3[t][r0 + 2], c3[t][r0 + 3], mw[k].x, mw[k].y);
            }
            P3 eb[4];
            SplitState ss;
            int blk = 0;   // 32 split blocks + 4 x 6 stores behind 48 MFMAs
            auto filler = [&]() {
                if (blk < 32) {
                    const int s = blk >> 3;
                    split_block(d1[s >> 1], s & 1, blk & 7, eb[s], ss);
                    if ((blk & 7) == 7) store_pieces(IMG_D, s >> 1, s & 1, eb[s]);
                }
                ++blk;
            };
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const P3& A = dA[t2][s];
                    const P3 &B0 = hB[0][s], &B1 = hB[1][s];
                    f32x16 &a0 = aW2[t2][0], &a1 = aW2[t2][1];
                    X3S_MFMA32_A_AA(a0, A.p[2], B0.p[0]); filler(); X3S_MFMA32_A_AA(a1, A.p[2], B1.p[0]); filler();
                    X3S_MFMA32_A_AA(a0, A.p[1], B0.p[1]); filler(); X3S_MFMA32_A_AA(a1, A.p[1], B1.p[1]); filler();
                    X3S_MFMA32_A_AA(a0, A.p[0], B0.p[2]); filler(); X3S_MFMA32_A_AA(a1, A.p[0], B1.p[2]); filler();
                    X3S_MFMA32_A_AA(a0, A.p[1], B0.p[0]); filler(); X3S_MFMA32_A_AA(a1, A.p[1], B1.p[0]); filler();
                    X3S_MFMA32_A_AA(a0, A.p[0], B0.p[1]); filler(); X3S_MFMA32_A_AA(a1, A.p[0], B1.p[1]); filler();
                    X3S_MFMA32_A_AA(a0, A.p[0], B0.p[0]); filler(); X3S_MFMA32_A_AA(a1, A.p[0], B1.p[0]); filler();
                }
        }
        landed();
        have_prev = true;
    }
    if (have_prev) {   // G1 of the last tile
        const u32x4 xb[3] = {xbn[0], xbn[1], xbn[2]};
        G1Ops g1o;
#pragma unroll
        for (int q = 0; q < 12; ++q) g1_gather_pair(g1o, q, h_base0, h_base1);
#pragma unroll
        for (int k = 0; k < 36; ++k) g1_step(k, g1o, xb, true);
    }

    // ---- the lane partials -> sums over the 32 sample lanes of each half, through the wave's (now free) images
    __syncthreads();
    float rb2[2], rw3[2], rw4[2];   // lane (l31 = unit within the 32-unit tile t... ) see below
    {
        float* const scr = reinterpret_cast<float*>(&sm.img[wave][0][0]);   // [96][64]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                scr[(0 * 32 + 16 * t + r) * 64 + lane] = pb2[t][r];
                scr[(1 * 32 + 16 * t + r) * 64 + lane] = pw3[t][r];
                scr[(2 * 32 + 16 * t + r) * 64 + lane] = pw4[t][r];
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // lane (u = l31, half q = lhi) sums unit 32 t + u of tile t = q ... : unit n = 32 t + row, row = (r & 3) + 8 (r >> 2) + 4 hi  <->  (t, r, hi)
        // this lane takes unit n = 32 lhi + l31 of each of the three vectors: t = lhi, row = l31: hi = (row >> 2) & 1, r = (row & 3) + 4 (row >> 3)
        const int row = l31, hi = (row >> 2) & 1, r = (row & 3) + 4 * (row >> 3), t = lhi;
        float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const float* src = scr + (v * 32 + 16 * t + r) * 64 + 32 * hi;
#pragma unroll
            for (int m = 0; m < 32; m += 4) {
                const v4f q = *reinterpret_cast<const v4f*>(src + m);
                s[v] += (q.x + q.y) + (q.z + q.w);
            }
        }
        rb2[0] = s[0]; rw3[0] = s[1]; rw4[0] = s[2];
        rb2[1] = rw3[1] = rw4[1] = 0.f;
    }
    __syncthreads();

    // ---- workgroup reduction of the 4 waves' partial gradients in LDS, then one coalesced row of `partial`
    constexpr int RP = (P + 3 + 3) & ~3;
    float s3 = adb3, s4 = adb4, sA = ACTOR ? st0 : st1, sB = st2, sC = st3;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s3 += __shfl_xor(s3, o, 64); s4 += __shfl_xor(s4, o, 64);
        sA += __shfl_xor(sA, o, 64); sB += __shfl_xor(sB, o, 64); sC += __shfl_xor(sC, o, 64);
    }
    float* const red0 = reinterpret_cast<float*>(&sm.img[0][0][0]);
    float* const rowp = red0 + wave * RP;
    {
        auto put = [&](int idx, float v) { rowp[idx] = v; };
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int t1 = 0; t1 < 2; ++t1)
#pragma unroll
                for (int r = 0; r < 16; ++r) put(OFF_W2 + (32 * t2 + c_row(r, lane)) * H + 32 * t1 + l31, aW2[t2][t1][r]);
        // unit 32 lhi + l31 of b2 / w3 / w4
        put(OFF_B2 + 32 * lhi + l31, rb2[0]);
        put(OFF_W3 + 32 * lhi + l31, rw3[0]);
        if (ACTOR) put(OFF_W4 + 32 * lhi + l31, rw4[0]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int r = 0; r < 4; ++r) put(OFF_W1 + (16 * u + 4 * kk + r) * 16 + l15, aW1[u][r]);
            if (l15 == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) put(OFF_B1 + 16 * u + 4 * kk + r, aB1[u][r]);
            }
        }
        if (lane == 0) {
            put(OFF_B3, s3);
            if (ACTOR) put(OFF_B4, s4);
            put(P + 0, sA);
            put(P + 1, sB);
            put(P + 2, sC);
        }
    }
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * P;
    const float* red = red0;
    for (int k = tid; k < P; k += NT) out[k] = (red[k] + red[RP + k]) + (red[2 * RP + k] + red[3 * RP + k]);
    if (tid < 3) stats_partial[blockIdx.x * 4 + tid] = (red[P + tid] + red[RP + P + tid]) + (red[2 * RP + P + tid] + red[3 * RP + P + tid]);
}

}  // namespace x3s

// both nets of one epoch in one launch (net_mask as mlp64_pass_both_x3), 16-column rows
__global__ __launch_bounds__(64 * x3s::SW) void mlp64_pass_both_x3s(const float* __restrict__ params, const unsigned char* __restrict__ prep,
                                                                    const float* __restrict__ act, const float* __restrict__ logp_old,
                                                                    const float* __restrict__ rtg, const float* __restrict__ adv,
                                                                    long long M, float var, float clip, float inv_n, int net_mask,
                                                                    float* __restrict__ partial_a, float* __restrict__ stats_partial_a,
                                                                    float* __restrict__ partial_c, float* __restrict__ stats_partial_c) {
    __shared__ __attribute__((aligned(128))) x3s::SmemS sm;
    if (net_mask & 1) x3s::pass_body<true>(sm, params, prep, act, logp_old, rtg, adv, M, var, clip, inv_n, partial_a, stats_partial_a);
    if (net_mask == 3) __syncthreads();
    if (net_mask & 2) x3s::pass_body<false>(sm, params + Layout<16>::P_ACTOR, prep, act, logp_old, rtg, adv, M, var, clip, inv_n, partial_c, stats_partial_c);
}
