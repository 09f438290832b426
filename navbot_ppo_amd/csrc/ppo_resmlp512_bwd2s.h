// ppo_resmlp512_bwd2s.h -- round 6: the backward of the second residual block (rb2, 32 -> 512 -> 32) of the 512-wide nets as ONE
// hand-placed instruction stream per wave, software-pipelined over half chunks of 16 hidden units.  Included by ppo_resmlp512.hip inside
// its anonymous namespace, behind resmlp_bwd (same arguments, same partial rows, same Q partials: a drop-in for resmlp_bwd<32, 2, 4>).
//
// What round 5's kernel left on the table (4.3 ms per epoch, MFMA pipe 44 % busy): per chunk it ran [48 MFMAs of H^T / dH^T] ->
// [leaky block, 80 vector instructions] -> [splits of H / dH, 176] -> [48 MFMAs of dW2 / dW1] -> [16 float32 MFMAs of Q through a
// float32 LDS tile], one group after the other (sched_barriers kept the compiler from making it worse), and split X^T / dY^T a second
// time per tile from float32 LDS tiles.  Here, with the technique of ppo_mlp64_x3s.h (every MFMA and every vector block an
// `asm volatile` statement with register-class constraints: the compiler allocates registers and inserts the s_waitcnt of the loads it
// issues, but cannot re-order the stream):
//   * the vector work rides behind the MFMAs of the neighbouring stage -- stage A(h) = H^T / dH^T of half chunk h + 1 (24 MFMAs) with the
//     pair splits of half chunk h around them, stage B(h) = dW2 / dW1 of half chunk h (24 MFMAs, + the 12 of Q behind the chunk's second
//     half) with the leaky block of half chunk h + 1 behind them; the pipeline runs across tiles (A(7) computes half chunk 0 of the
//     wave's NEXT tile).  Half chunks, not chunks: 16 + 16 + 24 live registers of H / dH / pieces instead of 32 + 32 + 48;
//   * X and dY are split ONCE per tile, two or three pairs per B stage one tile ahead (the rows are requested a tile and a half ahead):
//     the pieces go to a wave-private LDS image [piece][sample][feature | output], from which they come back (a) as the lane's own
//     eight values (24 ds_read_b64 in B(6), behind their last use for the tile before) and (b) transposed (lane = unit, k = sample: the
//     operands of dW1 / dW2) through ds_read_b64_tr_b16 -- no float32 tiles, no second split (176 vector instructions and 64
//     ds_write_b32 per tile less);
//   * Q = W1[:, 16:32]^T dH runs on the bf16 MFMA too: dH's pieces (split anyway for dW1) take the same trip through an image
//     [piece][hidden unit][sample] and come back with lane = sample, k = hidden unit: 12 MFMAs of 16 cycles per chunk instead of 16
//     of 32;
//   * the accumulators of H^T start as the bias (one rounding of sum + bias, as in the 64-wide heads: DESIGN 5f), read as one
//     ds_read_b128 from a four-fold copy of b1 (no vector moves in front of the MFMA that reads them);
//   * the 32 accumulator tiles of dW2 / dW1 are pinned to a[0:127] (see B2S_ACC_CASES).
// Where it stands (profiles/r06_resmlp_update_kernel_stats.csv, r06_b2s_pmc.txt): 1970 instructions per tile of 32 samples around 432 MFMAs,
// 3.1-3.2 ms per epoch (4.3 before), MFMA pipe 53 % busy.  One wave per SIMD issues one instruction per ~4 cycles and an MFMA costs ~13 of
// its own (tools/ubench/gen_mfma16_stream.py), so this mix -- 3.6 others per MFMA, most of them the splits -- is issue-bound at ~27-29
// cycles per MFMA slot.  32x32x16 tiles would halve the MFMA issue cost but make the whole chunk the pipeline's unit again: written, does
// not fit the register file (tools/variants/r06_resmlp512_bwd2t.h.txt).
// MFMA <-> vector hazards the compiler cannot see inside the asm statements: accumulators are read by the vector unit >= 3 MFMAs
// (>= 12 issue cycles) after their last MFMA or behind an s_nop 7; tools/verify/mfma_hazard_lint.py checks the listing of both
// instantiations (tests/test_isa_lint_cpu.py).
#pragma once

namespace b2s {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int SW = 4;                    // waves per workgroup: one per SIMD, 512 registers each
constexpr int kPieceB = 32 * 128;        // one piece of an image: 32 rows of 128 bytes
constexpr int kImgB = 3 * kPieceB;

struct Smem {
    uint4 W1p[3 * HS * 4];               // [piece][hidden j][k-group q] = eight bf16: W1[j][4 q + r], W1[j][16 + 4 q + r]   (B of H^T)
    uint4 W2Tp[3 * HS * 4];              // the same of W2^T                                                          (B of dH^T)
    uint4 W1Qp[3 * NCH * 64];            // [piece][chunk][lane (i, q)] = eight bf16 W1[32 c + 4 q + r | 16 + 4 q + r][16 + i]  (A of Q)
    float b1x4[HS * 4];                  // b1[j] four times: the accumulator tile H^T starts from (one ds_read_b128, no vector moves)
    unsigned char img[SW][2][kImgB];     // per wave: [0] rows = samples: X pieces (bytes 0..63) | dY pieces (64..127); [1] rows = hidden units: dH pieces
};
static_assert(sizeof(Smem) <= 160 * 1024, "LDS");
static_assert(offsetof(Smem, img) % 128 == 0 && kImgB % 128 == 0, "image bases keep the low 7 address bits of a lane offset (the XOR swizzles)");

// ---- the stream's building blocks.  Constraint letters: v = VGPR, a = AGPR (the compiler loads LDS operands straight into AGPRs).
// An MFMA statement optionally carries one PAIR SPLIT around it -- [first half of the split, 5 instructions] MFMA [second half, 6] --
// in the SAME asm statement: the compiler puts a wait state between two asm statements when the later one uses a register the earlier
// one defines and nothing but asm statements lies between (it assumes a dst-forwarding hazard of gfx950 inside what it cannot see);
// with the residuals of a pair internal to one statement, and the lane's bias-gradient sum left to the compiler, the stream has no
// such pairs (round 6's first version: 88 s_nop per tile = 4 % of its issue slots).
#ifndef B2S_GUARD
#define B2S_GUARD ""   // ("s_nop 0\n\t": a wait state in front of the stage-opening MFMAs, for A/B runs; the lint decides)
#endif
struct SplitIO {
    unsigned p0, p1, p2;
    float ra, rb, t0, t1, a, b;
};
#define B2S_SA                                 \
    "v_cvt_pk_bf16_f32 %[p0], %[xa], %[xb]\n\t" \
    "v_lshlrev_b32 %[ra], 16, %[p0]\n\t"        \
    "v_and_b32 %[rb], 0xffff0000, %[p0]\n\t"    \
    "v_sub_f32 %[ra], %[xa], %[ra]\n\t"         \
    "v_sub_f32 %[rb], %[xb], %[rb]\n\t"
#define B2S_SB                                     \
    "\n\tv_cvt_pk_bf16_f32 %[p1], %[ra], %[rb]\n\t" \
    "v_lshlrev_b32 %[t0], 16, %[p1]\n\t"            \
    "v_and_b32 %[t1], 0xffff0000, %[p1]\n\t"        \
    "v_sub_f32 %[t0], %[ra], %[t0]\n\t"             \
    "v_sub_f32 %[t1], %[rb], %[t1]\n\t"             \
    "v_cvt_pk_bf16_f32 %[p2], %[t0], %[t1]"
#define B2S_SOUT(S) [p0] "=&v"(S.p0), [p1] "=&v"(S.p1), [p2] "=&v"(S.p2), [ra] "=&v"(S.ra), [rb] "=&v"(S.rb), [t0] "=&v"(S.t0), [t1] "=&v"(S.t1)
#define B2S_SIN(S) [xa] "v"(S.a), [xb] "v"(S.b)
#define B2S_COMMA ,
#define B2S_EMIT(TXT, OUTS, INS, S, wrap)                                                        \
    do {                                                                                         \
        if (wrap) asm volatile(B2S_SA TXT B2S_SB : OUTS, B2S_SOUT(S) : INS, B2S_SIN(S));         \
        else asm volatile(TXT : OUTS : INS);                                                     \
    } while (0)
#define B2S_MF "v_mfma_f32_16x16x32_bf16 %[ac], %[sa], %[sb], "
// H^T / dH^T: accumulator in VGPRs (the vector unit reads it next), A = the tile's pieces (VGPR), B = weight pieces (AGPR, from LDS)
#define B2S_P(acc, A, B, S, wrap) B2S_EMIT(B2S_MF "%[ac]", [ac] "+v"(acc), [sa] "v"(A) B2S_COMMA [sb] "a"(B), S, wrap)
#define B2S_P_C(acc, A, B, C, S, wrap) \
    B2S_EMIT(B2S_GUARD B2S_MF "%[sc]", [ac] "=&v"(acc), [sa] "v"(A) B2S_COMMA [sb] "a"(B) B2S_COMMA [sc] "v"(C), S, wrap)
#define B2S_P_Z(acc, A, B, S, wrap) B2S_EMIT(B2S_GUARD B2S_MF "0", [ac] "=&v"(acc), [sa] "v"(A) B2S_COMMA [sb] "a"(B), S, wrap)
// dW2 (A = dY^T pieces, AGPR; B = H pieces, VGPR) and dW1 (A = dH pieces, VGPR; B = X^T pieces, AGPR).  The 32 accumulator tiles are
// PINNED to a[0:127] by physical-register constraints: left to itself the allocator parks the accumulators of the chunks that are not
// being worked on in free VGPRs and copies them back (128 v_accvgpr moves per tile of 2092 instructions, and copies right in front of
// MFMAs that read them: the hazard of the header comment).  A constraint string must be a literal: hence one case per tile.
#define B2S_ACC_CASES(X) X(0, 0, 3) X(1, 4, 7) X(2, 8, 11) X(3, 12, 15) X(4, 16, 19) X(5, 20, 23) X(6, 24, 27) X(7, 28, 31) X(8, 32, 35) X(9, 36, 39) X(10, 40, 43) X(11, 44, 47) X(12, 48, 51) X(13, 52, 55) X(14, 56, 59) X(15, 60, 63) X(16, 64, 67) X(17, 68, 71) X(18, 72, 75) X(19, 76, 79) X(20, 80, 83) X(21, 84, 87) X(22, 88, 91) X(23, 92, 95) X(24, 96, 99) X(25, 100, 103) X(26, 104, 107) X(27, 108, 111) X(28, 112, 115) X(29, 116, 119) X(30, 120, 123) X(31, 124, 127)
#define B2S_CASE_W2(i, lo, hi) \
    case i: B2S_EMIT(B2S_MF "%[ac]", [ac] "+{a[" #lo ":" #hi "]}"(w_acc), [sa] "a"(w_A) B2S_COMMA [sb] "v"(w_B), S, w_wrap); break;
#define B2S_CASE_W1(i, lo, hi) \
    case i: B2S_EMIT(B2S_MF "%[ac]", [ac] "+{a[" #lo ":" #hi "]}"(w_acc), [sa] "v"(w_A) B2S_COMMA [sb] "a"(w_B), S, w_wrap); break;
// Q: both operands from LDS
#define B2S_Q(acc, A, B, S, wrap) B2S_EMIT(B2S_MF "%[ac]", [ac] "+v"(acc), [sa] "a"(A) B2S_COMMA [sb] "a"(B), S, wrap)
#define B2S_Q_Z(acc, A, B, S, wrap) B2S_EMIT(B2S_GUARD B2S_MF "0", [ac] "=&v"(acc), [sa] "a"(A) B2S_COMMA [sb] "a"(B), S, wrap)

// a pair split on its own (the prologue): 11 instructions
__device__ __forceinline__ void split_pair_asm(SplitIO& S) { asm volatile(B2S_SA B2S_SB : B2S_SOUT(S) : B2S_SIN(S)); }
// one value of the leaky block: f = H > 0 ? 1 : 0.2 ; h = H f ; g = G f    (4 instructions; `one` / `slope` hold 1.0f / 0.2f: a literal
// next to vcc would be a second constant-bus operand)
__device__ __forceinline__ void leaky1(float& h, float& g, const float H, const float G, const float one, const float slope) {
    float f;
    asm volatile("v_cmp_lt_f32 vcc, 0, %3\n\t"
                 "v_cndmask_b32 %2, %6, %5, vcc\n\t"
                 "v_mul_f32 %0, %3, %2\n\t"
                 "v_mul_f32 %1, %4, %2"
                 : "=&v"(h), "=&v"(g), "=&v"(f) : "v"(H), "v"(G), "v"(one), "v"(slope) : "vcc");
}

struct P3 {   // eight values as three packed pieces
    u32x4 p[3];
};

// XOR swizzle of the 8-byte slots of a 128-byte image row (ppo_mlp64_x3s.h): conflict-free for the ds_write_b64 of 16 consecutive row
// lanes and for the 8-row x 32-byte gathers of the 16x16x32 operands
__device__ __forceinline__ int fswz(const int m) { return (m & 3) | ((((m >> 2) ^ (m >> 3)) & 1) << 2) | ((((m >> 1) ^ (m >> 3)) & 1) << 3); }

typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
__device__ __forceinline__ u32x2 tr_read(const unsigned char* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)p);
    u32x2 r;
    __builtin_memcpy(&r, &v, 8);
    return r;
}
__device__ __forceinline__ u32x4 cat(const u32x2 lo, const u32x2 hi) { return u32x4{lo.x, lo.y, hi.x, hi.y}; }

}  // namespace b2s

#ifndef RESMLP_BWD2S
#define RESMLP_BWD2S 1
#endif

// OBS_F16: the observation rows are float16 (navsim_cfg.obs_f16) -- a template parameter, not an argument: the stream has no room for branches
template <bool OBS_F16>
__global__ __launch_bounds__(64 * b2s::SW) void resmlp_bwd2s(const float* __restrict__ params, int n_nets, const void* __restrict__ obs,
                                                             const float* __restrict__ h1buf, const float* __restrict__ dypre,
                                                             long long n, int groups, float* __restrict__ wpart,
                                                             float* __restrict__ qout) {
    using namespace b2s;
    __shared__ __attribute__((aligned(128))) Smem sm;
    constexpr int IN = 32;
    constexpr int kWv = SW, kThr = 64 * SW;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, q = lane >> 4;
    const WG wg = decode_block(blockIdx.x, n_nets, groups);
    const float* __restrict__ pn = params + (wg.net_i ? rp::P_ACTOR : 0);

    // ---- the slice's weights as pieces
    for (int k = tid; k < HS * 4; k += kThr) {
        const int j = k >> 2, kq = k & 3;
        const float* wr = pn + Blk<IN>::W1 + (wg.sl * HS + j) * IN;
        const float4 lo = ld4(wr + 4 * kq), hi = ld4(wr + 16 + 4 * kq);
        const float* wc = pn + Blk<IN>::W2 + wg.sl * HS + j;
        const float v1[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        float v2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v2[e] = wc[(size_t)((e < 4 ? 0 : 16) + 4 * kq + (e & 3)) * rp::HID];
        const bf16x3::Pieces P1 = bf16x3::split8(v1), P2 = bf16x3::split8(v2);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            sm.W1p[(i * HS + j) * 4 + kq] = P1.p[i];
            sm.W2Tp[(i * HS + j) * 4 + kq] = P2.p[i];
        }
    }
    {   // A of Q: lane (i, qq) of chunk c holds W1[32 c + (4 qq + r | 16 + 4 qq + r)][16 + i]
        const int c = tid >> 6, L = tid & 63, ii = L & 15, qq = L >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = pn[Blk<IN>::W1 + (wg.sl * HS + c * 32 + (e < 4 ? 0 : 16) + 4 * qq + (e & 3)) * IN + 16 + ii];
        const bf16x3::Pieces P = bf16x3::split8(v);
#pragma unroll
        for (int i = 0; i < 3; ++i) sm.W1Qp[(i * NCH + c) * 64 + L] = P.p[i];
    }
    static_assert(NCH * 64 == kThr, "one W1Qp entry per thread");
    for (int k = tid; k < HS * 4; k += kThr) sm.b1x4[k] = pn[Blk<IN>::B1 + wg.sl * HS + (k >> 2)];
    __syncthreads();

    // ---- lane addresses into the wave's images (everything else is an immediate offset)
    const unsigned char* const smb = reinterpret_cast<const unsigned char*>(&sm);
    unsigned char* const smw = reinterpret_cast<unsigned char*>(&sm);
    const int img0 = (int)offsetof(Smem, img) + wave * 2 * kImgB;
    constexpr int IMG_X = 0, IMG_D = kImgB;
    // piece stores: lane (row l15 of a 16-row half, q) owns the slots qb + q, qb = 0 / 4 / 8 / 12
    const int st_base = img0 + l15 * 128 + ((fswz(l15) ^ q) << 3);
    // operand gathers: rows 4 q + (l15 >> 2) of a 16-row half, slots qb + (l15 & 3)
    const int g_row = 4 * q + (l15 >> 2);
    const int g_base = img0 + g_row * 128 + ((fswz(g_row) ^ (l15 & 3)) << 3);

    const float* __restrict__ h1n = h1buf + (size_t)wg.net_i * n * 16;
    const float* __restrict__ dyn = dypre + (size_t)wg.net_i * n * 32;
    float* __restrict__ qo = qout + (size_t)(wg.net_i * NSL + wg.sl) * n * 16;

    // ---- accumulators that persist over the wave's tiles (AGPRs)
    f32x4 aW2[NCH][2][2];   // dW2[o block][j block of chunk c]
    f32x4 aW1[NCH][2][2];   // dW1[j block of chunk c][i block]
    float adb1[NCH][2];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            adb1[c][jb] = 0.f;
#pragma unroll
            for (int b = 0; b < 2; ++b) aW2[c][b][jb] = aW1[c][jb][b] = zero4();
        }

    const long long n_tiles = (n + 31) / 32, stride = (long long)groups * kWv;
    // [feature block][sample tile]: lane (sample l15, q) holds columns 4 q .. 4 q + 3 of the block.  The observation columns stay as
    // LOADED (float32 bits, or four float16 in .xy: widened where the pair is split -- a conversion here would wait for the load)
    struct Raw { f32x4 X[2][2], G[2][2]; };
    auto request = [&](Raw& w, const long long tile) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const long long s = tile * 32 + 16 * st + l15;
            const bool v = s < n;
            if (OBS_F16) {
                const uint2 u = v ? *reinterpret_cast<const uint2*>(reinterpret_cast<const __half*>(obs) + s * 16 + 4 * q) : make_uint2(0u, 0u);
                w.X[0][st] = f32x4{__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f};
            } else {
                w.X[0][st] = v ? v4(ld4(reinterpret_cast<const float*>(obs) + s * 16 + 4 * q)) : zero4();
            }
            w.X[1][st] = v ? v4(ld4(h1n + s * 16 + 4 * q)) : zero4();
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) w.G[ob][st] = v ? v4(ld4(dyn + s * 32 + 16 * ob + 4 * q)) : zero4();
        }
    };

    constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};   // bf16x3 term order: small terms first, a0 b0 last
    constexpr int QA[6] = {0, 1, 2, 0, 1, 0}, QB[6] = {2, 1, 0, 1, 0, 0};   // Q: the same classes, B's small pieces first (they are let go first)
    Raw raw;                // the rows of the wave's NEXT tile (split two or three pairs per stage while this tile is worked on)
    P3 XP[2], DYP[2];       // [sample tile]: the tile's rows / output gradients, lane = sample (A of H^T / dH^T): read back from the image
    unsigned tw0 = 0, tw1 = 0, tw2 = 0;   // the pieces of an even pair waiting for the odd one (together: one 8-byte slot of the image per piece)
    P3 XTP[2], DYTP[2];     // the same pieces, lane = unit, k = sample (B of dW1 / A of dW2): gathered from the image
    P3 HP, DHP;             // the half chunk's H^T / dH^T, lane = hidden unit, k = sample
    float hv[8], gv[8];     // the half chunk's H^T / dH^T behind the leaky block: [sample tile * 4 + r]
    u32x4 Wn1[3], Wn2[3];   // weight pieces of the half chunk whose H^T / dH^T comes next
    f32x4 bvec;
    f32x4 H[2], dH[2];      // [sample tile]: lane = hidden unit l15 of the half chunk, register r = sample 16 st + 4 q + r
    f32x4 dXa[2];
    u32x4 W1Q[3];
    P3 DHT;   // B of Q: sample tile 0's pieces, then (piece by piece, as Q's chain of tile 0 lets go of them) sample tile 1's
    SplitIO S;
    S.a = S.b = 0.f;
    float one = 1.0f, slope = 0.2f;
    asm volatile("" : "+v"(one), "+v"(slope));

    // half chunk h of 0..7 = (chunk h >> 1, 16-unit block h & 1)
    auto load_weights = [&](const int h) {
        const int j = 16 * h + l15;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            Wn1[i] = *reinterpret_cast<const u32x4*>(&sm.W1p[(i * HS + j) * 4 + q]);
            Wn2[i] = *reinterpret_cast<const u32x4*>(&sm.W2Tp[(i * HS + j) * 4 + q]);
        }
        bvec = *reinterpret_cast<const f32x4*>(&sm.b1x4[j * 4]);
    };
    // pair pi of 0..15 of the next tile's split: (X | dY, sample tile, pair of the lane's eight values)
    auto tile_pair_in = [&](const int pi) {
        const int arr = pi >> 3, st = (pi >> 2) & 1, e2 = pi & 3;
        const f32x4& src = arr ? raw.G[e2 >> 1][st] : raw.X[e2 >> 1][st];
        if (arr == 0 && e2 < 2 && OBS_F16) {   // two float16 in one word
            const unsigned w = __float_as_uint(src[e2]);
            const __half2 hh = *reinterpret_cast<const __half2*>(&w);
            const float2 f = __half22float2(hh);
            S.a = f.x;
            S.b = f.y;
        } else {
            S.a = src[2 * (e2 & 1)];
            S.b = src[2 * (e2 & 1) + 1];
        }
    };
    // ... to the image: two pairs = the four values of one 8-byte slot, per piece
    auto tile_pair_out = [&](const int pi) {
        const int arr = pi >> 3, st = (pi >> 2) & 1, e2 = pi & 3, g = 2 * arr + (e2 >> 1);   // g: X lo | X hi | dY lo | dY hi
        if (pi & 1) {
            unsigned char* const d = smw + (st_base ^ ((4 * g) << 3)) + st * 2048 + IMG_X;
            *reinterpret_cast<u32x2*>(d) = u32x2{tw0, S.p0};
            *reinterpret_cast<u32x2*>(d + kPieceB) = u32x2{tw1, S.p1};
            *reinterpret_cast<u32x2*>(d + 2 * kPieceB) = u32x2{tw2, S.p2};
        } else {
            tw0 = S.p0;
            tw1 = S.p1;
            tw2 = S.p2;
        }
    };
    // the lane's own pieces back from the image: load k of 0..23 = (sample tile, piece, slot group)
    auto tile_load_own = [&](const int k) {
        const int st = k / 12, i = (k / 4) % 3, g = k & 3;
        const u32x2 v = *reinterpret_cast<const u32x2*>(smb + (st_base ^ ((4 * g) << 3)) + st * 2048 + i * kPieceB + IMG_X);
        P3& dst = (g & 2) ? DYP[st] : XP[st];
        dst.p[i][2 * (g & 1)] = v.x;
        dst.p[i][2 * (g & 1) + 1] = v.y;
    };
    // gather k of 0..11 of the tile's transposed operands: (operand, piece)
    auto tile_gather = [&](const int k) {
        const int which = k / 3, i = k % 3, qb = 4 * which;   // X^T i-block 0, 1, dY^T o-block 0, 1
        const int a = (g_base ^ (qb << 3)) + i * kPieceB + IMG_X;
        const u32x4 v = cat(tr_read(smb + a), tr_read(smb + a + 2048));
        if (which < 2) XTP[which].p[i] = v;
        else DYTP[which - 2].p[i] = v;
    };
    // pair pi of 0..7 of the half chunk's split (dH first: its pieces go to the image)
    auto half_pair_in = [&](const int pi) {
        const float* src = pi < 4 ? gv : hv;
        S.a = src[2 * (pi & 3)];
        S.b = src[2 * (pi & 3) + 1];
    };
    auto half_pair_out = [&](const int pi) {
        P3& dst = pi < 4 ? DHP : HP;
        dst.p[0][pi & 3] = S.p0;
        dst.p[1][pi & 3] = S.p1;
        dst.p[2][pi & 3] = S.p2;
    };
    auto half_store = [&](const int k, const int jb) {   // k of 0..5: (piece, half of the lane's eight samples)
        const int i = k >> 1, half = k & 1;
        const u32x2 v = half ? u32x2{DHP.p[i][2], DHP.p[i][3]} : u32x2{DHP.p[i][0], DHP.p[i][1]};
        *reinterpret_cast<u32x2*>(smw + (st_base ^ ((4 * half) << 3)) + jb * 2048 + i * kPieceB + IMG_D) = v;
    };
    auto dht_gather = [&](const int st, const int i) {   // B of Q for sample tile st: lane = sample, k = hidden unit
        const int a = (g_base ^ ((4 * st) << 3)) + i * kPieceB + IMG_D;
        DHT.p[i] = cat(tr_read(smb + a), tr_read(smb + a + 2048));
    };
    auto leaky_block = [&](const int e, const int hn) {   // e of 0..7; hn = the half chunk (0..7) whose bias gradient takes the sum
        const int k = e >> 2, r = e & 3;
        leaky1(hv[e], gv[e], H[k][r], dH[k][r], one, slope);
        adb1[hn >> 1][hn & 1] += gv[e];   // (the compiler's instruction: see the note at the macros)
    };
    // H^T / dH^T of the half chunk whose weights are in Wn1 / Wn2, 24 MFMAs.  WRAP: the first 16 carry the
    // 8 pair splits of the half chunk before (pair k: first half in front of MFMA 2 k + 1, second half behind it); filler(slot) behind
    // each MFMA
    auto p_stage = [&](auto wrap_c, auto&& filler) {
        constexpr bool WRAP = decltype(wrap_c)::value;
        int slot = 0;
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int k = a & 1;
                const bool wrap = WRAP && slot < 16 && (slot & 1);
                if (wrap) half_pair_in(slot >> 1);
                if (a < 2) {
                    if (t == 0) { B2S_P_C(H[k], XP[k].p[TA[0]], Wn1[TB[0]], bvec, S, wrap); }
                    else { B2S_P(H[k], XP[k].p[TA[t]], Wn1[TB[t]], S, wrap); }
                } else {
                    if (t == 0) { B2S_P_Z(dH[k], DYP[k].p[TA[0]], Wn2[TB[0]], S, wrap); }
                    else { B2S_P(dH[k], DYP[k].p[TA[t]], Wn2[TB[t]], S, wrap); }
                }
                if (wrap) half_pair_out(slot >> 1);
                filler(slot++);
            }
    };

    // ---- prologue: the wave's first tile up to the leaky block of its half chunk 0; the rows of its second tile on the way
    long long tile = (long long)wg.grp * kWv + wave;
    request(raw, tile);
#pragma unroll
    for (int pi = 0; pi < 16; ++pi) {
        tile_pair_in(pi);
        split_pair_asm(S);
        tile_pair_out(pi);
    }
    request(raw, tile + stride);
#pragma unroll
    for (int k = 0; k < 24; ++k) tile_load_own(k);
    load_weights(0);
    p_stage(std::false_type{}, [](int) {});
    asm volatile("s_nop 7" : "+v"(H[0]), "+v"(H[1]), "+v"(dH[0]), "+v"(dH[1]));
#pragma unroll
    for (int e = 0; e < 8; ++e) leaky_block(e, 0);
    load_weights(1);

    // stage pair h of 0..7 of a tile; h is a compile-time constant (the 8 instances are written out: the body is beyond the size up to
    // which `#pragma unroll` is honoured)
    auto half_stages = [&](auto hc) {
        constexpr int h = decltype(hc)::value;
        constexpr int c = h >> 1, jb = h & 1, hn = (h + 1) & 7;
        // the NEXT tile's pairs split behind stage B(h): [start[h], start[h + 1]); B(6) reads the lane's pieces back into XP / DYP (their
        // last use for this tile is A(6)), A(7) starts the next tile with them
        constexpr int kPairStart[9] = {0, 3, 6, 9, 11, 13, 16, 16, 16};
        // ================================================================ A(h): H^T / dH^T of half chunk h + 1 | splits of half chunk h
        p_stage(std::true_type{}, [&](const int slot) {
            if (jb == 1 && slot < 3) W1Q[slot] = *reinterpret_cast<const u32x4*>(&sm.W1Qp[(slot * NCH + c) * 64 + lane]);
            if (h == 0 && slot >= 3 && slot < 15) tile_gather(slot - 3);   // (before the next tile's pieces overwrite the image)
            if (slot >= 8 && slot < 14) half_store(slot - 8, jb);
            if (jb == 1 && slot >= 16 && slot < 19) dht_gather(0, 18 - slot);   // B of Q, sample tile 0 (pieces 2, 1, 0: Q's order)
        });
        // ================================================================ B(h): dW2 / dW1 of half chunk h (+ Q of the chunk) | leaky block of half chunk h + 1
        {
            int slot = 0;
            int pair = kPairStart[h];
            auto emit = [&](auto&& mfma) {
                // the next tile's pairs around the MFMAs of slots 6, 14, 22
                const bool wrap = (slot == 6 || slot == 14 || slot == 22) && pair < kPairStart[h + 1];
                if (wrap) tile_pair_in(pair);
                mfma(wrap);
                if (wrap) {
                    tile_pair_out(pair);
                    ++pair;
                }
                const int s = slot++;
                if (h == 6 && s < 12) {
                    tile_load_own(2 * s);
                    tile_load_own(2 * s + 1);
                }
                if (h == 6 && s == 12) request(raw, tile + 2 * stride);   // (its registers are free: the next tile's rows are split)
                if (s == 10) load_weights((h + 2) & 7);   // (Wn1 / Wn2 are idle during B)
                if (s >= 3 && s < 19 && (s & 1)) leaky_block((s - 3) >> 1, hn);
                if (jb == 1) {
                    // B of Q: tile 1's piece i takes the registers once tile 0's chain has used piece i last
                    // (QB: slots 2 5 8 11 14 17 use pieces 2 1 0 1 0 0)
                    if (s == 3) dht_gather(1, 2);
                    if (s == 12) dht_gather(1, 1);
                    if (s == 18) dht_gather(1, 0);
                }
            };
            // (accumulator tile (c, ob, jb) of dW2 = tile 4 c + 2 ob + jb, (c, jb, ib) of dW1 = tile 16 + 4 c + 2 jb + ib)
            auto w2 = [&](const int idx, f32x4& w_acc, const u32x4& w_A, const u32x4& w_B, const bool w_wrap) {
                switch (idx) { B2S_ACC_CASES(B2S_CASE_W2) }
            };
            auto w1 = [&](const int idx, f32x4& w_acc, const u32x4& w_A, const u32x4& w_B, const bool w_wrap) {
                switch (idx) { B2S_ACC_CASES(B2S_CASE_W1) }
            };
            int nq = 0;
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (a < 2) {
                        const int ob = a;
                        emit([&](const bool w) { w2(4 * c + 2 * ob + jb, aW2[c][ob][jb], DYTP[ob].p[TA[t]], HP.p[TB[t]], w); });
                    } else {
                        const int ib = a & 1;
                        emit([&](const bool w) { w1(16 + 4 * c + 2 * jb + ib, aW1[c][jb][ib], DHP.p[TA[t]], XTP[ib].p[TB[t]], w); });
                    }
                    if (jb == 1 && (a & 1)) {   // one MFMA of Q behind every two of dW
                        const int st = nq / 6, tt = nq % 6;
                        ++nq;
                        if (c == 0 && tt == 0) emit([&](const bool w) { B2S_Q_Z(dXa[st], W1Q[QA[0]], DHT.p[QB[0]], S, w); });
                        else emit([&](const bool w) { B2S_Q(dXa[st], W1Q[QA[tt]], DHT.p[QB[tt]], S, w); });
                    }
                }
        }
    };
    static_assert(NCH == 4, "eight half chunks");
    for (; tile < n_tiles; tile += stride) {
        half_stages(std::integral_constant<int, 0>{});
        half_stages(std::integral_constant<int, 1>{});
        half_stages(std::integral_constant<int, 2>{});
        half_stages(std::integral_constant<int, 3>{});
        half_stages(std::integral_constant<int, 4>{});
        half_stages(std::integral_constant<int, 5>{});
        half_stages(std::integral_constant<int, 6>{});
        half_stages(std::integral_constant<int, 7>{});
        // Q partial of this slice: lane (sample l15, q) holds inputs 4 q .. 4 q + 3
        asm volatile("s_nop 7" : "+v"(dXa[0]), "+v"(dXa[1]));
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const long long s = tile * 32 + 16 * st + l15;
            if (s < n) *reinterpret_cast<float4*>(qo + s * 16 + 4 * q) = make_float4(dXa[st][0], dXa[st][1], dXa[st][2], dXa[st][3]);
        }
    }

    // ---- one partial-gradient row per WAVE (resmlp_reduce sums the rows in a fixed order)
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        asm volatile("s_nop 7" : "+a"(aW2[c][0][0]), "+a"(aW2[c][0][1]), "+a"(aW2[c][1][0]), "+a"(aW2[c][1][1]), "+a"(aW1[c][0][0]),
                     "+a"(aW1[c][0][1]), "+a"(aW1[c][1][0]), "+a"(aW1[c][1][1]));
    float* __restrict__ row = wpart + ((size_t)(wg.net_i * (groups * kWv)) + (size_t)wg.grp * kWv + wave) * PSTRIDE;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int j0 = wg.sl * HS + c * 32 + 16 * jb;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    row[Blk<IN>::W2 + (16 * b + 4 * q + r) * rp::HID + j0 + l15] = aW2[c][b][jb][r];   // dW2[o][j]: lane n = j
                    row[Blk<IN>::W1 + (j0 + 4 * q + r) * IN + 16 * b + l15] = aW1[c][jb][b][r];        // dW1[j][i]: lane n = i
                }
            float v = adb1[c][jb];   // lane (j, q) holds the sum over its samples: add the four sample groups
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (q == 0) row[Blk<IN>::B1 + j0 + l15] = v;
        }
}
