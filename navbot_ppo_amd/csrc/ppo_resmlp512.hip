// ppo_resmlp512.hip -- fused PPO update + rollout-time policy step for the reference's ACTIVE nets on gfx950 (f32 MFMA).
//
// Nets (project_ppo/src/net_actor.py:16-144, net_critic.py:13-130), D = 16 observation floats:
//   rb1   h1 = leaky(x + W2a leaky(W1a x + b1a) + b2a)            W1a [512,16], W2a [16,512]
//   X1    = cat[x, h1]                                             (32)
//   rb2   h2 = leaky(X1 + W2b leaky(W1b X1 + b1b) + b2b)           W1b [512,32], W2b [32,512]
//   actor mean = (sigmoid(wo1 . h2 + bo1), tanh(wo2 . h2 + bo2)) ;  critic V = wo . h2 + bo        LeakyReLU(0.2) throughout
// and one epoch of the update loop (project_ppo/src/ppo.py:305-397): evaluate() (:708-737), ratios / clipped surrogate
// (:316-320,342), MSE critic loss (:343), both backward() calls (:349,386) and both Adam steps (:381,392).
//
// Why kernels: through PyTorch every [2.1 M, 512] f32 hidden activation (4.3 GB) makes several HBM round trips per epoch
// (round 2: 41 ms per epoch, 19 % of the f32-MFMA peak).  Here a 512-wide hidden activation never leaves the registers of
// the wave that produced it.
//
// What does not fit on a CU, and the decomposition that follows from it.  One net has 192 KB of weights and as many weight
// gradients; a CU has 160 KB of LDS and 512 KB of registers.  So the hidden dimension is cut into NSL = 4 SLICES of 128
// units and a workgroup owns (net, slice, group of sample tiles): its slice of the weights sits in LDS (<= 37 KB), its slice
// of the weight gradients in accumulator registers (<= 128 per lane) for the whole launch, and a WAVE owns a 32-sample
// tile end to end (no workgroup barrier in the tile loop, as in ppo_mlp64.hip).  A residual block's output is a sum over the
// hidden units, i.e. over the slices: each slice writes its partial [n, 16 | 32] sum to HBM, and whoever needs the block's
// output adds the four partials and applies the element-wise part (residual, bias, LeakyReLU ...) on the way in.  Per epoch:
//   fwd<16>  P1[s] = W2a[:, s] leaky(W1a[s] x + b1a[s])                 (MFMA)      s = slice
//   fwd<32>  h1 = leaky(x + b2a + sum_s P1[s]) ; P2[s] = W2b[:, s] leaky(W1b[s] [x, h1] + b1b[s])           (MFMA)
//   E2       h2 = leaky([x, h1] + b2b + sum_s P2[s]), heads, PPO loss / MSE, dpre2 = dL/d(pre-activation of h2), and the
//            sample sums d(heads), db2b, statistics                      (streaming kernel: 0.7 ms of an 11.3 ms epoch)
//   bwd<32>  per slice: hidden recomputed, dH = W2b[:, s]^T dpre2 . leaky', dW2b[:, s] += dpre2 H^T, dW1b[s] += dH X1^T,
//            db1b[s], Q[s] = W1b[s][:, 16:32]^T dH                       (MFMA)
//   bwd<16>  dpre1 = (sum_s Q[s] + dpre2[16:32]) . leaky'(h1), db2a ; then the same for rb1 (no input gradient)   (MFMA)
//   reduce   partial rows -> gradient (-> Adam in place)
// (Round 3 history: h1 and dpre1 first had streaming kernels of their own, 0.25 + 0.44 ms per epoch; folded into the
// consumers' tile prologues they cost ~0.1 ms of VALU time, redundantly in the four slice workgroups.)
// The hidden activation is recomputed in the backward kernels (K = 16 / 32: 16 % more MFMA work) instead of being stored
// (4 KB per sample and net).  MFMA work per sample and net: 2 x 155,648 MAC = 311 kFLOP -> 1.31 TFLOP per epoch at
// BASELINE configs[1] = 8.3 ms at the 157.3 TF f32-MFMA peak.
//
// The GEMMs were all v_mfma_f32_16x16x4_f32 through round 4 (exact k-ordered f32 fma chains; same rate as 32x32x2, and the 16-row
// output blocks of rb1 waste nothing).  Since round 5 the products that fill a whole k-step of v_mfma_f32_16x16x32_bf16 run as
// float32 products from three-piece bf16 splits ("bf16x3": csrc/bf16x3.h; six piece products, float32 accumulate, small terms
// first -- float32-equivalent, the same gates as before: tests/test_gpu_resmlp512.py):
//   * narrow x narrow (weights x the 16 / 32 values a sample carries in or out; nothing to split per hidden value): H of both
//     forward kernels (hidden_chunk_x3), H^T and dH^T of resmlp_bwd<32>;
//   * one operand hidden, K = 32 and 32 outputs (the split of a hidden register, 27 cycles, against 2 x 32 cycles of f32 MFMA it
//     replaces): Y = W2 leaky(H) of resmlp_fwd<32>, dW2 and dW1 of resmlp_bwd<32> (one 32-sample k-step each).
// resmlp_bwd<32> runs 4 waves x 512 registers for that (at 8 x 256 the pieces cost 66 -> 125 spilled registers and the gain).
// Per epoch: resmlp_fwd<32> 1781 -> 1143 us, resmlp_fwd<16> 876 -> 819, resmlp_bwd<32> 4988 -> 4304-4338; 11.18 -> 9.90 ms.  Left on
// the f32 MFMA because the count says so or the measurement did: rb1's 16-wide products with a hidden operand (the split costs
// more than it saves), resmlp_bwd<16> altogether (K = 16 fills half a k-step; both narrow products split: 2208 vs 2269-2308 us,
// no room in its 256 registers), Q = W1^T dH of resmlp_bwd<32> (dH would need a second, transposed split).
// Round 6: resmlp_bwd<32> is replaced at the launch by resmlp_bwd2s (csrc/ppo_resmlp512_bwd2s.h: the same products as one hand-placed,
// half-chunk-pipelined instruction stream; X / dY split once per tile, their transposes and Q's operand through bf16 LDS images and
// ds_read_b64_tr_b16; Q on the bf16 MFMA): 4304 -> 3204 us, 9.90 -> 8.81 ms per epoch.  resmlp_bwd<32, 2, 4> stays in the file as the
// compiler-scheduled statement of the same arithmetic (-DRESMLP_BWD2S=0 selects it).
// Layout ("16-layout"): a [rows, 32 samples] activation lives as f32x4 v[rows / 16][2]: lane
// (n = lane & 15, q = lane >> 4) holds rows 16 b + 4 q + r (r = 0..3) of sample 16 st + n -- which is both the C/D layout of
// the MFMA and, register r taken as the B operand of step r, a legal k-pairing when the A operand (weights) is read
// k-permuted: lane (m, q) fetches columns 16 b + 4 q .. + 3 of its row with ONE ds_read_b128.  Products that contract over
// samples (the weight gradients) need the transposed tiles: see resmlp_bwd.
//
// What bounds these kernels (profiles/r03_resmlp512_*): f32-input MFMA runs on the SIMD's FMA lanes, so every VALU
// instruction (4 cycles per wave) adds to the MFMA time instead of hiding under it, and the loop sustains ~2.2 GHz, not the
// nominal 2.4: time ~ (32 cycles x MFMAs + 4 cycles x VALU) / 2.2 GHz, which the four MFMA kernels reach to 92-100 %.
//
// Arithmetic: float32; sums over hidden units are taken slice by slice and over samples tile by tile, so results agree with
// PyTorch autograd to f32 round-off (tests: <= 2e-4 of each tensor's scale), not bit for bit.  Deterministic: no atomics.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <string>
#include <type_traits>

#include "mlp64_policy.h"   // Philox / Box-Muller noise of the rollout policy step (same stream as the mlp64x2 path)
#include "navppo.h"
#include "navppo_internal.h"
#include "bf16x3.h"          // float32 values as three bf16 pieces (the H product of rb2 below)
#include "resmlp_policy.h"   // the rollout-time policy step + the parameter layout (shared with the persistent rollout kernel)

namespace {

using resmlp::f32x4;
using resmlp::ld4;
using resmlp::leaky;
using resmlp::mfma16;
using resmlp::v4;
using resmlp::zero4;
namespace rp = resmlp::rp;
static_assert(rp::P_ACTOR == NAVPPO_RESMLP512_ACTOR_PARAMS && rp::P_CRITIC == NAVPPO_RESMLP512_CRITIC_PARAMS, "layout");

constexpr int NSL = 4;               // hidden slices
constexpr int HS = rp::HID / NSL;    // 128 hidden units per slice
constexpr int NCH = HS / 32;         // chunks of 32 hidden units per slice
// Round 6: a FORWARD workgroup runs FSP = 2 slices of its tile back to back into the same accumulators and writes ONE partial sum for the
// pair: NSLF = 2 partials per sample and net instead of 4 (resmlp_fwd<32> reads half as many of rb1's, resmlp_e2 -- HBM-bound, 0.69 ms per
// epoch, two thirds of it the four partials of rb2 -- half as many of rb2's), and a tile's input split is shared by 8 chunks instead of 4.
// The backward kernels keep one slice per workgroup (their LDS is full).
constexpr int FSP = 2, NSLF = NSL / FSP;
constexpr int kWaves = 8, kThreads = 64 * kWaves;
#ifndef RESMLP_BWD2_WAVES
#define RESMLP_BWD2_WAVES 4
#endif
#ifndef RESMLP_BWD1_STACK
#define RESMLP_BWD1_STACK 1
#endif
#ifndef RESMLP_BWD1_WAVES
#define RESMLP_BWD1_WAVES 8
#endif
constexpr int kBwd1Waves = RESMLP_BWD1_WAVES;   // waves per workgroup of resmlp_bwd<16>
constexpr int kBwd2Waves = RESMLP_BWD2_WAVES;   // waves per workgroup of resmlp_bwd<32> (4: with its narrow products on the bf16 MFMA)
constexpr int LT = 36;               // row pitch of the wave tiles (floats): 16-byte rows, conflict-free ds_read_b128
constexpr int kMaxWG = 256;          // one persistent workgroup per CU
constexpr int kWRows = kMaxWG / NSL * kWaves;   // partial-gradient rows: 2 nets x 32 groups x 8 waves
constexpr int PSTRIDE = 50304;       // pitch of a partial-gradient row (>= P_ACTOR, multiple of 64)
constexpr int EP = 128;              // pitch of a streaming-kernel partial row
constexpr int kEMaxBlocks = 1024, kEThreads = 256;
// columns of a streaming-kernel partial row
constexpr int EC_B2B = 16, EC_WO1 = 48, EC_BO1 = 80, EC_WO2 = 81, EC_BO2 = 113, EC_STAT = 120;
constexpr int kGnSlotsR = 1600;   // squared-norm slots of resmlp_reduce per (parity, net): one per block (1572); columns 0 .. 7 of 2 x kEMaxBlocks partial rows
static_assert((rp::P_ACTOR + rp::P_CRITIC + 63) / 64 <= kGnSlotsR && 4 * kGnSlotsR <= 8 * 2 * kEMaxBlocks, "slots");

template <int IN> struct Blk;
template <> struct Blk<16> { static constexpr int W1 = rp::W1A, B1 = rp::B1A, W2 = rp::W2A; };
template <> struct Blk<32> { static constexpr int W1 = rp::W1B, B1 = rp::B1B, W2 = rp::W2B; };

__device__ __forceinline__ float dleaky(float g, float act) { return act > 0.f ? g : 0.2f * g; }   // sign(leaky(x)) == sign(x)
__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wave are performed in issue order: this only stops the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct WG { int net_i, sl, grp; };
// blockIdx -> (net, slice, group).  With a multiple of 8 groups the 4 (or 8) workgroups that stream the SAME sample tiles
// (the slices of a group, both nets) get block ids that differ by multiples of 8, i.e. the same XCD and the same L2.
__device__ __forceinline__ WG decode_block(int b, int n_nets, int groups, int nsl = NSL) {
    const int nsc = nsl * n_nets;
    int ns, grp;
    if ((groups & 7) == 0) {
        const int xcd = b & 7, rest = b >> 3;
        ns = rest % nsc;
        grp = xcd + 8 * (rest / nsc);
    } else {
        ns = b % nsc;
        grp = b / nsc;
    }
    return WG{ns / nsl, ns % nsl, grp};
}

// Four consecutive columns of an observation row, element index idx = s * 16 + 4 q: float32 rows, or float16 rows (obs_f16: BASELINE
// configs[4]'s "fp16 obs buffers", navsim_cfg.obs_f16) widened at the load -- everything behind the load is the same float32 arithmetic.
__device__ __forceinline__ float4 ldx4(const void* __restrict__ obs, const long long idx, const int obs_f16) {
    if (obs_f16) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const __half*>(obs) + idx);
        const __half2 lo = *reinterpret_cast<const __half2*>(&u.x), hi = *reinterpret_cast<const __half2*>(&u.y);
        const float2 a = __half22float2(lo), b = __half22float2(hi);
        return make_float4(a.x, a.y, b.x, b.y);
    }
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(obs) + idx);
}

// the wave's [IN, 32] input tile in the 16-layout: rows 0..15 = observation, rows 16..31 = h1 (IN == 32)
template <int IN>
__device__ __forceinline__ void load_x(f32x4 (&X)[IN / 16][2], const void* __restrict__ obs, const int obs_f16, const float* __restrict__ h1n,
                                       long long n, long long m0, int l15, int q) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const long long s = m0 + 16 * st + l15;
        const bool v = s < n;
        X[0][st] = v ? v4(ldx4(obs, s * 16 + 4 * q, obs_f16)) : zero4();
        if (IN == 32) X[IN / 16 - 1][st] = v ? v4(ld4(h1n + s * 16 + 4 * q)) : zero4();
    }
}

// H = leaky(W1[chunk] X + b1[chunk]): 32 hidden units x 32 samples, as [2 j-blocks][2 sample tiles].  (Rounds 3-4 ran it as
// v_mfma_f32_16x16x4_f32 chains -- four independent accumulators back to back, the element-wise block in one piece behind them because
// f32 MFMA and vector work share the SIMD's FMA lanes; the backward kernels still form their H^T that way.)
// Since round 5: on the bf16 MFMA from three-piece splits ("bf16x3": csrc/bf16x3.h): K = 32 (rb2) is one k-step of
// v_mfma_f32_16x16x32_bf16, K = 16 (rb1) half of one.  BOTH operands are narrow here -- weights, split once per launch, and the 32 input values of a
// sample, split once per tile -- so the product runs at the bf16 rate with no per-hidden-value split: 6 MFMAs of 16 cycles per
// 16 x 16 tile against 8 f32 MFMAs of 32.  Lane (n, q) of the B operand holds ITS OWN eight registers X[0][st][0..3], X[1][st][0..3]
// (features 4 q + r and 16 + 4 q + r) as the eight k-slots of k-group q; the weight pieces are stored in that order.  Small terms
// first, the a0 b0 term last, the bias on the vector unit behind it (the order of the update kernels of the 64-wide heads).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma16bf(const uint4 a, const uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf16x3::as_bf(a), bf16x3::as_bf(b), c, 0, 0, 0);
}
// K16 (rb1: 16 inputs fill HALF a k-step): TWO piece products share one MFMA, stacked along k -- the operands arrive as the three
// "stacks" (a2 | a1) (a0 | a1) (a0 | a0) of the weights and (b0 | b1) (b2 | b0) (b1 | b0) of the rows, so a2 b0 + a1 b1, a0 b2 + a1 b0,
// a0 b1 + a0 b0 are THREE instructions instead of six (same six products, same small-to-large order; inside one instruction the MFMA
// aligns all addends to the largest and rounds once: tools/ubench/bf16_mfma_rounding.hip).
template <bool K16 = false>
__device__ __forceinline__ void hidden_chunk_x3(const uint4* W1p /* [3][HS][4] */, const float* b1s, int c, const bf16x3::Pieces (&XP)[2],
                                                f32x4 (&H)[2][2], int l15, int q) {
    bf16x3::Pieces A[2];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int i = 0; i < 3; ++i) A[jb].p[i] = W1p[(i * HS + c * 32 + 16 * jb + l15) * 4 + q];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) H[jb][0] = H[jb][1] = zero4();
    // four independent accumulators take turns
#define RESMLP_X3_TERM(ia, ib)                                        \
    H[0][0] = mfma16bf(A[0].p[ia], XP[0].p[ib], H[0][0]);             \
    H[0][1] = mfma16bf(A[0].p[ia], XP[1].p[ib], H[0][1]);             \
    H[1][0] = mfma16bf(A[1].p[ia], XP[0].p[ib], H[1][0]);             \
    H[1][1] = mfma16bf(A[1].p[ia], XP[1].p[ib], H[1][1]);
    if constexpr (K16) {   // (A[jb].p[s], XP[st].p[s] hold stack s)
        RESMLP_X3_TERM(0, 0) RESMLP_X3_TERM(1, 1) RESMLP_X3_TERM(2, 2)
    } else {
        RESMLP_X3_TERM(2, 0) RESMLP_X3_TERM(1, 1) RESMLP_X3_TERM(0, 2) RESMLP_X3_TERM(1, 0) RESMLP_X3_TERM(0, 1) RESMLP_X3_TERM(0, 0)
    }
#undef RESMLP_X3_TERM
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
        const f32x4 b = v4(ld4(b1s + c * 32 + 16 * jb + 4 * q));
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) H[jb][st][r] = leaky(H[jb][st][r] + b[r]);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------- forward partial of one residual block
template <int IN>
struct FwdSmem {
    // (every array: FSP slices one behind the other)
    float W2s[(IN == 32) ? 4 : FSP * IN * (HS + 4)];    // [output o][hidden j] (rb1; rb2 holds W2 as bf16 pieces)
    float b1s[FSP * HS];
    // rb2: W2 as pieces in the order the Y product consumes them: [piece][chunk c][output block ob][lane (m, q)] = eight bf16
    // W2[16 ob + m][32 c + 4 q + r], W2[16 ob + m][32 c + 16 + 4 q + r] -- the k-slots in which lane (n, q) holds its own H registers
    uint4 W2p[(IN == 32) ? FSP * 3 * NCH * 2 * 64 : 1];
    uint4 W1p[FSP * 3 * HS * 4];       // [piece][hidden j][k-group q] = eight bf16: W1[j][4 q + r], then W1[j][16 + 4 q + r] (rb1: [stack][j][q], two pieces of W1[j][4 q + r])
};

// pout[net][slice][n][IN] = W2[:, slice] leaky(W1[slice] X + b1[slice])
// IN == 32: rows 16..31 of the input are h1 = leaky(x + b2a + sum_s P1[s]) (net_actor.py:41-53), formed here from the four
// partials of resmlp_fwd<16> (p1) instead of by a streaming kernel of its own (0.25 ms per epoch); the slice-0 workgroups
// store it to h1buf for the kernels behind.
template <int IN>
__global__ __launch_bounds__(kThreads) void resmlp_fwd(const float* __restrict__ params, int net_base, int n_nets,
                                                       const void* __restrict__ obs, const float* __restrict__ p1,
                                                       float* __restrict__ h1buf, long long n, int groups,
                                                       float* __restrict__ pout, int obs_f16) {
    __shared__ __attribute__((aligned(16))) FwdSmem<IN> sm;
    constexpr int S2 = HS + 4, NB = IN / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, q = lane >> 4;
    const WG wg = decode_block(blockIdx.x, n_nets, groups, NSLF);   // wg.sl: the PAIR of slices FSP wg.sl, FSP wg.sl + 1
    const float* __restrict__ pn = params + ((net_base + wg.net_i) ? rp::P_ACTOR : 0);
    constexpr int W1P_F = 3 * HS * 4, W2P_F = 3 * NCH * 2 * 64, W2S_F = IN * S2;   // one slice's share of the arrays
    for (int fs = 0; fs < FSP; ++fs) {
        const int sl = wg.sl * FSP + fs;
        uint4* const W1p = sm.W1p + fs * W1P_F;
        for (int k = tid; k < HS * 4; k += kThreads) {
            const int j = k >> 2, kq = k & 3;
            const float* wr = pn + Blk<IN>::W1 + (sl * HS + j) * IN;
            const float4 lo = ld4(wr + 4 * kq), hi = (IN == 32) ? ld4(wr + (IN == 32 ? 16 : 0) + 4 * kq) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const bf16x3::Pieces P = bf16x3::split8(v);
            if constexpr (IN == 16) {   // the stacks (a2 | a1) (a0 | a1) (a0 | a0): see hidden_chunk_x3<true>
                W1p[(0 * HS + j) * 4 + kq] = make_uint4(P.p[2].x, P.p[2].y, P.p[1].x, P.p[1].y);
                W1p[(1 * HS + j) * 4 + kq] = make_uint4(P.p[0].x, P.p[0].y, P.p[1].x, P.p[1].y);
                W1p[(2 * HS + j) * 4 + kq] = make_uint4(P.p[0].x, P.p[0].y, P.p[0].x, P.p[0].y);
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i) W1p[(i * HS + j) * 4 + kq] = P.p[i];
            }
        }
        if constexpr (IN == 32) {
            uint4* const W2p = sm.W2p + fs * W2P_F;
            for (int k = tid; k < NCH * NB * 64; k += kThreads) {
                const int ln = k & 63, ob = (k >> 6) % NB, c = k / (64 * NB), m = ln & 15, kq = ln >> 4;
                const float* wr = pn + Blk<IN>::W2 + (size_t)(16 * ob + m) * rp::HID + sl * HS + 32 * c + 4 * kq;
                const float4 lo = ld4(wr), hi = ld4(wr + 16);
                const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                const bf16x3::Pieces P = bf16x3::split8(v);
#pragma unroll
                for (int i = 0; i < 3; ++i) W2p[((i * NCH + c) * NB + ob) * 64 + ln] = P.p[i];
            }
        } else {
            float* const W2s = sm.W2s + fs * W2S_F;
            for (int k = tid; k < IN * HS; k += kThreads) W2s[(k / HS) * S2 + (k % HS)] = pn[Blk<IN>::W2 + (k / HS) * rp::HID + sl * HS + (k % HS)];
        }
        if (tid < HS) sm.b1s[fs * HS + tid] = pn[Blk<IN>::B1 + sl * HS + tid];
    }
    __syncthreads();
    float* __restrict__ po = pout + (size_t)(wg.net_i * NSLF + wg.sl) * n * IN;
    const long long n_tiles = (n + 31) / 32, stride = (long long)groups * kWaves;
    // rb2: the four rb1 partials of this net and the output bias of rb1 (rows 4 q .. 4 q + 3 of the lane)
    const float* __restrict__ pp = IN == 32 ? p1 + (size_t)wg.net_i * NSLF * n * 16 : nullptr;
    float* __restrict__ ho = (IN == 32 && wg.sl == 0) ? h1buf + (size_t)wg.net_i * n * 16 : nullptr;
    f32x4 b2 = zero4(), b2o[NB];   // rb1's output bias (h1 is formed here); rb2's own output bias, rows 16 ob + 4 q + r (pair 0 adds it)
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) b2o[ob] = zero4();
    if (IN == 32)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            b2[r] = pn[rp::B2A + 4 * q + r];
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) b2o[ob][r] = pn[rp::B2B + 16 * ob + 4 * q + r];
        }
    f32x4 xn[2], pa[IN == 32 ? NSLF : 1][2];   // the next tile's rows, requested one tile ahead
    auto request = [&](long long t) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const long long s = t * 32 + 16 * st + l15;
            const bool v = s < n;
            xn[st] = v ? v4(ldx4(obs, s * 16 + 4 * q, obs_f16)) : zero4();
            if (IN == 32)
#pragma unroll
                for (int k = 0; k < NSLF; ++k) pa[k][st] = v ? v4(ld4(pp + ((size_t)k * n + s) * 16 + 4 * q)) : zero4();
        }
    };
    long long tile = (long long)wg.grp * kWaves + wave;
    if (tile < n_tiles) request(tile);
    for (; tile < n_tiles; tile += stride) {
        f32x4 X[NB][2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            X[0][st] = xn[st];
            if (IN == 32) {
                // the arithmetic of net_actor.py:41-53 summed slice pair by slice pair: (x + b) + (p01 + p23)
                f32x4 h = (xn[st] + b2) + (pa[0][st] + pa[IN == 32 ? 1 : 0][st]);
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = leaky(h[r]);
                X[NB - 1][st] = h;
                const long long s = tile * 32 + 16 * st + l15;
                if (ho && s < n) *reinterpret_cast<float4*>(ho + s * 16 + 4 * q) = make_float4(h[0], h[1], h[2], h[3]);
            }
        }
        if (tile + stride < n_tiles) request(tile + stride);   // next tile streams in
        f32x4 Y[NB][2];
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) Y[ob][0] = Y[ob][1] = zero4();
        bf16x3::Pieces XP[2];   // the tile's input rows as the B operand of the H product, split once per tile (rb1: k-slots 4 .. 7 empty)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if constexpr (IN == 32) {
                const float v[8] = {X[0][st][0], X[0][st][1], X[0][st][2], X[0][st][3], X[NB - 1][st][0], X[NB - 1][st][1], X[NB - 1][st][2], X[NB - 1][st][3]};
                XP[st] = bf16x3::split8(v);
            } else {
                uint32_t w[3][2];   // the stacks (b0 | b1) (b2 | b0) (b1 | b0) of the lane's four values
                bf16x3::split_pair(X[0][st][0], X[0][st][1], w[0][0], w[1][0], w[2][0]);
                bf16x3::split_pair(X[0][st][2], X[0][st][3], w[0][1], w[1][1], w[2][1]);
                XP[st].p[0] = make_uint4(w[0][0], w[0][1], w[1][0], w[1][1]);
                XP[st].p[1] = make_uint4(w[2][0], w[2][1], w[0][0], w[0][1]);
                XP[st].p[2] = make_uint4(w[1][0], w[1][1], w[0][0], w[0][1]);
            }
        }
#pragma unroll 1
        for (int fs = 0; fs < FSP; ++fs) {   // the pair's slices one after the other, into the same Y
        const uint4* const W1p = sm.W1p + fs * W1P_F;
        const float* const b1s = sm.b1s + fs * HS;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            f32x4 H[2][2];
            hidden_chunk_x3<IN == 16>(W1p, b1s, c, XP, H, l15, q);
            if constexpr (IN == 32) {
                // Y += W2[:, chunk] leaky(H): the chunk's 32 hidden units are ONE k-step; a lane's eight H registers per sample tile
                // (units 4 q + r and 16 + 4 q + r) are its k-slots, split here -- 2 x 44 vector instructions against 24 MFMAs of 16
                // cycles instead of 32 of 32 (rb1's 16 outputs would not pay for the split: it keeps the f32 chain below)
                bf16x3::Pieces HP[2];
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const float v[8] = {H[0][st][0], H[0][st][1], H[0][st][2], H[0][st][3], H[1][st][0], H[1][st][1], H[1][st][2], H[1][st][3]};
                    HP[st] = bf16x3::split8(v);
                }
                constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int ob = 0; ob < NB; ++ob) {
                        const uint4 a = sm.W2p[(IN == 32 ? fs * W2P_F : 0) + ((TA[t] * NCH + c) * NB + ob) * 64 + lane];
                        Y[ob][0] = mfma16bf(a, HP[0].p[TB[t]], Y[ob][0]);
                        Y[ob][1] = mfma16bf(a, HP[1].p[TB[t]], Y[ob][1]);
                    }
            } else
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                f32x4 a[NB];
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) a[ob] = v4(ld4(sm.W2s + (IN == 32 ? 0 : fs * W2S_F) + (16 * ob + l15) * S2 + c * 32 + 16 * jb + 4 * q));
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ob = 0; ob < NB; ++ob) {
                        Y[ob][0] = mfma16(a[ob][r], H[jb][0][r], Y[ob][0]);
                        Y[ob][1] = mfma16(a[ob][r], H[jb][1][r], Y[ob][1]);
                    }
            }
        }
        }
        if (IN == 32 && wg.sl == 0) {   // rb2, pair 0: its partial carries the residual input and the block's output bias (resmlp_e2 reads neither)
#pragma unroll
            for (int ob = 0; ob < NB; ++ob)
#pragma unroll
                for (int st = 0; st < 2; ++st) Y[ob][st] += X[ob][st] + b2o[ob];
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const long long s = tile * 32 + 16 * st + l15;
            if (s < n)
#pragma unroll
                for (int ob = 0; ob < NB; ++ob)
                    *reinterpret_cast<float4*>(po + s * IN + 16 * ob + 4 * q) = make_float4(Y[ob][st][0], Y[ob][st][1], Y[ob][st][2], Y[ob][st][3]);
        }
    }
}

// ---------------------------------------------------------------- backward of one residual block, one hidden slice
// The weight gradients contract over SAMPLES, the activations were produced contracting over FEATURES.  Instead of
// transposing H and dH through LDS, the backward kernel computes them already transposed: the MFMA's output has its n index
// on lanes and its m index on registers, so swapping the two operands of the same instruction,
//   mfma(A = X  (lane = sample, registers = features), B = W1 rows (lane = hidden unit))   ->  H^T: lane = hidden unit,
// yields the tile with lane = hidden unit j and registers = samples ("R-layout"), which IS the B operand of
// dW2[o][j] += sum_s dY[o][s] H[j][s] and, for dH, the A operand of dW1[j][i] += sum_s dH[j][s] X[i][s].  Their partners
// dY^T / X^T (lane = output / input unit, registers = samples) come from two wave-private LDS tiles written once per tile.
// Only rb2's input gradient Q = W1[:, 16:32]^T dH contracts over hidden units again: dH makes one trip through an LDS tile
// (ds_write_b128 rows, conflict-free ds_read_b32 columns).
template <int IN, int NWV>
struct BwdSmem {
    static constexpr bool X3 = (IN == 32 && NWV == 4);   // rb2 on 4 waves: H^T and dH^T as bf16x3 products (below)
    // rb1 (round 6): its two narrow products H^T = X W1^T and dH^T = dpre1^T W2 have K = 16 = HALF a k-step of v_mfma_f32_16x16x32_bf16, so
    // two piece products share an instruction, stacked along k (hidden_chunk_x3<true>): 3 MFMAs of 16 cycles per 16 x 16 tile instead of
    // 4 float32 MFMAs of 32 (round 5's unstacked six: 2208 vs 2269-2308 us, not kept).  The weight-gradient products stay float32.
    static constexpr bool KS = (IN == 16) && RESMLP_BWD1_STACK;
    float W1s[HS * (IN + 4)];     // [hidden j][input i]   (X3: still the source of Q's A operand)
    float W2Ts[X3 ? 4 : HS * (IN + 4)];    // [hidden j][output o]
    float b1s[HS];
    uint4 W1p[(X3 || KS) ? 3 * HS * 4 : 1], W2Tp[(X3 || KS) ? 3 * HS * 4 : 1];   // X3: [piece][hidden j][k-group q] = eight bf16 (hidden_chunk_x3's order); KS: [stack][j][q]
    float tiles[NWV * (2 * IN + (IN == 32 ? 32 : 0)) * LT];   // per wave: TX [IN][32] | TDY [IN][32] | TH [32][32] (rb2)
};
static_assert(sizeof(BwdSmem<32, 8>) <= 160 * 1024 && sizeof(BwdSmem<32, 4>) <= 160 * 1024, "LDS");

// dypre [net][n][32] = dL / d(pre-activation of h2) (the streaming kernel E2 writes it).  rb2 (IN == 32) uses it as is and
// writes qout[net][slice][n][16] = W1[slice][:, 16:32]^T dH; rb1 (IN == 16) forms ITS output gradient on the way in,
//   dpre1 = (sum_s Q[s] + dpre2[16:32]) . leaky'(h1)        (the residual path of rb2 + its hidden path, net_actor.py:41-53)
// from qin (the four partials rb2's launch wrote), dypre and h1buf, and its slice-0 workgroups add up db2a = sum dpre1
// (this was a streaming kernel of its own: 0.44 ms per epoch).  Both write this slice's weight-gradient partials into row
// (net, grp * 8 + wave) of wpart.
// NST = sample tiles of a chunk per pass.  rb2 keeps 128 accumulator registers per lane and the compiler spills ~70 more next
// to a whole 32 x 32 chunk of H / dH (NST = 2), mostly outside the tile loop (27 scratch accesses per tile); measured
// alternatives, all slower: half chunks (NST = 1: no fewer spills, twice the weight reads, 5.54 vs 5.10 ms), 4 waves x 512
// registers with two tiles in flight (no spills, but ~270 AGPR <-> VGPR moves per tile: 5.53 ms).
// NWV = waves per workgroup.  rb2 runs 4 (one per SIMD, 512 registers each): its two products with narrow operands -- H^T = X W1^T and
// dH^T = dY^T W2, K = 32 -- then fit as float32 products from three-piece bf16 splits (hidden_chunk_x3's scheme with the operands
// swapped: A = the lane's own eight X / dY registers split once per tile, B = the weight pieces), which the 8-wave build has no
// registers for (66 -> 125 spilled).
template <int IN, int NST, int NWV>
__global__ __launch_bounds__(64 * NWV) void resmlp_bwd(const float* __restrict__ params, int n_nets, const void* __restrict__ obs,
                                                       const float* __restrict__ h1buf, const float* __restrict__ dypre,
                                                       long long n, int groups, float* __restrict__ wpart,
                                                       float* __restrict__ qout, const float* __restrict__ qin, int obs_f16) {
    __shared__ __attribute__((aligned(16))) BwdSmem<IN, NWV> sm;
    constexpr int S1 = IN + 4, NB = IN / 16;
    constexpr int kWaves = NWV, kThreads = 64 * NWV;   // (shadow the 8-wave constants of the file)
    constexpr bool X3 = BwdSmem<IN, NWV>::X3, KS = BwdSmem<IN, NWV>::KS;
    constexpr bool NEED_DX = IN == 32;
    constexpr bool PREFETCH = IN == 16;   // rb2 has no registers to spare for the next tile's rows
    constexpr int TILE_F = (2 * IN + (IN == 32 ? 32 : 0)) * LT;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, q = lane >> 4;
    const WG wg = decode_block(blockIdx.x, n_nets, groups);
    const float* __restrict__ pn = params + (wg.net_i ? rp::P_ACTOR : 0);
    for (int k = tid; k < HS * IN; k += kThreads) sm.W1s[(k / IN) * S1 + (k % IN)] = pn[Blk<IN>::W1 + (wg.sl * HS + k / IN) * IN + (k % IN)];
    if constexpr (X3) {
        for (int k = tid; k < HS * 4; k += kThreads) {
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
    } else {
        for (int k = tid; k < IN * HS; k += kThreads) sm.W2Ts[(k % HS) * S1 + (k / HS)] = pn[Blk<IN>::W2 + (k / HS) * rp::HID + wg.sl * HS + (k % HS)];
    }
    if constexpr (KS) {   // the weights' stacks (a2 | a1) (a0 | a1) (a0 | a0) of a lane's four k-values
        for (int k = tid; k < HS * 4; k += kThreads) {
            const int j = k >> 2, kq = k & 3;
            const float4 w1 = ld4(pn + Blk<IN>::W1 + (wg.sl * HS + j) * IN + 4 * kq);
            const float* wc = pn + Blk<IN>::W2 + wg.sl * HS + j;
            const float w2[4] = {wc[(size_t)(4 * kq) * rp::HID], wc[(size_t)(4 * kq + 1) * rp::HID], wc[(size_t)(4 * kq + 2) * rp::HID], wc[(size_t)(4 * kq + 3) * rp::HID]};
            uint32_t a[3][2], b[3][2];
            bf16x3::split_pair(w1.x, w1.y, a[0][0], a[1][0], a[2][0]);
            bf16x3::split_pair(w1.z, w1.w, a[0][1], a[1][1], a[2][1]);
            bf16x3::split_pair(w2[0], w2[1], b[0][0], b[1][0], b[2][0]);
            bf16x3::split_pair(w2[2], w2[3], b[0][1], b[1][1], b[2][1]);
            sm.W1p[(0 * HS + j) * 4 + kq] = make_uint4(a[2][0], a[2][1], a[1][0], a[1][1]);
            sm.W1p[(1 * HS + j) * 4 + kq] = make_uint4(a[0][0], a[0][1], a[1][0], a[1][1]);
            sm.W1p[(2 * HS + j) * 4 + kq] = make_uint4(a[0][0], a[0][1], a[0][0], a[0][1]);
            sm.W2Tp[(0 * HS + j) * 4 + kq] = make_uint4(b[2][0], b[2][1], b[1][0], b[1][1]);
            sm.W2Tp[(1 * HS + j) * 4 + kq] = make_uint4(b[0][0], b[0][1], b[1][0], b[1][1]);
            sm.W2Tp[(2 * HS + j) * 4 + kq] = make_uint4(b[0][0], b[0][1], b[0][0], b[0][1]);
        }
    }
    if (tid < HS) sm.b1s[tid] = pn[Blk<IN>::B1 + wg.sl * HS + tid];
    __syncthreads();
    float* const TX = sm.tiles + wave * TILE_F;
    float* const TDY = TX + IN * LT;
    float* const TH = TDY + IN * LT;          // rb2 only
    const int wr = 4 * q * LT + l15;          // S-layout store: tile[row 16 b + 4 q + r][sample 16 st + l15]: + (16 b + r) * LT + 16 st
    const int rr = l15 * LT + 4 * q;          // R-layout access: row 16 b + l15, samples 16 st + 4 q .. + 3:       + 16 b * LT + 16 st

    const float* __restrict__ h1n = h1buf + (size_t)wg.net_i * n * 16;
    const float* __restrict__ dyn = dypre + (size_t)wg.net_i * n * 32;
    float* __restrict__ qo = NEED_DX ? qout + (size_t)(wg.net_i * NSL + wg.sl) * n * 16 : nullptr;
    const float* __restrict__ qq = NEED_DX ? nullptr : qin + (size_t)wg.net_i * NSL * n * 16;
    f32x4 adb2 = zero4();   // rb1: db2a rows 4 q .. 4 q + 3, summed over this lane's samples

    f32x4 aW2[NCH][NB][2];   // dW2[o block][j block of chunk c]
    f32x4 aW1[NCH][2][NB];   // dW1[j block of chunk c][i block]
    float adb1[NCH][2];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            adb1[c][jb] = 0.f;
#pragma unroll
            for (int b = 0; b < NB; ++b) aW2[c][b][jb] = aW1[c][jb][b] = zero4();
        }

    const long long n_tiles = (n + 31) / 32, stride = (long long)groups * kWaves;
    // raw rows of a tile (past n: zeros, they add nothing).  rb2: X = [x, h1], G = dpre2.  rb1: X = x, G[0] = dpre2[16:32],
    // RQ = h1 and the four Q partials; finish() turns them into the block's output gradient.
    struct Raw { f32x4 X[NB][2], G[NB][2], RQ[NEED_DX ? 1 : 1 + NSL][2]; };
    auto request = [&](Raw& w, long long tile) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const long long s = tile * 32 + 16 * st + l15;
            const bool v = s < n;
            w.X[0][st] = v ? v4(ldx4(obs, s * 16 + 4 * q, obs_f16)) : zero4();
            if (NEED_DX) {
                w.X[NB - 1][st] = v ? v4(ld4(h1n + s * 16 + 4 * q)) : zero4();
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) w.G[ob][st] = v ? v4(ld4(dyn + s * 32 + 16 * ob + 4 * q)) : zero4();
            } else {
                w.G[0][st] = v ? v4(ld4(dyn + s * 32 + 16 + 4 * q)) : zero4();
                w.RQ[0][st] = v ? v4(ld4(h1n + s * 16 + 4 * q)) : zero4();
#pragma unroll
                for (int k = 0; k < NSL; ++k) w.RQ[NEED_DX ? 0 : 1 + k][st] = v ? v4(ld4(qq + ((size_t)k * n + s) * 16 + 4 * q)) : zero4();
            }
        }
    };
    auto finish = [&](const Raw& w, f32x4 (&X)[NB][2], f32x4 (&DY)[NB][2]) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int b = 0; b < NB; ++b) X[b][st] = w.X[b][st];
            if (NEED_DX) {
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) DY[ob][st] = w.G[ob][st];
            } else {
                f32x4 d = w.G[0][st] + ((w.RQ[NEED_DX ? 0 : 1][st] + w.RQ[NEED_DX ? 0 : 2][st]) + (w.RQ[NEED_DX ? 0 : 3][st] + w.RQ[NEED_DX ? 0 : 4][st]));
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r] = dleaky(d[r], w.RQ[0][st][r]);
                DY[0][st] = d;
                adb2 += d;
            }
        }
    };
    f32x4 X[NB][2], DY[NB][2];   // S-layout: lane = sample, registers = feature rows
    long long tile = (long long)wg.grp * kWaves + wave;
    Raw raw;
    request(raw, tile);
    for (; tile < n_tiles; tile += stride) {
        finish(raw, X, DY);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    TX[wr + (16 * b + r) * LT + 16 * st] = X[b][st][r];
                    TDY[wr + (16 * b + r) * LT + 16 * st] = DY[b][st][r];
                }
        f32x4 dXa[2] = {zero4(), zero4()};
        bf16x3::Pieces XP[2], DYP[2];   // X3: the tile's rows / output gradients as A operands, split once per tile
        if constexpr (X3) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const float vx[8] = {X[0][st][0], X[0][st][1], X[0][st][2], X[0][st][3], X[NB - 1][st][0], X[NB - 1][st][1], X[NB - 1][st][2], X[NB - 1][st][3]};
                const float vd[8] = {DY[0][st][0], DY[0][st][1], DY[0][st][2], DY[0][st][3], DY[NB - 1][st][0], DY[NB - 1][st][1], DY[NB - 1][st][2], DY[NB - 1][st][3]};
                XP[st] = bf16x3::split8(vx);
                DYP[st] = bf16x3::split8(vd);
            }
        }
        if constexpr (KS) {   // the stacks (b0 | b1) (b2 | b0) (b1 | b0) of the lane's four values of a row / of an output gradient
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                uint32_t w[3][2], g[3][2];
                bf16x3::split_pair(X[0][st][0], X[0][st][1], w[0][0], w[1][0], w[2][0]);
                bf16x3::split_pair(X[0][st][2], X[0][st][3], w[0][1], w[1][1], w[2][1]);
                bf16x3::split_pair(DY[0][st][0], DY[0][st][1], g[0][0], g[1][0], g[2][0]);
                bf16x3::split_pair(DY[0][st][2], DY[0][st][3], g[0][1], g[1][1], g[2][1]);
                XP[st].p[0] = make_uint4(w[0][0], w[0][1], w[1][0], w[1][1]);
                XP[st].p[1] = make_uint4(w[2][0], w[2][1], w[0][0], w[0][1]);
                XP[st].p[2] = make_uint4(w[1][0], w[1][1], w[0][0], w[0][1]);
                DYP[st].p[0] = make_uint4(g[0][0], g[0][1], g[1][0], g[1][1]);
                DYP[st].p[1] = make_uint4(g[2][0], g[2][1], g[0][0], g[0][1]);
                DYP[st].p[2] = make_uint4(g[1][0], g[1][1], g[0][0], g[0][1]);
            }
        }
        constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};   // bf16x3 term order: small terms first, a0 b0 last
        // X3: the weight-gradient products contract over the tile's 32 samples = ONE k-step of v_mfma_f32_16x16x32_bf16.  Their narrow
        // partners -- dY^T (A of dW2) and X^T (B of dW1), lane = unit, k-group q = samples 4 q + r and 16 + 4 q + r -- are split once
        // per tile; H^T / dH^T (the other operands) are split per chunk from the registers they are in, in the same k-slot order.
        bf16x3::Pieces DYTP[NB], XTP[NB];
        if constexpr (X3) {
            wave_lds_fence();
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float4 d0 = ld4(TDY + rr + 16 * b * LT), d1 = ld4(TDY + rr + 16 * b * LT + 16);
                const float4 x0 = ld4(TX + rr + 16 * b * LT), x1 = ld4(TX + rr + 16 * b * LT + 16);
                const float vd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                const float vx[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                DYTP[b] = bf16x3::split8(vd);
                XTP[b] = bf16x3::split8(vx);
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int s0 = 0; s0 < 2; s0 += NST) {   // sample tiles s0 .. s0 + NST - 1 of the chunk
                // (compiler only) tile rows are re-read per pass instead of being held in registers across passes, and a later
                // pass's operands are not hoisted into this one
                wave_lds_fence();
                __builtin_amdgcn_sched_barrier(0);
                // H^T (R-layout: lane = hidden unit 16 jb + l15 of the chunk, register r = sample 16 st + 4 q + r)
                f32x4 H[2][NST], dH[2][NST];
                if constexpr (KS) {
                    bf16x3::Pieces Wp[2];
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) Wp[jb].p[i] = sm.W1p[(i * HS + c * 32 + 16 * jb + l15) * 4 + q];
                        const float bj = sm.b1s[c * 32 + 16 * jb + l15];
#pragma unroll
                        for (int k = 0; k < NST; ++k) H[jb][k] = f32x4{bj, bj, bj, bj};
                    }
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int k = 0; k < NST; ++k) {
                            H[0][k] = mfma16bf(XP[s0 + k].p[t], Wp[0].p[t], H[0][k]);
                            H[1][k] = mfma16bf(XP[s0 + k].p[t], Wp[1].p[t], H[1][k]);
                        }
                } else if constexpr (X3) {
                    bf16x3::Pieces Wp[2];
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                        for (int i = 0; i < 3; ++i) Wp[jb].p[i] = sm.W1p[(i * HS + c * 32 + 16 * jb + l15) * 4 + q];
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                        for (int k = 0; k < NST; ++k) H[jb][k] = zero4();
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int k = 0; k < NST; ++k) {
                            H[0][k] = mfma16bf(XP[s0 + k].p[TA[t]], Wp[0].p[TB[t]], H[0][k]);
                            H[1][k] = mfma16bf(XP[s0 + k].p[TA[t]], Wp[1].p[TB[t]], H[1][k]);
                        }
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        const float bj = sm.b1s[c * 32 + 16 * jb + l15];
#pragma unroll
                        for (int k = 0; k < NST; ++k)
#pragma unroll
                            for (int r = 0; r < 4; ++r) H[jb][k][r] += bj;
                    }
                } else {
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {   // (the bias 8-fold in LDS, one ds_read_b128 per tile instead of 4 v_mov: measured no gain)
                    const float bj = sm.b1s[c * 32 + 16 * jb + l15];
#pragma unroll
                    for (int k = 0; k < NST; ++k) H[jb][k] = f32x4{bj, bj, bj, bj};
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const f32x4 w0 = v4(ld4(sm.W1s + (c * 32 + l15) * S1 + 16 * b + 4 * q));
                    const f32x4 w1 = v4(ld4(sm.W1s + (c * 32 + 16 + l15) * S1 + 16 * b + 4 * q));
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int k = 0; k < NST; ++k) {
                            H[0][k] = mfma16(X[b][s0 + k][r], w0[r], H[0][k]);
                            H[1][k] = mfma16(X[b][s0 + k][r], w1[r], H[1][k]);
                        }
                }
                }
                // dH^T = (dY^T W2[:, chunk]) . leaky'
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int k = 0; k < NST; ++k) dH[jb][k] = zero4();
                if constexpr (KS) {
                    bf16x3::Pieces Wp[2];
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                        for (int i = 0; i < 3; ++i) Wp[jb].p[i] = sm.W2Tp[(i * HS + c * 32 + 16 * jb + l15) * 4 + q];
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int k = 0; k < NST; ++k) {
                            dH[0][k] = mfma16bf(DYP[s0 + k].p[t], Wp[0].p[t], dH[0][k]);
                            dH[1][k] = mfma16bf(DYP[s0 + k].p[t], Wp[1].p[t], dH[1][k]);
                        }
                } else if constexpr (X3) {
                    bf16x3::Pieces Wp[2];
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                        for (int i = 0; i < 3; ++i) Wp[jb].p[i] = sm.W2Tp[(i * HS + c * 32 + 16 * jb + l15) * 4 + q];
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int k = 0; k < NST; ++k) {
                            dH[0][k] = mfma16bf(DYP[s0 + k].p[TA[t]], Wp[0].p[TB[t]], dH[0][k]);
                            dH[1][k] = mfma16bf(DYP[s0 + k].p[TA[t]], Wp[1].p[TB[t]], dH[1][k]);
                        }
                } else {
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) {
                    const f32x4 w0 = v4(ld4(sm.W2Ts + (c * 32 + l15) * S1 + 16 * ob + 4 * q));
                    const f32x4 w1 = v4(ld4(sm.W2Ts + (c * 32 + 16 + l15) * S1 + 16 * ob + 4 * q));
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int k = 0; k < NST; ++k) {
                            dH[0][k] = mfma16(DY[ob][s0 + k][r], w0[r], dH[0][k]);
                            dH[1][k] = mfma16(DY[ob][s0 + k][r], w1[r], dH[1][k]);
                        }
                }
                }
                if (PREFETCH && c == NCH - 1 && s0 + NST == 2) request(raw, tile + stride);   // X / DY are dead: the next tile streams in
                float db[2] = {0.f, 0.f};
                __builtin_amdgcn_sched_barrier(0);   // one element-wise block per pass (see hidden_chunk)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int k = 0; k < NST; ++k)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // leaky(x) = x f, leaky'(x) = f, f = 1 (x > 0) or 0.2: compare + select + two multiplies
                            const float f = H[jb][k][r] > 0.f ? 1.0f : 0.2f;
                            H[jb][k][r] *= f;
                            dH[jb][k][r] *= f;
                            db[jb] += dH[jb][k][r];
                        }
                adb1[c][0] += db[0];
                adb1[c][1] += db[1];
                __builtin_amdgcn_sched_barrier(0);
                if (NEED_DX) {   // dH rows to the tile (read back as columns below, behind the dW products)
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                        for (int k = 0; k < NST; ++k)
                            *reinterpret_cast<float4*>(TH + rr + 16 * jb * LT + 16 * (s0 + k)) =
                                make_float4(dH[jb][k][0], dH[jb][k][1], dH[jb][k][2], dH[jb][k][3]);
                    wave_lds_fence();
                }
                if constexpr (X3 && NST == 2) {
                    // dW2 / dW1 on the bf16 MFMA: the chunk's H^T and dH^T registers (samples 4 q + r of both sample tiles = the eight
                    // k-slots of k-group q) split here, four 16 x 16 outputs each, six piece products per output
                    bf16x3::Pieces HP[2], DHP[2];
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        const float vh[8] = {H[jb][0][0], H[jb][0][1], H[jb][0][2], H[jb][0][3], H[jb][1][0], H[jb][1][1], H[jb][1][2], H[jb][1][3]};
                        const float vg[8] = {dH[jb][0][0], dH[jb][0][1], dH[jb][0][2], dH[jb][0][3], dH[jb][1][0], dH[jb][1][1], dH[jb][1][2], dH[jb][1][3]};
                        HP[jb] = bf16x3::split8(vh);
                        DHP[jb] = bf16x3::split8(vg);
                    }
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int ob = 0; ob < NB; ++ob) {
                            aW2[c][ob][0] = mfma16bf(DYTP[ob].p[TA[t]], HP[0].p[TB[t]], aW2[c][ob][0]);
                            aW2[c][ob][1] = mfma16bf(DYTP[ob].p[TA[t]], HP[1].p[TB[t]], aW2[c][ob][1]);
                        }
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int ib = 0; ib < NB; ++ib) {
                            aW1[c][0][ib] = mfma16bf(DHP[0].p[TA[t]], XTP[ib].p[TB[t]], aW1[c][0][ib]);
                            aW1[c][1][ib] = mfma16bf(DHP[1].p[TA[t]], XTP[ib].p[TB[t]], aW1[c][1][ib]);
                        }
                } else {
                // dW2[o][j] += sum_s dY[o][s] H[j][s]: A = dY^T from the tile, B = the H^T registers
#pragma unroll
                for (int ob = 0; ob < NB; ++ob)
#pragma unroll
                    for (int k = 0; k < NST; ++k) {
                        const f32x4 d = v4(ld4(TDY + rr + 16 * ob * LT + 16 * (s0 + k)));
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            aW2[c][ob][0] = mfma16(d[r], H[0][k][r], aW2[c][ob][0]);
                            aW2[c][ob][1] = mfma16(d[r], H[1][k][r], aW2[c][ob][1]);
                        }
                    }
                // dW1[j][i] += sum_s dH[j][s] X[i][s]: A = the dH^T registers, B = X^T from the tile
#pragma unroll
                for (int ib = 0; ib < NB; ++ib)
#pragma unroll
                    for (int k = 0; k < NST; ++k) {
                        const f32x4 x = v4(ld4(TX + rr + 16 * ib * LT + 16 * (s0 + k)));
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            aW1[c][0][ib] = mfma16(dH[0][k][r], x[r], aW1[c][0][ib]);
                            aW1[c][1][ib] = mfma16(dH[1][k][r], x[r], aW1[c][1][ib]);
                        }
                    }
                }
                if (NEED_DX) {   // Q[i][s] += sum_j W1[j][16 + i] dH[j][s]: A = a column of W1, B = a column of the dH tile, per step
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float a = sm.W1s[(c * 32 + 16 * jb + 4 * q + r) * S1 + 16 + l15];
#pragma unroll
                            for (int k = 0; k < NST; ++k)
                                dXa[s0 + k] = mfma16(a, TH[(16 * jb + 4 * q + r) * LT + 16 * (s0 + k) + l15], dXa[s0 + k]);
                        }
                }
            }
        }
        if (NEED_DX) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const long long s = tile * 32 + 16 * st + l15;
                if (s < n) *reinterpret_cast<float4*>(qo + s * 16 + 4 * q) = make_float4(dXa[st][0], dXa[st][1], dXa[st][2], dXa[st][3]);
            }
        }
        wave_lds_fence();   // the next tile's TX / TDY stores stay behind this tile's reads
        if (!PREFETCH) request(raw, tile + stride);
    }

    // one partial-gradient row per WAVE (no cross-wave reduction here; resmlp_reduce sums the rows in a fixed order)
    float* __restrict__ row = wpart + ((size_t)(wg.net_i * (groups * kWaves)) + (size_t)wg.grp * kWaves + wave) * PSTRIDE;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int j0 = wg.sl * HS + c * 32 + 16 * jb;
#pragma unroll
            for (int b = 0; b < NB; ++b)
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
    if (!NEED_DX && wg.sl == 0) {   // db2a: lane (sample, q) holds rows 4 q + r summed over its samples: add the 16 sample lanes
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = adb2[r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);
            if (l15 == 0) row[rp::B2A + 4 * q + r] = v;
        }
    }
}

#include "ppo_resmlp512_bwd2s.h"   // round 6: resmlp_bwd<32, 2, 4> as one hand-placed, chunk-pipelined instruction stream

// ---------------------------------------------------------------- streaming kernels (element-wise + reductions over samples)
// block-wide sum of per-thread accumulators over the threads that share (tid & (LPS - 1)); result in row[col0 + 4 * og + k]
template <int LPS, int NV>
__device__ __forceinline__ void block_sum_to_row(float (&acc)[NV], float* red /* [kEThreads / 64][LPS][NV] */, float* __restrict__ row,
                                                 const int* cols /* NV column bases; value v of group og -> cols[v] + og * stride[v] */,
                                                 const int* strides, const bool* og0_only) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        float x = acc[v];
#pragma unroll
        for (int o = LPS; o < 64; o <<= 1) x += __shfl_xor(x, o, 64);
        acc[v] = x;
    }
    if (lane < LPS)
#pragma unroll
        for (int v = 0; v < NV; ++v) red[(wave * LPS + lane) * NV + v] = acc[v];
    __syncthreads();
    for (int k = threadIdx.x; k < LPS * NV; k += kEThreads) {
        const int og = k / NV, v = k % NV;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kEThreads / 64; ++w) s += red[(w * LPS + og) * NV + v];
        if (!og0_only[v] || og == 0) row[cols[v] + og * strides[v]] = s;
    }
}

// E2: h2, heads, PPO loss / MSE, dpre2, and the sample sums d(heads), db2b, statistics.  thread = (sample, 4 of the 32 units)
// HEAD_ONLY: V = critic(obs).squeeze() only (ppo.py:275, :724).
template <bool HEAD_ONLY>
__global__ __launch_bounds__(kEThreads) void resmlp_e2(const float* __restrict__ params, int net_base, const void* __restrict__ obs,
                                                       const float* __restrict__ h1buf, const float* __restrict__ p2,
                                                       const float* __restrict__ act, const float* __restrict__ logp_old,
                                                       const float* __restrict__ rtg, const float* __restrict__ adv, long long n,
                                                       float var, float clip, float inv_n, float* __restrict__ dy2,
                                                       float* __restrict__ epart, float* __restrict__ v_out, int obs_f16) {
    __shared__ float red[(kEThreads / 64) * 8 * 17];
    const int net_i = blockIdx.y, net = net_base + net_i;
    const bool actor = net == 0;
    const float* __restrict__ pn = params + (net ? rp::P_ACTOR : 0);
    const int og = threadIdx.x & 7;
    float w1[4], w2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        w1[k] = pn[rp::WO1 + 4 * og + k];
        w2[k] = actor ? pn[rp::WO2 + 4 * og + k] : 0.f;
    }
    const float bo1 = pn[rp::BO1], bo2 = actor ? pn[rp::BO2] : 0.f;
    const float* __restrict__ pp = p2 + (size_t)net_i * NSLF * n * 32;   // (obs / h1buf: no longer read here -- pair 0's partial carries them)
    float* __restrict__ dyo = HEAD_ONLY ? nullptr : dy2 + (size_t)net_i * n * 32;
    // accumulators: db2b[4] dwo1[4] dwo2[4] dbo1 dbo2 st0 st1 st2   (17 values)
    float acc[17];
#pragma unroll
    for (int v = 0; v < 17; ++v) acc[v] = 0.f;
    const long long total = n * 8, step = (long long)gridDim.x * kEThreads;
    for (long long g = (long long)blockIdx.x * kEThreads + threadIdx.x; g < total; g += step) {
        const long long s = g >> 3, o = g * 4;   // o == s * 32 + 4 * og
        // the two slice pairs' partials; pair 0's carries the residual input [x, h1] and the bias b2b (resmlp_fwd<32>)
        const float4 a0 = ld4(pp + o), a1 = ld4(pp + (size_t)n * 32 + o);
        static_assert(NSLF == 2, "two forward partials per sample and net");
        float h[4];
        h[0] = leaky(a0.x + a1.x);
        h[1] = leaky(a0.y + a1.y);
        h[2] = leaky(a0.z + a1.z);
        h[3] = leaky(a0.w + a1.w);
        float z3 = 0.f, z4 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            z3 = fmaf(h[k], w1[k], z3);
            z4 = fmaf(h[k], w2[k], z4);
        }
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
            z3 += __shfl_xor(z3, m, 64);
            z4 += __shfl_xor(z4, m, 64);
        }
        z3 += bo1;
        z4 += bo2;
        if (HEAD_ONLY) {
            if (og == 0) v_out[s] = z3;
            continue;
        }
        const float own = og == 0 ? 1.f : 0.f;   // statistics are counted once per sample
        float g3, g4 = 0.f;
        if (actor) {
            const float2 a = reinterpret_cast<const float2*>(act)[s];
            const float mu0 = 1.0f / (1.0f + expf(-z3));   // F.sigmoid, net_actor.py:141
            const float mu1 = tanhf(z4);                    // F.tanh, net_actor.py:142
            const float d0 = a.x - mu0, d1 = a.y - mu1;
            // MultivariateNormal(mean, var * I).log_prob, ppo.py:734-735
            const float lp = -0.5f * ((d0 * d0 + d1 * d1) / var) - 1.8378770664093453f - logf(var);
            const float lr = lp - logp_old[s];
            const float ratio = expf(lr);                  // ppo.py:316
            const float A = adv[s];
            const float s1 = ratio * A;                     // ppo.py:319
            const float rc = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
            const float s2 = rc * A;                        // ppo.py:320
            acc[14] += own * -fminf(s1, s2);                // ppo.py:342 (mean taken by inv_n at the end)
            acc[15] += own * ((ratio - 1.0f) - lr);         // approx KL, ppo.py:326
            acc[16] += (fabsf(ratio - 1.0f) > clip) ? own : 0.f;   // clip fraction, ppo.py:335
            const bool inside = (ratio >= 1.0f - clip) && (ratio <= 1.0f + clip);
            const float dL_dratio = (inside || s1 < s2) ? -A : 0.f;
            const float dL_dlp = dL_dratio * ratio * inv_n;
            g3 = dL_dlp * (d0 / var) * (mu0 * (1.0f - mu0));
            g4 = dL_dlp * (d1 / var) * (1.0f - mu1 * mu1);
        } else {
            const float e = z3 - rtg[s];                    // critic(obs).squeeze(), ppo.py:724
            acc[14] += own * (e * e);                       // MSELoss, ppo.py:343
            g3 = 2.0f * e * inv_n;
        }
        float d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            d[k] = dleaky(fmaf(g3, w1[k], g4 * w2[k]), h[k]);
            acc[k] += d[k];
            acc[4 + k] = fmaf(g3, h[k], acc[4 + k]);
            acc[8 + k] = fmaf(g4, h[k], acc[8 + k]);
        }
        acc[12] += own * g3;
        acc[13] += own * g4;
        *reinterpret_cast<float4*>(dyo + o) = make_float4(d[0], d[1], d[2], d[3]);
    }
    if (HEAD_ONLY) return;
    // value v of unit group og -> column cols[v] + og * strides[v]
    __shared__ int cols[17], strides[17];
    __shared__ bool og0[17];
    if (threadIdx.x < 17) {
        const int v = threadIdx.x;
        const int c = v < 4 ? EC_B2B + v : v < 8 ? EC_WO1 + (v - 4) : v < 12 ? EC_WO2 + (v - 8)
                      : v == 12 ? EC_BO1 : v == 13 ? EC_BO2 : EC_STAT + (v - 14);
        cols[v] = c;
        strides[v] = v < 12 ? 4 : 0;
        og0[v] = v >= 12;
    }
    __syncthreads();
    block_sum_to_row<8, 17>(acc, red, epart + ((size_t)net_i * gridDim.x + blockIdx.x) * EP, cols, strides, og0);
}

// ---------------------------------------------------------------- partial rows -> gradient (-> Adam)
// grad[q] = sum of the partial rows that hold parameter q, rows in a fixed order (deterministic); ADAM: torch.optim.Adam's
// step in place (ppo.py:116-117,381,392: betas (0.9, 0.999), eps 1e-8, no weight decay).
constexpr int kRedGroups = 16;
template <bool ADAM>
__global__ __launch_bounds__(64 * kRedGroups) void resmlp_reduce(const float* __restrict__ wpart, int w_rows1, int w_rows2, const float* __restrict__ epart,
                                                                  int e_rows, float inv_n, float* __restrict__ grad, float* __restrict__ stats,
                                                                  float* __restrict__ params, float* __restrict__ m, float* __restrict__ v,
                                                                  float lr, float beta1, float beta2, float eps, float bc1, float bc2_sqrt,
                                                                  float* __restrict__ gnbase, int parity) {
    // gnbase / parity: the per-net squared gradient norms of the epoch BEFORE into stats[3] / stats[7] (reduce_adam of ppo_mlp64.hip has
    // the scheme); the slots are columns 0 .. 7 of the streaming kernels' partial rows (those kernels write columns >= EC_B2B)
    auto gn_slot = [&](const int par, const int net, const int b) -> float* {
        const int i = (par * 2 + net) * kGnSlotsR + b;
        return gnbase + (size_t)(i >> 3) * EP + (i & 7);
    };
    __shared__ float part[kRedGroups][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6, qi = blockIdx.x * 64 + lane;   // qi: index into actor | critic
    const bool valid = qi < rp::P_ACTOR + rp::P_CRITIC;
    const int net_i = qi < rp::P_ACTOR ? 0 : 1, p = qi - net_i * rp::P_ACTOR;
    // where parameter p lives: biases of the block outputs and the heads come from the streaming kernels' rows
    const bool from_e = p >= rp::B2B;                 // b2b and the heads: summed over samples by the streaming kernel E2
    const int ecol = EC_B2B + (p - rp::B2B);          // b2b | wo1 | bo1 | wo2 | bo2, same order as the parameters
    const int w_rows = p < rp::W1B ? w_rows1 : w_rows2;   // rb1's and rb2's backward kernels run different numbers of waves
    const float* __restrict__ src = from_e ? epart + (size_t)net_i * e_rows * EP + ecol : wpart + (size_t)net_i * w_rows * PSTRIDE + p;
    const size_t pitch = from_e ? EP : PSTRIDE;
    const int rows = from_e ? e_rows : w_rows;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (valid) {
        int b = g;
        for (; b + 3 * kRedGroups < rows; b += 4 * kRedGroups) {
            s0 += src[(size_t)b * pitch];
            s1 += src[(size_t)(b + kRedGroups) * pitch];
            s2 += src[(size_t)(b + 2 * kRedGroups) * pitch];
            s3 += src[(size_t)(b + 3 * kRedGroups) * pitch];
        }
        for (; b < rows; b += kRedGroups) s0 += src[(size_t)b * pitch];
    }
    part[g][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && valid) {
        float gr = 0.f;
#pragma unroll
        for (int k = 0; k < kRedGroups; ++k) gr += part[k][lane];
        grad[qi] = gr;
        if (ADAM) {
            const float mm = m[qi] + (gr - m[qi]) * (1.0f - beta1);          // exp_avg.lerp_(grad, 1 - beta1)
            const float vv = beta2 * v[qi] + (1.0f - beta2) * (gr * gr);     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
            m[qi] = mm;
            v[qi] = vv;
            params[qi] -= (lr / bc1) * (mm / (sqrtf(vv) / bc2_sqrt + eps));
        }
    }
    if (parity >= 0 && g == 0) {
        const float gr2 = valid ? grad[qi] * grad[qi] : 0.f;   // (this lane's own store above)
        float sa = net_i == 0 ? gr2 : 0.f, sc = net_i == 0 ? 0.f : gr2;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            sa += __shfl_xor(sa, o, 64);
            sc += __shfl_xor(sc, o, 64);
        }
        if (lane == 0) {
            *gn_slot(parity, 0, blockIdx.x) = sa;
            *gn_slot(parity, 1, blockIdx.x) = sc;
        }
    }
    if (parity >= 0 && blockIdx.x == 0 && (g == 3 || g == 7)) {
        float s = 0.f;
        for (int b = lane; b < (int)gridDim.x; b += 64) s += *gn_slot(1 - parity, g >> 2, b);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) stats[g] = s;
    }
    if (blockIdx.x == 0 && g < 8 && (g & 3) < 3) {   // stats[0..2] actor (loss, approx KL, clip fraction), stats[4] critic loss
        const float* sp = epart + (size_t)(g >> 2) * e_rows * EP + EC_STAT + (g & 3);
        float s = 0.f;
        for (int b = lane; b < e_rows; b += 64) s += sp[(size_t)b * EP];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) stats[g] = s * inv_n;
    }
}

// ---------------------------------------------------------------- rollout-time policy step (PPO.get_action, ppo.py:673-706)
// One workgroup = 16 envs on 8 waves: csrc/resmlp_policy.h (shared with the persistent rollout kernel of navsim.hip: same bits).
constexpr int kActEnvs = resmlp::kPolEnvs;
static_assert(kWaves == resmlp::kPolWaves, "the policy step's workgroup");
__global__ __launch_bounds__(kThreads) void resmlp_act(const float* __restrict__ pa, const void* __restrict__ obs,
                                                       const float* __restrict__ noise, long long n, const float* __restrict__ var_ptr,
                                                       uint64_t seed, uint64_t env_id_base, const uint32_t* __restrict__ step_base,
                                                       uint32_t step_offset, float* __restrict__ act, float* __restrict__ logp,
                                                       float* __restrict__ mean_out, int obs_f16) {
    __shared__ __attribute__((aligned(16))) resmlp::PolicySmem ps;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, q = lane >> 4;
    const long long e = (long long)blockIdx.x * kActEnvs + l15;
    const bool valid = e < n;
    const f32x4 xq = valid ? v4(ldx4(obs, e * 16 + 4 * q, obs_f16)) : zero4();
    float z3, z4;
    resmlp::policy_preact(pa, xq, lane, w, ps, z3, z4);
    if (w != 0) return;
    if (q == 0 && valid) {
        float e0, e1;
        if (noise) {
            e0 = noise[2 * e];
            e1 = noise[2 * e + 1];
        } else {
            mlp64::policy_noise((step_base ? *step_base : 0u) + step_offset, seed, env_id_base + (uint64_t)e, e0, e1);
        }
        const resmlp::Action o = resmlp::policy_finish(pa + rp::B2B, z3, z4, *var_ptr, e0, e1);
        act[2 * e] = o.a0;
        act[2 * e + 1] = o.a1;
        logp[e] = o.logp;
        if (mean_out) {
            mean_out[2 * e] = o.mu0;
            mean_out[2 * e + 1] = o.mu1;
        }
    }
}

// ---------------------------------------------------------------- host side
thread_local std::string g_err;

struct Plan {
    int groups, wgs, e_blocks;
    float *p1, *h1, *p2, *dy2, *qb, *wpart, *epart;
};

size_t ws_floats(int64_t n) {
    const size_t N = (size_t)n;
    return 2 * N * (NSL * 16 + 16 + NSL * 32 + 32 + NSL * 16) + (size_t)kWRows * PSTRIDE + (size_t)2 * kEMaxBlocks * EP + 64;
}

Plan make_plan(void* ws, int64_t n, int n_nets) {
    Plan p;
    const long long tiles = (n + 31) / 32;
    long long g = (tiles + kWaves - 1) / kWaves;
    const int gmax = kMaxWG / (NSL * n_nets);
    if (g > gmax) g = gmax;
    if (g >= 8) g &= ~7LL;
    p.groups = (int)g;
    p.wgs = p.groups * NSL * n_nets;
    long long eb = (n + 31) / 32;
    p.e_blocks = (int)(eb < 1 ? 1 : eb > kEMaxBlocks ? kEMaxBlocks : eb);
    float* f = reinterpret_cast<float*>(((uintptr_t)ws + 15) & ~(uintptr_t)15);
    const size_t N = (size_t)n;
    p.p1 = f; f += 2 * NSL * N * 16;
    p.h1 = f; f += 2 * N * 16;
    p.p2 = f; f += 2 * NSL * N * 32;
    p.dy2 = f; f += 2 * N * 32;
    p.qb = f; f += 2 * NSL * N * 16;
    p.wpart = f; f += (size_t)kWRows * PSTRIDE;
    p.epart = f;
    return p;
}

bool launch_ok(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_err = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    }
    return true;
}

// forward of `n_nets` nets starting at net_base (0 = actor, 1 = critic) up to the partial sums of rb2
void launch_forward(const Plan& p, const float* params, int net_base, int n_nets, const void* obs, int obs_f16, int64_t n, hipStream_t st) {
    // (a forward workgroup = a PAIR of slices: FSP times the groups on the same number of workgroups)
    hipLaunchKernelGGL(resmlp_fwd<16>, dim3(p.wgs), dim3(kThreads), 0, st, params, net_base, n_nets, obs, (const float*)nullptr,
                       (float*)nullptr, (long long)n, p.groups * FSP, p.p1, obs_f16);
    hipLaunchKernelGGL(resmlp_fwd<32>, dim3(p.wgs), dim3(kThreads), 0, st, params, net_base, n_nets, obs, (const float*)p.p1, p.h1,
                       (long long)n, p.groups * FSP, p.p2, obs_f16);
}

int loss_grad_impl(const char* name, bool adam, float* params, const void* obs, int32_t obs_f16, const float* act, const float* logp_old,
                   const float* rtg, const float* adv, int64_t n, float var, float clip, float lr, float beta1, float beta2,
                   float eps, int32_t step, float* adam_m, float* adam_v, float* grad, float* stats, void* ws, void* stream) {
    if (!params || !obs || !act || !logp_old || !rtg || !adv || !grad || !stats || !ws || n < 1 || !(var > 0.f) ||
        (adam && (!adam_m || !adam_v || step < 1))) {
        g_err = std::string(name) + ": bad argument";
        return -1;
    }
    if (((uintptr_t)obs & 15) || ((uintptr_t)act & 7)) {
        g_err = std::string(name) + ": obs must be 16-byte and act 8-byte aligned";
        return -1;
    }
    hipStream_t st = (hipStream_t)stream;
    const Plan p = make_plan(ws, n, 2);
    const float inv_n = 1.0f / (float)n;
    const int f16 = obs_f16 != 0;
    launch_forward(p, params, 0, 2, obs, f16, n, st);
    hipLaunchKernelGGL(resmlp_e2<false>, dim3(p.e_blocks, 2), dim3(kEThreads), 0, st, (const float*)params, 0, obs, (const float*)p.h1,
                       (const float*)p.p2, act, logp_old, rtg, adv, (long long)n, var, clip, inv_n, p.dy2, p.epart, (float*)nullptr, f16);
    if constexpr (RESMLP_BWD2S && kBwd2Waves == b2s::SW) {
        if (f16)
            hipLaunchKernelGGL(resmlp_bwd2s<true>, dim3(p.wgs), dim3(64 * b2s::SW), 0, st, (const float*)params, 2, obs, (const float*)p.h1,
                               (const float*)p.dy2, (long long)n, p.groups, p.wpart, p.qb);
        else
            hipLaunchKernelGGL(resmlp_bwd2s<false>, dim3(p.wgs), dim3(64 * b2s::SW), 0, st, (const float*)params, 2, obs, (const float*)p.h1,
                               (const float*)p.dy2, (long long)n, p.groups, p.wpart, p.qb);
    } else
        hipLaunchKernelGGL((resmlp_bwd<32, 2, kBwd2Waves>), dim3(p.wgs), dim3(64 * kBwd2Waves), 0, st, (const float*)params, 2, obs, (const float*)p.h1,
                           (const float*)p.dy2, (long long)n, p.groups, p.wpart, p.qb, (const float*)nullptr, f16);
    hipLaunchKernelGGL((resmlp_bwd<16, 2, kBwd1Waves>), dim3(p.wgs), dim3(64 * kBwd1Waves), 0, st, (const float*)params, 2, obs, (const float*)p.h1,
                       (const float*)p.dy2, (long long)n, p.groups, p.wpart, (float*)nullptr, (const float*)p.qb, f16);
    const int rblocks = (rp::P_ACTOR + rp::P_CRITIC + 63) / 64;
    if (adam) {
        const float bc1 = (float)(1.0 - std::pow((double)beta1, (double)step));
        const float bc2_sqrt = (float)std::sqrt(1.0 - std::pow((double)beta2, (double)step));
        hipLaunchKernelGGL(resmlp_reduce<true>, dim3(rblocks), dim3(64 * kRedGroups), 0, st, (const float*)p.wpart, p.groups * kBwd1Waves,
                           p.groups * kBwd2Waves, (const float*)p.epart, p.e_blocks, inv_n, grad, stats, params, adam_m, adam_v, lr, beta1, beta2, eps, bc1, bc2_sqrt,
                           p.epart, (int)(step & 1));
    } else {
        hipLaunchKernelGGL(resmlp_reduce<false>, dim3(rblocks), dim3(64 * kRedGroups), 0, st, (const float*)p.wpart, p.groups * kBwd1Waves,
                           p.groups * kBwd2Waves, (const float*)p.epart, p.e_blocks, inv_n, grad, stats, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f,
                           0.f, 0.f, 0.f, 1.f, 1.f, (float*)nullptr, -1);
    }
    return launch_ok(name) ? 0 : -2;
}

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

size_t navppo_resmlp512_workspace_bytes(int64_t n_samples) { return n_samples < 1 ? 0 : ws_floats(n_samples) * sizeof(float); }

int navppo_resmlp512_loss_grad(const float* params_dev, const void* obs_dev, int32_t obs_f16, const float* act_dev, const float* logp_old_dev,
                               const float* rtg_dev, const float* adv_dev, int64_t n_samples, float var, float clip,
                               float* grad_dev, float* stats_dev, void* workspace_dev, void* stream) {
    const int rc = loss_grad_impl("navppo_resmlp512_loss_grad", false, const_cast<float*>(params_dev), obs_dev, obs_f16, act_dev, logp_old_dev,
                                  rtg_dev, adv_dev, n_samples, var, clip, 0.f, 0.f, 0.f, 0.f, 1, nullptr, nullptr, grad_dev, stats_dev,
                                  workspace_dev, stream);
    if (rc != 0) navppo_set_error(g_err.c_str());
    return rc;
}

int navppo_resmlp512_update_epoch(float* params_dev, const void* obs_dev, int32_t obs_f16, const float* act_dev, const float* logp_old_dev,
                                  const float* rtg_dev, const float* adv_dev, int64_t n_samples, float var, float clip, float lr,
                                  float beta1, float beta2, float eps, int32_t step, float* adam_m_dev, float* adam_v_dev,
                                  float* grad_dev, float* stats_dev, void* workspace_dev, void* stream) {
    const int rc = loss_grad_impl("navppo_resmlp512_update_epoch", true, params_dev, obs_dev, obs_f16, act_dev, logp_old_dev, rtg_dev, adv_dev,
                                  n_samples, var, clip, lr, beta1, beta2, eps, step, adam_m_dev, adam_v_dev, grad_dev, stats_dev,
                                  workspace_dev, stream);
    if (rc != 0) navppo_set_error(g_err.c_str());
    return rc;
}

int navppo_resmlp512_value(const float* critic_params_dev, const void* obs_dev, int32_t obs_f16, int64_t n_samples, float* value_dev,
                           void* workspace_dev, void* stream) {
    if (!critic_params_dev || !obs_dev || !value_dev || !workspace_dev || n_samples < 1 || ((uintptr_t)obs_dev & 15)) {
        navppo_set_error("navppo_resmlp512_value: bad argument (obs must be 16-byte aligned)");
        return -1;
    }
    hipStream_t st = (hipStream_t)stream;
    const Plan p = make_plan(workspace_dev, n_samples, 1);
    // the kernels index the flat [actor | critic] buffer by net: hand them the address the actor would have
    const float* base = critic_params_dev - rp::P_ACTOR;
    launch_forward(p, base, 1, 1, obs_dev, obs_f16 != 0, n_samples, st);
    hipLaunchKernelGGL(resmlp_e2<true>, dim3(p.e_blocks, 1), dim3(kEThreads), 0, st, base, 1, obs_dev, (const float*)p.h1,
                       (const float*)p.p2, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (long long)n_samples, 1.f, 0.f, 0.f, (float*)nullptr, (float*)nullptr, value_dev, (int)(obs_f16 != 0));
    if (!launch_ok("navppo_resmlp512_value")) {
        navppo_set_error(g_err.c_str());
        return -2;
    }
    return 0;
}

int navppo_resmlp512_act(const float* actor_params_dev, const void* obs_dev, int32_t obs_f16, const float* noise_dev, int64_t n_envs,
                         const float* var_dev, uint64_t seed, uint64_t env_id_base, const uint32_t* step_base_dev,
                         uint32_t step_offset, float* act_dev, float* logp_dev, float* mean_dev, void* stream) {
    if (!actor_params_dev || !obs_dev || !act_dev || !logp_dev || n_envs < 1 || !var_dev) {
        navppo_set_error("navppo_resmlp512_act: bad argument");
        return -1;
    }
    if (((uintptr_t)actor_params_dev & 15) || ((uintptr_t)obs_dev & 15)) {
        navppo_set_error("navppo_resmlp512_act: params and obs must be 16-byte aligned");
        return -1;
    }
    const int blocks = (int)((n_envs + kActEnvs - 1) / kActEnvs);
    hipLaunchKernelGGL(resmlp_act, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, actor_params_dev, obs_dev, noise_dev,
                       (long long)n_envs, var_dev, seed, env_id_base, step_base_dev, step_offset, act_dev, logp_dev, mean_dev, (int)(obs_f16 != 0));
    if (!launch_ok("navppo_resmlp512_act")) {
        navppo_set_error(g_err.c_str());
        return -2;
    }
    return 0;
}

}  // extern "C"
#pragma GCC visibility pop
