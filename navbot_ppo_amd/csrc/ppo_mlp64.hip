// ppo_mlp64.hip -- fused PPO loss + gradient for the 16-64-64 actor / critic heads on gfx950 (f32 MFMA).
//
// Replaces, for one epoch of the update loop of the reference (project_ppo/src/ppo.py:305-397):
//   evaluate()            ppo.py:708-737   V = critic(obs), mean = actor(obs), log_prob(batch_acts)
//   ratios / surrogates   ppo.py:316-320
//   actor_loss, critic_loss and both backward() calls   ppo.py:342-343, 349, 386
// for the "mlp64x2" policy (BASELINE.json configs[1]; 16-64-64 MLP of NetActor_old / the tutorial network,
// project_ppo/src/net_actor.py:147-189, graph_code/ppo_for_beginners/network.py:11-50).
//
// Why a kernel: in PyTorch every layer's activations ([2.1 M, 64] f32 = 537 MB) make a round trip through HBM per op
// (profiles/r01_bench_v4_kernel_stats.csv: 3.8 ms per epoch, relu-backward alone 1 ms).  Here activations never leave
// the CU: per epoch the batch is read once per net (84 B/sample) and nothing but one partial-gradient row per
// workgroup is written.
//
// Arithmetic: float32 throughout.  GEMMs use v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32, which are exact
// k-ordered f32 fma chains (no reduced-precision inputs); sums over samples are taken in tile order, so results agree
// with PyTorch autograd to fp32 round-off, not bit for bit.
//
// One launch per net (template ACTOR), one persistent 8-wave workgroup per CU; a WAVE owns a 32-sample tile end to end
// (mlp64_pass_w below):
//   F1  H1 = relu(X W1^T + b1)      [32x16]x[16x64]
//   F2  H2 = relu(H1 W2^T + b2)     [32x64]x[64x64]
//   heads + PPO loss / MSE (VALU)   -> g3, g4 (dL/dz of the output units) ; dW3, dW4, db3, db4
//   dH2 = (g3 w3 + g4 w4) . [H2 > 0]                    (VALU, elementwise) ; db2
//   B2  dH1 = (dH2 W2) . [H1 > 0]   [32x64]x[64x64]     ; db1
//   G2  dW2 += dH2^T H1             [64x32]x[32x64]     accumulators persist over the wave's tiles
//   G1  dW1 += dH1^T X              [64x32]x[32x16]
// At the end every workgroup writes its partial gradient (one row of `partial`), and `reduce_adam` sums rows.
#include <hip/hip_runtime.h>

#include <hip/hip_fp16.h>

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <string>
#include <type_traits>

#include "navppo.h"
#include "navppo_internal.h"
#include "mlp64_policy.h"
#include "bf16x3.h"

namespace {

using namespace mlp64;
static_assert(P_ACTOR == NAVPPO_MLP64_ACTOR_PARAMS && P_CRITIC == NAVPPO_MLP64_CRITIC_PARAMS, "layout");
static_assert(Layout<42>::P_ACTOR == NAVPPO_MLP64_ACTOR_PARAMS_D(42) && Layout<42>::P_CRITIC == NAVPPO_MLP64_CRITIC_PARAMS_D(42) &&
              Layout<16>::P_ACTOR == NAVPPO_MLP64_ACTOR_PARAMS_D(16), "layout");

typedef float f32x16 __attribute__((ext_vector_type(16)));


__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// C/D element (row, col) of accumulator register r for lane l (32x32 shapes; cdna_hip_programming.md section 3)
__device__ __forceinline__ int c_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------- the pass kernel
// A WAVE owns a 32-sample tile end to end and never meets a workgroup barrier inside the tile loop (round-1 history:
// 4-wave workgroups that shared 64-sample tiles through LDS needed 7 barriers and ~370 ds_read_b32 per tile and ran
// 1.43 ms per epoch; this layout 1.13 ms).  Measured on gfx950 (tools/ubench/mfma_valu_overlap.hip): f32 MFMA and
// VALU / LDS work of two waves on one SIMD do NOT overlap -- the times add -- so the kernel is organised to minimise
// the number of non-MFMA instructions, not to hide them.  Everything is computed transposed (sample = lane):
//   F1  H1^T[n][m] = relu(b1 + W1 X^T)         A = W1 rows from LDS, B = the lane's own 8 observation floats (registers)
//   F2  H2^T       = relu(b2 + W2 H1^T)        B = the F1 accumulators themselves: accumulator register r of lane
//   B2  dH1^T      = (W2^T dH2^T) . [H1 > 0]       (m, hi) holds row (r&3) + 8 (r>>2) + 4 hi, which is exactly a legal
//                                                  k-pairing for the next MFMA when A is read k-permuted (ds_read_b128)
//   G2 / G1 (contraction over samples = lanes) are the only products that need a transpose: H1^T, dH2^T, dH1^T and X
//   pass through three 32x32 wave-private LDS tiles and come back as k-contiguous ds_read_b128 operands; the same
//   operand registers give db2 / db1, and one more tile round trip of H2^T gives dW3 / dW4.
// LDS: W1 5 KB + W2 and W2^T 17 KB each + vectors 1 KB + 8 x 13.75 KB wave tiles = 150 KB -> one workgroup per CU.
//
// Observation width (template IN) and row type (F16): IN = 16 is BASELINE configs[1] / [2] / [4] (10 beams), IN = 42 configs[3]
// (36 beams: 42-D rows, padded to 48 columns on chip -- three 16-column tiles of dW1, 24 k-steps of F1 per half; W1 takes 13 KB
// of LDS, 158 KB in all).  F16: the rows are float16 (configs[4], navsim_cfg.obs_f16) and are widened to float32 as they are
// loaded; everything behind the load is the same float32 arithmetic.  The 16-wide float32 instantiation is the round-2 kernel
// unchanged (next tile's rows prefetched into registers, X kept in registers for G1); the 42-wide ones load a tile's rows at its
// start and read X again -- from L2, directly in G1's operand layout -- when G1 needs it: 48 registers of dW1 accumulators
// leave no room for 2 x 24 registers of rows held across the tile.
template <int IN>
struct Pad {
    static constexpr int INP = (IN == 16) ? 16 : 48;   // columns on chip
    static constexpr int KH = INP / 2;                 // features per lane half in F1 (32x32x2: k = lane >> 5)
    static constexpr int NC = INP / 16;                // 16-column tiles of dW1
    static constexpr int LW1 = INP + 4;                // row stride of W1 in LDS (floats): 16-byte aligned rows, conflict-free ds_read_b128
#ifndef MLP64_WIDE_WAVES
#define MLP64_WIDE_WAVES 4
#endif
    // waves per workgroup (one workgroup per CU).  16 columns: 8 = two per SIMD, 256 registers each.  42 columns: 48 more registers of
    // dW1 accumulators and 24 + 24 of rows do not fit 256 (the 8-wave build spilled ~450 registers and re-read X from L2 for G1:
    // 2242 us per epoch = 0.43 of the f32 MFMA peak) -- 4 = one per SIMD, 512 registers each, the 16-column structure unchanged
    static constexpr int NW = (IN == 16) ? 8 : MLP64_WIDE_WAVES;
    static_assert(IN == 16 || IN == 42, "observation widths of the reference's sensor configurations: 10 or 36 beams + 6");
};
constexpr int LW2 = 68;
constexpr int LT = 36;
constexpr int TILE_F = 32 * LT;
constexpr int WAVE_F = 3 * TILE_F + 64;   // T0 | T1 | TD | g3[32] g4[32]
constexpr int kWWaves = 8;

constexpr int kWMaxBlocks = 256;          // one persistent workgroup per CU

template <int IN>
struct SmemW {
    float W1s[H * Pad<IN>::LW1];
    float W2s[H * LW2];
    float W2Ts[H * LW2];
    float b1[H], b2[H], w3[H], w4[H];
    // the wave tiles; at the end of a pass also the four rows of the workgroup's gradient reduction (42 columns: from W2s on)
    static constexpr int kRedF = 4 * ((Layout<IN>::P_ACTOR + 6) & ~3) - ((IN == 16) ? 0 : 2 * H * LW2 + 4 * H);
    float wv[(Pad<IN>::NW * WAVE_F > kRedF) ? Pad<IN>::NW * WAVE_F : kRedF];
};
static_assert(sizeof(SmemW<16>) <= 160 * 1024 && sizeof(SmemW<42>) <= 160 * 1024, "LDS");
static_assert(2 * kWMaxBlocks <= NAVPPO_MLP64_MAX_BLOCKS, "workspace rows for both nets");
static_assert(offsetof(SmemW<42>, wv) == offsetof(SmemW<42>, W2s) + sizeof(float) * (2 * H * LW2 + 4 * H), "reduction rows span W2s .. wv");

__device__ __forceinline__ float4 lds4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wave are performed in issue order: this only stops the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The KH observation entries X[m][KH lhi .. KH lhi + KH - 1] of one sample as the lane (m, lhi) of a tile holds them: requested
// into raw dwords (float32, or packed pairs of float16), widened to float32 at the use.  Entries past the row (IN = 42: columns
// 42 .. 47) are zero.  Rows are 16-byte (IN = 16) / 8-byte (IN = 42 float32) / 4-byte (IN = 42 float16) aligned.
template <int IN, bool F16>
struct XRow {
    static constexpr int KH = Pad<IN>::KH;
    static constexpr int NRAW = F16 ? KH / 2 : KH;
    uint32_t raw[NRAW];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int j = 0; j < NRAW; ++j) raw[j] = 0u;
    }
    __device__ __forceinline__ void request(const void* __restrict__ obs, const long long m, const int lhi) {
        if constexpr (IN == 16 && !F16) {
            const uint4* xp = reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(obs) + m * IN + 8 * lhi);
            const uint4 q0 = xp[0], q1 = xp[1];
            raw[0] = q0.x; raw[1] = q0.y; raw[2] = q0.z; raw[3] = q0.w; raw[4] = q1.x; raw[5] = q1.y; raw[6] = q1.z; raw[7] = q1.w;
        } else if constexpr (IN == 16 && F16) {
            const uint4 q = *reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(obs) + m * IN + 8 * lhi);
            raw[0] = q.x; raw[1] = q.y; raw[2] = q.z; raw[3] = q.w;
        } else if constexpr (!F16) {   // 42 float32 columns: pairs, the half lhi = 1 holds 18 of them
            const uint2* xp = reinterpret_cast<const uint2*>(reinterpret_cast<const float*>(obs) + m * IN + KH * lhi);
#pragma unroll
            for (int j = 0; j < KH / 2; ++j) {
                uint2 q = make_uint2(0u, 0u);
                if (KH * lhi + 2 * j < IN) q = xp[j];
                raw[2 * j] = q.x;
                raw[2 * j + 1] = q.y;
            }
        } else {
            const uint32_t* xp = reinterpret_cast<const uint32_t*>(reinterpret_cast<const __half*>(obs) + m * IN + KH * lhi);
#pragma unroll
            for (int j = 0; j < KH / 2; ++j) raw[j] = (KH * lhi + 2 * j < IN) ? xp[j] : 0u;
        }
    }
    __device__ __forceinline__ void get(float (&x)[KH]) const {
#pragma unroll
        for (int j = 0; j < KH; ++j) {
            if constexpr (F16) {
                const uint32_t w = raw[j >> 1];
                x[j] = __half2float(__ushort_as_half((unsigned short)((j & 1) ? (w >> 16) : (w & 0xffffu))));
            } else {
                x[j] = __uint_as_float(raw[j]);
            }
        }
    }
};

// one observation entry as float32
template <bool F16>
__device__ __forceinline__ float obs_at(const void* __restrict__ obs, const long long idx) {
    if constexpr (F16) return __half2float(reinterpret_cast<const __half*>(obs)[idx]);
    else return reinterpret_cast<const float*>(obs)[idx];
}

// FWD: forward only (critic): V[m] = critic(obs[m]) is written to v_out and everything behind the output unit is skipped.
template <bool ACTOR, bool FWD = false, int IN = 16, bool F16 = false>
__device__ __forceinline__ void pass_body(SmemW<IN>& sm, const float* __restrict__ params, const void* __restrict__ obs,
                                          const float* __restrict__ act, const float* __restrict__ logp_old,
                                          const float* __restrict__ rtg, const float* __restrict__ adv,
                                          long long M, float var, float clip, float inv_n,
                                          float* __restrict__ partial, float* __restrict__ stats_partial,
                                          float* __restrict__ grad_zero, float* __restrict__ stats_zero,
                                          float* __restrict__ v_out = nullptr) {
    static_assert(!(FWD && ACTOR), "forward-only pass is the critic's");
    using L = Layout<IN>;
    constexpr int OFF_W1 = L::OFF_W1, OFF_B1 = L::OFF_B1, OFF_W2 = L::OFF_W2, OFF_B2 = L::OFF_B2, OFF_W3 = L::OFF_W3,
                  OFF_B3 = L::OFF_B3, OFF_W4 = L::OFF_W4, OFF_B4 = L::OFF_B4;
    constexpr int P = ACTOR ? L::P_ACTOR : L::P_CRITIC;
    constexpr int INP = Pad<IN>::INP, KH = Pad<IN>::KH, NC = Pad<IN>::NC, LW1 = Pad<IN>::LW1;
    constexpr int NW = Pad<IN>::NW;
    constexpr bool kKeepX = (IN == 16) || NW == 4;   // X stays in registers from F1 to G1, the next tile's rows are prefetched (see Pad)
    constexpr int NT = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5, l15 = lane & 15, kk = lane >> 4;

    (void)grad_zero;
    (void)stats_zero;
    if constexpr (INP != IN)
        for (int k = tid; k < H * (INP - IN); k += NT) sm.W1s[(k / (INP - IN)) * LW1 + IN + (k % (INP - IN))] = 0.f;
    for (int k = tid; k < H * IN; k += NT) sm.W1s[(k / IN) * LW1 + (k % IN)] = params[OFF_W1 + k];
    for (int k = tid; k < H * H; k += NT) {
        const float w = params[OFF_W2 + k];
        sm.W2s[(k / H) * LW2 + (k % H)] = w;
        sm.W2Ts[(k % H) * LW2 + (k / H)] = w;
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

    float* const T0 = sm.wv + wave * WAVE_F;   // H1^T rows 0..31  (later: X as [k][m])
    float* const T1 = T0 + TILE_F;             // H1^T rows 32..63
    float* const TD = T0 + 2 * TILE_F;         // H2^T / dH2^T / dH1^T, one 32-row tile at a time
    float* const gs = T0 + 3 * TILE_F;         // g3[m] | g4[m]
    // accumulator register r = 4 g + j of lane (m = l31, lhi) is row j + 8 g + 4 lhi of its 32x32 tile
    const int wr_base = 4 * lhi * LT + l31;    // tile[row][m] write: + (8 g + j) * LT
    const int rd32 = l31 * LT + 16 * lhi;      // row l31, samples 16 lhi .. + 15 (32x32x2 operands: k = sample)
    const int rd16 = l15 * LT + 8 * kk;        // row l15, samples 8 kk .. + 7   (16x16x4 operands)
    const int vec_off = 4 * lhi;               // b1 / b2 / w3 / w4 [32 t + 8 g + 4 lhi + j]
    const float* const w1row = sm.W1s + l31 * LW1 + KH * lhi;
    const float* const w2row = sm.W2s + l31 * LW2 + 4 * lhi;
    const float* const w2trow = sm.W2Ts + l31 * LW2 + 4 * lhi;

    // accumulators that persist over this wave's tiles
    f32x16 aW2[2][2];   // dW2 quadrant [n2 tile][n tile]
    f32x4 aW1[4][NC];   // dW1 rows 16 u .. + 15, columns 16 c .. + 15
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) aW2[a][b] = zero16();
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < NC; ++c) aW1[u][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float adb2[2] = {0.f, 0.f}, adw3[2] = {0.f, 0.f}, adw4[2] = {0.f, 0.f}, adb1[4] = {0.f, 0.f, 0.f, 0.f};
    float adb3 = 0.f, adb4 = 0.f, st0 = 0.f, st1 = 0.f, st2 = 0.f, st3 = 0.f;

    const long long n_tiles = (M + 31) / 32;
    const long long gw = (long long)blockIdx.x * NW + wave, stride = (long long)gridDim.x * NW;
    XRow<IN, F16> xpre;
    xpre.zero();
    float pre_a0 = 0.f, pre_a1 = 0.f, pre_lp = 0.f, pre_t = 0.f;
    auto prefetch_tile = [&](long long tile) {   // next tile's rows stream in while the current one is computed
        const long long m = tile * 32 + l31;
        xpre.zero();
        if (m < M) {
            xpre.request(obs, m, lhi);
            if (ACTOR) {
                const float2 a = reinterpret_cast<const float2*>(act)[m];
                pre_a0 = a.x;
                pre_a1 = a.y;
                pre_lp = logp_old[m];
                pre_t = adv[m];
            } else if (!FWD) {
                pre_t = rtg[m];
            }
        }
    };
    if (gw < n_tiles) prefetch_tile(gw);
    for (long long tile = gw; tile < n_tiles; tile += stride) {
        const bool valid = tile * 32 + l31 < M;
        float xr[KH];   // X[m][KH lhi + s]
        xpre.get(xr);
        const float cur_a0 = pre_a0, cur_a1 = pre_a1, cur_lp = pre_lp, cur_t = pre_t;
        if (kKeepX && tile + stride < n_tiles) prefetch_tile(tile + stride);

        // ---- F1: both 32-row tiles of H1^T interleaved (independent accumulators), eight k-steps per group of operand reads
        f32x16 c1[2];
        {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b = lds4(sm.b1 + 32 * t + 8 * g + vec_off);
                    c1[t][4 * g] = b.x; c1[t][4 * g + 1] = b.y; c1[t][4 * g + 2] = b.z; c1[t][4 * g + 3] = b.w;
                }
#pragma unroll
            for (int ch = 0; ch < KH / 8; ++ch) {
                float wa[2][8];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float4 w0 = lds4(w1row + 32 * t * LW1 + 8 * ch), w1 = lds4(w1row + 32 * t * LW1 + 8 * ch + 4);
                    wa[t][0] = w0.x; wa[t][1] = w0.y; wa[t][2] = w0.z; wa[t][3] = w0.w;
                    wa[t][4] = w1.x; wa[t][5] = w1.y; wa[t][6] = w1.z; wa[t][7] = w1.w;
                }
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    c1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[0][s], xr[8 * ch + s], c1[0], 0, 0, 0);
                    c1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[1][s], xr[8 * ch + s], c1[1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                c1[0][r] = relu_bits(c1[0][r]);
                c1[1][r] = relu_bits(c1[1][r]);
                const int o = wr_base + (8 * (r >> 2) + (r & 3)) * LT;
                T0[o] = c1[0][r];
                T1[o] = c1[1][r];
            }
        }

        // ---- F2: H2^T, B operands are the H1^T accumulators
        f32x16 c2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = lds4(sm.b2 + 32 * t + 8 * g + vec_off);
                c2[t][4 * g] = b.x; c2[t][4 * g + 1] = b.y; c2[t][4 * g + 2] = b.z; c2[t][4 * g + 3] = b.w;
            }
        {   // operands of group i + 1 are requested before the MFMAs of group i are issued
            float4 w0n = lds4(w2row), w1n = lds4(w2row + 32 * LW2);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int t1 = i >> 2, g = i & 3;
                const float4 w0 = w0n, w1 = w1n;
                if (i + 1 < 8) {
                    w0n = lds4(w2row + 8 * (i + 1));
                    w1n = lds4(w2row + 32 * LW2 + 8 * (i + 1));
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the requests above the MFMAs they overlap with
                const float a0[4] = {w0.x, w0.y, w0.z, w0.w}, a1[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], c1[t1][4 * g + j], c2[0], 0, 0, 0);
                    c2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], c1[t1][4 * g + j], c2[1], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            c2[0][r] = relu_bits(c2[0][r]);
            c2[1][r] = relu_bits(c2[1][r]);
        }

        // ---- output units + loss (every lane: the two halves of a sample hold 32 hidden units each)
        float g3 = 0.f, g4 = 0.f;
        {
            float z3 = 0.f, z4 = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 w = lds4(sm.w3 + 32 * t + 8 * g + vec_off);
                    z3 = fmaf(c2[t][4 * g], w.x, z3); z3 = fmaf(c2[t][4 * g + 1], w.y, z3);
                    z3 = fmaf(c2[t][4 * g + 2], w.z, z3); z3 = fmaf(c2[t][4 * g + 3], w.w, z3);
                    if (ACTOR) {
                        const float4 v = lds4(sm.w4 + 32 * t + 8 * g + vec_off);
                        z4 = fmaf(c2[t][4 * g], v.x, z4); z4 = fmaf(c2[t][4 * g + 1], v.y, z4);
                        z4 = fmaf(c2[t][4 * g + 2], v.z, z4); z4 = fmaf(c2[t][4 * g + 3], v.w, z4);
                    }
                }
            z3 += __shfl_xor(z3, 32, 64);
            if (ACTOR) z4 += __shfl_xor(z4, 32, 64);
            if (FWD) {
                if (valid && lhi == 0) v_out[tile * 32 + l31] = z3 + b3;   // critic(obs).squeeze(), ppo.py:275,724
                if (!kKeepX && tile + stride < n_tiles) prefetch_tile(tile + stride);
                continue;
            }
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
                    // MultivariateNormal(mean, var*I).log_prob, ppo.py:734-735
                    const float lp = -0.5f * ((d0 * d0 + d1 * d1) / var) - 1.8378770664093453f - logf(var);
                    const float lr = lp - cur_lp;
                    const float ratio = expf(lr);                  // ppo.py:316
                    const float A = cur_t;
                    const float s1 = ratio * A;                     // ppo.py:319
                    const float rc = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
                    const float s2 = rc * A;                        // ppo.py:320
                    st0 += own * -fminf(s1, s2);                    // ppo.py:342 (mean taken by inv_n at the end)
                    st2 += own * ((ratio - 1.0f) - lr);             // approx KL, ppo.py:326
                    st3 += (fabsf(ratio - 1.0f) > clip) ? own : 0.f;  // clip fraction, ppo.py:335
                    // d(-min(s1,s2))/d ratio: -A through s1 when s1 <= s2 inside the clip range (tie: both halves),
                    // or when s1 < s2 outside it; 0 when the clipped (constant) branch is the minimum
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
            if (lhi == 0) {
                gs[l31] = g3;
                if (ACTOR) gs[32 + l31] = g4;
            }
        }

        // ---- per 32-row tile of the second layer: dW3/dW4 from H2^T, dH2^T in place, then its two dW2 quadrants
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) TD[wr_base + (8 * (r >> 2) + (r & 3)) * LT] = c2[t2][r];
            wave_lds_fence();
#pragma unroll
            for (int q = 0; q < 4; ++q) {   // lane (n2 = 32 t2 + l31, half lhi): sum over its 16 samples
                const float4 h = lds4(TD + rd32 + 4 * q);
                const float4 a = lds4(gs + 16 * lhi + 4 * q);
                adw3[t2] = fmaf(h.x, a.x, adw3[t2]); adw3[t2] = fmaf(h.y, a.y, adw3[t2]);
                adw3[t2] = fmaf(h.z, a.z, adw3[t2]); adw3[t2] = fmaf(h.w, a.w, adw3[t2]);
                if (ACTOR) {
                    const float4 b = lds4(gs + 32 + 16 * lhi + 4 * q);
                    adw4[t2] = fmaf(h.x, b.x, adw4[t2]); adw4[t2] = fmaf(h.y, b.y, adw4[t2]);
                    adw4[t2] = fmaf(h.z, b.z, adw4[t2]); adw4[t2] = fmaf(h.w, b.w, adw4[t2]);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 w = lds4(sm.w3 + 32 * t2 + 8 * g + vec_off);
                const float wv3[4] = {w.x, w.y, w.z, w.w};
                float wv4[4] = {0.f, 0.f, 0.f, 0.f};
                if (ACTOR) {
                    const float4 v = lds4(sm.w4 + 32 * t2 + 8 * g + vec_off);
                    wv4[0] = v.x; wv4[1] = v.y; wv4[2] = v.z; wv4[3] = v.w;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = ACTOR ? fmaf(g3, wv3[j], g4 * wv4[j]) : g3 * wv3[j];
                    c2[t2][4 * g + j] = (c2[t2][4 * g + j] > 0.f) ? d : 0.f;
                }
            }
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r) TD[wr_base + (8 * (r >> 2) + (r & 3)) * LT] = c2[t2][r];
            wave_lds_fence();
            {
                float4 an = lds4(TD + rd32), h0n = lds4(T0 + rd32), h1n = lds4(T1 + rd32);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = an, h0 = h0n, h1 = h1n;
                    if (q + 1 < 4) {
                        an = lds4(TD + rd32 + 4 * (q + 1));
                        h0n = lds4(T0 + rd32 + 4 * (q + 1));
                        h1n = lds4(T1 + rd32 + 4 * (q + 1));
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep the requests above the MFMAs they overlap with
                    adb2[t2] += (a.x + a.y) + (a.z + a.w);
                    const float ar[4] = {a.x, a.y, a.z, a.w}, b0[4] = {h0.x, h0.y, h0.z, h0.w}, b1v[4] = {h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        aW2[t2][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[j], b0[j], aW2[t2][0], 0, 0, 0);
                        aW2[t2][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[j], b1v[j], aW2[t2][1], 0, 0, 0);
                    }
                }
            }
            wave_lds_fence();
        }

        // ---- B2: dH1^T = (W2^T dH2^T) . [H1 > 0]; B operands are the dH2^T registers
        f32x16 c3[2] = {zero16(), zero16()};
        {
            float4 w0n = lds4(w2trow), w1n = lds4(w2trow + 32 * LW2);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int t2 = i >> 2, g = i & 3;
                const float4 w0 = w0n, w1 = w1n;
                if (i + 1 < 8) {
                    w0n = lds4(w2trow + 8 * (i + 1));
                    w1n = lds4(w2trow + 32 * LW2 + 8 * (i + 1));
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the requests above the MFMAs they overlap with
                const float a0[4] = {w0.x, w0.y, w0.z, w0.w}, a1[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c3[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], c2[t2][4 * g + j], c3[0], 0, 0, 0);
                    c3[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], c2[t2][4 * g + j], c3[1], 0, 0, 0);
                }
            }
        }

        // relu mask of layer 1: H1^T is still in T0 / T1 (same lane, same slot it was written from)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = wr_base + (8 * (r >> 2) + (r & 3)) * LT;
            c3[0][r] = (T0[o] > 0.f) ? c3[0][r] : 0.f;
            c3[1][r] = (T1[o] > 0.f) ? c3[1][r] : 0.f;
        }
        wave_lds_fence();
        // ---- G1: dH1^T through TD 32 rows at a time; X as [k = sample][column]: IN = 16 from the registers through T0 (the H1^T
        // tiles are no longer needed), IN = 42 read again from L2 straight in the operand layout (lane (l15, kk): samples 8 kk + s),
        // one 16-column tile at a time, the next tile's eight values requested ahead of the current tile's MFMAs
        if constexpr (kKeepX) {   // (42 columns: 48 rows, T0 and the first half of T1 -- the tiles are contiguous)
#pragma unroll
            for (int s = 0; s < KH; ++s) T0[(KH * lhi + s) * LT + l31] = xr[s];
        }
        int xrow[8];   // (IN = 42) row offsets of the lane's eight samples; rows past the batch re-read the last one (dH1 is zero there)
        if constexpr (!kKeepX) {
            const long long m0 = tile * 32 + 8 * kk;
            const int lim = (int)((M - 1 - m0 < 7) ? ((M - 1 - m0 < 0) ? 0 : M - 1 - m0) : 7);
#pragma unroll
            for (int s = 0; s < 8; ++s) xrow[s] = min(s, lim) * IN;
        }
        const long long xbase = (tile * 32 + 8 * kk < M ? tile * 32 + 8 * kk : M - 1) * IN + l15;
        auto load_xb = [&](const int c, float (&dst)[8]) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 8; ++s) dst[s] = (16 * c + l15 < IN) ? obs_at<F16>(obs, xbase + xrow[s] + 16 * c) : 0.f;
        };
        float xb[2][8];   // X[m = 8 kk + s][16 c + l15], two column tiles in turn
        if constexpr (!kKeepX) load_xb(0, xb[0]);
#pragma unroll
        for (int t1 = 0; t1 < 2; ++t1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) TD[wr_base + (8 * (r >> 2) + (r & 3)) * LT] = c3[t1][r];
            wave_lds_fence();
            if constexpr (kKeepX && NC == 1) {
                const float4 x0 = lds4(T0 + rd16), x1 = lds4(T0 + rd16 + 4);
                xb[0][0] = x0.x; xb[0][1] = x0.y; xb[0][2] = x0.z; xb[0][3] = x0.w;
                xb[0][4] = x1.x; xb[0][5] = x1.y; xb[0][6] = x1.z; xb[0][7] = x1.w;   // X[m = 8 kk + s][k = l15]
            }
            {
                const float4 d0 = lds4(TD + rd16), d1 = lds4(TD + rd16 + 4);
                const float4 e0 = lds4(TD + 16 * LT + rd16), e1 = lds4(TD + 16 * LT + rd16 + 4);
                const float da[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                const float ea[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
                adb1[2 * t1] += ((d0.x + d0.y) + (d0.z + d0.w)) + ((d1.x + d1.y) + (d1.z + d1.w));
                adb1[2 * t1 + 1] += ((e0.x + e0.y) + (e0.z + e0.w)) + ((e1.x + e1.y) + (e1.z + e1.w));
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int cur = kKeepX ? 0 : ((t1 * NC + c) & 1);
                    if constexpr (kKeepX && NC > 1) {   // X[m = 8 kk + s][16 c + l15] out of the staged rows
                        const float4 x0 = lds4(T0 + 16 * c * LT + rd16), x1 = lds4(T0 + 16 * c * LT + rd16 + 4);
                        xb[0][0] = x0.x; xb[0][1] = x0.y; xb[0][2] = x0.z; xb[0][3] = x0.w;
                        xb[0][4] = x1.x; xb[0][5] = x1.y; xb[0][6] = x1.z; xb[0][7] = x1.w;
                    }
                    if constexpr (!kKeepX) {
                        if (t1 * NC + c + 1 < 2 * NC) load_xb((c + 1) % NC, xb[cur ^ 1]);
                        __builtin_amdgcn_sched_barrier(0);   // keep the requests above the MFMAs they overlap with
                    }
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        aW1[2 * t1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(da[s], xb[cur][s], aW1[2 * t1][c], 0, 0, 0);
                        aW1[2 * t1 + 1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[s], xb[cur][s], aW1[2 * t1 + 1][c], 0, 0, 0);
                    }
                    if constexpr (!kKeepX) __builtin_amdgcn_sched_barrier(0);
                }
            }
            wave_lds_fence();
        }
        if (!kKeepX && tile + stride < n_tiles) prefetch_tile(tile + stride);
    }

    if (FWD) return;
    // ---- workgroup reduction of the 8 waves' partial gradients in LDS, then one coalesced row of `partial`.
    // No atomics (round 1 added all 8 waves' 5.4 k accumulators into one LDS row with ds_add_f32: 8-way same-address
    // conflicts, ~60 us per launch = 10 % of the epoch): waves 0-3 STORE their accumulators into four private rows, waves 4-7
    // then ADD theirs on top (wave w + 4 onto row w: every index of a row has exactly one owner lane per wave), then all
    // threads sum the four rows.  Accumulators that two or four lanes of a wave share (the bias / output-weight sums of the
    // two sample halves, the loss statistics) are combined with shuffles first.
    __syncthreads();
    constexpr int RP = (P + 3 + 3) & ~3;   // row pitch (floats): P parameters + 3 statistics, a multiple of 4
    // the four rows lie over the wave tiles; the 42-wide nets' rows (4 x 7048 floats) start at W2s, which nothing reads any more
    float* const red_base = (IN == 16) ? sm.wv : sm.W2s;
    static_assert(4 * RP <= (int)(sizeof(sm.wv) / sizeof(float)) + ((IN == 16) ? 0 : 2 * H * LW2 + 4 * H), "reduction rows fit");
    float s3 = adb3, s4 = adb4, sA = ACTOR ? st0 : st1, sB = st2, sC = st3;   // held per lane (lhi == 0 lanes non-zero)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s3 += __shfl_xor(s3, o, 64); s4 += __shfl_xor(s4, o, 64);
        sA += __shfl_xor(sA, o, 64); sB += __shfl_xor(sB, o, 64); sC += __shfl_xor(sC, o, 64);
    }
    float hb2[2], hw3[2], hw4[2], qb1[4];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
        hb2[t2] = adb2[t2] + __shfl_xor(adb2[t2], 32, 64);
        hw3[t2] = adw3[t2] + __shfl_xor(adw3[t2], 32, 64);
        hw4[t2] = adw4[t2] + __shfl_xor(adw4[t2], 32, 64);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float v = adb1[u];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        qb1[u] = v;
    }
    float* const row = red_base + (wave & 3) * RP;
#pragma unroll
    for (int pass = 0; pass < NW / 4; ++pass) {
        if ((wave >> 2) == pass) {
            const bool add = pass >= 1;
            auto put = [&](int idx, float v) { row[idx] = add ? row[idx] + v : v; };
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
                for (int t1 = 0; t1 < 2; ++t1)
#pragma unroll
                    for (int r = 0; r < 16; ++r) put(OFF_W2 + (32 * t2 + c_row(r, lane)) * H + 32 * t1 + l31, aW2[t2][t1][r]);
                if (lhi == 0) {
                    put(OFF_B2 + 32 * t2 + l31, hb2[t2]);
                    put(OFF_W3 + 32 * t2 + l31, hw3[t2]);
                    if (ACTOR) put(OFF_W4 + 32 * t2 + l31, hw4[t2]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (INP == IN || 16 * c + l15 < IN) put(OFF_W1 + (16 * u + 4 * kk + r) * IN + 16 * c + l15, aW1[u][c][r]);
                if (kk == 0) put(OFF_B1 + 16 * u + l15, qb1[u]);
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
    }
    float* out = partial + (size_t)blockIdx.x * P;
    const float* red = red_base;
    for (int k = tid; k < P; k += NT) out[k] = (red[k] + red[RP + k]) + (red[2 * RP + k] + red[3 * RP + k]);
    if (tid < 3) stats_partial[blockIdx.x * 4 + tid] = (red[P + tid] + red[RP + P + tid]) + (red[2 * RP + P + tid] + red[3 * RP + P + tid]);
}

template <bool ACTOR, bool FWD, int IN, bool F16>
__global__ __launch_bounds__(64 * Pad<IN>::NW) void mlp64_pass_w(const float* __restrict__ params, const void* __restrict__ obs,
                                                          const float* __restrict__ act, const float* __restrict__ logp_old,
                                                          const float* __restrict__ rtg, const float* __restrict__ adv,
                                                          long long M, float var, float clip, float inv_n,
                                                          float* __restrict__ partial, float* __restrict__ stats_partial,
                                                          float* __restrict__ grad_zero, float* __restrict__ stats_zero,
                                                          float* __restrict__ v_out = nullptr) {
    __shared__ __attribute__((aligned(16))) SmemW<IN> sm;
    pass_body<ACTOR, FWD, IN, F16>(sm, params, obs, act, logp_old, rtg, adv, M, var, clip, inv_n, partial, stats_partial, grad_zero,
                                   stats_zero, v_out);
}

// both nets of one epoch in one launch (single-GPU path): the actor's tiles, then the critic's, by the same workgroups
template <int IN, bool F16>
__global__ __launch_bounds__(64 * Pad<IN>::NW) void mlp64_pass_both(const float* __restrict__ params, const void* __restrict__ obs,
                                                             const float* __restrict__ act, const float* __restrict__ logp_old,
                                                             const float* __restrict__ rtg, const float* __restrict__ adv,
                                                             long long M, float var, float clip, float inv_n,
                                                             float* __restrict__ partial_a, float* __restrict__ stats_partial_a,
                                                             float* __restrict__ partial_c, float* __restrict__ stats_partial_c,
                                                             float* __restrict__ grad, float* __restrict__ stats) {
    __shared__ __attribute__((aligned(16))) SmemW<IN> sm;
    pass_body<true, false, IN, F16>(sm, params, obs, act, logp_old, rtg, adv, M, var, clip, inv_n, partial_a, stats_partial_a, grad,
                                    stats);
    __syncthreads();
    pass_body<false, false, IN, F16>(sm, params + Layout<IN>::P_ACTOR, obs, act, logp_old, rtg, adv, M, var, clip, inv_n, partial_c,
                                     stats_partial_c, grad + Layout<IN>::P_ACTOR, stats + 4);
}

// ================================================================ the split-bf16 pass ("bf16x3")
// The same pass -- same tile, same order of operations, float32 results -- with every matrix product evaluated on the bf16 MFMA
// (v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16: 16 x the contraction depth per unit time of the f32-input MFMA) from
// operands split into three bf16 pieces:  a = a0 + a1 + a2 EXACTLY (a0 = bf16(a) round-to-nearest-even, a1 = bf16(a - a0),
// a2 = a - a0 - a1, which has at most 8 significant bits left), and
//     a b  ~  a2 b0 + a1 b1 + a0 b2 + a1 b0 + a0 b1 + a0 b0          (the three dropped terms are <= 2^-23 |a b|)
// every bf16 x bf16 product exact in float32, float32 accumulation inside the MFMA.  Order: small terms first -- the five small
// terms of ALL k-steps of a product, then its a0 b0 terms, then (on the vector unit) the bias -- so only a handful of accumulations
// happen at the magnitude of the result.  Measured against float64 (tools/design/bf16_split_error.py,
// profiles/r04_bf16_split_mfma_device.txt, profiles/r05_bf16x3_error.txt): not worse than the f32 MFMA chain on any of the
// kernel's contraction shapes.  What is split where:
//   X        once per update by mlp64_split_obs (the observations do not change over the 50 epochs): row-major pieces for F1 (lane =
//            sample) and tile-transposed pieces for G1 (k = sample)
//   W1, W2   once per launch into LDS (W2 also transposed for B2), k-permuted to match the accumulator layout (below)
//   H1, dH2  as B operands of F2 / B2: split in-lane from the accumulator registers -- register r = 8 j + e of lane (m, hi) is row
//            16 j + 8 (e >> 2) + 4 hi + (e & 3) of its 32-row tile, so k-step (t, j) takes registers 8 j .. 8 j + 7 as ITS eight
//            k-slots and the weights are stored with the middle two quads of every 16 columns swapped
//   H1, dH2, dH1  as operands of the weight-gradient products (contraction over samples): through the wave's float32 LDS tiles as
//            in the f32 kernel (the only transposes), split after the read
// LDS: weight pieces 55 KB + 8 x 12.25 KB wave tiles (32-float rows, 16-byte slots XOR-swizzled by the row instead of padded).
using namespace bf16x3;   // cvt_pk_bf16, split_pair, Pieces, split8, as_bf (csrc/bf16x3.h)
#ifdef X3_EXPERIMENT_NO_MFMA   // (timing experiment: the kernel without its MFMAs, operands kept alive; results are wrong)
#define X3_MFMA32(a, b, c) ([&] { asm volatile("" ::"v"(a), "v"(b)); return c; }())
#define X3_MFMA16(a, b, c) ([&] { asm volatile("" ::"v"(a), "v"(b)); return c; }())
#else
#define X3_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define X3_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
// Two accumulators take turns in every group below (an MFMA that reads the accumulator the MFMA right in front of it writes waits
// for that result).
// the five small terms of two products with a common B: acc_t += a2 b0 + a1 b1 + a0 b2 + a1 b0 + a0 b1
__device__ __forceinline__ void mma5_small2(f32x16& acc0, f32x16& acc1, const Pieces& A0, const Pieces& A1, const Pieces& B) {
    acc0 = X3_MFMA32(as_bf(A0.p[2]), as_bf(B.p[0]), acc0);
    acc1 = X3_MFMA32(as_bf(A1.p[2]), as_bf(B.p[0]), acc1);
    acc0 = X3_MFMA32(as_bf(A0.p[1]), as_bf(B.p[1]), acc0);
    acc1 = X3_MFMA32(as_bf(A1.p[1]), as_bf(B.p[1]), acc1);
    acc0 = X3_MFMA32(as_bf(A0.p[0]), as_bf(B.p[2]), acc0);
    acc1 = X3_MFMA32(as_bf(A1.p[0]), as_bf(B.p[2]), acc1);
    acc0 = X3_MFMA32(as_bf(A0.p[1]), as_bf(B.p[0]), acc0);
    acc1 = X3_MFMA32(as_bf(A1.p[1]), as_bf(B.p[0]), acc1);
    acc0 = X3_MFMA32(as_bf(A0.p[0]), as_bf(B.p[1]), acc0);
    acc1 = X3_MFMA32(as_bf(A1.p[0]), as_bf(B.p[1]), acc1);
}
// the big term: acc += a0 b0
__device__ __forceinline__ void mma1_big(f32x16& acc, const uint4 a0, const uint4 b0) {
    acc = X3_MFMA32(as_bf(a0), as_bf(b0), acc);
}
// running accumulators (weight gradients: the sum over all tiles is already in them), small terms first; common A
__device__ __forceinline__ void mma6_acc2(f32x16& acc0, f32x16& acc1, const Pieces& A, const Pieces& B0, const Pieces& B1) {
    acc0 = X3_MFMA32(as_bf(A.p[2]), as_bf(B0.p[0]), acc0);
    acc1 = X3_MFMA32(as_bf(A.p[2]), as_bf(B1.p[0]), acc1);
    acc0 = X3_MFMA32(as_bf(A.p[1]), as_bf(B0.p[1]), acc0);
    acc1 = X3_MFMA32(as_bf(A.p[1]), as_bf(B1.p[1]), acc1);
    acc0 = X3_MFMA32(as_bf(A.p[0]), as_bf(B0.p[2]), acc0);
    acc1 = X3_MFMA32(as_bf(A.p[0]), as_bf(B1.p[2]), acc1);
    acc0 = X3_MFMA32(as_bf(A.p[1]), as_bf(B0.p[0]), acc0);
    acc1 = X3_MFMA32(as_bf(A.p[1]), as_bf(B1.p[0]), acc1);
    acc0 = X3_MFMA32(as_bf(A.p[0]), as_bf(B0.p[1]), acc0);
    acc1 = X3_MFMA32(as_bf(A.p[0]), as_bf(B1.p[1]), acc1);
    acc0 = X3_MFMA32(as_bf(A.p[0]), as_bf(B0.p[0]), acc0);
    acc1 = X3_MFMA32(as_bf(A.p[0]), as_bf(B1.p[0]), acc1);
}
// 16 x 16 outputs, K = 32; common B
__device__ __forceinline__ void mma6_acc16_2(f32x4& acc0, f32x4& acc1, const Pieces& A0, const Pieces& A1, const Pieces& B) {
    acc0 = X3_MFMA16(as_bf(A0.p[2]), as_bf(B.p[0]), acc0);
    acc1 = X3_MFMA16(as_bf(A1.p[2]), as_bf(B.p[0]), acc1);
    acc0 = X3_MFMA16(as_bf(A0.p[1]), as_bf(B.p[1]), acc0);
    acc1 = X3_MFMA16(as_bf(A1.p[1]), as_bf(B.p[1]), acc1);
    acc0 = X3_MFMA16(as_bf(A0.p[0]), as_bf(B.p[2]), acc0);
    acc1 = X3_MFMA16(as_bf(A1.p[0]), as_bf(B.p[2]), acc1);
    acc0 = X3_MFMA16(as_bf(A0.p[1]), as_bf(B.p[0]), acc0);
    acc1 = X3_MFMA16(as_bf(A1.p[1]), as_bf(B.p[0]), acc1);
    acc0 = X3_MFMA16(as_bf(A0.p[0]), as_bf(B.p[1]), acc0);
    acc1 = X3_MFMA16(as_bf(A1.p[0]), as_bf(B.p[1]), acc1);
    acc0 = X3_MFMA16(as_bf(A0.p[0]), as_bf(B.p[0]), acc0);
    acc1 = X3_MFMA16(as_bf(A1.p[0]), as_bf(B.p[0]), acc1);
}

// ---- the pre-split observations (navppo_mlp64_bf16x3_prepare): per 32-sample tile, INP = 16 | 48 columns on chip (42-column rows:
// columns 42 .. 47 are zero), KX = INP / 16 k-steps of F1 / column tiles of dW1
//   rows  [32 m][3 pieces][INP f] bf16   F1's B operand: lane (m, kh) reads the eight features 16 ks + 8 kh .. + 7 of piece i (16 bytes)
//   cols  [3 pieces][INP f][32 m] bf16   G1's B operand: lane (f, kb) reads the eight samples 8 kb .. 8 kb + 7 of feature 16 c + f
#ifndef X3_WAVES
#define X3_WAVES 8   // waves per workgroup of the 16-column split pass (one workgroup per CU: 8 = two per SIMD; measured with 4, one per
#endif               // SIMD and 512 registers each: 1069 us per epoch against 924)
template <int IN>
struct XPad {
    static constexpr int INP = Pad<IN>::INP, KX = INP / 16;
    static constexpr int W1P = (INP == 16) ? 16 : 56;       // pitch of a W1 piece row in LDS (bf16): 112-byte rows, conflict-free 16-byte reads
    // 42 columns: the wave tiles of 8 waves + 21 KB of W1 pieces do not fit the LDS beside W2 / W2^T -- 4 waves (one per SIMD, 512
    // registers each: the 48 dW1 accumulators and 36 + 36 registers of row pieces fit without spills), as the f32 pass of that width
    static constexpr int NW = (IN == 16) ? X3_WAVES : 4;
    static constexpr int kRowsBytes = 32 * 3 * INP * 2;     // 3072 | 9216: offset of `cols` inside a tile
    static constexpr int kTileBytes = 2 * kRowsBytes;       // 6144 | 18432
};

template <bool F16, int IN>
__global__ __launch_bounds__(64) void mlp64_split_obs(const void* __restrict__ obs, long long M, unsigned char* __restrict__ prep) {
    constexpr int INP = XPad<IN>::INP, KX = XPad<IN>::KX;
    __shared__ __attribute__((aligned(16))) uint16_t colsT[3][INP][32];
    const int lane = threadIdx.x, m = lane & 31, kh = lane >> 5;
    const long long tile = blockIdx.x, row = tile * 32 + m;
    unsigned char* const t = prep + (size_t)tile * XPad<IN>::kTileBytes;
#pragma unroll
    for (int ks = 0; ks < KX; ++ks) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int f = 16 * ks + 8 * kh + e;
            v[e] = (row < M && f < IN) ? obs_at<F16>(obs, row * IN + min(f, IN - 1)) : 0.f;
        }
        const Pieces P = split8(v);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            *reinterpret_cast<uint4*>(t + ((m * 3 + i) * INP + 16 * ks + 8 * kh) * 2) = P.p[i];
            const uint32_t w[4] = {P.p[i].x, P.p[i].y, P.p[i].z, P.p[i].w};
#pragma unroll
            for (int e = 0; e < 8; ++e)
                colsT[i][16 * ks + 8 * kh + e][m] = (uint16_t)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu));
        }
    }
    __syncthreads();
    // 3 x INP x 32 bf16 = INP x 12 x 16 bytes
    const uint4* src = reinterpret_cast<const uint4*>(&colsT[0][0][0]);
    uint4* dst = reinterpret_cast<uint4*>(t + XPad<IN>::kRowsBytes);
    for (int k = lane; k < 12 * INP; k += 64) dst[k] = src[k];
}

// ---- LDS of the split pass
constexpr int XT = 32;                    // wave tile: 32 rows x 32 floats, 16-byte slot c of row r stored at slot c ^ ((r >> 1) & 7)
constexpr int X_TILE_F = 32 * XT;
constexpr int X_WAVE_F = 3 * X_TILE_F + 64;   // T0 | T1 | TD | g3[32] g4[32]
template <int IN>
struct SmemX {
    // weight pieces, bf16, [piece][row][k-position]; 16-byte slot s of row r stored at slot s ^ ((r >> 1) & 7)
    uint16_t W2p[3][H][H];    // rows = layer-2 units (A operand of F2), k = layer-1 units, permuted
    uint16_t W2Tp[3][H][H];   // rows = layer-1 units (A operand of B2), k = layer-2 units, permuted
    uint16_t W1p[3][H][XPad<IN>::W1P];   // rows = layer-1 units (A operand of F1), k = features
    float b1[H], b2[H], w3[H], w4[H];
    // the wave tiles; at the end of a pass also the four rows of the workgroup's gradient reduction -- 16 columns: in wv; 42 columns
    // (4 x 7048 floats): from the start of the struct on, nothing reads the weights any more at that point
    static constexpr int kRedRow = (Layout<IN>::P_ACTOR + 6) & ~3;
    static constexpr int kWvF = (IN == 16 && 4 * kRedRow > XPad<IN>::NW * X_WAVE_F) ? 4 * kRedRow : XPad<IN>::NW * X_WAVE_F;
    float wv[kWvF];
};
static_assert(sizeof(SmemX<16>) <= 160 * 1024 && sizeof(SmemX<42>) <= 160 * 1024, "LDS");
static_assert(sizeof(SmemX<42>) >= 4 * SmemX<42>::kRedRow * sizeof(float), "the reduction rows of the 42-column nets fit the struct");

// element (row, col) of a swizzled wave tile
__device__ __forceinline__ int xt(const int row, const int col) { return row * XT + ((((col >> 2) ^ (row >> 1)) & 7) << 2) + (col & 3); }
// k-position of column k in a weight row: the middle two quads of every 16 columns are swapped (see the header)
__device__ __forceinline__ int kpos(const int k) { return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1); }
// address (in uint16 units) of k-position p of row r
__device__ __forceinline__ int wslot(const int r, const int p) { return r * H + ((((p >> 3) ^ (r >> 1)) & 7) << 3) + (p & 7); }

template <bool ACTOR, int IN>
__device__ __forceinline__ void pass_body_x3(SmemX<IN>& sm, const float* __restrict__ params, const unsigned char* __restrict__ prep,
                                             const float* __restrict__ act, const float* __restrict__ logp_old,
                                             const float* __restrict__ rtg, const float* __restrict__ adv, long long M, float var,
                                             float clip, float inv_n, float* __restrict__ partial, float* __restrict__ stats_partial) {
    using L = Layout<IN>;   // (the names below shadow the 16-column constants of the namespace)
    constexpr int OFF_W1 = L::OFF_W1, OFF_B1 = L::OFF_B1, OFF_W2 = L::OFF_W2, OFF_B2 = L::OFF_B2, OFF_W3 = L::OFF_W3, OFF_B3 = L::OFF_B3,
                  OFF_W4 = L::OFF_W4, OFF_B4 = L::OFF_B4;
    constexpr int P = ACTOR ? L::P_ACTOR : L::P_CRITIC;
    constexpr int INP = XPad<IN>::INP, KX = XPad<IN>::KX, kXWaves = XPad<IN>::NW, kX3TileBytes = XPad<IN>::kTileBytes,
                  kX3RowsBytes = XPad<IN>::kRowsBytes;
    constexpr int NT = 64 * kXWaves;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5, l15 = lane & 15, kk = lane >> 4;

    // ---- weights -> bf16 pieces in LDS (once per launch: 5120 values, 10 per thread)
    for (int k = tid; k < H * H / 2; k += NT) {   // pairs of adjacent columns
        const int r = (2 * k) / H, c = (2 * k) % H;
        uint32_t p0, p1, p2;
        split_pair(params[OFF_W2 + r * H + c], params[OFF_W2 + r * H + c + 1], p0, p1, p2);
        const uint32_t pc[3] = {p0, p1, p2};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            // W2p[i][r][kpos(c)], [kpos(c + 1)] (c even: the pair stays adjacent under kpos) ; W2Tp[i][c][kpos(r)], W2Tp[i][c + 1][kpos(r)]
            *reinterpret_cast<uint32_t*>(&sm.W2p[i][0][0] + wslot(r, kpos(c))) = pc[i];
            (&sm.W2Tp[i][0][0])[wslot(c, kpos(r))] = (uint16_t)(pc[i] & 0xffffu);
            (&sm.W2Tp[i][0][0])[wslot(c + 1, kpos(r))] = (uint16_t)(pc[i] >> 16);
        }
    }
    for (int k = tid; k < H * INP / 2; k += NT) {   // (42 columns: the pair (c, c + 1) is inside the row or past it as a whole)
        const int r = (2 * k) / INP, c = (2 * k) % INP;
        uint32_t p0, p1, p2;
        const int cc = min(c, IN - 2);
        const float wa_ = params[OFF_W1 + r * IN + cc], wb_ = params[OFF_W1 + r * IN + cc + 1];
        split_pair(c < IN ? wa_ : 0.f, c < IN ? wb_ : 0.f, p0, p1, p2);
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

    float* const T0 = sm.wv + wave * X_WAVE_F;   // H1^T rows 0..31
    float* const T1 = T0 + X_TILE_F;             // H1^T rows 32..63
    float* const TD = T0 + 2 * X_TILE_F;         // H2^T / dH2^T / dH1^T, one 32-row tile at a time
    float* const gs = T0 + 3 * X_TILE_F;         // g3[m] | g4[m]
    const int vec_off = 4 * lhi;                 // b1 / b2 / w3 / w4 [32 t + 8 g + 4 lhi + j]
    // accumulator register r = 4 g + j of lane (m = l31, lhi) is row j + 8 g + 4 lhi of its 32 x 32 tile: tile[row][m]
    auto wr = [&](const int r) { return xt(8 * (r >> 2) + 4 * lhi + (r & 3), l31); };
    // the A operand of weight step (row tile tt, k-step s of 4, piece i): 16 bytes
    auto ldw = [&](const uint16_t* base, const int tt, const int s, const int i) {
        const int r = 32 * tt + l31;
        return *reinterpret_cast<const uint4*>(base + i * H * H + wslot(r, 16 * s + 8 * lhi));
    };

    // accumulators that persist over this wave's tiles
    f32x16 aW2[2][2];   // dW2 quadrant [n2 tile][n tile]
    f32x4 aW1[KX][4];   // dW1 rows 16 u .. + 15 of column tile c
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) aW2[a][b] = zero16();
#pragma unroll
    for (int c = 0; c < KX; ++c)
#pragma unroll
        for (int u = 0; u < 4; ++u) aW1[c][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    float adb2[2] = {0.f, 0.f}, adw3[2] = {0.f, 0.f}, adw4[2] = {0.f, 0.f}, adb1[4] = {0.f, 0.f, 0.f, 0.f};
    float adb3 = 0.f, adb4 = 0.f, st0 = 0.f, st1 = 0.f, st2 = 0.f, st3 = 0.f;

    const long long n_tiles = (M + 31) / 32;
    const long long gw = (long long)blockIdx.x * kXWaves + wave, stride = (long long)gridDim.x * kXWaves;
    Pieces xp[KX];   // the tile's observation rows as F1's B operand (prefetched), one set of pieces per k-step
#pragma unroll
    for (int ks = 0; ks < KX; ++ks) xp[ks].p[0] = xp[ks].p[1] = xp[ks].p[2] = make_uint4(0u, 0u, 0u, 0u);
    float pre_a0 = 0.f, pre_a1 = 0.f, pre_lp = 0.f, pre_t = 0.f;
    auto prefetch_tile = [&](long long tile) {
        const long long m = tile * 32 + l31;
        const unsigned char* t = prep + (size_t)tile * kX3TileBytes + ((l31 * 3) * INP + 8 * lhi) * 2;
#pragma unroll
        for (int ks = 0; ks < KX; ++ks)
#pragma unroll
            for (int i = 0; i < 3; ++i)   // rows past the batch are zero in `prep`
                xp[ks].p[i] = *reinterpret_cast<const uint4*>(t + (i * INP + 16 * ks) * 2);
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
    if (gw < n_tiles) prefetch_tile(gw);
    for (long long tile = gw; tile < n_tiles; tile += stride) {
        const bool valid = tile * 32 + l31 < M;
        Pieces xr[KX];
#pragma unroll
        for (int ks = 0; ks < KX; ++ks) xr[ks] = xp[ks];
        const float cur_a0 = pre_a0, cur_a1 = pre_a1, cur_lp = pre_lp, cur_t = pre_t;
        if (tile + stride < n_tiles) prefetch_tile(tile + stride);

        // ---- F1: H1^T = relu(b1 + W1 X^T), K = 16 per k-step (one for 16 columns, three for 42)
        f32x16 c1[2] = {zero16(), zero16()};
        {
#pragma unroll
            for (int ks = 0; ks < KX; ++ks) {
                Pieces wa[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int i = 0; i < 3; ++i) wa[t].p[i] = *reinterpret_cast<const uint4*>(&sm.W1p[i][32 * t + l31][16 * ks + 8 * lhi]);
                mma5_small2(c1[0], c1[1], wa[0], wa[1], xr[ks]);
            }
#pragma unroll
            for (int ks = 0; ks < KX; ++ks) {
                mma1_big(c1[0], *reinterpret_cast<const uint4*>(&sm.W1p[0][l31][16 * ks + 8 * lhi]), xr[ks].p[0]);
                mma1_big(c1[1], *reinterpret_cast<const uint4*>(&sm.W1p[0][32 + l31][16 * ks + 8 * lhi]), xr[ks].p[0]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b = lds4(sm.b1 + 32 * t + 8 * g + vec_off);
                    c1[t][4 * g] = relu_bits(c1[t][4 * g] + b.x); c1[t][4 * g + 1] = relu_bits(c1[t][4 * g + 1] + b.y);
                    c1[t][4 * g + 2] = relu_bits(c1[t][4 * g + 2] + b.z); c1[t][4 * g + 3] = relu_bits(c1[t][4 * g + 3] + b.w);
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                T0[wr(r)] = c1[0][r];
                T1[wr(r)] = c1[1][r];
            }
        }

        // ---- F2: H2^T = relu(b2 + W2 H1^T); B operands: the H1^T accumulators split in-lane, eight registers per k-step
        f32x16 c2[2] = {zero16(), zero16()};
        {
            uint4 hb0[4];   // the leading pieces of the four k-steps, for the big terms at the end
#pragma unroll
            for (int s = 0; s < 4; ++s) {   // k-step (t1, j) = (s >> 1, s & 1): registers 8 j .. 8 j + 7 of c1[t1]
                float hv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = c1[s >> 1][8 * (s & 1) + e];
                const Pieces hb = split8(hv);
                hb0[s] = hb.p[0];
                Pieces w0, w1;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    w0.p[i] = ldw(&sm.W2p[0][0][0], 0, s, i);
                    w1.p[i] = ldw(&sm.W2p[0][0][0], 1, s, i);
                }
                mma5_small2(c2[0], c2[1], w0, w1, hb);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                mma1_big(c2[0], ldw(&sm.W2p[0][0][0], 0, s, 0), hb0[s]);
                mma1_big(c2[1], ldw(&sm.W2p[0][0][0], 1, s, 0), hb0[s]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b = lds4(sm.b2 + 32 * t + 8 * g + vec_off);
                    c2[t][4 * g] = relu_bits(c2[t][4 * g] + b.x); c2[t][4 * g + 1] = relu_bits(c2[t][4 * g + 1] + b.y);
                    c2[t][4 * g + 2] = relu_bits(c2[t][4 * g + 2] + b.z); c2[t][4 * g + 3] = relu_bits(c2[t][4 * g + 3] + b.w);
                }
        }

        // ---- output units + loss (every lane: the two halves of a sample hold 32 hidden units each) -- as in the f32 pass
        float g3 = 0.f, g4 = 0.f;
        {
            float z3 = 0.f, z4 = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 w = lds4(sm.w3 + 32 * t + 8 * g + vec_off);
                    z3 = fmaf(c2[t][4 * g], w.x, z3); z3 = fmaf(c2[t][4 * g + 1], w.y, z3);
                    z3 = fmaf(c2[t][4 * g + 2], w.z, z3); z3 = fmaf(c2[t][4 * g + 3], w.w, z3);
                    if (ACTOR) {
                        const float4 v = lds4(sm.w4 + 32 * t + 8 * g + vec_off);
                        z4 = fmaf(c2[t][4 * g], v.x, z4); z4 = fmaf(c2[t][4 * g + 1], v.y, z4);
                        z4 = fmaf(c2[t][4 * g + 2], v.z, z4); z4 = fmaf(c2[t][4 * g + 3], v.w, z4);
                    }
                }
            z3 += __shfl_xor(z3, 32, 64);
            if (ACTOR) z4 += __shfl_xor(z4, 32, 64);
            const float own = (lhi == 0) ? 1.f : 0.f;   // statistics are counted once per sample
            if (valid) {
                z3 += b3;
                z4 += b4;
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
            if (lhi == 0) {
                gs[l31] = g3;
                if (ACTOR) gs[32 + l31] = g4;
            }
        }

        // ---- H1 as the B operand of the dW2 products: lane (n = l31, half lhi) of tile t1, samples 16 s + 8 lhi .. + 7
        Pieces h1b[2][2];
#pragma unroll
        for (int t1 = 0; t1 < 2; ++t1)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float* src = (t1 ? T1 : T0);
                const float4 q0 = lds4(src + xt(l31, 16 * s + 8 * lhi)), q1 = lds4(src + xt(l31, 16 * s + 8 * lhi + 4));
                const float hv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                h1b[t1][s] = split8(hv);
            }

        // ---- per 32-row tile of the second layer: dW3/dW4 from H2^T, dH2^T in place, then its two dW2 quadrants
        Pieces d2b[4];   // dH2^T as the B operand of B2: k-step s = 2 t2 + j
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) TD[wr(r)] = c2[t2][r];
            wave_lds_fence();
#pragma unroll
            for (int q = 0; q < 4; ++q) {   // lane (n2 = 32 t2 + l31, half lhi): sum over its 16 samples
                const float4 h = lds4(TD + xt(l31, 16 * lhi + 4 * q));
                const float4 a = lds4(gs + 16 * lhi + 4 * q);
                adw3[t2] = fmaf(h.x, a.x, adw3[t2]); adw3[t2] = fmaf(h.y, a.y, adw3[t2]);
                adw3[t2] = fmaf(h.z, a.z, adw3[t2]); adw3[t2] = fmaf(h.w, a.w, adw3[t2]);
                if (ACTOR) {
                    const float4 b = lds4(gs + 32 + 16 * lhi + 4 * q);
                    adw4[t2] = fmaf(h.x, b.x, adw4[t2]); adw4[t2] = fmaf(h.y, b.y, adw4[t2]);
                    adw4[t2] = fmaf(h.z, b.z, adw4[t2]); adw4[t2] = fmaf(h.w, b.w, adw4[t2]);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 w = lds4(sm.w3 + 32 * t2 + 8 * g + vec_off);
                const float wv3[4] = {w.x, w.y, w.z, w.w};
                float wv4[4] = {0.f, 0.f, 0.f, 0.f};
                if (ACTOR) {
                    const float4 v = lds4(sm.w4 + 32 * t2 + 8 * g + vec_off);
                    wv4[0] = v.x; wv4[1] = v.y; wv4[2] = v.z; wv4[3] = v.w;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = ACTOR ? fmaf(g3, wv3[j], g4 * wv4[j]) : g3 * wv3[j];
                    c2[t2][4 * g + j] = (c2[t2][4 * g + j] > 0.f) ? d : 0.f;
                }
            }
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r) TD[wr(r)] = c2[t2][r];
            wave_lds_fence();
#pragma unroll
            for (int j = 0; j < 2; ++j) {   // B2's B operand, in-lane
                float dv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) dv[e] = c2[t2][8 * j + e];
                d2b[2 * t2 + j] = split8(dv);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {   // dW2[32 t2 ..][.] += dH2^T H1 over samples 16 s .. 16 s + 15
                const float4 q0 = lds4(TD + xt(l31, 16 * s + 8 * lhi)), q1 = lds4(TD + xt(l31, 16 * s + 8 * lhi + 4));
                const float av[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                adb2[t2] += ((q0.x + q0.y) + (q0.z + q0.w)) + ((q1.x + q1.y) + (q1.z + q1.w));
                const Pieces da = split8(av);
                mma6_acc2(aW2[t2][0], aW2[t2][1], da, h1b[0][s], h1b[1][s]);
            }
            wave_lds_fence();
        }

        // ---- B2: dH1^T = (W2^T dH2^T) . [H1 > 0]
        f32x16 c3[2] = {zero16(), zero16()};
        {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                Pieces w0, w1;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    w0.p[i] = ldw(&sm.W2Tp[0][0][0], 0, s, i);
                    w1.p[i] = ldw(&sm.W2Tp[0][0][0], 1, s, i);
                }
                mma5_small2(c3[0], c3[1], w0, w1, d2b[s]);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                mma1_big(c3[0], ldw(&sm.W2Tp[0][0][0], 0, s, 0), d2b[s].p[0]);
                mma1_big(c3[1], ldw(&sm.W2Tp[0][0][0], 1, s, 0), d2b[s].p[0]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {   // relu mask of layer 1: H1^T is still in T0 / T1 (same lane, same slot it was written from)
                c3[0][r] = (T0[wr(r)] > 0.f) ? c3[0][r] : 0.f;
                c3[1][r] = (T1[wr(r)] > 0.f) ? c3[1][r] : 0.f;
            }
        }
        wave_lds_fence();

        // ---- G1: dW1 += dH1^T X over the tile's 32 samples (16 x 16 x 32: one k-step); X^T pieces from the pre-split buffer
        Pieces xb[KX];   // lane (f = l15, kb = kk): samples 8 kk .. 8 kk + 7 of feature 16 c + f
        {
            const unsigned char* t = prep + (size_t)tile * kX3TileBytes + kX3RowsBytes + (l15 * 32 + 8 * kk) * 2;
#pragma unroll
            for (int c = 0; c < KX; ++c)
#pragma unroll
                for (int i = 0; i < 3; ++i) xb[c].p[i] = *reinterpret_cast<const uint4*>(t + (i * INP + 16 * c) * 32 * 2);
        }
#pragma unroll
        for (int t1 = 0; t1 < 2; ++t1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) TD[wr(r)] = c3[t1][r];
            wave_lds_fence();
            Pieces da[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {   // rows 32 t1 + 16 u + l15, samples 8 kk .. 8 kk + 7
                const float4 q0 = lds4(TD + xt(16 * u + l15, 8 * kk)), q1 = lds4(TD + xt(16 * u + l15, 8 * kk + 4));
                const float dv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                adb1[2 * t1 + u] += ((q0.x + q0.y) + (q0.z + q0.w)) + ((q1.x + q1.y) + (q1.z + q1.w));
                da[u] = split8(dv);
            }
#pragma unroll
            for (int c = 0; c < KX; ++c) mma6_acc16_2(aW1[c][2 * t1], aW1[c][2 * t1 + 1], da[0], da[1], xb[c]);
            wave_lds_fence();
        }
    }

    // ---- workgroup reduction of the 8 waves' partial gradients in LDS, then one coalesced row of `partial` (as the f32 pass)
    __syncthreads();
    constexpr int RP = (P + 3 + 3) & ~3;
    float s3 = adb3, s4 = adb4, sA = ACTOR ? st0 : st1, sB = st2, sC = st3;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s3 += __shfl_xor(s3, o, 64); s4 += __shfl_xor(s4, o, 64);
        sA += __shfl_xor(sA, o, 64); sB += __shfl_xor(sB, o, 64); sC += __shfl_xor(sC, o, 64);
    }
    float hb2[2], hw3[2], hw4[2], qb1[4];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
        hb2[t2] = adb2[t2] + __shfl_xor(adb2[t2], 32, 64);
        hw3[t2] = adw3[t2] + __shfl_xor(adw3[t2], 32, 64);
        hw4[t2] = adw4[t2] + __shfl_xor(adw4[t2], 32, 64);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float v = adb1[u];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        qb1[u] = v;
    }
    float* const red0 = (IN == 16) ? sm.wv : reinterpret_cast<float*>(&sm);   // (42 columns: the rows start at the dead weight pieces)
    float* const row = red0 + (wave & 3) * RP;
#pragma unroll
    for (int pass = 0; pass < kXWaves / 4; ++pass) {
        if ((wave >> 2) == pass) {
            const bool add = pass >= 1;
            auto put = [&](int idx, float v) { row[idx] = add ? row[idx] + v : v; };
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
                for (int t1 = 0; t1 < 2; ++t1)
#pragma unroll
                    for (int r = 0; r < 16; ++r) put(OFF_W2 + (32 * t2 + c_row(r, lane)) * H + 32 * t1 + l31, aW2[t2][t1][r]);
                if (lhi == 0) {
                    put(OFF_B2 + 32 * t2 + l31, hb2[t2]);
                    put(OFF_W3 + 32 * t2 + l31, hw3[t2]);
                    if (ACTOR) put(OFF_W4 + 32 * t2 + l31, hw4[t2]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int c = 0; c < KX; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * c + l15 < IN) put(OFF_W1 + (16 * u + 4 * kk + r) * IN + 16 * c + l15, aW1[c][u][r]);
                if (kk == 0) put(OFF_B1 + 16 * u + l15, qb1[u]);
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
    }
    float* out = partial + (size_t)blockIdx.x * P;
    const float* red = red0;
    for (int k = tid; k < P; k += NT) out[k] = (red[k] + red[RP + k]) + (red[2 * RP + k] + red[3 * RP + k]);
    if (tid < 3) stats_partial[blockIdx.x * 4 + tid] = (red[P + tid] + red[RP + P + tid]) + (red[2 * RP + P + tid] + red[3 * RP + P + tid]);
}

// both nets of one epoch in one launch; net_mask: bit 0 = actor, bit 1 = critic (the one-net launches of the multi-GPU pipeline)
template <int IN>
__global__ __launch_bounds__(64 * XPad<IN>::NW) void mlp64_pass_both_x3(const float* __restrict__ params, const unsigned char* __restrict__ prep,
                                                                const float* __restrict__ act, const float* __restrict__ logp_old,
                                                                const float* __restrict__ rtg, const float* __restrict__ adv,
                                                                long long M, float var, float clip, float inv_n, int net_mask,
                                                                float* __restrict__ partial_a, float* __restrict__ stats_partial_a,
                                                                float* __restrict__ partial_c, float* __restrict__ stats_partial_c) {
    __shared__ __attribute__((aligned(16))) SmemX<IN> sm;
    if (net_mask & 1) pass_body_x3<true, IN>(sm, params, prep, act, logp_old, rtg, adv, M, var, clip, inv_n, partial_a, stats_partial_a);
    if (net_mask == 3) __syncthreads();
    if (net_mask & 2) pass_body_x3<false, IN>(sm, params + Layout<IN>::P_ACTOR, prep, act, logp_old, rtg, adv, M, var, clip, inv_n, partial_c, stats_partial_c);
}

// round 6: the 16-column split pass as a hand-placed stream (4 waves x 512 registers, pieces transposed through LDS); the kernel above
// stays for the 42-column rows and, under -DX3_SCHED=0, as the 16-column A/B partner (tools/build_mlp64_variant.py)
#ifndef X3_SCHED
#define X3_SCHED 1
#endif
#include "ppo_mlp64_x3s.h"

// grad[p] = sum over the workgroups' partial rows, for BOTH nets; ADAM (single-GPU epoch): then torch.optim.Adam's update (ppo.py:116-117,381,392; defaults betas
// (0.9, 0.999), eps 1e-8, no weight decay) applied in place -- one launch instead of two reductions + an optimiser launch.
// One block owns 64 parameters and ALL rows (no atomics): grad[p] is stored, not accumulated.  pa / pc: parameters of the actor /
// the critic (Layout<IN>), the pitch of their partial rows.
constexpr int kRedGroups = 16;   // row groups per block: 1024 threads, every thread sums n_blocks / 16 rows, 4 loads in flight
constexpr int kGnSlots = 256;    // per (parity, net): one slot per block of reduce_adam (<= 235 blocks at 42 columns)
template <bool ADAM>
__global__ __launch_bounds__(64 * kRedGroups) void reduce_adam(const float* __restrict__ partial_a, const float* __restrict__ stats_partial_a,
                                                   const float* __restrict__ partial_c, const float* __restrict__ stats_partial_c,
                                                   int n_blocks, float inv_n, float* __restrict__ grad, float* __restrict__ stats,
                                                   float* __restrict__ params, float* __restrict__ m, float* __restrict__ v,
                                                   float lr, float beta1, float beta2, float eps, float bc1, float bc2_sqrt,
                                                   int pa, int pc, int q_begin, int q_end, float* __restrict__ gn, int parity) {
    // gn / parity (whole-update epochs only; parity < 0: off): the squared gradient norms per net, for the per-epoch MEANS the reference
    // logs (ppo.py:351-352, 389-390: clip_grad_norm_(inf) of each net in every epoch, averaged) without a norm launch per epoch and
    // without atomics: every block leaves the sums of its 64 squared gradients (actor part, critic part) in slot `parity`, and block 0
    // adds up the slots of the OTHER parity -- the epoch before -- into stats[3] (actor) / stats[7] (critic).  The last epoch's norms
    // are taken from grad_dev by the caller.
    __shared__ float part[kRedGroups][64];
    const int lane = threadIdx.x & 63, q = q_begin + blockIdx.x * 64 + lane, g = threadIdx.x >> 6;   // q: index into actor | critic
    const bool actor = q < pa;
    const float* __restrict__ partial = actor ? partial_a : partial_c;
    const int P = actor ? pa : pc, p = actor ? q : q - pa;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (q < q_end) {
        int b = g;
        for (; b + 3 * kRedGroups < n_blocks; b += 4 * kRedGroups) {   // the row reads of a (b, lane) pair are 256 B per wave
            s0 += partial[(size_t)b * P + p];
            s1 += partial[(size_t)(b + kRedGroups) * P + p];
            s2 += partial[(size_t)(b + 2 * kRedGroups) * P + p];
            s3 += partial[(size_t)(b + 3 * kRedGroups) * P + p];
        }
        for (; b < n_blocks; b += kRedGroups) s0 += partial[(size_t)b * P + p];
    }
    part[g][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && q < q_end) {
        float gr = 0.f;
#pragma unroll
        for (int k = 0; k < kRedGroups; ++k) gr += part[k][lane];
        grad[q] = gr;
        if (ADAM) {
            const float mm = m[q] + (gr - m[q]) * (1.0f - beta1);          // exp_avg.lerp_(grad, 1 - beta1)
            const float vv = beta2 * v[q] + (1.0f - beta2) * (gr * gr);    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
            m[q] = mm;
            v[q] = vv;
            const float denom = sqrtf(vv) / bc2_sqrt + eps;
            params[q] -= (lr / bc1) * (mm / denom);
        }
    }
    if (parity >= 0 && g == 0) {
        const float gr2 = q < q_end ? grad[q] * grad[q] : 0.f;   // (this lane's own store above)
        float sa = actor ? gr2 : 0.f, sc = actor ? 0.f : gr2;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            sa += __shfl_xor(sa, o, 64);
            sc += __shfl_xor(sc, o, 64);
        }
        if (lane == 0) {
            gn[(parity * 2 + 0) * kGnSlots + blockIdx.x] = sa;
            gn[(parity * 2 + 1) * kGnSlots + blockIdx.x] = sc;
        }
    }
    if (parity >= 0 && blockIdx.x == 0 && (g == 3 || g == 7)) {
        const float* sp = gn + ((1 - parity) * 2 + (g >> 2)) * kGnSlots;
        float s = 0.f;
        for (int b = lane; b < (int)gridDim.x; b += 64) s += sp[b];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) stats[g] = s;
    }
    const bool net_here = (g < 4) ? (q_begin == 0) : (q_end > pa);   // a one-net launch leaves the other net's statistics alone
    if (blockIdx.x == 0 && g < 8 && (g & 3) < 3 && net_here) {   // stats[0..2] actor, stats[4..6] critic: wave g sums statistic g over the rows
        const float* sp = (g < 4) ? stats_partial_a : stats_partial_c;
        float s = 0.f;
        for (int b = lane; b < n_blocks; b += 64) s += sp[b * 4 + (g & 3)];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) stats[g] = s * inv_n;
    }
}

// torch.optim.Adam's update on a flat buffer, gradient pre-scaled (multi-GPU: grad = all-reduced sum x 1 / world)
__global__ void adam_step_kernel(float* __restrict__ params, const float* __restrict__ grad, float* __restrict__ m,
                                 float* __restrict__ v, int n, float grad_scale, float lr, float beta1, float beta2, float eps,
                                 float bc1, float bc2_sqrt) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const float gr = grad[q] * grad_scale;
    const float mm = m[q] + (gr - m[q]) * (1.0f - beta1);
    const float vv = beta2 * v[q] + (1.0f - beta2) * (gr * gr);
    m[q] = mm;
    v[q] = vv;
    params[q] -= (lr / bc1) * (mm / (sqrtf(vv) / bc2_sqrt + eps));
}

// ---------------------------------------------------------------- rollout-time policy step (PPO.get_action, ppo.py:673-706)
// PPO.get_action for all envs, one launch: one wave = 16 envs, the policy step itself is mlp64_policy.h (shared with the
// persistent rollout kernel of navsim.hip so that both produce the same bits)
template <int IN, bool F16>
__global__ __launch_bounds__(64) void mlp64_act(const float* __restrict__ params, const void* __restrict__ obs,
                                                const float* __restrict__ noise, long long n,
                                                const float* __restrict__ var_ptr, uint64_t seed, uint64_t env_id_base,
                                                const uint32_t* __restrict__ step_base, uint32_t step_offset,
                                                float* __restrict__ act, float* __restrict__ logp,
                                                float* __restrict__ mean_out) {
    using L = Layout<IN>;
    const int lane = threadIdx.x, l15 = lane & 15, kk = lane >> 4;
    const long long m = (long long)blockIdx.x * kActEnvs + l15;
    const bool valid = m < n;
    float xs[L::KS];
    if constexpr (IN == 16 && !F16) {
        const float4 xq = valid ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(obs) + m * IN + 4 * kk)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        xs[0] = xq.x; xs[1] = xq.y; xs[2] = xq.z; xs[3] = xq.w;
    } else {
#pragma unroll
        for (int s = 0; s < L::KS; ++s) {
            const int f = L::KS * kk + s;
            xs[s] = (valid && f < IN) ? obs_at<F16>(obs, m * IN + min(f, IN - 1)) : 0.f;
        }
    }
    const uint32_t step = (step_base ? *step_base : 0u) + step_offset;
    const PolicyOut o = policy_wave16<L>(params, xs, lane, *var_ptr, (noise && valid) ? noise + 2 * m : nullptr, step, seed,
                                         env_id_base + (uint64_t)m);
    if (kk == 0 && valid) {
        act[2 * m] = o.a0;
        act[2 * m + 1] = o.a1;
        logp[m] = o.logp;
        if (mean_out) {
            mean_out[2 * m] = o.mu0;
            mean_out[2 * m + 1] = o.mu1;
        }
    }
}

// ---------------------------------------------------------------- episode sums of one rollout (the iteration's logging line)
// ppo.py:552-560,833: episodes, successes (arrive), collisions (done and not arrive), timeouts (neither), the sum of episode
// lengths and the sum of episode returns over the [T, N] buffers a rollout fills.  One pass over 22 MB instead of a dozen
// PyTorch reductions; per-block partials summed in a fixed order, so the result does not depend on the schedule.
constexpr int kSumBlocks = 256, kSumThreads = 256;

template <bool VEC>   // VEC: four entries per thread and load (n % 4 == 0, 16-byte aligned buffers)
__global__ __launch_bounds__(kSumThreads) void episode_sums_partial(const uint8_t* __restrict__ ended, const uint8_t* __restrict__ arrive,
                                                                     const uint8_t* __restrict__ done, const int32_t* __restrict__ ep_len,
                                                                     const float* __restrict__ ep_ret, long long n,
                                                                     double* __restrict__ partial) {
    __shared__ double red[kSumThreads / 64][6];
    unsigned c_ep = 0, c_ok = 0, c_hit = 0, c_tmo = 0;
    long long len = 0;
    double ret = 0;
    auto one = [&](unsigned eb, unsigned ab, unsigned db, int l, float r) {
        const bool e = eb != 0, a = (ab != 0) && e, d = (db != 0) && e;
        c_ep += e;
        c_ok += a;
        c_hit += d && !a;
        c_tmo += e && !d && !a;
        len += l;
        ret += e ? (double)r : 0.0;
    };
    const long long stride = (long long)gridDim.x * kSumThreads;
    if (VEC) {
        for (long long q = (long long)blockIdx.x * kSumThreads + threadIdx.x; q < n / 4; q += stride) {
            const unsigned e4 = reinterpret_cast<const unsigned*>(ended)[q], a4 = reinterpret_cast<const unsigned*>(arrive)[q],
                           d4 = reinterpret_cast<const unsigned*>(done)[q];
            const int4 l4 = reinterpret_cast<const int4*>(ep_len)[q];
            const float4 r4 = reinterpret_cast<const float4*>(ep_ret)[q];
            one(e4 & 0xffu, a4 & 0xffu, d4 & 0xffu, l4.x, r4.x);
            one((e4 >> 8) & 0xffu, (a4 >> 8) & 0xffu, (d4 >> 8) & 0xffu, l4.y, r4.y);
            one((e4 >> 16) & 0xffu, (a4 >> 16) & 0xffu, (d4 >> 16) & 0xffu, l4.z, r4.z);
            one(e4 >> 24, a4 >> 24, d4 >> 24, l4.w, r4.w);
        }
    } else {
        for (long long k = (long long)blockIdx.x * kSumThreads + threadIdx.x; k < n; k += stride)
            one(ended[k], arrive[k], done[k], ep_len[k], ep_ret[k]);
    }
    double v[6] = {(double)c_ep, (double)c_ok, (double)c_hit, (double)c_tmo, (double)len, ret};
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v[j] += __shfl_xor(v[j], m, 64);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (lane == 0)
#pragma unroll
        for (int j = 0; j < 6; ++j) red[wave][j] = v[j];
    __syncthreads();
    if (threadIdx.x < 6) {
        double t = 0;
        for (int w = 0; w < kSumThreads / 64; ++w) t += red[w][threadIdx.x];
        partial[(size_t)blockIdx.x * 6 + threadIdx.x] = t;
    }
}

// the partial rows of up to kSumBlocks workgroups, pairwise in a fixed tree
__global__ __launch_bounds__(kSumBlocks) void episode_sums_final(const double* __restrict__ partial, int blocks, double* __restrict__ out) {
    __shared__ double red[kSumBlocks][6];
    const int b = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 6; ++j) red[b][j] = (b < blocks) ? partial[(size_t)b * 6 + j] : 0.0;
    __syncthreads();
    for (int h = kSumBlocks / 2; h >= 1; h >>= 1) {
        if (b < h)
#pragma unroll
            for (int j = 0; j < 6; ++j) red[b][j] += red[b + h][j];
        __syncthreads();
    }
    if (b < 6) out[b] = red[0][b];
}

thread_local std::string g_err;

}  // namespace

void navppo_set_error(const char* msg) { g_err = msg ? msg : ""; }

#pragma GCC visibility push(default)
extern "C" {

const char* navppo_last_error(void) { return g_err.c_str(); }

size_t navppo_mlp64_workspace_bytes(int32_t obs_dim) {
    if (obs_dim != 16 && obs_dim != 42) return 0;
    return (size_t)NAVPPO_MLP64_MAX_BLOCKS * (NAVPPO_MLP64_ACTOR_PARAMS_D(obs_dim) + 4) * sizeof(float);
}

}  // extern "C"
#pragma GCC visibility pop

namespace {

// the instantiation for (observation width, row type): f(std::integral_constant<int, IN>, std::integral_constant<bool, F16>)
template <class F>
bool for_obs(int32_t obs_dim, int32_t obs_f16, F f) {
    if (obs_dim == 16) {
        if (obs_f16) f(std::integral_constant<int, 16>{}, std::true_type{});
        else f(std::integral_constant<int, 16>{}, std::false_type{});
        return true;
    }
    if (obs_dim == 42) {
        if (obs_f16) f(std::integral_constant<int, 42>{}, std::true_type{});
        else f(std::integral_constant<int, 42>{}, std::false_type{});
        return true;
    }
    return false;
}

// rows of `obs` are read with 16-byte (16 columns), 8-byte (42 float32 columns) or 4-byte (42 float16 columns) loads
bool obs_aligned(const void* obs, int32_t obs_dim, int32_t obs_f16) {
    const uintptr_t mask = obs_dim == 16 ? 15 : (obs_f16 ? 3 : 7);
    return ((uintptr_t)obs & mask) == 0;
}

struct PassPlan {
    float *partial, *stats_partial, *partial_c, *stats_partial_c, *gn;
    float inv_n;
    int blocks, pa, pc;
};
PassPlan plan_pass(void* workspace_dev, int64_t n_samples, int32_t obs_dim) {
    PassPlan pl;
    pl.pa = NAVPPO_MLP64_ACTOR_PARAMS_D(obs_dim);
    pl.pc = NAVPPO_MLP64_CRITIC_PARAMS_D(obs_dim);
    pl.partial = reinterpret_cast<float*>(workspace_dev);
    pl.stats_partial = pl.partial + (size_t)NAVPPO_MLP64_MAX_BLOCKS * pl.pa;
    pl.partial_c = pl.partial + (size_t)kWMaxBlocks * pl.pa;   // the workspace has NAVPPO_MLP64_MAX_BLOCKS = 2 kWMaxBlocks rows
    pl.stats_partial_c = pl.stats_partial + (size_t)kWMaxBlocks * 4;
    // the critic's rows are shorter than the actor's: the tail of its half of the row area is free -- the squared-norm slots of reduce_adam
    pl.gn = pl.partial_c + (size_t)kWMaxBlocks * pl.pc;
    static_assert((Layout<42>::P_ACTOR + Layout<42>::P_CRITIC + 63) / 64 <= kGnSlots && 4 * kGnSlots <= kWMaxBlocks * (Layout<16>::P_ACTOR - Layout<16>::P_CRITIC),
                  "the squared-norm slots fit the free tail of the critic's rows");
    pl.inv_n = 1.0f / (float)n_samples;
    const long long wtiles = (n_samples + 31) / 32;
    const long long want = (wtiles + kWWaves - 1) / kWWaves;
    pl.blocks = (int)(want < kWMaxBlocks ? want : kWMaxBlocks);
    return pl;
}

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

int navppo_mlp64_loss_grad(const float* params_dev, const void* obs_dev, int32_t obs_dim, int32_t obs_f16, const float* act_dev,
                           const float* logp_old_dev, const float* rtg_dev, const float* adv_dev, int64_t n_samples,
                           float var, float clip, float* grad_dev, float* stats_dev, void* workspace_dev, void* stream) {
    if (!params_dev || !obs_dev || !act_dev || !logp_old_dev || !rtg_dev || !adv_dev || !grad_dev || !stats_dev ||
        !workspace_dev || n_samples < 1 || !(var > 0.f) || (obs_dim != 16 && obs_dim != 42)) {
        g_err = "navppo_mlp64_loss_grad: bad argument (obs_dim is 16 or 42)";
        return -1;
    }
    if (!obs_aligned(obs_dev, obs_dim, obs_f16) || ((uintptr_t)act_dev & 7)) {
        g_err = "navppo_mlp64_loss_grad: obs must be 16-byte (42 columns: 8-byte, float16: 4-byte) and act 8-byte aligned";
        return -1;
    }
    hipStream_t st = (hipStream_t)stream;
    const PassPlan pl = plan_pass(workspace_dev, n_samples, obs_dim);
    for_obs(obs_dim, obs_f16, [&](auto in, auto f16) {
        hipLaunchKernelGGL((mlp64_pass_both<decltype(in)::value, decltype(f16)::value>), dim3(pl.blocks), dim3(64 * Pad<decltype(in)::value>::NW), 0, st, params_dev,
                           obs_dev, act_dev, logp_old_dev, rtg_dev, adv_dev, (long long)n_samples, var, clip, pl.inv_n, pl.partial,
                           pl.stats_partial, pl.partial_c, pl.stats_partial_c, grad_dev, stats_dev);
    });
    hipLaunchKernelGGL(reduce_adam<false>, dim3((pl.pa + pl.pc + 63) / 64), dim3(64 * kRedGroups), 0, st, pl.partial, pl.stats_partial,
                       pl.partial_c, pl.stats_partial_c, pl.blocks, pl.inv_n, grad_dev, stats_dev, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f,
                       0.f, 1.f, 1.f, pl.pa, pl.pc, 0, pl.pa + pl.pc, (float*)nullptr, -1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_err = std::string("navppo_mlp64_loss_grad: ") + hipGetErrorString(e);
        return -2;
    }
    return 0;
}

int navppo_mlp64_loss_grad_net(int32_t net, const float* params_dev, const void* obs_dev, int32_t obs_dim, int32_t obs_f16,
                               const float* act_dev, const float* logp_old_dev, const float* rtg_dev, const float* adv_dev,
                               int64_t n_samples, float var, float clip, float* grad_dev, float* stats_dev, void* workspace_dev,
                               void* stream) {
    if ((net != 0 && net != 1) || !params_dev || !obs_dev || !act_dev || !logp_old_dev || !rtg_dev || !adv_dev || !grad_dev ||
        !stats_dev || !workspace_dev || n_samples < 1 || !(var > 0.f) || (obs_dim != 16 && obs_dim != 42)) {
        g_err = "navppo_mlp64_loss_grad_net: bad argument (obs_dim is 16 or 42)";
        return -1;
    }
    if (!obs_aligned(obs_dev, obs_dim, obs_f16) || ((uintptr_t)act_dev & 7)) {
        g_err = "navppo_mlp64_loss_grad_net: obs must be 16-byte (42 columns: 8-byte, float16: 4-byte) and act 8-byte aligned";
        return -1;
    }
    hipStream_t st = (hipStream_t)stream;
    const PassPlan pl = plan_pass(workspace_dev, n_samples, obs_dim);
    for_obs(obs_dim, obs_f16, [&](auto in, auto f16) {
        constexpr int IN = decltype(in)::value;
        constexpr bool F16 = decltype(f16)::value;
        if (net == 0)
            hipLaunchKernelGGL((mlp64_pass_w<true, false, IN, F16>), dim3(pl.blocks), dim3(64 * Pad<IN>::NW), 0, st, params_dev, obs_dev, act_dev,
                               logp_old_dev, rtg_dev, adv_dev, (long long)n_samples, var, clip, pl.inv_n, pl.partial, pl.stats_partial,
                               grad_dev, stats_dev, (float*)nullptr);
        else
            hipLaunchKernelGGL((mlp64_pass_w<false, false, IN, F16>), dim3(pl.blocks), dim3(64 * Pad<IN>::NW), 0, st, params_dev + pl.pa, obs_dev,
                               act_dev, logp_old_dev, rtg_dev, adv_dev, (long long)n_samples, var, clip, pl.inv_n, pl.partial_c,
                               pl.stats_partial_c, grad_dev + pl.pa, stats_dev + 4, (float*)nullptr);
    });
    const int q0 = net == 0 ? 0 : pl.pa, q1 = net == 0 ? pl.pa : pl.pa + pl.pc;
    hipLaunchKernelGGL(reduce_adam<false>, dim3((q1 - q0 + 63) / 64), dim3(64 * kRedGroups), 0, st, pl.partial, pl.stats_partial, pl.partial_c,
                       pl.stats_partial_c, pl.blocks, pl.inv_n, grad_dev, stats_dev, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f, 1.f, 1.f,
                       pl.pa, pl.pc, q0, q1, (float*)nullptr, -1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_err = std::string("navppo_mlp64_loss_grad_net: ") + hipGetErrorString(e);
        return -2;
    }
    return 0;
}

int navppo_adam_step(float* params_dev, const float* grad_dev, float* adam_m_dev, float* adam_v_dev, int64_t n, float grad_scale,
                     float lr, float beta1, float beta2, float eps, int32_t step, void* stream) {
    if (!params_dev || !grad_dev || !adam_m_dev || !adam_v_dev || n < 1 || step < 1) {
        g_err = "navppo_adam_step: bad argument";
        return -1;
    }
    const float bc1 = (float)(1.0 - std::pow((double)beta1, (double)step));
    const float bc2_sqrt = (float)std::sqrt(1.0 - std::pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params_dev, grad_dev,
                       adam_m_dev, adam_v_dev, (int)n, grad_scale, lr, beta1, beta2, eps, bc1, bc2_sqrt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_err = std::string("navppo_adam_step: ") + hipGetErrorString(e);
        return -2;
    }
    return 0;
}

int navppo_mlp64_value(const float* critic_params_dev, const void* obs_dev, int32_t obs_dim, int32_t obs_f16, int64_t n_samples,
                       float* value_dev, void* stream) {
    if (!critic_params_dev || !obs_dev || !value_dev || n_samples < 1 || (obs_dim != 16 && obs_dim != 42) ||
        !obs_aligned(obs_dev, obs_dim, obs_f16)) {
        g_err = "navppo_mlp64_value: bad argument (obs_dim is 16 or 42; obs 16-byte aligned, 42 columns: 8-byte, float16: 4-byte)";
        return -1;
    }
    const long long wtiles = (n_samples + 31) / 32;
    const long long want = (wtiles + kWWaves - 1) / kWWaves;
    const int blocks = (int)(want < kWMaxBlocks ? want : kWMaxBlocks);
    for_obs(obs_dim, obs_f16, [&](auto in, auto f16) {
        hipLaunchKernelGGL((mlp64_pass_w<false, true, decltype(in)::value, decltype(f16)::value>), dim3(blocks), dim3(64 * Pad<decltype(in)::value>::NW), 0,
                           (hipStream_t)stream, critic_params_dev, obs_dev, nullptr, nullptr, nullptr, nullptr, (long long)n_samples, 1.f,
                           0.f, 0.f, nullptr, nullptr, nullptr, nullptr, value_dev);
    });
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_err = std::string("navppo_mlp64_value: ") + hipGetErrorString(e);
        return -2;
    }
    return 0;
}

int navppo_mlp64_update_epoch(float* params_dev, const void* obs_dev, int32_t obs_dim, int32_t obs_f16, const float* act_dev,
                              const float* logp_old_dev, const float* rtg_dev, const float* adv_dev, int64_t n_samples, float var,
                              float clip, float lr, float beta1, float beta2, float eps, int32_t step, float* adam_m_dev,
                              float* adam_v_dev, float* grad_dev, float* stats_dev, void* workspace_dev, void* stream) {
    if (!params_dev || !obs_dev || !act_dev || !logp_old_dev || !rtg_dev || !adv_dev || !grad_dev || !stats_dev ||
        !workspace_dev || !adam_m_dev || !adam_v_dev || n_samples < 1 || !(var > 0.f) || step < 1 || (obs_dim != 16 && obs_dim != 42)) {
        g_err = "navppo_mlp64_update_epoch: bad argument (obs_dim is 16 or 42)";
        return -1;
    }
    if (!obs_aligned(obs_dev, obs_dim, obs_f16) || ((uintptr_t)act_dev & 7)) {
        g_err = "navppo_mlp64_update_epoch: obs must be 16-byte (42 columns: 8-byte, float16: 4-byte) and act 8-byte aligned";
        return -1;
    }
    hipStream_t st = (hipStream_t)stream;
    const PassPlan pl = plan_pass(workspace_dev, n_samples, obs_dim);
    const float bc1 = (float)(1.0 - std::pow((double)beta1, (double)step));
    const float bc2_sqrt = (float)std::sqrt(1.0 - std::pow((double)beta2, (double)step));
    for_obs(obs_dim, obs_f16, [&](auto in, auto f16) {
        hipLaunchKernelGGL((mlp64_pass_both<decltype(in)::value, decltype(f16)::value>), dim3(pl.blocks), dim3(64 * Pad<decltype(in)::value>::NW), 0, st, params_dev,
                           obs_dev, act_dev, logp_old_dev, rtg_dev, adv_dev, (long long)n_samples, var, clip, pl.inv_n, pl.partial,
                           pl.stats_partial, pl.partial_c, pl.stats_partial_c, grad_dev, stats_dev);
    });
    hipLaunchKernelGGL(reduce_adam<true>, dim3((pl.pa + pl.pc + 63) / 64), dim3(64 * kRedGroups), 0, st, pl.partial, pl.stats_partial,
                       pl.partial_c, pl.stats_partial_c, pl.blocks, pl.inv_n, grad_dev, stats_dev, params_dev, adam_m_dev, adam_v_dev, lr,
                       beta1, beta2, eps, bc1, bc2_sqrt, pl.pa, pl.pc, 0, pl.pa + pl.pc, pl.gn, (int)(step & 1));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_err = std::string("navppo_mlp64_update_epoch: ") + hipGetErrorString(e);
        return -2;
    }
    return 0;
}

size_t navppo_mlp64_bf16x3_prep_bytes(int64_t n_samples, int32_t obs_dim) {
    if (n_samples < 1 || (obs_dim != 16 && obs_dim != 42)) return 0;
    return (size_t)((n_samples + 31) / 32) * (obs_dim == 16 ? XPad<16>::kTileBytes : XPad<42>::kTileBytes);
}

int navppo_mlp64_bf16x3_prepare(const void* obs_dev, int32_t obs_dim, int32_t obs_f16, int64_t n_samples, void* prep_dev, void* stream) {
    if (!obs_dev || !prep_dev || n_samples < 1 || (obs_dim != 16 && obs_dim != 42) || !obs_aligned(obs_dev, obs_dim, obs_f16) ||
        ((uintptr_t)prep_dev & 15)) {
        g_err = "navppo_mlp64_bf16x3_prepare: bad argument (obs_dim is 16 or 42; obs aligned as for navppo_mlp64_loss_grad, prep 16-byte aligned)";
        return -1;
    }
    const unsigned tiles = (unsigned)((n_samples + 31) / 32);
    for_obs(obs_dim, obs_f16, [&](auto in, auto f16) {
        hipLaunchKernelGGL((mlp64_split_obs<decltype(f16)::value, decltype(in)::value>), dim3(tiles), dim3(64), 0, (hipStream_t)stream, obs_dev,
                           (long long)n_samples, reinterpret_cast<unsigned char*>(prep_dev));
    });
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_err = std::string("navppo_mlp64_bf16x3_prepare: ") + hipGetErrorString(e);
        return -2;
    }
    return 0;
}

// net_mask 3: both nets; 1 / 2: the actor's / the critic's pass and its slice of the reduction; step >= 1: Adam in the reduction
static int x3_epoch(const char* who, float* params_dev, const void* prep_dev, int32_t obs_dim, const float* act_dev, const float* logp_old_dev,
                    const float* rtg_dev, const float* adv_dev, int64_t n_samples, float var, float clip, int net_mask, int32_t step,
                    float lr, float beta1, float beta2, float eps, float* adam_m_dev, float* adam_v_dev, float* grad_dev,
                    float* stats_dev, void* workspace_dev, void* stream) {
    if (!params_dev || !prep_dev || !act_dev || !logp_old_dev || !rtg_dev || !adv_dev || !grad_dev || !stats_dev || !workspace_dev ||
        n_samples < 1 || !(var > 0.f) || (obs_dim != 16 && obs_dim != 42) || ((uintptr_t)prep_dev & 15) || ((uintptr_t)act_dev & 7)) {
        g_err = std::string(who) + ": bad argument (obs_dim is 16 or 42; prep 16-byte, act 8-byte aligned)";
        return -1;
    }
    hipStream_t st = (hipStream_t)stream;
    const PassPlan pl = plan_pass(workspace_dev, n_samples, obs_dim);
    if (obs_dim == 16 && X3_SCHED)
        hipLaunchKernelGGL(mlp64_pass_both_x3s, dim3(pl.blocks), dim3(64 * x3s::SW), 0, st, params_dev,
                           reinterpret_cast<const unsigned char*>(prep_dev), act_dev, logp_old_dev, rtg_dev, adv_dev, (long long)n_samples, var, clip,
                           pl.inv_n, net_mask, pl.partial, pl.stats_partial, pl.partial_c, pl.stats_partial_c);
    else if (obs_dim == 16)
        hipLaunchKernelGGL(mlp64_pass_both_x3<16>, dim3(pl.blocks), dim3(64 * XPad<16>::NW), 0, st, params_dev,
                           reinterpret_cast<const unsigned char*>(prep_dev), act_dev, logp_old_dev, rtg_dev, adv_dev, (long long)n_samples, var, clip,
                           pl.inv_n, net_mask, pl.partial, pl.stats_partial, pl.partial_c, pl.stats_partial_c);
    else
        hipLaunchKernelGGL(mlp64_pass_both_x3<42>, dim3(pl.blocks), dim3(64 * XPad<42>::NW), 0, st, params_dev,
                           reinterpret_cast<const unsigned char*>(prep_dev), act_dev, logp_old_dev, rtg_dev, adv_dev, (long long)n_samples, var, clip,
                           pl.inv_n, net_mask, pl.partial, pl.stats_partial, pl.partial_c, pl.stats_partial_c);
    const int q0 = (net_mask & 1) ? 0 : pl.pa, q1 = (net_mask & 2) ? pl.pa + pl.pc : pl.pa;
    if (step >= 1) {
        const float bc1 = (float)(1.0 - std::pow((double)beta1, (double)step));
        const float bc2_sqrt = (float)std::sqrt(1.0 - std::pow((double)beta2, (double)step));
        hipLaunchKernelGGL(reduce_adam<true>, dim3((q1 - q0 + 63) / 64), dim3(64 * kRedGroups), 0, st, pl.partial, pl.stats_partial, pl.partial_c,
                           pl.stats_partial_c, pl.blocks, pl.inv_n, grad_dev, stats_dev, params_dev, adam_m_dev, adam_v_dev, lr, beta1, beta2,
                           eps, bc1, bc2_sqrt, pl.pa, pl.pc, q0, q1, pl.gn, net_mask == 3 ? (int)(step & 1) : -1);
    } else {
        hipLaunchKernelGGL(reduce_adam<false>, dim3((q1 - q0 + 63) / 64), dim3(64 * kRedGroups), 0, st, pl.partial, pl.stats_partial, pl.partial_c,
                           pl.stats_partial_c, pl.blocks, pl.inv_n, grad_dev, stats_dev, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f, 1.f,
                           1.f, pl.pa, pl.pc, q0, q1, (float*)nullptr, -1);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_err = std::string(who) + ": " + hipGetErrorString(e);
        return -2;
    }
    return 0;
}

int navppo_mlp64_bf16x3_loss_grad(const float* params_dev, const void* prep_dev, int32_t obs_dim, const float* act_dev,
                                  const float* logp_old_dev, const float* rtg_dev, const float* adv_dev, int64_t n_samples, float var, float clip,
                                  float* grad_dev, float* stats_dev, void* workspace_dev, void* stream) {
    return x3_epoch("navppo_mlp64_bf16x3_loss_grad", const_cast<float*>(params_dev), prep_dev, obs_dim, act_dev, logp_old_dev, rtg_dev, adv_dev,
                    n_samples, var, clip, 3, 0, 0.f, 0.f, 0.f, 0.f, nullptr, nullptr, grad_dev, stats_dev, workspace_dev, stream);
}

int navppo_mlp64_bf16x3_loss_grad_net(int32_t net, const float* params_dev, const void* prep_dev, int32_t obs_dim, const float* act_dev,
                                      const float* logp_old_dev, const float* rtg_dev, const float* adv_dev, int64_t n_samples, float var,
                                      float clip, float* grad_dev, float* stats_dev, void* workspace_dev, void* stream) {
    if (net != 0 && net != 1) {
        g_err = "navppo_mlp64_bf16x3_loss_grad_net: net is 0 (actor) or 1 (critic)";
        return -1;
    }
    return x3_epoch("navppo_mlp64_bf16x3_loss_grad_net", const_cast<float*>(params_dev), prep_dev, obs_dim, act_dev, logp_old_dev, rtg_dev, adv_dev,
                    n_samples, var, clip, net == 0 ? 1 : 2, 0, 0.f, 0.f, 0.f, 0.f, nullptr, nullptr, grad_dev, stats_dev, workspace_dev, stream);
}

int navppo_mlp64_bf16x3_update_epoch(float* params_dev, const void* prep_dev, int32_t obs_dim, const float* act_dev, const float* logp_old_dev,
                                     const float* rtg_dev, const float* adv_dev, int64_t n_samples, float var, float clip, float lr,
                                     float beta1, float beta2, float eps, int32_t step, float* adam_m_dev, float* adam_v_dev,
                                     float* grad_dev, float* stats_dev, void* workspace_dev, void* stream) {
    if (step < 1 || !adam_m_dev || !adam_v_dev) {
        g_err = "navppo_mlp64_bf16x3_update_epoch: bad argument";
        return -1;
    }
    return x3_epoch("navppo_mlp64_bf16x3_update_epoch", params_dev, prep_dev, obs_dim, act_dev, logp_old_dev, rtg_dev, adv_dev, n_samples, var, clip,
                    3, step, lr, beta1, beta2, eps, adam_m_dev, adam_v_dev, grad_dev, stats_dev, workspace_dev, stream);
}

int navppo_episode_sums(const uint8_t* ended_dev, const uint8_t* arrive_dev, const uint8_t* done_dev, const int32_t* ep_length_dev,
                        const float* ep_return_dev, int64_t n, double* sums_dev, void* workspace_dev, void* stream) {
    if (!ended_dev || !arrive_dev || !done_dev || !ep_length_dev || !ep_return_dev || !sums_dev || !workspace_dev || n < 0) {
        g_err = "navppo_episode_sums: bad argument";
        return -1;
    }
    double* partial = reinterpret_cast<double*>(workspace_dev);
    const long long want = (n + kSumThreads - 1) / kSumThreads;
    const int blocks = (int)(want < 1 ? 1 : (want < kSumBlocks ? want : kSumBlocks));
    const bool vec = (n % 4 == 0) && !(((uintptr_t)ended_dev | (uintptr_t)arrive_dev | (uintptr_t)done_dev) & 3) &&
                     !(((uintptr_t)ep_length_dev | (uintptr_t)ep_return_dev) & 15);
    if (vec)
        hipLaunchKernelGGL(episode_sums_partial<true>, dim3(blocks), dim3(kSumThreads), 0, (hipStream_t)stream, ended_dev, arrive_dev,
                           done_dev, ep_length_dev, ep_return_dev, (long long)n, partial);
    else
        hipLaunchKernelGGL(episode_sums_partial<false>, dim3(blocks), dim3(kSumThreads), 0, (hipStream_t)stream, ended_dev, arrive_dev,
                           done_dev, ep_length_dev, ep_return_dev, (long long)n, partial);
    hipLaunchKernelGGL(episode_sums_final, dim3(1), dim3(kSumBlocks), 0, (hipStream_t)stream, partial, blocks, sums_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_err = std::string("navppo_episode_sums: ") + hipGetErrorString(e);
        return -2;
    }
    return 0;
}

int navppo_mlp64_act(const float* actor_params_dev, const void* obs_dev, int32_t obs_dim, int32_t obs_f16, const float* noise_dev,
                     int64_t n_envs, const float* var_dev, uint64_t seed, uint64_t env_id_base, const uint32_t* step_base_dev,
                     uint32_t step_offset, float* act_dev, float* logp_dev, float* mean_dev, void* stream) {
    if (!actor_params_dev || !obs_dev || !act_dev || !logp_dev || n_envs < 1 || !var_dev || (obs_dim != 16 && obs_dim != 42)) {
        g_err = "navppo_mlp64_act: bad argument (obs_dim is 16 or 42)";
        return -1;
    }
    if (((uintptr_t)actor_params_dev & 15) || !obs_aligned(obs_dev, obs_dim, obs_f16)) {
        g_err = "navppo_mlp64_act: params must be 16-byte aligned, obs 16-byte (42 columns: 8-byte, float16: 4-byte)";
        return -1;
    }
    const int blocks = (int)((n_envs + kActEnvs - 1) / kActEnvs);
    for_obs(obs_dim, obs_f16, [&](auto in, auto f16) {
        hipLaunchKernelGGL((mlp64_act<decltype(in)::value, decltype(f16)::value>), dim3(blocks), dim3(64), 0, (hipStream_t)stream,
                           actor_params_dev, obs_dev, noise_dev, (long long)n_envs, var_dev, seed, env_id_base, step_base_dev, step_offset,
                           act_dev, logp_dev, mean_dev);
    });
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_err = std::string("navppo_mlp64_act: ") + hipGetErrorString(e);
        return -2;
    }
    return 0;
}

}  // extern "C"
#pragma GCC visibility pop
