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
// (profiles/r01_bench_v4_kernel_stats.csv: 3.8 ms per epoch, relu-backward alone 1 ms).  Here a workgroup
// keeps a 128-sample tile and the net's weights in LDS (132 KB of the CU's 160 KB): per epoch the batch is read
// once per net (84 B/sample) and nothing but 10,691 gradient floats per workgroup is written.
//
// Arithmetic: float32 throughout.  GEMMs use v_mfma_f32_32x32x2_f32, which is an exact k-ordered f32 fma chain
// (no reduced-precision inputs); sums over samples are taken in tile order, so results agree with PyTorch
// autograd to fp32 round-off, not bit for bit.
//
// One launch per net (template ACTOR): grid = persistent workgroups (1 per CU), each loops over tiles.
//   F1  H1 = relu(X W1^T + b1)      [128x16]x[16x64]    wave w owns rows 32w..32w+31
//   F2  H2 = relu(H1 W2^T + b2)     [128x64]x[64x64]
//   heads + PPO loss / MSE (VALU, thread = sample) -> g3, g4 (dL/dz of the output units)
//   dH2 = (g3 w3 + g4 w4) . [H2 > 0]                    (VALU, elementwise) ; db2, dW3, dW4, db3, db4 partial sums
//   B2  dH1 = (dH2 W2) . [H1 > 0]   [128x64]x[64x64]    ; db1 partial sums
//   G2  dW2 += dH2^T H1             [64x128]x[128x64]   wave w owns one 32x32 quadrant, accumulators persist over tiles
//   G1  dW1 += dH1^T X              [64x128]x[128x16]
// At the end every workgroup writes its partial gradient (one row of `partial`), and `reduce_partials` sums rows.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "navppo.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TM = 128;    // samples per tile
constexpr int H = 64;      // hidden width
constexpr int IN = 16;     // observation width
constexpr int LDH = H + 1; // padded LDS row strides (odd: conflict-free for row-per-lane and column-per-lane reads)
constexpr int LDX = IN + 1;
constexpr int kThreads = 256;

// flat parameter layout of one net (nn.Module.named_parameters order: layer1.weight, layer1.bias, layer2.weight,
// layer2.bias, layer3.weight, layer3.bias [, layer4.weight, layer4.bias])
constexpr int OFF_W1 = 0, OFF_B1 = OFF_W1 + H * IN, OFF_W2 = OFF_B1 + H, OFF_B2 = OFF_W2 + H * H, OFF_W3 = OFF_B2 + H,
              OFF_B3 = OFF_W3 + H, OFF_W4 = OFF_B3 + 1, OFF_B4 = OFF_W4 + H;
constexpr int P_ACTOR = OFF_B4 + 1;   // 5378
constexpr int P_CRITIC = OFF_B3 + 1;  // 5313
static_assert(P_ACTOR == NAVPPO_MLP64_ACTOR_PARAMS && P_CRITIC == NAVPPO_MLP64_CRITIC_PARAMS, "layout");

template <int TM_>
struct SmemT {
    float X[TM_ * LDX];
    float HB[3 * TM_ * LDH];  // H1 | H2 (later dH1) | dH2 ; at the very end: the gradient reduction buffer (5382 floats)
    float W1[H * LDX];
    float W2[H * LDH];
    float b1[H], b2[H], w3[H], w4[H];
    float g3[TM_], g4[TM_];
};

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// C/D element (row, col) of accumulator register r for lane l (32x32 shapes; cdna_hip_programming.md section 3)
__device__ __forceinline__ int c_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// PT = samples per tile, 4 threads per sample: PT = 128 -> 8 waves, 132 KB LDS, 1 workgroup per CU;
// PT = 64 -> 4 waves, 76 KB LDS, 2 independent workgroups per CU (their MFMA / VALU / barrier phases interleave).
constexpr int kPassTile = 64;
constexpr int kPassThreads = 4 * kPassTile;

template <bool ACTOR, int PT>
__global__ __launch_bounds__(4 * PT) void mlp64_pass(const float* __restrict__ params, const float* __restrict__ obs,
                                                           const float* __restrict__ act, const float* __restrict__ logp_old,
                                                           const float* __restrict__ rtg, const float* __restrict__ adv,
                                                           long long M, float var, float clip, float inv_n,
                                                           float* __restrict__ partial, float* __restrict__ stats_partial,
                                                           float* __restrict__ grad_zero, float* __restrict__ stats_zero) {
    constexpr int P = ACTOR ? P_ACTOR : P_CRITIC;
    constexpr int NT = 4 * PT;       // threads
    constexpr int NW = NT / 64;      // waves: PT/32 row strips x 2 column tiles
    constexpr int TM = PT;           // (shadows the namespace constant inside this kernel)
    static_assert(PT == 64 || PT == 128, "tile");
    __shared__ SmemT<PT> sm;
    static_assert(3 * PT * LDH >= P_ACTOR + 4, "reduction buffer");
    float* const sH1 = sm.HB;
    float* const sH2 = sm.HB + PT * LDH;
    float* const sdH2 = sm.HB + 2 * PT * LDH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int strip = wave >> 1, ct = wave & 1;  // F1/F2/B2: 32-row strip and 32-column tile owned by this wave

    if (blockIdx.x == 0) {  // the reduction that follows this launch accumulates with atomics: clear its targets here
        for (int k = tid; k < P; k += NT) grad_zero[k] = 0.f;
        if (tid < 3) stats_zero[tid] = 0.f;
    }
    // ---- weights -> LDS (once per workgroup)
    for (int k = tid; k < H * IN; k += NT) sm.W1[(k / IN) * LDX + (k % IN)] = params[OFF_W1 + k];
    for (int k = tid; k < H * H; k += NT) sm.W2[(k / H) * LDH + (k % H)] = params[OFF_W2 + k];
    if (tid < H) {
        sm.b1[tid] = params[OFF_B1 + tid];
        sm.b2[tid] = params[OFF_B2 + tid];
        sm.w3[tid] = params[OFF_W3 + tid];
        sm.w4[tid] = ACTOR ? params[OFF_W4 + tid] : 0.f;
    }
    const float b3 = params[OFF_B3];
    const float b4 = ACTOR ? params[OFF_B4] : 0.f;

    // ---- accumulators that persist over this workgroup's tiles
    f32x16 accW2 = zero16();   // quadrant (wave & 3) of dW2 over the samples of half (wave >> 2) of each tile
    f32x16 accW1 = zero16();   // half (wave & 1) of dW1 over the samples of quarter (wave >> 1) of each tile
    float acc_db1 = 0.f;       // column 32 ct + l31 of the dH1 rows this lane sees in B2
    float acc_db2 = 0.f, acc_dw3 = 0.f, acc_dw4 = 0.f;  // column (tid & 63), rows (tid >> 6) + 8 i
    float acc_db3 = 0.f, acc_db4 = 0.f;                 // sample-owner threads (tid & 3) == 0
    float st0 = 0.f, st1 = 0.f, st2 = 0.f, st3 = 0.f;   // loss, kl, clip-frac sums

    // Software prefetch: the next tile's observations (PT*16/NT = 4 floats per thread) and per-sample scalars are
    // loaded into registers while the current tile is being processed, so no phase waits on HBM latency.
    constexpr int XPT = TM * IN / NT;
    float xpre[XPT];
    float pre_a0 = 0.f, pre_a1 = 0.f, pre_lp = 0.f, pre_t = 0.f;  // act, logp_old, (adv | rtg) of sample tid >> 2
    auto prefetch_tile = [&](long long tile) {
        const long long mb = tile * TM;
#pragma unroll
        for (int j = 0; j < XPT; ++j) {
            const int k = tid + j * NT;
            const long long m = mb + k / IN;
            xpre[j] = (m < M) ? obs[m * IN + (k % IN)] : 0.f;
        }
        if ((tid & 3) == 0) {
            const long long m = mb + (tid >> 2);
            if (m < M) {
                if (ACTOR) {
                    pre_a0 = act[2 * m];
                    pre_a1 = act[2 * m + 1];
                    pre_lp = logp_old[m];
                    pre_t = adv[m];
                } else {
                    pre_t = rtg[m];
                }
            }
        }
    };
    const long long n_tiles = (M + TM - 1) / TM;
    if (blockIdx.x < n_tiles) prefetch_tile(blockIdx.x);
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long m_base = tile * TM;
        __syncthreads();  // previous tile's LDS readers are done
        // ---- X tile: registers -> LDS; then start loading the next tile
#pragma unroll
        for (int j = 0; j < XPT; ++j) {
            const int k = tid + j * NT;
            sm.X[(k / IN) * LDX + (k % IN)] = xpre[j];
        }
        const float cur_a0 = pre_a0, cur_a1 = pre_a1, cur_lp = pre_lp, cur_t = pre_t;
        if (tile + gridDim.x < n_tiles) prefetch_tile(tile + gridDim.x);
        __syncthreads();

        // ---- F1: H1 = relu(X W1^T + b1), K = 16
        {
            f32x16 c = zero16();
            const float* a_ptr = sm.X + (32 * strip + l31) * LDX + lhi;
            const float* b_ptr = sm.W1 + (32 * ct + l31) * LDX + lhi;
#pragma unroll
            for (int k0 = 0; k0 < IN; k0 += 2) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ptr[k0], b_ptr[k0], c, 0, 0, 0);
            const float bias = sm.b1[32 * ct + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sH1[(32 * strip + c_row(r, lane)) * LDH + 32 * ct + l31] = fmaxf(c[r] + bias, 0.f);
        }
        __syncthreads();

        // ---- F2: H2 = relu(H1 W2^T + b2), K = 64
        {
            f32x16 c = zero16();
            const float* a_ptr = sH1 + (32 * strip + l31) * LDH + lhi;
            const float* b_ptr = sm.W2 + (32 * ct + l31) * LDH + lhi;
#pragma unroll 16
            for (int k0 = 0; k0 < H; k0 += 2) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ptr[k0], b_ptr[k0], c, 0, 0, 0);
            const float bias = sm.b2[32 * ct + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sH2[(32 * strip + c_row(r, lane)) * LDH + 32 * ct + l31] = fmaxf(c[r] + bias, 0.f);
        }
        __syncthreads();

        // ---- output units + loss: 4 threads per sample, 16 hidden units each, then a 4-lane butterfly.
        //      Thread part p reads units 16 p + ((j + 8 (p >> 1)) & 15): conflict-free over each 32-lane group.
        {
            const int ms = tid >> 2, p = tid & 3;
            const long long m = m_base + ms;
            const float* h2 = sH2 + ms * LDH + 16 * p;
            const int rot = 8 * (p >> 1);
            float z3 = 0.f, z4 = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int k = (j + rot) & 15;
                const float h = h2[k];
                z3 = fmaf(h, sm.w3[16 * p + k], z3);
                if (ACTOR) z4 = fmaf(h, sm.w4[16 * p + k], z4);
            }
            z3 += __shfl_xor(z3, 1, 64);
            z3 += __shfl_xor(z3, 2, 64);
            if (ACTOR) {
                z4 += __shfl_xor(z4, 1, 64);
                z4 += __shfl_xor(z4, 2, 64);
            }
            if (p == 0) {
                float g3 = 0.f, g4 = 0.f;
                if (m < M) {
                    z3 += b3;
                    z4 += b4;
                    if (ACTOR) {
                        const float mu0 = 1.0f / (1.0f + expf(-z3));   // torch.sigmoid, net_actor.py:185
                        const float mu1 = tanhf(z4);                    // net_actor.py:186
                        const float a0 = cur_a0, a1 = cur_a1;
                        const float d0 = a0 - mu0, d1 = a1 - mu1;
                        // MultivariateNormal(mean, var*I).log_prob, ppo.py:734-735
                        const float lp = -0.5f * ((d0 * d0 + d1 * d1) / var) - 1.8378770664093453f - logf(var);
                        const float lr = lp - cur_lp;
                        const float ratio = expf(lr);                  // ppo.py:316
                        const float A = cur_t;
                        const float s1 = ratio * A;                     // ppo.py:319
                        const float rc = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
                        const float s2 = rc * A;                        // ppo.py:320
                        st0 += -fminf(s1, s2);                          // ppo.py:342 (mean taken by inv_n at the end)
                        st2 += (ratio - 1.0f) - lr;                     // approx KL, ppo.py:326
                        st3 += (fabsf(ratio - 1.0f) > clip) ? 1.f : 0.f;  // clip fraction, ppo.py:335
                        // d(-min(s1,s2))/d ratio: -A through s1 when s1 <= s2 inside the clip range (tie: both halves),
                        // or when s1 < s2 outside it; 0 when the clipped (constant) branch is the minimum
                        const bool inside = (ratio >= 1.0f - clip) && (ratio <= 1.0f + clip);
                        const float dL_dratio = (inside || s1 < s2) ? -A : 0.f;
                        const float dL_dlp = dL_dratio * ratio * inv_n;
                        g3 = dL_dlp * (d0 / var) * (mu0 * (1.0f - mu0));
                        g4 = dL_dlp * (d1 / var) * (1.0f - mu1 * mu1);
                    } else {
                        const float V = z3;                             // critic(obs).squeeze(), ppo.py:724
                        const float e = V - cur_t;
                        st1 += e * e;                                   // MSELoss, ppo.py:343
                        g3 = 2.0f * e * inv_n;
                    }
                }
                sm.g3[ms] = g3;
                sm.g4[ms] = g4;
                acc_db3 += g3;
                acc_db4 += g4;
            }
        }
        __syncthreads();

        // ---- dH2 = (g3 w3 + g4 w4) . [H2 > 0] ; column sums for db2, dW3, dW4
        {
            const int k = tid & 63;
            const float w3k = sm.w3[k], w4k = sm.w4[k];
#pragma unroll 4
            for (int i = 0; i < TM / NW; ++i) {
                const int m = (tid >> 6) + NW * i;
                const float h = sH2[m * LDH + k];
                const float g3 = sm.g3[m], g4 = sm.g4[m];
                const float d = (h > 0.f) ? fmaf(g3, w3k, g4 * w4k) : 0.f;
                sdH2[m * LDH + k] = d;
                acc_db2 += d;
                acc_dw3 = fmaf(g3, h, acc_dw3);
                acc_dw4 = fmaf(g4, h, acc_dw4);
            }
        }
        __syncthreads();

        // ---- B2: dH1 = (dH2 W2) . [H1 > 0] -> stored over H2 ; db1 partial sums
        {
            f32x16 c = zero16();
            const float* a_ptr = sdH2 + (32 * strip + l31) * LDH + lhi;
            const float* b_ptr = sm.W2 + lhi * LDH + 32 * ct + l31;
#pragma unroll 16
            for (int n0 = 0; n0 < H; n0 += 2) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ptr[n0], b_ptr[n0 * LDH], c, 0, 0, 0);
            // every wave finished reading H2 before the barrier above; each wave overwrites only its own tile
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int idx = (32 * strip + c_row(r, lane)) * LDH + 32 * ct + l31;
                const float d = (sH1[idx] > 0.f) ? c[r] : 0.f;
                sH2[idx] = d;
                acc_db1 += d;
            }
        }
        __syncthreads();

        // ---- G2: dW2[n][k] += sum_m dH2[m][n] H1[m][k]; quadrant (nt, kt) from wave & 3, samples 64 (wave>>2) .. +64
        {
            const int q = wave & 3, mh = 64 * (wave >> 2);  // NW/4 sample groups of 64
            const float* a_ptr = sdH2 + (mh + lhi) * LDH + 32 * (q >> 1) + l31;
            const float* b_ptr = sH1 + (mh + lhi) * LDH + 32 * (q & 1) + l31;
#pragma unroll 16
            for (int m0 = 0; m0 < 64; m0 += 2)
                accW2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ptr[m0 * LDH], b_ptr[m0 * LDH], accW2, 0, 0, 0);
        }
        // ---- G1: dW1[n][k] += sum_m dH1[m][n] X[m][k] (k < 16; columns 16..31 of the 32-wide tile are zero padding);
        //      half nt = wave & 1, samples 32 (wave>>1) .. +32
        {
            const int mq = 32 * (wave >> 1);
            const float* a_ptr = sH2 + (mq + lhi) * LDH + 32 * (wave & 1) + l31;
            const float* b_ptr = sm.X + (mq + lhi) * LDX + (l31 & 15);
            const bool live = l31 < IN;
#pragma unroll 16
            for (int m0 = 0; m0 < 32; m0 += 2) {
                const float b = live ? b_ptr[m0 * LDX] : 0.f;
                accW1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ptr[m0 * LDH], b, accW1, 0, 0, 0);
            }
        }
    }

    // ---- workgroup reduction of the partial gradient in LDS, then one coalesced row of `partial`
    __syncthreads();
    float* red = sm.HB;
    for (int k = tid; k < P + 4; k += NT) red[k] = 0.f;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = c_row(r, lane);
        const int q = wave & 3;
        atomicAdd(&red[OFF_W2 + (32 * (q >> 1) + row) * H + 32 * (q & 1) + l31], accW2[r]);
        if (l31 < IN) atomicAdd(&red[OFF_W1 + (32 * (wave & 1) + row) * IN + l31], accW1[r]);
    }
    atomicAdd(&red[OFF_B1 + 32 * ct + l31], acc_db1);
    atomicAdd(&red[OFF_B2 + (tid & 63)], acc_db2);
    atomicAdd(&red[OFF_W3 + (tid & 63)], acc_dw3);
    if (ACTOR) atomicAdd(&red[OFF_W4 + (tid & 63)], acc_dw4);
    if ((tid & 3) == 0) {
        atomicAdd(&red[OFF_B3], acc_db3);
        if (ACTOR) atomicAdd(&red[OFF_B4], acc_db4);
        atomicAdd(&red[P + 0], ACTOR ? st0 : st1);
        atomicAdd(&red[P + 1], st2);
        atomicAdd(&red[P + 2], st3);
    }
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * P;
    for (int k = tid; k < P; k += NT) out[k] = red[k];
    if (tid < 3) stats_partial[blockIdx.x * 4 + tid] = red[P + tid];
}

// grad[p] += sum over a slice of the workgroups' partial rows ; stats += slice sums * inv_n.  grid = (P/64, kRedSlices):
// block = 64 parameters x 4 row groups of one slice; rows are read 256 B per wave; one atomicAdd per parameter per slice
// (grad / stats were zeroed by workgroup 0 of the pass kernel that produced the partials).
constexpr int kRedSlices = 8;

__global__ __launch_bounds__(256) void reduce_partials(const float* __restrict__ partial, const float* __restrict__ stats_partial,
                                                       int n_blocks, int P, float inv_n, float* __restrict__ grad,
                                                       float* __restrict__ stats) {
    __shared__ float part[4][64];
    const int p = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int per = (n_blocks + kRedSlices - 1) / kRedSlices;
    const int b_lo = blockIdx.y * per, b_hi = min(n_blocks, b_lo + per);
    float s0 = 0.f, s1 = 0.f;
    if (p < P) {
        int b = b_lo + g;
        for (; b + 4 < b_hi; b += 8) {
            s0 += partial[(size_t)b * P + p];
            s1 += partial[(size_t)(b + 4) * P + p];
        }
        if (b < b_hi) s0 += partial[(size_t)b * P + p];
    }
    part[g][threadIdx.x & 63] = s0 + s1;
    __syncthreads();
    if (g == 0 && p < P)
        atomicAdd(&grad[p], (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
    if (blockIdx.x == 0 && threadIdx.x < 3) {
        float s = 0.f;
        for (int b = b_lo; b < b_hi; ++b) s += stats_partial[b * 4 + threadIdx.x];
        atomicAdd(&stats[threadIdx.x], s * inv_n);
    }
}

// ---------------------------------------------------------------- rollout-time policy step (PPO.get_action, ppo.py:673-706)
// mean = actor(obs) (same F1/F2 MFMA tiles as above), action = clamp(mean + sqrt(var) * eps), log-prob of the CLAMPED
// action under N(mean, var I).  eps comes from `noise` ([n,2], e.g. torch.randn) or, when noise == nullptr, from
// Philox4x32-10 keyed by (seed, global env id, step) + Box-Muller, so a rollout step is one launch.
__device__ __forceinline__ void philox10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                         uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ __launch_bounds__(kThreads) void mlp64_act(const float* __restrict__ params, const float* __restrict__ obs,
                                                      const float* __restrict__ noise, long long n,
                                                      const float* __restrict__ var_ptr, uint64_t seed,
                                                      uint64_t env_id_base, const uint32_t* __restrict__ step_base,
                                                      uint32_t step_offset,
                                                      float* __restrict__ act, float* __restrict__ logp,
                                                      float* __restrict__ mean_out) {
    __shared__ float X[TM * LDX], H1[TM * LDH], H2[TM * LDH], W1[H * LDX], W2[H * LDH], b1s[H], b2s[H], w3s[H], w4s[H];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const long long m_base = (long long)blockIdx.x * TM;
    const float var = *var_ptr;
    const uint32_t step = (step_base ? *step_base : 0u) + step_offset;
    for (int k = tid; k < H * IN; k += kThreads) W1[(k / IN) * LDX + (k % IN)] = params[OFF_W1 + k];
    for (int k = tid; k < H * H; k += kThreads) W2[(k / H) * LDH + (k % H)] = params[OFF_W2 + k];
    if (tid < H) {
        b1s[tid] = params[OFF_B1 + tid]; b2s[tid] = params[OFF_B2 + tid];
        w3s[tid] = params[OFF_W3 + tid]; w4s[tid] = params[OFF_W4 + tid];
    }
    for (int k = tid; k < TM * IN; k += kThreads) {
        const int m = k / IN, c = k % IN;
        X[m * LDX + c] = (m_base + m < n) ? obs[(m_base + m) * IN + c] : 0.f;
    }
    __syncthreads();
    {
        f32x16 c0 = zero16(), c1 = zero16();
        const float* a_ptr = X + (32 * wave + l31) * LDX + lhi;
#pragma unroll
        for (int k0 = 0; k0 < IN; k0 += 2) {
            const float a = a_ptr[k0];
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, W1[l31 * LDX + lhi + k0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, W1[(32 + l31) * LDX + lhi + k0], c1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * wave + c_row(r, lane);
            H1[row * LDH + l31] = fmaxf(c0[r] + b1s[l31], 0.f);
            H1[row * LDH + 32 + l31] = fmaxf(c1[r] + b1s[32 + l31], 0.f);
        }
    }
    __syncthreads();
    {
        f32x16 c0 = zero16(), c1 = zero16();
        const float* a_ptr = H1 + (32 * wave + l31) * LDH + lhi;
#pragma unroll 8
        for (int k0 = 0; k0 < H; k0 += 2) {
            const float a = a_ptr[k0];
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, W2[l31 * LDH + lhi + k0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, W2[(32 + l31) * LDH + lhi + k0], c1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * wave + c_row(r, lane);
            H2[row * LDH + l31] = fmaxf(c0[r] + b2s[l31], 0.f);
            H2[row * LDH + 32 + l31] = fmaxf(c1[r] + b2s[32 + l31], 0.f);
        }
    }
    __syncthreads();
    if (tid < TM && m_base + tid < n) {
        const long long m = m_base + tid;
        float z3 = params[OFF_B3], z4 = params[OFF_B4];
#pragma unroll 8
        for (int k = 0; k < H; ++k) {
            const float h = H2[tid * LDH + k];
            z3 = fmaf(h, w3s[k], z3);
            z4 = fmaf(h, w4s[k], z4);
        }
        const float mu0 = 1.0f / (1.0f + expf(-z3)), mu1 = tanhf(z4);
        float e0, e1;
        if (noise) {
            e0 = noise[2 * m];
            e1 = noise[2 * m + 1];
        } else {
            const uint64_t gid = env_id_base + (uint64_t)m;
            uint32_t r[4];
            philox10((uint32_t)gid, (uint32_t)(gid >> 32), step, 0x61637473u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
            const float u1 = ((float)(r[0] >> 8) + 1.0f) * 0x1.0p-24f;  // (0, 1]
            const float u2 = (float)(r[1] >> 8) * 0x1.0p-24f;           // [0, 1)
            const float rad = sqrtf(-2.0f * logf(u1));
            e0 = rad * cosf(6.283185307179586f * u2);
            e1 = rad * sinf(6.283185307179586f * u2);
        }
        const float sd = sqrtf(var);
        const float a0 = fminf(fmaxf(fmaf(sd, e0, mu0), 0.f), 1.f);    // ppo.py:698-703
        const float a1 = fminf(fmaxf(fmaf(sd, e1, mu1), -1.f), 1.f);
        const float d0 = a0 - mu0, d1 = a1 - mu1;
        act[2 * m] = a0;
        act[2 * m + 1] = a1;
        logp[m] = -0.5f * ((d0 * d0 + d1 * d1) / var) - 1.8378770664093453f - logf(var);  // ppo.py:704
        if (mean_out) {
            mean_out[2 * m] = mu0;
            mean_out[2 * m + 1] = mu1;
        }
    }
}

thread_local std::string g_err;

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

const char* navppo_last_error(void) { return g_err.c_str(); }

size_t navppo_mlp64_workspace_bytes(void) {
    return (size_t)NAVPPO_MLP64_MAX_BLOCKS * (NAVPPO_MLP64_ACTOR_PARAMS + 4) * sizeof(float);
}

int navppo_mlp64_loss_grad(const float* params_dev, const float* obs_dev, const float* act_dev,
                           const float* logp_old_dev, const float* rtg_dev, const float* adv_dev, int64_t n_samples,
                           float var, float clip, float* grad_dev, float* stats_dev, void* workspace_dev, void* stream) {
    if (!params_dev || !obs_dev || !act_dev || !logp_old_dev || !rtg_dev || !adv_dev || !grad_dev || !stats_dev ||
        !workspace_dev || n_samples < 1 || !(var > 0.f)) {
        g_err = "navppo_mlp64_loss_grad: bad argument";
        return -1;
    }
    hipStream_t st = (hipStream_t)stream;
    const long long tiles = (n_samples + kPassTile - 1) / kPassTile;
    const int blocks = (int)(tiles < NAVPPO_MLP64_MAX_BLOCKS ? tiles : NAVPPO_MLP64_MAX_BLOCKS);
    float* partial = reinterpret_cast<float*>(workspace_dev);
    float* stats_partial = partial + (size_t)NAVPPO_MLP64_MAX_BLOCKS * NAVPPO_MLP64_ACTOR_PARAMS;
    const float inv_n = 1.0f / (float)n_samples;
    hipLaunchKernelGGL((mlp64_pass<true, kPassTile>), dim3(blocks), dim3(kPassThreads), 0, st, params_dev, obs_dev, act_dev, logp_old_dev,
                       rtg_dev, adv_dev, (long long)n_samples, var, clip, inv_n, partial, stats_partial, grad_dev, stats_dev);
    hipLaunchKernelGGL(reduce_partials, dim3((P_ACTOR + 63) / 64, kRedSlices), dim3(256), 0, st, partial, stats_partial, blocks,
                       P_ACTOR, inv_n, grad_dev, stats_dev);
    hipLaunchKernelGGL((mlp64_pass<false, kPassTile>), dim3(blocks), dim3(kPassThreads), 0, st, params_dev + P_ACTOR, obs_dev, act_dev,
                       logp_old_dev, rtg_dev, adv_dev, (long long)n_samples, var, clip, inv_n, partial, stats_partial,
                       grad_dev + P_ACTOR, stats_dev + 4);
    hipLaunchKernelGGL(reduce_partials, dim3((P_CRITIC + 63) / 64, kRedSlices), dim3(256), 0, st, partial, stats_partial, blocks,
                       P_CRITIC, inv_n, grad_dev + P_ACTOR, stats_dev + 4);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_err = std::string("navppo_mlp64_loss_grad: ") + hipGetErrorString(e);
        return -2;
    }
    return 0;
}

int navppo_mlp64_act(const float* actor_params_dev, const float* obs_dev, const float* noise_dev, int64_t n_envs,
                     const float* var_dev, uint64_t seed, uint64_t env_id_base, const uint32_t* step_base_dev,
                     uint32_t step_offset, float* act_dev, float* logp_dev, float* mean_dev, void* stream) {
    if (!actor_params_dev || !obs_dev || !act_dev || !logp_dev || n_envs < 1 || !var_dev) {
        g_err = "navppo_mlp64_act: bad argument";
        return -1;
    }
    const int blocks = (int)((n_envs + TM - 1) / TM);
    hipLaunchKernelGGL(mlp64_act, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, actor_params_dev, obs_dev, noise_dev,
                       (long long)n_envs, var_dev, seed, env_id_base, step_base_dev, step_offset, act_dev, logp_dev, mean_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_err = std::string("navppo_mlp64_act: ") + hipGetErrorString(e);
        return -2;
    }
    return 0;
}

}  // extern "C"
#pragma GCC visibility pop
