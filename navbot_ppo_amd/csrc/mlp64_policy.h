// mlp64_policy.h -- one policy step of the D-64-64 actor (D = 16: 10 beams, D = 42: 36 beams) for 16 envs on ONE wave (gfx950,
// f32 MFMA), shared by
//   navppo_mlp64_act        (ppo_mlp64.hip: one launch per rollout step) and
//   navsim_rollout_mlp64    (navsim.hip: the persistent rollout kernel, all T steps in one launch)
// so that both produce the same bits.
//
// PPO.get_action (project_ppo/src/ppo.py:673-706): mean = actor(obs), action = clamp(mean + sqrt(var) * eps) with
// a0 in [0, 1], a1 in [-1, 1] (:700-703), log-prob of the CLAMPED action under N(mean, var I) (:704).  eps comes from
// `noise_row` (two floats, e.g. torch.randn) or, when it is null, from Philox4x32-10 keyed by (seed, global env id, step)
// + Box-Muller.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdint>

namespace mlp64 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 64;      // hidden width
constexpr int IN = 16;     // observation width of BASELINE configs[1] (10 beams); Layout<D> below carries any width

// flat parameter layout of one net (nn.Module.named_parameters order: layer1.weight, layer1.bias, layer2.weight,
// layer2.bias, layer3.weight, layer3.bias [, layer4.weight, layer4.bias]) for observation width D
template <int D>
struct Layout {
    static constexpr int IN = D;
    static constexpr int OFF_W1 = 0, OFF_B1 = OFF_W1 + H * D, OFF_W2 = OFF_B1 + H, OFF_B2 = OFF_W2 + H * H, OFF_W3 = OFF_B2 + H,
                         OFF_B3 = OFF_W3 + H, OFF_W4 = OFF_B3 + 1, OFF_B4 = OFF_W4 + H;
    static constexpr int P_ACTOR = OFF_B4 + 1;    // D = 16: 5378, D = 42: 7042
    static constexpr int P_CRITIC = OFF_B3 + 1;   // D = 16: 5313, D = 42: 6977
    // policy step: features per lane group (the 16x16x4 MFMA takes k = lane >> 4; lane group kk feeds features KS kk .. KS kk + KS - 1)
    static constexpr int KS = (D + 3) / 4;        // 4 | 11 (42 pads to 44: the two features past the row are zero on both operands)
    static_assert(D % 2 == 0 && (OFF_B1 % 4) == 0 && (OFF_W2 % 4) == 0, "rows 8-byte aligned, vectors 16-byte aligned");
};
constexpr int OFF_W1 = Layout<IN>::OFF_W1, OFF_B1 = Layout<IN>::OFF_B1, OFF_W2 = Layout<IN>::OFF_W2, OFF_B2 = Layout<IN>::OFF_B2,
              OFF_W3 = Layout<IN>::OFF_W3, OFF_B3 = Layout<IN>::OFF_B3, OFF_W4 = Layout<IN>::OFF_W4, OFF_B4 = Layout<IN>::OFF_B4;
constexpr int P_ACTOR = Layout<IN>::P_ACTOR;   // 5378
constexpr int P_CRITIC = Layout<IN>::P_CRITIC;  // 5313

// relu as an integer max on the bit pattern: negative floats (sign bit set, incl. -0) -> +0, positive unchanged.
// One v_max_i32 instead of the canonicalise + v_max_f32 pair the compiler emits for fmaxf(x, 0).
__device__ __forceinline__ float relu_bits(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

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

constexpr int kActEnvs = 16;   // envs per wave

struct PolicyOut {
    float a0, a1, logp, mu0, mu1;   // meaningful on lanes 0..15 (lane = env of the wave)
};

// The policy step of 16 envs, computed transposed on v_mfma_f32_16x16x4_f32 (lane = (l15 = lane & 15: env, kk = lane >> 4)):
//   H1^T[n][m] = relu(b1 + W1 X^T)   A = W1 rows (k-permuted: lane group kk reads columns KS kk .. KS kk + KS - 1; D = 16: one
//                                     dwordx4), B = the lane's own KS observation floats xs = obs[env l15][KS kk .. KS kk + KS - 1]
//   H2^T       = relu(b2 + W2 H1^T)  B = the H1^T accumulators: register r of lane (m, kk) is row 4 kk + r of its tile
//   z3, z4     = w3 . H2 + b3, w4 . H2 + b4: per 16-row tile t2 of H2 a partial sum (in-lane fma chain over the lane's four
//                rows, then the four lane groups added by xor-shuffles 16, 32), the four partials added in tile order.
// The pieces below are what one wave (policy_wave16) or four waves (the persistent rollout kernel: every wave computes H1,
// wave t2 its tile of H2 and its partial sums) execute; the arithmetic and its order are the same, hence the same bits.
// `params` may be global memory (weights from L2 / L1, 21-28 KB shared by every wave) or an LDS copy.  L = Layout<D>.

// the lane's KS observation entries out of a float32 row (entries past the row: 0).  half_rows: the observation buffers hold
// float16 (navsim_cfg.obs_f16) -- the policy then sees what a reader of those buffers sees, the entries rounded to half
template <class L>
__device__ __forceinline__ void policy_row(const float* __restrict__ row, const int kk, const bool half_rows, float (&xs)[L::KS]) {
#pragma unroll
    for (int s = 0; s < L::KS; ++s) {
        const int f = L::KS * kk + s;
        float v = row[min(f, L::IN - 1)];
        if (f >= L::IN) v = 0.f;
        if (half_rows) v = __half2float(__float2half_rn(v));
        xs[s] = v;
    }
}

// layer 1: all 64 lanes
template <class L>
__device__ __forceinline__ void policy_hidden1(const float* __restrict__ params, const float (&xs)[L::KS], const int lane, f32x4 (&c1)[4]) {
    const int l15 = lane & 15, kk = lane >> 4;
    auto ld4 = [](const float* p) { return *reinterpret_cast<const float4*>(p); };
    float w1[4][L::KS];
    float4 b1q[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float* wrow = params + L::OFF_W1 + (16 * t + l15) * L::IN;   // W1[16 t + l15][.]
        if constexpr (L::IN == 16) {
            const float4 q = ld4(wrow + 4 * kk);
            w1[t][0] = q.x; w1[t][1] = q.y; w1[t][2] = q.z; w1[t][3] = q.w;
        } else {
#pragma unroll
            for (int s = 0; s < L::KS; ++s) {
                const int f = L::KS * kk + s;
                w1[t][s] = wrow[min(f, L::IN - 1)];
                if (f >= L::IN) w1[t][s] = 0.f;
            }
        }
        b1q[t] = ld4(params + L::OFF_B1 + 16 * t + 4 * kk);                 // rows 16 t + 4 kk + r of the accumulator
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) c1[t] = f32x4{b1q[t].x, b1q[t].y, b1q[t].z, b1q[t].w};
#pragma unroll
    for (int s = 0; s < L::KS; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)   // four independent accumulators back to back
            c1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[t][s], xs[s], c1[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) c1[t][r] = relu_bits(c1[t][r]);
}

// rows 16 t2 .. 16 t2 + 15 of layer 2 and their share of the two output units; all 64 lanes, result on every lane
template <class L>
__device__ __forceinline__ void policy_tile2(const float* __restrict__ params, const f32x4 (&c1)[4], const int lane, const int t2,
                                             float& pz3, float& pz4) {
    const int l15 = lane & 15, kk = lane >> 4;
    auto ld4 = [](const float* p) { return *reinterpret_cast<const float4*>(p); };
    float4 w2q[4];
#pragma unroll
    for (int t1 = 0; t1 < 4; ++t1) w2q[t1] = ld4(params + L::OFF_W2 + (16 * t2 + l15) * H + 16 * t1 + 4 * kk);
    const float4 b2q = ld4(params + L::OFF_B2 + 16 * t2 + 4 * kk);
    const float4 w3q = ld4(params + L::OFF_W3 + 16 * t2 + 4 * kk);
    // layer4.weight starts one float after layer3.bias: not 16-byte aligned, so four dword loads
    const float* w4p = params + L::OFF_W4 + 16 * t2 + 4 * kk;
    const float w4v[4] = {w4p[0], w4p[1], w4p[2], w4p[3]};
    f32x4 c2 = f32x4{b2q.x, b2q.y, b2q.z, b2q.w};
#pragma unroll
    for (int t1 = 0; t1 < 4; ++t1)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 w = w2q[t1];   // W2[16 t2 + l15][16 t1 + 4 kk + r]
            const float a = r == 0 ? w.x : r == 1 ? w.y : r == 2 ? w.z : w.w;
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, c1[t1][r], c2, 0, 0, 0);
        }
    const float w3v[4] = {w3q.x, w3q.y, w3q.z, w3q.w};
    float z3 = 0.f, z4 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float h = relu_bits(c2[r]);
        z3 = fmaf(h, w3v[r], z3);
        z4 = fmaf(h, w4v[r], z4);
    }
    z3 += __shfl_xor(z3, 16, 64);
    z4 += __shfl_xor(z4, 16, 64);
    z3 += __shfl_xor(z3, 32, 64);
    z4 += __shfl_xor(z4, 32, 64);
    pz3 = z3;
    pz4 = z4;
}

// the two standard-normal draws of env `gid` at rollout step `step`: Philox4x32-10 + Box-Muller
__device__ __forceinline__ void policy_noise(const uint32_t step, const uint64_t seed, const uint64_t gid, float& e0, float& e1) {
    uint32_t r[4];
    philox10((uint32_t)gid, (uint32_t)(gid >> 32), step, 0x61637473u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float u1 = ((float)(r[0] >> 8) + 1.0f) * 0x1.0p-24f;  // (0, 1]
    const float u2 = (float)(r[1] >> 8) * 0x1.0p-24f;           // [0, 1)
    const float rad = sqrtf(-2.0f * logf(u1));
    e0 = rad * cosf(6.283185307179586f * u2);
    e1 = rad * sinf(6.283185307179586f * u2);
}

// pz3 / pz4: the four tile partials in tile order.  sd = sqrtf(var), log_var = logf(var): functions of the launch's variance alone,
// which a caller that takes many steps with one variance evaluates once (policy_finish below evaluates them per call: same bits).
template <class L>
__device__ __forceinline__ PolicyOut policy_finish_pre(const float* __restrict__ params, const float (&pz3)[4], const float (&pz4)[4],
                                                       const float var, const float sd, const float log_var, const float e0,
                                                       const float e1) {
    const float z3 = (((pz3[0] + pz3[1]) + pz3[2]) + pz3[3]) + params[L::OFF_B3];
    const float z4 = (((pz4[0] + pz4[1]) + pz4[2]) + pz4[3]) + params[L::OFF_B4];
    PolicyOut o;
    o.mu0 = 1.0f / (1.0f + expf(-z3));
    o.mu1 = tanhf(z4);
    o.a0 = fminf(fmaxf(fmaf(sd, e0, o.mu0), 0.f), 1.f);    // ppo.py:698-703
    o.a1 = fminf(fmaxf(fmaf(sd, e1, o.mu1), -1.f), 1.f);
    const float d0 = o.a0 - o.mu0, d1 = o.a1 - o.mu1;
    o.logp = -0.5f * ((d0 * d0 + d1 * d1) / var) - 1.8378770664093453f - log_var;  // ppo.py:704
    return o;
}
template <class L>
__device__ __forceinline__ PolicyOut policy_finish(const float* __restrict__ params, const float (&pz3)[4], const float (&pz4)[4],
                                                   const float var, const float e0, const float e1) {
    return policy_finish_pre<L>(params, pz3, pz4, var, sqrtf(var), logf(var), e0, e1);
}

// One wave = 16 envs, no LDS and no barrier.  All 64 lanes must call (MFMA); `noise_row`, `gid` are per env (used on lanes
// < 16 only, where the result is meaningful).
template <class L>
__device__ __forceinline__ PolicyOut policy_wave16(const float* __restrict__ params, const float (&xs)[L::KS], const int lane,
                                                   const float var, const float* __restrict__ noise_row, const uint32_t step,
                                                   const uint64_t seed, const uint64_t gid) {
    f32x4 c1[4];
    policy_hidden1<L>(params, xs, lane, c1);
    float pz3[4], pz4[4];
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2) policy_tile2<L>(params, c1, lane, t2, pz3[t2], pz4[t2]);
    PolicyOut o = {0.f, 0.f, 0.f, 0.f, 0.f};
    if ((lane >> 4) == 0) {
        float e0, e1;
        if (noise_row) {
            e0 = noise_row[0];
            e1 = noise_row[1];
        } else {
            policy_noise(step, seed, gid, e0, e1);
        }
        o = policy_finish<L>(params, pz3, pz4, var, e0, e1);
    }
    return o;
}

}  // namespace mlp64
