// resmlp_policy.h -- one policy step of the reference's ACTIVE actor (NetActor, project_ppo/src/net_actor.py:56-144: two residual
// blocks of 512 hidden units, LeakyReLU(0.2), heads sigmoid / tanh) for 16 envs on ONE 8-wave workgroup (gfx950, f32 MFMA), shared by
//   navppo_resmlp512_act     (ppo_resmlp512.hip: one launch per rollout step) and
//   navsim_rollout_resmlp512 (navsim.hip: the persistent rollout kernel, all T steps in one launch)
// so that both produce the same bits.  PPO.get_action (project_ppo/src/ppo.py:673-706) around it is the caller's.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace resmlp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace rp {   // flat parameter layout of one net = nn.Module.named_parameters() order without the unused BatchNorm entries
constexpr int D = 16, HID = 512;
constexpr int W1A = 0, B1A = W1A + HID * D, W2A = B1A + HID, B2A = W2A + D * HID, W1B = B2A + D, B1B = W1B + HID * 2 * D,
              W2B = B1B + HID, B2B = W2B + 2 * D * HID, WO1 = B2B + 2 * D, BO1 = WO1 + 2 * D, WO2 = BO1 + 1, BO2 = WO2 + 2 * D;
constexpr int P_ACTOR = BO2 + 1, P_CRITIC = BO1 + 1;   // 50290, 50257
}  // namespace rp

constexpr int kPolWaves = 8, kPolEnvs = 16;   // waves of the workgroup (each owns 64 hidden units), envs per workgroup

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ f32x4 v4(const float4 a) { return f32x4{a.x, a.y, a.z, a.w}; }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
// nn.LeakyReLU(0.2) (net_actor.py:38) = max(x, 0.2 x) = the median of (x, 0.2 x, FLT_MAX-ish): v_mul + v_med3_f32, two
// instructions.  fmaxf() costs three (the compiler quiets a possible signalling NaN with v_max x, x first; it also rewrites a
// median against +inf into that max).  A hand-written v_max_f32 in inline asm is two as well, but the hazard recogniser does
// not see an asm statement as a VALU write and the MFMA behind it read a stale register (the rollout policy step did).
__device__ __forceinline__ float leaky(float x) { return __builtin_amdgcn_fmed3f(x, 0.2f * x, 3.0e38f); }
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// LDS scratch of one policy step: the two block outputs are summed over the workgroup's waves through it
struct PolicySmem {
    float part1[kPolWaves][256];
    float part2[kPolWaves][2][256];
};

// every weight one wave needs for a policy step: its 64 hidden units of both blocks (32 dwordx4 per lane, 128 registers)
struct Weights {
    f32x4 w1a[4], b1a[4], w2a[4], w1b[4][2], b1b[4], w2b[2][4];
};
// requested in the order the step consumes them (block 1 first: its MFMAs start while block 2's weights are still in flight).  The
// per-step kernel (resmlp_act) streams both blocks from L2; the persistent rollout kernel keeps block 1 in LDS (Block1Smem below)
// and streams block 2 only (navsim.hip: rollout_resmlp_kernel has the measurements).
__device__ __forceinline__ void load_weights_a(const float* __restrict__ pa, const int lane, const int w, Weights& W) {   // block 1
    const int l15 = lane & 15, q = lane >> 4;
    const int j0 = 64 * w;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        W.w1a[jb] = v4(ld4(pa + rp::W1A + (j0 + 16 * jb + l15) * 16 + 4 * q));
        W.b1a[jb] = v4(ld4(pa + rp::B1A + j0 + 16 * jb + 4 * q));
    }
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) W.w2a[jb] = v4(ld4(pa + rp::W2A + l15 * rp::HID + j0 + 16 * jb + 4 * q));
}
__device__ __forceinline__ void load_weights_b(const float* __restrict__ pa, const int lane, const int w, Weights& W) {   // block 2
    const int l15 = lane & 15, q = lane >> 4;
    const int j0 = 64 * w;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        W.w1b[jb][0] = v4(ld4(pa + rp::W1B + (j0 + 16 * jb + l15) * 32 + 4 * q));
        W.w1b[jb][1] = v4(ld4(pa + rp::W1B + (j0 + 16 * jb + l15) * 32 + 16 + 4 * q));
        W.b1b[jb] = v4(ld4(pa + rp::B1B + j0 + 16 * jb + 4 * q));
    }
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        W.w2b[0][jb] = v4(ld4(pa + rp::W2B + l15 * rp::HID + j0 + 16 * jb + 4 * q));
        W.w2b[1][jb] = v4(ld4(pa + rp::W2B + (16 + l15) * rp::HID + j0 + 16 * jb + 4 * q));
    }
}
// Block 1's weights (66 KB) kept in LDS by the persistent rollout kernel, in the order the waves consume them: one ds_read_b128 per
// operand register quad, consecutive lanes 16 bytes apart.
struct Block1Smem {
    float4 w1a[kPolWaves][4][64], w2a[kPolWaves][4][64];
    float b1a[rp::HID];
    float b2a[rp::D];
    float tail[rp::P_ACTOR - rp::B2B + 2];
};
__device__ __forceinline__ void stage_block1(const float* __restrict__ pa, const int lane, const int w, Block1Smem& bs) {   // all 8 waves
    Weights W;
    load_weights_a(pa, lane, w, W);
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        bs.w1a[w][jb][lane] = make_float4(W.w1a[jb][0], W.w1a[jb][1], W.w1a[jb][2], W.w1a[jb][3]);
        bs.w2a[w][jb][lane] = make_float4(W.w2a[jb][0], W.w2a[jb][1], W.w2a[jb][2], W.w2a[jb][3]);
    }
    bs.b1a[64 * w + lane] = pa[rp::B1A + 64 * w + lane];
    if (w == 0 && lane < rp::D) bs.b2a[lane] = pa[rp::B2A + lane];
    if (w == 1)
        for (int k = lane; k < rp::P_ACTOR - rp::B2B; k += 64) bs.tail[k] = pa[rp::B2B + k];
}
__device__ __forceinline__ void load_weights_a_lds(const Block1Smem& bs, const int lane, const int w, Weights& W) {
    const int q = lane >> 4;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        W.w1a[jb] = v4(bs.w1a[w][jb][lane]);
        W.b1a[jb] = v4(ld4(&bs.b1a[64 * w + 16 * jb + 4 * q]));
        W.w2a[jb] = v4(bs.w2a[w][jb][lane]);
    }
}
__device__ __forceinline__ void load_weights(const float* __restrict__ pa, const int lane, const int w, Weights& W) {
    load_weights_a(pa, lane, w, W);
    load_weights_b(pa, lane, w, W);
}

// The pre-activations (z3, z4) of the two output units for the workgroup's 16 envs.  ALL 8 waves of the workgroup call it (two
// workgroup barriers inside); lane (l15 = lane & 15: env, q = lane >> 4) passes xq = obs[env][4 q .. 4 q + 3] (zeros for an env
// past the shard).  One workgroup = 16 envs; its 8 waves split the hidden units 8 ways (64 each), weights come straight from global
// memory (L2: all workgroups read the same 197 KB), the two block outputs are summed over the waves through LDS.  The result is
// valid on wave 0 (every lane of an env's column holds the env's sums); the other waves return zeros.
// b2a = &params[rp::B2A] and tail = &params[rp::B2B] (block 2's output bias and the two heads: 98 floats), or copies of them in LDS.
__device__ __forceinline__ void policy_preact_w(const Weights& W, const float* __restrict__ b2a, const float* __restrict__ tail, const f32x4 xq,
                                                const int lane, const int w, PolicySmem& ps, float& z3_out, float& z4_out) {
    const int q = lane >> 4;
    const f32x4 (&w1a)[4] = W.w1a, (&b1a)[4] = W.b1a, (&w2a)[4] = W.w2a, (&b1b)[4] = W.b1b;
    const f32x4 (&w1b)[4][2] = W.w1b, (&w2b)[2][4] = W.w2b;
    // rb1, this wave's 64 hidden units
    f32x4 y1[2] = {zero4(), zero4()};
    {
        f32x4 H[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) H[jb] = b1a[jb];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) H[jb] = mfma16(w1a[jb][r], xq[r], H[jb]);
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) y1[jb & 1] = mfma16(w2a[jb][r], leaky(H[jb][r]), y1[jb & 1]);
    }
    *reinterpret_cast<float4*>(&ps.part1[w][4 * lane]) = make_float4(y1[0][0] + y1[1][0], y1[0][1] + y1[1][1], y1[0][2] + y1[1][2], y1[0][3] + y1[1][3]);
    __syncthreads();
    f32x4 h1;
    {
        const float4 b = ld4(b2a + 4 * q);
        f32x4 s = v4(ld4(&ps.part1[0][4 * lane]));
#pragma unroll
        for (int k = 1; k < kPolWaves; ++k) s += v4(ld4(&ps.part1[k][4 * lane]));
        h1 = xq + v4(b) + s;
#pragma unroll
        for (int r = 0; r < 4; ++r) h1[r] = leaky(h1[r]);
    }
    // rb2
    f32x4 y2[2] = {zero4(), zero4()};
    {
        f32x4 H[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) H[jb] = b1b[jb];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) H[jb] = mfma16(w1b[jb][0][r], xq[r], H[jb]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) H[jb] = mfma16(w1b[jb][1][r], h1[r], H[jb]);
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float hl = leaky(H[jb][r]);
                y2[0] = mfma16(w2b[0][jb][r], hl, y2[0]);
                y2[1] = mfma16(w2b[1][jb][r], hl, y2[1]);
            }
    }
    *reinterpret_cast<float4*>(&ps.part2[w][0][4 * lane]) = make_float4(y2[0][0], y2[0][1], y2[0][2], y2[0][3]);
    *reinterpret_cast<float4*>(&ps.part2[w][1][4 * lane]) = make_float4(y2[1][0], y2[1][1], y2[1][2], y2[1][3]);
    __syncthreads();
    float z3 = 0.f, z4 = 0.f;
    if (w == 0) {
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            f32x4 s = v4(ld4(&ps.part2[0][ob][4 * lane]));
#pragma unroll
            for (int k = 1; k < kPolWaves; ++k) s += v4(ld4(&ps.part2[k][ob][4 * lane]));
            const f32x4 x1 = ob == 0 ? xq : h1;
            const float4 b = ld4(tail + 16 * ob + 4 * q), u = ld4(tail + (rp::WO1 - rp::B2B) + 16 * ob + 4 * q);
            const float* w2p = tail + (rp::WO2 - rp::B2B) + 16 * ob + 4 * q;   // out2.weight starts one float after out1.bias: not 16-byte aligned
            const f32x4 h2 = x1 + v4(b) + s;
            const float uv[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float hl = leaky(h2[r]);
                z3 = fmaf(hl, uv[r], z3);
                z4 = fmaf(hl, w2p[r], z4);
            }
        }
        z3 += __shfl_xor(z3, 16, 64);
        z4 += __shfl_xor(z4, 16, 64);
        z3 += __shfl_xor(z3, 32, 64);
        z4 += __shfl_xor(z4, 32, 64);
    }
    z3_out = z3;
    z4_out = z4;
}
__device__ __forceinline__ void policy_preact(const float* __restrict__ pa, const f32x4 xq, const int lane, const int w, PolicySmem& ps,
                                              float& z3_out, float& z4_out) {
    Weights W;
    load_weights(pa, lane, w, W);
    policy_preact_w(W, pa + rp::B2A, pa + rp::B2B, xq, lane, w, ps, z3_out, z4_out);
}

// ppo.py:698-704 on the sums: means, clamped sample, log-prob of the CLAMPED action.  (e0, e1): the standard-normal draws.
struct Action {
    float a0, a1, logp, mu0, mu1;
};
__device__ __forceinline__ Action policy_finish(const float* __restrict__ tail, float z3, float z4, const float var, const float e0,
                                                const float e1) {   // tail = &params[rp::B2B] or its copy in LDS
    z3 += tail[rp::BO1 - rp::B2B];
    z4 += tail[rp::BO2 - rp::B2B];
    Action o;
    o.mu0 = 1.0f / (1.0f + expf(-z3));
    o.mu1 = tanhf(z4);
    const float sd = sqrtf(var);
    o.a0 = fminf(fmaxf(fmaf(sd, e0, o.mu0), 0.f), 1.f);    // ppo.py:698-703
    o.a1 = fminf(fmaxf(fmaf(sd, e1, o.mu1), -1.f), 1.f);
    const float d0 = o.a0 - o.mu0, d1 = o.a1 - o.mu1;
    o.logp = -0.5f * ((d0 * d0 + d1 * d1) / var) - 1.8378770664093453f - logf(var);  // log-prob of the CLAMPED action, ppo.py:704
    return o;
}

}  // namespace resmlp
