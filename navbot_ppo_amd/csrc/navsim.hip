// navsim.hip -- gfx950 (MI355X / CDNA4) kernels + the C ABI of include/navsim.h.
//
// What runs here replaces, for N envs at once, the reference's
//   Env.step / Env.reset            project_ppo/src/environment_new.py:272-382
//   Env.getOdometry/getState/setReward               ... :138-270
//   the Gazebo diff-drive + ray sensor they wait on   turtlebot3_fake.cpp:110-180 (motion model),
//                                                     turtlebot3_burger.gazebo.xacro:104-127 (LiDAR)
//   PPO.compute_rtgs                project_ppo/src/ppo.py:643-671
//
// Arithmetic contract (DESIGN.md "numerics"): pose, goal geometry, angles, reward in float64
// (the reference's rounding rules -- round-half-even yaw, 1- and 2-decimal Python round() --
// cannot be met in float32); ray/segment tests in float32 with explicit fmaf and one
// correctly-rounded division per hit; min over hits is exact, so the result does not depend
// on how segments are split over lanes.  Compiled with -ffp-contract=off: every fused
// multiply-add is written out.
//
// Kernel layout: one workgroup = 4 waves = EPB (16) consecutive envs.
//   phase 1  wave 0, lane = env : integrate pose (f64), sensor origin + B beam directions -> LDS
//   phase 2  all 4 waves        : nearest hit per (env, beam)
//              shared map  : segments staged in LDS tiles, lane = ray (env, beam); every lane of a wave
//                            reads the same segment at the same time: LDS broadcast, no conflicts
//              per-env map : wave w takes envs 4w..4w+3; lane = segment (16 B/lane coalesced
//                            HBM loads, each segment read exactly once), B running minima per lane,
//                            wavefront min-reduce per beam
//   phase 3  wave 0, lane = env : getState / obs / reward / flags / timeout / auto-reset, state write-back
//   phase 4  all 4 waves        : the block's 64 x (B+6) observation tile leaves LDS as one contiguous,
//                                 fully coalesced store
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "navsim.h"
#include "mlp64_policy.h"
#include "resmlp_policy.h"

namespace {

constexpr double kWheelRadius = 0.033;  // turtlebot3_fake.h:39
constexpr double kWheelSep = 0.160;     // turtlebot3_fake.cpp:44
constexpr double kLidarX = -0.032;      // turtlebot3_burger.urdf.xacro:137
constexpr double kAngleMin = -1.5707975, kAngleMax = 1.5707975;  // gazebo.xacro:113-114
constexpr float kRangeMin = 0.12f, kRangeMax = 3.5f;             // gazebo.xacro:118-119
constexpr int kSubsteps = 6;     // 30 Hz drive updates per 5 Hz scan (gazebo.xacro:62,107)
constexpr int kMaxRects = 16;
constexpr int kMaxGoalTries = 64;

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(NAVSIM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

struct Rects {
    double r[2][kMaxRects][4];  // [which][k][xmin,xmax,ymin,ymax]
    int n[2];
};

struct Params {
    // The first 128 bytes hold what the first memory requests of a step launch need (the pose waves' state loads, the ray waves'
    // first segment tiles): they arrive with the first round of scalar kernarg loads.  In declaration order the compiler reached
    // the state pointers in its fourth dependent round, ~1 us into the launch.
    int N, S, per_env;        // per_env: bit 0 = one map per env, bits 1 / 2 = stream it with non-temporal loads (navsim_set_map)
    int seg_pack_log2;        // cast: log2(lanes per env in one 64-lane pass) = 6, or log2(pow2ceil(S)) when S <= 32
    const float4* seg;        // [S] or [N][S]
    double *th, *x, *y, *gx, *gy, *past_dist, *ep_ret, *ep_path;
    float2* past_action;
    int32_t* ep_step;         // low 30 bits: steps taken in the episode; kRecValid: the records below are current
    uint32_t* rng_ctr;
    const double* beam_cs;    // [2][B]: cos(phi_b), sin(phi_b)
    int B, respawn;
    // ---- 128 bytes
    int max_ep_steps, auto_reset, obs_f16, below_min_mode;
    float sigma;  // LiDAR range noise (0 = off)
    uint32_t key0, key1;
    uint64_t env_id_base;
    double thr, spawn_x, spawn_y, spawn_yaw, goal_lo, goal_hi, diag;
    // next-episode records, written by the step kernel when an env's goal stream has moved, read otherwise:
    double* rec_g;            // [2][3][N] goal x, y and start-to-goal distance
    float4* rec_tail;         // [2][N] the reset observation's (dist/diag, yaw/360, rel_theta/360, diff/180)
    uint2* rec_ck;            // [2][N] (draw counter after the reset, start-pose index)
    double* rsp_g;            // [2][N] the arrival re-spawn goal (respawn_on_arrive)
    uint32_t* rsp_ctr;        // [N] draw counter after the re-spawn draw
    const float4* tile_box;   // shared maps of 65..4096 segments (kept in Morton order by navsim_set_map): per 64-segment tile
                              // the bounding box (xmin, ymin, xmax, ymax) of its segments; else null
    const float* spawn_scan;  // [K][B] or [N][K][B] nearest hits (+inf = none) at the K start poses (K = 1: the cfg spawn pose)
    const float* spawn_obs;   // same shape: the noise-free lidar entries of the reset observation (sanitised scan / 3.5)
    const double* starts;     // [K][3] start poses (x, y, yaw)
    const double* starts_sc;  // [K] the integer-degree yaw of Env.getOdometry at each start pose, evaluated on the device
    const double* goals;      // [G][2] goal points; G == 0: uniform goal box + rejection rectangles
    int K, G;
    double min_dist, max_dist;
    const Rects* rects;
};
static_assert(offsetof(Params, max_ep_steps) == 128, "the hot head of Params");

// ---------------------------------------------------------------- device helpers

// q / s for an integer-valued q with |q| < 2^26 and s in {10, 100} (also k/100 values over 180 or 360): one multiply by
// the rounded reciprocal plus one fma correction gives the correctly rounded quotient -- verified exhaustively over those
// domains by tools/verify/div_by_const_exact.c -- instead of the ~30-instruction IEEE float64 division sequence.
__device__ __forceinline__ double div_const(double q, double s, double inv_s) {
    const double r0 = q * inv_s;
    return fma(fma(-s, r0, q), inv_s, r0);
}

// Python round(x, nd), s = 10^nd: decimal rounding, ties-to-even on the exact binary value
// (environment_new.py:149-150,169-176).  x*s is held exactly as p + err.  Branch-free except for the out-of-domain
// quotient (|x s| >= 2^26: never with map coordinates in metres and angles in degrees).
__device__ __forceinline__ double py_round(double x, double s, double inv_s) {
    const double p = x * s;
    const double err = fma(x, s, -p);
    const double fl = floor(p);
    const double frac = p - fl;
    const double h = fl * 0.5;
    const bool odd = h != floor(h);                     // fl odd -> the even neighbour is fl + 1
    const bool up = (frac > 0.5) || ((frac == 0.5) && ((err > 0.0) || ((err == 0.0) && odd)));
    const double q = up ? fl + 1.0 : fl;
    double r = div_const(q, s, inv_s);
    if (__builtin_expect(!(fabs(q) < 0x1p26), 0)) r = q / s;   // also NaN / inf
    if (r == 0.0) r = copysign(0.0, x);
    return isfinite(x) ? r : x;
}

// yaw of Env.getOdometry, environment_new.py:142-147, for the yaw-only quaternion (qz, qw) = (sin, cos)(theta / 2)
__device__ __forceinline__ double yaw_from_quat4(double qx, double qy, double qz, double qw) {
    const double rad2deg = 180.0 / 3.14159265358979323846;
    double yaw = rint(atan2(2 * (qx * qy + qw * qz), 1 - 2 * (qy * qy + qz * qz)) * rad2deg);  // :142
    if (!(yaw >= 0)) yaw = yaw + 360;                                                            // :144-147
    return yaw + 0.0;   // Python's round() returns the int 0 for -0.4: no negative zero
}
__device__ __forceinline__ double yaw_from_quat(double qz, double qw) { return yaw_from_quat4(0.0, 0.0, qz, qw); }

// The same integer without the quaternion round trip: atan2(sin th, cos th) is th wrapped to (-pi, pi], so the yaw is
// rint(th in degrees, wrapped to [-180, 180]).  The two routes differ by < 1e-9 degree for |th| < 1e7 rad, so they round
// to the same integer unless the angle sits within 1e-6 of a half degree: then (*exact = false) the caller takes the
// quaternion route.
__device__ __forceinline__ double yaw_fast(double th, bool* exact) {
    const double rad2deg = 180.0 / 3.14159265358979323846;
    const double d = th * rad2deg;
    const double m = d - 360.0 * rint(d * (1.0 / 360.0));
    double yaw = rint(m);
    *exact = (fabs(d) < 5e8) && (fabs(fabs(m - yaw) - 0.5) > 1e-6);
    if (!(yaw >= 0)) yaw = yaw + 360;
    return yaw + 0.0;
}

// rel_theta and diff_angle of Env.getOdometry, environment_new.py:149-176; branch-free
__device__ __forceinline__ void goal_rel(double x, double y, double gx, double gy, double yaw, double& rel_theta, double& diff) {
    const double kPi = 3.14159265358979323846;
    const double rad2deg = 180.0 / kPi;
    const double dx = py_round(gx - x, 10.0, 0.1);                                       // :149
    const double dy = py_round(gy - y, 10.0, 0.1);                                       // :150
    const double at = atan(dy / dx);                                                     // :153-168
    double theta = kPi;
    theta = (dy == 0 && dx > 0) ? 0.0 : theta;
    theta = (dx == 0 && dy < 0) ? 1.5 * kPi : theta;
    theta = (dx == 0 && dy > 0) ? 0.5 * kPi : theta;
    theta = (dx < 0 && (dy < 0 || dy > 0)) ? kPi + at : theta;
    theta = (dx > 0 && dy < 0) ? 2 * kPi + at : theta;
    theta = (dx > 0 && dy > 0) ? at : theta;
    rel_theta = py_round(theta * rad2deg, 100.0, 0.01);                                  // :169
    const double d = yaw - rel_theta;                                                    // :170
    const double w = (d >= -180 && d <= 180) ? d : ((d < -180) ? 360 + d : -360 + d);    // :171-176
    diff = py_round(w, 100.0, 0.01);
}

// Env.getOdometry, environment_new.py:138-181, for the yaw-only quaternion Gazebo publishes.
__device__ __forceinline__ void goal_angles_q(double x, double y, double qz, double qw, double gx, double gy,
                                              double& yaw, double& rel_theta, double& diff) {
    yaw = yaw_from_quat(qz, qw);
    goal_rel(x, y, gx, gy, yaw, rel_theta, diff);
}

__device__ __forceinline__ void goal_angles(double x, double y, double th, double gx, double gy,
                                            double& yaw, double& rel_theta, double& diff) {
    goal_angles_q(x, y, sin(th / 2), cos(th / 2), gx, gy, yaw, rel_theta, diff);
}

// Parametric ray / segment test.  o + t d = a + u e  =>  t = cross(a-o, e) / cross(d, e),
// u = cross(a-o, d) / cross(d, e); hit iff t >= 0 and 0 <= u <= 1, decided on the signs of the
// numerators so the only division is the (correctly rounded) t of an actual hit.
__device__ __forceinline__ float ray_seg(float rx, float ry, float ex, float ey, float k, float c, float s) {
    const float den = fmaf(c, ey, -(s * ex));
    const float un = fmaf(rx, s, -(ry * c));
    const bool pos = (den > 0.0f) && (k >= 0.0f) && (un >= 0.0f) && (un <= den);
    const bool neg = (den < 0.0f) && (k <= 0.0f) && (un <= 0.0f) && (un >= den);
    return (pos || neg) ? (k / den) : INFINITY;
}

// LaserScan value from the nearest hit `best` (+inf if none), gazebo.xacro:117-126:
//   >= range_max -> +inf ; < range_min -> range_min (mode 0) or -inf (mode 1, Gazebo's ray sensor) ;
//   otherwise best + sigma * n clamped to [range_min, range_max]  (n = 0 when the noise is off)
__device__ __forceinline__ float sensor_value(float best, float sigma, float n, int below_min_mode) {
    if (!(best < kRangeMax)) return INFINITY;
    if (best < kRangeMin) return below_min_mode ? -INFINITY : kRangeMin;
    const float r = fmaf(sigma, n, best);
    return fminf(fmaxf(r, kRangeMin), kRangeMax);
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Standard-normal draw for LiDAR beam b of env `gid` at (goal draws so far, episode step): Philox4x32-10, one call per
// four beams, Box-Muller in float32.
__device__ __forceinline__ float lidar_noise(uint32_t k0, uint32_t k1, uint64_t gid, uint32_t ctr, uint32_t ep_step, int b) {
    uint32_t r[4];
    philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), ctr, 0x4C000000u | ((ep_step & 0xFFFFu) << 8) | (uint32_t)(b >> 2),
                  k0, k1, r);
    const int pr = (b >> 1) & 1;
    const float u1 = ((float)(r[2 * pr] >> 8) + 1.0f) * 0x1.0p-24f;  // (0, 1]
    const float u2 = (float)(r[2 * pr + 1] >> 8) * 0x1.0p-24f;       // [0, 1)
    const float rad = sqrtf(-2.0f * logf(u1));
    const float ang = 6.283185307179586f * u2;
    return (b & 1) ? rad * sinf(ang) : rad * cosf(ang);
}

// No early exit: the loop bounds and addresses are then wave-uniform (scalar loads from the constant cache, all
// rectangles in flight at once) instead of one dependent L2 round trip per rectangle per lane.
// (`R` is the workgroup's LDS copy in the step kernel -- from global memory every rectangle of every attempt was a
// dependent scalar-load round trip on the one wave the whole workgroup ends up waiting for, 0.8 us per attempt with the
// eight rectangles of stage_2 -- and the table in global memory in the reset kernel.)
__device__ __forceinline__ bool goal_rejected(const Rects& R, int which, double gx, double gy) {
    const int n = R.n[which];
    bool rej = false;
    for (int k = 0; k < n; ++k) {
        const double* q = R.r[which][k];
        rej |= (q[0] <= gx) & (gx <= q[1]) & (q[2] <= gy) & (gy <= q[3]);
    }
    return rej;
}

// goal ~ U(lo,hi)^2 with rejection (environment_new.py:337-345 reset, :245-253 respawn);
// one Philox call per attempt, counter = (env id, draws so far).
template <class PRef>
__device__ __forceinline__ void sample_goal(PRef P, const Rects& R, int i, int which, uint32_t& ctr, double& gx, double& gy) {
    const uint64_t gid = P.env_id_base + (uint64_t)i;
    gx = 0;
    gy = 0;
    for (int tries = 0; tries < kMaxGoalTries; ++tries) {
        uint32_t r[4];
        philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), ctr, 0x6e617673u, P.key0, P.key1, r);
        ctr += 1;
        const double ux = (double)((((uint64_t)r[0] << 32) | r[1]) >> 11) * 0x1.0p-53;
        const double uy = (double)((((uint64_t)r[2] << 32) | r[3]) >> 11) * 0x1.0p-53;
        gx = P.goal_lo + (P.goal_hi - P.goal_lo) * ux;
        gy = P.goal_lo + (P.goal_hi - P.goal_lo) * uy;
        if (!goal_rejected(R, which, gx, gy)) break;
    }
}

// Episode start: which start pose k (index into P.starts) and which goal.
//   G == 0  the reference Env.reset: the single spawn pose and a uniform goal with rectangle rejection.
//   G  > 0  GoalSpawnSampler.sample_start_and_goal (project_ppo/src/spawn_goal_sampler.py:52-62): uniform picks from the
//           start-pose and goal tables until min_dist <= |start - goal| <= max_dist, at most 100 attempts, then one
//           unconditional pick.  One Philox call per attempt.
template <class PRef>
__device__ __forceinline__ void sample_episode(PRef P, const Rects& R, int i, uint32_t& ctr, int& k, double& gx, double& gy) {
    if (P.G == 0) {
        k = 0;
        sample_goal<PRef>(P, R, i, 0, ctr, gx, gy);
        return;
    }
    const uint64_t gid = P.env_id_base + (uint64_t)i;
    for (int tries = 0; tries <= 100; ++tries) {
        uint32_t r[4];
        philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), ctr, 0x6e617673u, P.key0, P.key1, r);
        ctr += 1;
        k = (int)(((uint64_t)r[0] * (uint64_t)P.K) >> 32);
        const int g = (int)(((uint64_t)r[1] * (uint64_t)P.G) >> 32);
        gx = P.goals[2 * g];
        gy = P.goals[2 * g + 1];
        const double dx = P.starts[3 * k] - gx, dy = P.starts[3 * k + 1] - gy;
        const double dist = sqrt(dx * dx + dy * dy);  // np.linalg.norm, spawn_goal_sampler.py:57
        if (tries == 100 || (P.min_dist <= dist && dist <= P.max_dist)) break;
    }
}

constexpr uint32_t kRecValid = 0x40000000u;   // bit of the ep_step word: the cached next-episode records of this env are current
constexpr uint32_t kStepMask = 0x3FFFFFFFu;

template <int NB, int EPB, int NW = 4>
struct StepSmem {
    float2 org[EPB];             // sensor origin per env
    float2 hd[EPB];              // heading (cos, sin) as float32: the cull only (never the ranges)
    float2 dir[NB * EPB];        // [beam][env] unit direction
    unsigned rng[NB * EPB];      // [beam][env] nearest hit as float bits (non-negative floats order like uints)
    float obs[EPB * (NB + 7)];   // [env][B+6 (+1 pad: odd row stride, conflict-free)] the block's output tile
    float noise[NB * EPB];       // [beam][env] standard-normal draws for the range noise (only written when sigma > 0)
    // env state parked by the part-1 lanes for the part-2/3 lanes
    double sv_d[13][EPB];        // x, y, th, gx, gy, past_dist, dist, (7-9 unused), ep_ret, step displacement, ep_path
    float2 sv_act[EPB], sv_pact[EPB];
    uint32_t sv_ctr[EPB];
    uint32_t sv_step[EPB];       // raw ep_step word (kRecValid | step)
    // next-episode records (cached in HBM, see Params::rec_*): [0] the reset that follows a collision / timeout,
    // [1] the reset that follows an arrival when the arrival re-spawn draw comes first (respawn_on_arrive)
    double sp_d[2][6][EPB];      // start x, y, yaw, goal x, y, distance
    float4 sp_tail[2][EPB];      // the four goal entries of the reset observation
    uint32_t sp_ctr[2][EPB];
    int sp_k[2][EPB];
    double sp_rg[2][EPB];        // the arrival re-spawn goal (environment_new.py:245-253) and the draw counter after it
    uint32_t sp_rctr[EPB];
    float sp_scan[2][EPB][NB];   // lidar entries of the reset observation (SENS: the raw nearest hits at the start pose)
    unsigned mn_bits[EPB];       // min over the sanitised scan as float bits, and whether a reading is negative (-inf, SENS)
    unsigned neg[EPB];
    // cast: per-wave queue of the segments that survive the cull (ring of 128)
    float4 q4[NW][128];
    unsigned qe[NW][128];
    float4 r4[NW][128];          // ... and of those whose angular extent holds at least one beam
    unsigned re[NW][128];        //     env | first beam of that extent << 8 | number of beams << 16
    unsigned pe[NW][(NB > 16) ? 128 : 1];   // 36 beams: stage-B work entries: r4 slot | first beam << 8 | beams to test << 16
    float4 tbox[64];             // Params::tile_box, staged once per launch (BOXES)
    Rects rects;                 // Params::rects, staged by the ray waves before barrier A (the spec lanes' goal rejection test)
    float2 act_l[EPB];           // persistent rollout: the action the policy phase chose for this step
    double beam[2 * NB];         // Params::beam_cs, staged by the pose waves (persistent rollout: once)
    // persistent rollout: the envs' state lives here between the steps (HBM sees it before the first and after the last)
    double st_d[8][EPB];         // x, y, th, gx, gy, past_dist, ep_ret, ep_path
    float2 st_pact[EPB];
    uint32_t st_step[EPB], st_ctr[EPB];
};

// Correctly rounded K / Dn for positive, normal-range operands (Dn in [2^-60, 2^20], K in {0} U [2^-60, 2^20]):
// the reciprocal-refinement sequence the compiler emits for an IEEE float32 divide (v_rcp, 2 fma to refine,
// quotient, two fma residual corrections), minus its exponent pre-scaling (v_div_scale x2) and special-case
// fix-up (v_div_fixup, v_div_fmas), which are no-ops in this range: 22 instead of 36 pipe cycles on gfx950
// (measured: v_fma/v_mul full rate, v_div_* half rate, v_rcp quarter rate).  Dn == 0 gives NaN.
__device__ __forceinline__ float div_pos(float K, float Dn) {
    float r = __builtin_amdgcn_rcpf(Dn);
    const float f0 = fmaf(-Dn, r, 1.0f);
    r = fmaf(f0, r, r);
    float q = K * r;
    float e = fmaf(-Dn, q, K);
    q = fmaf(e, r, q);
    e = fmaf(-Dn, q, K);
    return fmaf(e, r, q);
}

// One ray/segment test, same decisions and the same range bits as ray_seg(), built from full-rate VALU ops
// only (v_cmp / v_cndmask / v_min_f32 are half rate on gfx950):
//   p1 = k*den + 0 >= 0  <=>  k and den agree in sign or k == 0   (no underflow: |k|,|den| are 0 or >= 2^-56;
//   p2 = un*den + 0 >= 0,  w = |den| - |un| >= 0  <=>  0 <= u <= 1  the "+ 0" turns a -0 product into +0)
//   t  = |k| / |den|  ==  k / den  when the signs agree; >= +0, +inf or NaN, so its bit pattern orders like a uint.
// miss <=> sign bit of (p1 | p2 | w); an arithmetic shift makes that an all-ones mask OR-ed into t's bits, and the
// running nearest hit is an unsigned min over bit patterns (inf = 0x7f800000 is the identity, NaN never wins).
__device__ __forceinline__ unsigned ray_seg_bits(float rx, float ry, float ex, float ey, float k, float c, float s) {
    const float den = fmaf(c, ey, -(s * ex));
    const float un = fmaf(rx, s, -(ry * c));
    const float p1 = fmaf(k, den, 0.0f);
    const float p2 = fmaf(un, den, 0.0f);
    const float w = fabsf(den) - fabsf(un);
    const unsigned miss = (unsigned)((int)(__float_as_uint(p1) | __float_as_uint(p2) | __float_as_uint(w)) >> 31);
    return __float_as_uint(div_pos(fabsf(k), fabsf(den))) | miss;
}

// Beam coordinate of a point (x ahead, y left) in the robot frame: f = (theta + A) / delta, beam b sits at f = b.
// theta = 2 atan(u) with the half-angle tangent u = y / (r + x), monotone over (-pi, pi) and clamped to |u| <= 1.1 (just
// past the +-90 degree fan edge, where only the side matters); atan by a degree-9 odd polynomial, |error| < 2.4e-5 rad.
// Used ONLY to decide which segments are tested exactly, with a margin 40x above its error.  *bad: the point is within
// 3 cm of the sensor or within 4.5e-3 rad of straight behind it, where u is ill-conditioned -- the caller keeps the segment.
__device__ __forceinline__ float beam_coord(float x, float y, float two_inv_delta, float a_inv_delta, bool* bad) {
    const float r = __builtin_amdgcn_sqrtf(fmaf(x, x, y * y));
    const float d = r + x;
    *bad = !(d > 1e-5f * r) || (r < 0.03f);
    float u = y * __builtin_amdgcn_rcpf(d);
    u = __builtin_amdgcn_fmed3f(u, -1.1f, 1.1f);
    const float z = u * u;
    float p = fmaf(0.01608042f, z, -0.07488631f);
    p = fmaf(p, z, 0.1729194f);
    p = fmaf(p, z, -0.32846571f);
    p = fmaf(p, z, 0.99974538f);
    return fmaf(u * p, two_inv_delta, a_inv_delta);
}

// (A workgroup reserves ceil(waves / 4) slots on EVERY SIMD of its CU, so a 5-wave block costs as much residency as an 8-wave
// one: measured 1 block/CU.)

// lidar part of an observation row (environment_new.py:289-294) from nearest hits `best` (stride `bstride`);
// returns min(sanitised scan) for the collision rule (:200)
__device__ __forceinline__ float write_lidar(float* row, const float* best, int bstride, const float* noise, int nstride,
                                             float sigma, int below_min_mode, int B) {
    float mn = INFINITY;
    for (int b = 0; b < B; ++b) {
        float r = sensor_value(best[b * bstride], sigma, noise ? noise[b * nstride] : 0.f, below_min_mode);
        if (r == INFINITY) r = 3.5f;  // :193-194 (+inf only: -inf passes through like in the reference; NaN cannot occur)
        mn = r < mn ? r : mn;
        // (float)((double)r / 3.5) == r / 3.5f : double rounding is innocuous for a quotient of
        // two float32 values (53 >= 2*24+2), so the float32 divide gives the reference's bits.
        row[b] = r / 3.5f;            // :289
    }
    return mn;
}

// ---------------------------------------------------------------- the step kernel
// One workgroup = 4 waves = EPB consecutive envs (EPB = 16: waves 0-1 are the "pose waves", 2-3 the "ray waves").
//   part 1   pose waves, 8 lanes per env: state loads, float64 motion, the 8 sincos of the step one per lane -> LDS
//            sensor origin + beam directions; the env's state is parked in LDS for its part-2/3 lane.
//            ray waves: the first segment loads of their first work item (nothing there depends on the pose)
//   barrier A
//   part 2   lane-dense (a wave instruction costs the same with 8 or 64 active lanes): wave 0, lane = env: goal geometry
//            of the new pose; last pose wave: the next-episode records of every env -- read back from HBM, or, for an env
//            whose goal stream moved since they were written, recomputed (Philox draw + start-pose geometry) and stored.
//            Meanwhile the cast, one work item = one env (several envs when the map has <= 32 segments), taken from an
//            LDS counter; lane = segment, 16 B/lane coalesced, every segment read from HBM once, two tiles ahead:
//              stage A  cull: a segment that lies wholly behind the +-90 degree beam fan, or farther than the 3.5 m sensor
//                       range, cannot change the scan (ranges >= 3.5 read as "no return", gazebo.xacro:119); the cull
//                       is conservative (margins far above float32 noise) so the result keeps its bits.  Survivors
//                       (about a quarter) are compacted into a per-wave LDS queue.
//              stage B  whenever 64 survivors are queued: the exact ray/segment test of every beam, nearest hit per
//                       (env, beam) by LDS atomic-min over float bit patterns.
//   barrier B
//   part 3   wave 0, lane = env: rules of getState / step / setReward, episode logic, state stores
//   barrier C
//   all:     the EPB x (B+6) observation tile leaves LDS as one contiguous, fully coalesced store.
// SENS = false compiles the sensor-fidelity options (range noise, -inf below range_min) out.
// PERSIST: called once per step by the persistent rollout kernel; the action comes from the policy phase through LDS and
// the env state stays in LDS (sm.st_*) from step to step; `last_step` also writes it back to HBM.
// NW: waves per workgroup (4; the persistent rollout kernel, one workgroup per CU, runs 8 for a shorter cast).
// BOXES: shared map with tile bounding boxes (Params::tile_box): whole 64-segment tiles that lie behind the beam fan or out of
// range are skipped without being loaded (the house map: 32 tiles, ~5 of them near any one pose).
// row0: element offset of this step's row in the [T, N] output buffers `io` points at (navsim_step_seq: t N; the pointers are
// then read from the kernarg segment at their use instead of living in SGPRs through the body; 0 elsewhere).
// PRef / IORef: how the parameter block and the I/O pointers are reached.  The persistent rollout passes plain references to its
// own copies.  step_kernel passes references into the kernarg segment (constant address space) behind a compiler barrier: every
// field is then a scalar load AT ITS USE.  As ordinary by-value kernel parameters all ~100 dwords were loaded and spilled to
// VGPR lanes in the entry block (not enough SGPRs), in five dependent rounds, before the first state load could leave: 1.1 us.
struct StepIO {
    const float2* action;
    const float2* past_override;
    void* obs_out;
    float* reward;
    uint8_t *done, *arrive, *ended;
    float* ep_return;
    int32_t* ep_length;
    float* ep_path_out;
};
// Hook: called by EVERY wave right behind barrier B2 (the observation rows are complete but for a reset, which the rules lanes of
// wave 0 work out next): the persistent rollout kernels run the policy of the NEXT step there, on waves that would otherwise
// wait for the rules, reading the rows through next_obs_n below.  NoHook: nothing (step_kernel, steps_kernel).
struct NoHook {
    __device__ __forceinline__ void operator()(int, int) const {}
};
template <int NB, int EPB, bool SENS, bool PERSIST, int NW = 4, bool BOXES = false, int PAIR = 0, class PRef = const Params&,
          class IORef = const StepIO&, class Hook = NoHook>
__device__ __forceinline__ void step_body(PRef P, StepSmem<NB, EPB, NW>& sm, int& next_env, IORef io, const bool last_step = true,
                                          const size_t row0 = 0, Hook hook = Hook()) {
    static_assert(EPB <= 64 && EPB >= 4 && NB % 2 == 0, "EPB / NB");
    static_assert(!(PAIR && BOXES), "tile boxes describe 64-segment tiles");
    constexpr int kThreads = 64 * NW;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // readfirstlane tells the compiler what it cannot see: the wave index is wave-uniform, so the work positions, the loop
    // exits and every `wave < PW` test below live in SGPRs and scalar branches instead of VGPRs and exec-mask loops
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int base = blockIdx.x * EPB;
    constexpr int B = NB;       // host dispatch guarantees P.B == NB
    constexpr int D = B + 6;
    constexpr int DP = D + 1;   // padded LDS row stride
    const int nloc = min(EPB, P.N - base);  // envs in this block
    const unsigned kInfBits = 0x7f800000u;
    // Pose lanes: TWO lanes per env (four in the small shapes).  The step needs the heading's (cos, sin) at the six substep arguments th_k + dth/2 and at the
    // final heading; lane 0 of the pair evaluates sincos(final heading), lane 1 sincos(dth / 2), and the six substep values follow
    // by rotating back from the final heading (angle addition, explicit fma: |error| ~ 1e-15, i.e. 1e-17 m on the pose).  The beam
    // directions and the sensor origin -- everything the scan's bits depend on -- come from the library sincos of the final heading,
    // exactly as in the oracle.  Rounds 1-3 spent 8 lanes per env (one library sincos per substep and lane): 8 pose waves per 64
    // envs, i.e. four times the float64 instruction issue for the same chain length; at configs[2] the pose phase was a fifth of the
    // launch's vector-unit time and its second wave per SIMD reached barrier A 0.5 us behind the first.
    // Lane 0 of each pair owns the env's float64 state across the phases.
    constexpr int LPE = (EPB <= 16) ? 4 : 2;            // pose lanes per env (lanes 2, 3 of a quad only share the beam directions:
                                                        // the small shapes are latency chains, the big ones issue-bound)
    constexpr int PWP = (LPE * EPB + 63) / 64;          // pose waves
    double x = 0, y = 0, th = 0, gx = 0, gy = 0, pdist = 0, dist = 0, yaw = 0, rel_theta = 0, diff = 0, ret0 = 0, path0 = 0;
    float2 act = make_float2(0.f, 0.f), pact = make_float2(0.f, 0.f);
    uint32_t ctr = 0, stepw = 0;
    const int el_pose = (64 / LPE) * wave + (lane / LPE);  // env (local) this pose lane works for
    const int rr = lane % LPE;
    const bool pose_lane = (wave < PWP) && (el_pose < EPB);
    const int i = base + el_pose;
    constexpr int kSpecWave0 = (EPB <= 8) ? 0 : PWP;
    constexpr int kSpecLane0 = (EPB <= 8) ? EPB : 0;
    constexpr int PW = (EPB <= 8) ? 1 : PWP + (2 * EPB + 63) / 64;   // "front" waves: pose + spec; the others are ray waves
    static_assert(PW < NW, "front waves + at least one ray wave");
    static_assert(kSpecLane0 + ((EPB <= 8) ? 2 * EPB : 0) <= 64, "spec lanes of the small shapes fit wave 0");
    // ---- cast geometry: lane -> (env of the pass, segment of the tile)
    const int spl = P.seg_pack_log2;            // log2(lanes per env in one pass): 6, smaller when the map has <= 32 segments
    const int epp = 64 >> spl;                  // envs per pass
    const int sub = lane >> spl;
    const int j0 = lane & ((1 << spl) - 1);
    const int n_items = (nloc + epp - 1) / epp;
    // (locals: read through the kernarg segment, P's fields are scalar loads at every use -- these sit in the cast loop)
    const int S = P.S;
    const bool per_env = P.per_env != 0;
    // Non-temporal hint on the per-env segment stream (128-segment passes), by measurement (profiles/r04_nontemporal_loads_ablation.txt,
    // 16384 envs, us per step, default / non-temporal loads):
    //   one launch per step   S=128 13.11 / 12.64   S=256 18.58 / 17.62   S=512-1024 equal   S=1280 66.2 / 60.8   S=1536 82.4 / 71.8
    //   persistent (tape)     S=128  9.74 / 10.93   S=256 14.69 / 16.91   S=1024 41.6 / 48.2  S=1280 equal         S=1536 73.8 / 71.5
    // A launch reads every segment once and the next launch finds nothing of a >= 35 MB stream in the 8 x 4 MB of L2 (LRU), so
    // allocating the lines there is pure overhead: step_kernel streams (PAIR == 2) whenever the per-env stream is >= 32 MiB AND the shape is a
    // big 10-beam one (32 / 64 envs per workgroup: pick_shape; the ablation above was taken there -- the 8 / 16-env shapes of shards up to
    // 8192 envs and the 36-beam shapes keep the default policy).  A persistent workgroup re-reads ITS envs' segments every
    // step from its own XCD's L2 / the Infinity Cache (traffic 0.77 x algorithmic at configs[2]) and keeps the default policy until
    // one step's stream exceeds 1.25 x the Infinity Cache (bit 1 of per_env, navsim_set_map), where nothing can be re-used either.
    // (Below 32 MiB the stream fits the L2s and is left there: bit 2.)  The host picks the PAIR == 2 instantiation accordingly.
    const float4* const segs = P.seg;
    const int ntiles = (S + 63) >> 6;
    // PAIR (maps of more than 64 segments, one env per pass): a pass takes 128 segments, two per lane (j and j + 64) -- the
    // work-queue, address and loop instructions of a pass, about as many as its arithmetic, are paid once per 128 segments
    const int ntl = PAIR ? (ntiles + 1) >> 1 : ntiles;   // passes per item
    struct Pos { int it, t; };
    auto grab = [&]() __attribute__((always_inline)) {
        int v = 0;
        if (lane == 0) v = atomicAdd(&next_env, 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
    // Requests are unconditional (a position past the end re-reads a clamped, valid address and is flagged invalid): the count of
    // requests issued after any given one is then the same on every path, which is what lets `s_waitcnt vmcnt(n)` -- the counter
    // retires in order -- wait for the OLDEST slot only and leave the two younger requests in flight.  With exec-masked loads the
    // compiler had to assume the path that skipped them and waited for vmcnt(0) before every tile: no prefetch at all.
    auto ld = [&](const Pos p, float4& g, bool& v) __attribute__((always_inline)) {
        const int el = p.it * epp + sub, j = (p.t << 6) + j0;
        v = (p.it < n_items) && (el < nloc) && (j < S);
        g = segs[(per_env ? (size_t)(base + min(el, nloc - 1)) * (size_t)S : (size_t)0) + (size_t)min(j, S - 1)];
    };
    auto ld2 = [&](const Pos p, float4& ga, bool& va, float4& gb, bool& vb) __attribute__((always_inline)) {
        const int j = (p.t << 7) + lane;
        const bool item = (p.it < n_items) && (p.it < nloc);
        va = item && (j < S);
        vb = item && (j + 64 < S);
        const float4* src = segs + (per_env ? (size_t)(base + min(p.it, nloc - 1)) * (size_t)S : (size_t)0);
        if constexpr (PAIR == 2) {   // the stream carries the non-temporal hint (a compile-time variant: as a run-time branch the
                                     // compiler waited for vmcnt(0) in front of every request, i.e. no tile stayed in flight)
            typedef float v4f_nt __attribute__((ext_vector_type(4)));
            const v4f_nt* srcv = reinterpret_cast<const v4f_nt*>(src);
            const v4f_nt a4 = __builtin_nontemporal_load(srcv + min(j, S - 1));
            const v4f_nt b4 = __builtin_nontemporal_load(srcv + min(j + 64, S - 1));
            ga = make_float4(a4.x, a4.y, a4.z, a4.w);
            gb = make_float4(b4.x, b4.y, b4.z, b4.w);
        } else {
            ga = src[min(j, S - 1)];
            gb = src[min(j + 64, S - 1)];
        }
    };
    Pos p0 = {n_items, 0}, p1 = {n_items, 0}, p2 = {n_items, 0};
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0, g0b = g0, g1b = g0, g2b = g0;
    bool v0 = false, v1 = false, v2 = false, v0b = false, v1b = false, v2b = false;
    auto request = [&](const Pos p, float4& ga, bool& va, float4& gb, bool& vb) __attribute__((always_inline)) {
        if constexpr (PAIR) ld2(p, ga, va, gb, vb);
        else ld(p, ga, va);
    };

    // ---------------- the first memory requests of the launch, ahead of everything else: what they need sits in the first 128
    // bytes of Params, i.e. in the first round of scalar kernarg loads (the rest of the set-up below costs three more rounds)
    if (wave >= PW) {
        // ray waves: static positions k = 0, 1, 2 of ray wave r: item r + RW (k / ntiles), tile k % ntiles
        constexpr int RW = NW - PW;
        const int r = wave - PW;
        p0 = Pos{r, 0};
        p1 = (ntl >= 2) ? Pos{r, 1} : Pos{r + RW, 0};
        p2 = (ntl >= 3) ? Pos{r, 2} : ((ntl == 2) ? Pos{r + RW, 0} : Pos{r + 2 * RW, 0});
        request(p0, g0, v0, g0b, v0b);
        request(p1, g1, v1, g1b, v1b);
        request(p2, g2, v2, g2b, v2b);
    }
    // pose waves: the beam table goes through LDS (lane k requests entry k here, right behind the state; written to LDS behind the
    // sincos, read back per beam): held in registers per lane it
    // cost 4 NB / LPE VGPRs from here to the end of part 1.  (The persistent rollout stages it once, before its first step.)
    constexpr int kBeamLd = (2 * NB + 63) / 64;
    double beam_ld[kBeamLd];
    if (wave < PWP) {
        if (!PERSIST && pose_lane && el_pose < nloc) {
            th = P.th[i];
            act = io.action[i];
            ctr = P.rng_ctr[i];
            stepw = (uint32_t)P.ep_step[i];
            if (rr == 0) {
                x = P.x[i]; y = P.y[i];
                gx = P.gx[i]; gy = P.gy[i]; pdist = P.past_dist[i];
                pact = io.past_override ? io.past_override[i] : P.past_action[i];
                ret0 = P.ep_ret[i];
                path0 = P.ep_path[i];
            }
        }
        if (!PERSIST) {
#pragma unroll
            for (int q = 0; q < kBeamLd; ++q) beam_ld[q] = P.beam_cs[min(lane + 64 * q, 2 * NB - 1)];
        }
    }

    // sensor-fidelity options (range noise, -inf below range_min) are compiled out of the default instantiation:
    // carrying them as run-time branches cost 1.2 us per launch (28.2 -> 27.0 us, configs[2])
    const float sigma = SENS ? P.sigma : 0.f;
    const int below_min = SENS ? P.below_min_mode : 0;


    // Parts 2 and 3 run lane-dense (a wave instruction costs the same with 8 or 64 active lanes, and the float64 geometry is
    // ~1000 of them): the goal geometry and the rules of ALL the block's envs on lanes 0..EPB-1 of wave 0 (the "owners"); the
    // next-episode records on 2 EPB "spec" lanes -- behind the owners in wave 0 for the small shapes, else on the waves that
    // follow the pose waves (they have nothing else to do before barrier A, so their record requests leave at kernel entry).
    // State crosses from the part-1 lanes through LDS (sv_*).  PW = the "front" waves: pose + spec; the others are ray waves.
    const bool own = (wave == 0) && (lane < nloc);            // lane = env for the geometry / rules lanes
    const int n_rec = P.respawn ? 2 : 1;
    // spec lanes: record spec_c of env spec_e
    const int sl = (wave - kSpecWave0) * 64 + lane - kSpecLane0;
    const int spec_c = sl / EPB;
    const int spec_e = sl % EPB;
    const bool spec_lane = (wave >= kSpecWave0) && (wave < PW) && (sl >= 0) && (spec_c < n_rec) && (spec_e < nloc);

    // The spec lanes request the cached records at kernel entry, whether they will turn out current or not: the first of
    // the two dependent round trips of the record path then runs under the pose phase instead of after barrier A, where the
    // spec wave is the last one into the cast.
    const bool spec = spec_lane && (P.auto_reset || (spec_c == 1));
    double pf_rgx = 0, pf_rgy = 0, pf_g0 = 0, pf_g1 = 0, pf_g2 = 0;
    float4 pf_tl = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned long long pf_ck = 0ull;   // (counter, start-pose index) kept as ONE 64-bit value until it is used: a uint2 was split
                                       // right behind the load, which made the wave wait for the whole round trip there
    uint32_t pf_rctr = 0;
    auto prefetch_records = [&]() __attribute__((always_inline)) {
        if (spec) {   // (the persistent rollout calls this after barrier A: measured slower there at the top of the step)
            const int ie = base + spec_e, c = spec_c;
            const size_t N = (size_t)P.N;
            if (c == 1) {
                pf_rgx = P.rsp_g[ie]; pf_rgy = P.rsp_g[N + ie]; pf_rctr = P.rsp_ctr[ie];
            }
            if (P.auto_reset) {
                const double* rg = P.rec_g + (size_t)c * 3 * N + ie;
                pf_g0 = rg[0]; pf_g1 = rg[N]; pf_g2 = rg[2 * N];
                pf_tl = P.rec_tail[(size_t)c * N + ie];
                pf_ck = reinterpret_cast<const unsigned long long*>(P.rec_ck)[(size_t)c * N + ie];
            }
        }
    };

    if (wave < PW) {
        // ---------------- pose lanes, part 1: motion + sensor frame
        // spec lanes on a wave of their own: the cached records are requested at kernel entry
        if constexpr (EPB > 8) { if (!PERSIST) prefetch_records(); }
        double delta_s = 0, delta_theta = 0, arg = 0;
        if (pose_lane && el_pose < nloc) {
            if (PERSIST) {
                const int e = el_pose;
                th = sm.st_d[2][e];
                act = sm.act_l[e];
                ctr = sm.st_ctr[e];
                stepw = sm.st_step[e];
                if (rr == 0) {
                    x = sm.st_d[0][e]; y = sm.st_d[1][e];
                    gx = sm.st_d[3][e]; gy = sm.st_d[4][e]; pdist = sm.st_d[5][e];
                    pact = sm.st_pact[e];
                    ret0 = sm.st_d[6][e];
                    path0 = sm.st_d[7][e];
                }
            }
            // environment_new.py:273-278
            const double v = (double)act.x / 4;
            const double w = (double)act.y;
            // turtlebot3_fake.cpp:117-118, :133-146, :154-155
            const double vl = v - (w * kWheelSep / 2);
            const double vr = v + (w * kWheelSep / 2);
            const double dt = 1.0 / 30.0;
            const double wl = vl / kWheelRadius, wr = vr / kWheelRadius;
            const double wheel_l = wl * dt, wheel_r = wr * dt;
            delta_s = kWheelRadius * (wheel_r + wheel_l) / 2.0;
            delta_theta = kWheelRadius * (wheel_r - wheel_l) / kWheelSep;
            // the heading after the step is th0 + kSubsteps additions of delta_theta (same roundings as the serial loop, :160)
            for (int k = 0; k < kSubsteps; ++k) th += delta_theta;
            arg = (rr == 0) ? th : (delta_theta / 2.0);
        }
        // spec lanes that share wave 0 with the pose lanes (small shapes): behind the wheel arithmetic -- the wait for the state
        // loads is over, the sincos covers this round trip
        if constexpr (EPB <= 8) { if (!PERSIST) prefetch_records(); }
        double cth = 1.0, sth = 0.0;
        if (pose_lane) {
            double sn, cs;
            sincos(arg, &sn, &cs);
            // lane 0's and lane 1's pair to every lane of the env's group through DPP quad permutes: no LDS round trip on the
            // chain to barrier A
            auto bcast = [](const double vv, auto sel) __attribute__((always_inline)) {
                constexpr int kCtrl = (LPE == 4) ? (decltype(sel)::value ? 0x55 : 0x00)     // quad_perm [k, k, k, k]
                                                 : (decltype(sel)::value ? 0xF5 : 0xA0);    // quad_perm [k, k, 2 + k, 2 + k]
                const int lo = __builtin_amdgcn_mov_dpp(__double2loint(vv), kCtrl, 0xF, 0xF, true);
                const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(vv), kCtrl, 0xF, 0xF, true);
                return __hiloint2double(hi, lo);
            };
            cth = bcast(cs, std::integral_constant<int, 0>{});   // final heading
            sth = bcast(sn, std::integral_constant<int, 0>{});
            const double ch = bcast(cs, std::integral_constant<int, 1>{}), sh = bcast(sn, std::integral_constant<int, 1>{});    // half a substep's turn
            if (rr == 0) {
                // substep k integrates along th_k + dth / 2 = final heading - (kSubsteps - 1/2 - k) dth  (:158-159): rotate back
                // by dth / 2 once, then by dth (double angle) per substep; the displacements are summed as they come, last substep
                // first (the serial loop's order would keep all six pairs alive)
                const double cd = fma(ch, ch, -(sh * sh)), sd = 2.0 * (sh * ch);
                double c = fma(cth, ch, sth * sh), s = fma(sth, ch, -(cth * sh));
                double dx = delta_s * c, dy = delta_s * s;
#pragma unroll
                for (int k = kSubsteps - 2; k >= 0; --k) {
                    const double c1 = fma(c, cd, s * sd);
                    s = fma(s, cd, -(c * sd));
                    c = c1;
                    dx += delta_s * c;
                    dy += delta_s * s;
                }
                x += dx;
                y += dy;
                const double ox = x + kLidarX * cth;
                const double oy = y + kLidarX * sth;
                sm.org[el_pose] = make_float2((float)ox, (float)oy);
                sm.hd[el_pose] = make_float2((float)cth, (float)sth);
                if (el_pose < nloc) {  // hand the env over to its geometry / rules lane
                    const int e = el_pose;
                    sm.sv_d[0][e] = x; sm.sv_d[1][e] = y; sm.sv_d[2][e] = th; sm.sv_d[3][e] = gx; sm.sv_d[4][e] = gy;
                    sm.sv_d[5][e] = pdist; sm.sv_d[10][e] = ret0;
                    sm.sv_d[11][e] = sqrt(dx * dx + dy * dy);   // np.linalg.norm(curr_pos - prev_pos), ppo.py:536-537
                    sm.sv_d[12][e] = path0;
                    sm.sv_act[e] = act; sm.sv_pact[e] = pact; sm.sv_ctr[e] = ctr; sm.sv_step[e] = stepw;
                }
            }
        }
        if (!PERSIST && wave < PWP) {
#pragma unroll
            for (int q = 0; q < kBeamLd; ++q)
                if (lane + 64 * q < 2 * NB) sm.beam[lane + 64 * q] = beam_ld[q];
        }
        if (pose_lane) {
            // the table entries just written by this wave (LDS operations of a wave complete in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            constexpr int kBeamIt = (NB + LPE - 1) / LPE;
            double bc[kBeamIt], bs[kBeamIt];
#pragma unroll
            for (int q = 0; q < kBeamIt; ++q) {   // all reads first: one LDS latency
                const int b = min(rr + LPE * q, NB - 1);
                bc[q] = sm.beam[b];
                bs[q] = sm.beam[NB + b];
            }
#pragma unroll
            for (int q = 0; q < kBeamIt; ++q) {
                const int b = rr + LPE * q;
                if (b < NB) {
                    const double c = cth * bc[q] - sth * bs[q];
                    const double s = sth * bc[q] + cth * bs[q];
                    sm.dir[b * EPB + el_pose] = make_float2((float)c, (float)s);
                }
            }
        }
        // the ray waves' first three tiles are pre-assigned (requested before barrier A): 1, 2 or 3 items per ray wave
        if (tid == 0) next_env = (NW - PW) * (ntl == 1 ? 3 : (ntl == 2 ? 2 : 1));
    }
    if (wave >= PW) {
        // ---------------- other waves, part 1 (their tile requests went out at the top): LDS traffic under that round trip
        for (int k = tid - 64 * PW; k < NB * EPB; k += kThreads - 64 * PW) sm.rng[k] = kInfBits;
        if (tid - 64 * PW < EPB) {
            sm.mn_bits[tid - 64 * PW] = kInfBits;
            sm.neg[tid - 64 * PW] = 0u;
        }
    }
    // the goal rejection rectangles (the spec lanes' recompute path reads them after barrier A) and the tile boxes: staged by the
    // waves behind the pose waves -- spec waves where the shape has them (their only other work before barrier A is the record
    // request), else the ray waves, behind their tile requests.  A ray wave that waits for this round trip also waits for its
    // three tiles (the counter retires in order), which is why the big shapes keep it off them.  (The persistent rollout stages
    // the rectangles once, before its first step.)
    if (wave >= PWP) {
        constexpr int kStageThreads = 64 * (NW - PWP);
        const int st = tid - 64 * PWP;
        constexpr bool kOnFront = (PW > PWP);   // staging waves = the front waves behind the pose waves
        const bool mine = kOnFront ? (wave < PW) : true;
        constexpr int kStride = kOnFront ? 64 * (PW - PWP) : kStageThreads;
        if (mine) {
            if (BOXES) for (int k = st; k < ntiles; k += kStride) sm.tbox[k] = P.tile_box[k];
            static_assert(sizeof(Rects) % 8 == 0, "Rects is copied in 8-byte words");
            if (!PERSIST)
                for (int k = st; k < (int)(sizeof(Rects) / 8); k += kStride)
                    reinterpret_cast<uint64_t*>(&sm.rects)[k] = reinterpret_cast<const uint64_t*>(P.rects)[k];
        }
    }
    __syncthreads();  // barrier A: origins / directions visible, work counter set

    if (sigma > 0.f && pose_lane && el_pose < nloc) {
        // range noise for this step: lane rr draws beams rr, rr + 2, ... (off the critical path: the others ray-cast)
        for (int b = rr; b < B; b += LPE)
            sm.noise[b * EPB + el_pose] = lidar_noise(P.key0, P.key1, P.env_id_base + (uint64_t)i, ctr, stepw & kStepMask, b);
    }
    // ---------------- part 2 (the other waves are already ray-casting): goal geometry.
    // An owner lane works on the pose its env just moved to.  The spec lanes hold the episode that starts if this step
    // ends the current one (ppo.py:582-593 + Env.reset, environment_new.py:312-382): an episode draws from the goal stream
    // only when it ends, so its successor is known from the moment it starts -- record 0 for an end by collision or
    // timeout (and by arrival when respawn_on_arrive is off), record 1 for an end by arrival after the arrival re-spawn
    // draw of :245-253.  Records live in HBM; kRecValid in the env's ep_step word says they match its draw counter.
    // A spec lane of an env whose flag is clear (it was reset, or its goal stream moved) recomputes and stores them.
    // (wave < PW is implied; spelt out so that the ray waves' path to the cast holds no request the compiler would have to
    // count: their three tile requests stay the youngest ones and the first cull waits for vmcnt(2), not vmcnt(0))
    if ((wave < PW) && (own || spec)) {
        const int e = own ? lane : spec_e;
        const int ie = base + e;
        if (own) {
            const double px = sm.sv_d[0][e], py = sm.sv_d[1][e], tgx = sm.sv_d[3][e], tgy = sm.sv_d[4][e];
            // yaw: rint of the heading in degrees; the quaternion route of :142 only when the two could round differently
            const double pth = sm.sv_d[2][e];
            bool exact;
            yaw = yaw_fast(pth, &exact);
            if (__builtin_expect(!exact, 0)) yaw = yaw_from_quat(sin(pth / 2), cos(pth / 2));
            goal_rel(px, py, tgx, tgy, yaw, rel_theta, diff);
            dist = hypot(tgx - px, tgy - py);  // environment_new.py:203 ; getGoalDistace, :116-120
            sm.sv_d[6][e] = dist;
            // the six non-lidar entries of the observation row depend on the pose only: written here, under the cast, instead
            // of in part 3 (the float64 division and the three exact constant divisions were a fifth of its dependent chain)
            float* row = sm.obs + e * DP;
            const float2 pa = sm.sv_pact[e];
            row[B + 0] = pa.x;                          // environment_new.py:299-300
            row[B + 1] = pa.y;
            row[B + 2] = (float)(dist / P.diag);        // :301
            row[B + 3] = (float)div_const(yaw, 360.0, 1.0 / 360.0);        // yaw / 360, rel_theta / 360, diff_angle / 180:
            row[B + 4] = (float)div_const(rel_theta, 360.0, 1.0 / 360.0);  // correctly rounded over these operands' domains
            row[B + 5] = (float)div_const(diff, 180.0, 1.0 / 180.0);       // (integers, k/100), tools/verify
        } else {
            const int c = spec_c;
            const size_t N = (size_t)P.N;
            double px = 0, py = 0, pth = 0, tgx = 0, tgy = 0, rgx = 0, rgy = 0, rdist = 0;
            float4 tl = make_float4(0.f, 0.f, 0.f, 0.f);
            uint32_t sctr = sm.sv_ctr[e], rctr = sctr;
            int sk = 0;
            // the scan every reset at start pose k observes: rows of NB floats are 8-byte aligned (NB even), all NB / 2 requests
            // go out before the first is awaited (one memory latency instead of NB / 2)
            float2 row[NB / 2];
            auto request_row = [&](const int k) __attribute__((always_inline)) {
                const float2* sp = reinterpret_cast<const float2*>((SENS ? P.spawn_scan : P.spawn_obs) +
                                                                   ((P.per_env ? (size_t)ie * P.K : 0) + k) * B);
#pragma unroll
                for (int b = 0; b < NB / 2; ++b) row[b] = sp[b];
            };
            const bool current = (sm.sv_step[e] & kRecValid) != 0;
            if (PERSIST && current) prefetch_records();
            if (current) {   // the records requested at kernel entry; their start pose and scan row are requested here, ahead of
                             // the recomputation the other lanes of the wave may have to go through
                if (c == 1) {
                    rgx = pf_rgx; rgy = pf_rgy; rctr = pf_rctr;
                }
                if (P.auto_reset) {
                    tgx = pf_g0; tgy = pf_g1; rdist = pf_g2;
                    tl = pf_tl;
                    sctr = (uint32_t)pf_ck; sk = (int)(uint32_t)(pf_ck >> 32);
                    px = P.starts[3 * sk]; py = P.starts[3 * sk + 1]; pth = P.starts[3 * sk + 2];
                    request_row(sk);
                }
            }
            if (!current) {
                if (c == 1) {
                    sample_goal<PRef>(P, sm.rects, ie, 1, sctr, rgx, rgy);
                    rctr = sctr;
                    P.rsp_g[ie] = rgx; P.rsp_g[N + ie] = rgy; P.rsp_ctr[ie] = rctr;
                }
                if (P.auto_reset) {
                    sample_episode<PRef>(P, sm.rects, ie, sctr, sk, tgx, tgy);
                    request_row(sk);
                    px = P.starts[3 * sk]; py = P.starts[3 * sk + 1]; pth = P.starts[3 * sk + 2];
                    double ryaw, rrel, rdiff;
                    ryaw = P.starts_sc[sk];   // = yaw_from_quat((sin, cos)(pth / 2)), tabulated by starts_sc_kernel
                    goal_rel(px, py, tgx, tgy, ryaw, rrel, rdiff);
                    rdist = hypot(tgx - px, tgy - py);
                    tl = make_float4((float)(rdist / P.diag), (float)(ryaw / 360), (float)(rrel / 360), (float)(rdiff / 180));
                    double* rg = P.rec_g + (size_t)c * 3 * N + ie;
                    rg[0] = tgx; rg[N] = tgy; rg[2 * N] = rdist;
                    P.rec_tail[(size_t)c * N + ie] = tl;
                    P.rec_ck[(size_t)c * N + ie] = make_uint2(sctr, (uint32_t)sk);
                }
            }
            if (P.auto_reset) {
#pragma unroll
                for (int b = 0; b < NB / 2; ++b) {
                    sm.sp_scan[c][e][2 * b] = row[b].x;
                    sm.sp_scan[c][e][2 * b + 1] = row[b].y;
                }
            }
            sm.sp_d[c][0][e] = px; sm.sp_d[c][1][e] = py; sm.sp_d[c][2][e] = pth; sm.sp_d[c][3][e] = tgx;
            sm.sp_d[c][4][e] = tgy; sm.sp_d[c][5][e] = rdist;
            sm.sp_tail[c][e] = tl;
            sm.sp_ctr[c][e] = sctr;
            sm.sp_k[c][e] = sk;
            if (c == 1) {
                sm.sp_rg[0][e] = rgx; sm.sp_rg[1][e] = rgy;
                sm.sp_rctr[e] = rctr;
            }
        }
    }
    // ---------------- the cast
    {
        float4* const q4 = sm.q4[wave];
        unsigned* const qe = sm.qe[wave];
        float4* const r4 = sm.r4[wave];
        unsigned* const re = sm.re[wave];
        unsigned* const pe = sm.pe[wave];
        int qhead = 0, qtail = 0, rhead = 0, rtail = 0, phead = 0, ptail = 0;   // wave-uniform
        // Stage B, the exact tests.  With 10 beams every queued segment is tested against ALL beams (lane = segment: 206 vector
        // instructions per 64 segments).  With 36 beams that is 770 instructions in one wave, although stage A2 has just bounded
        // the beams the segment's arc can hold -- so there (kPairB) a queued segment is expanded into entries of kBeamsPerEntry
        // consecutive beams of that extent and a pass tests 64 ENTRIES (lane = entry): the same arithmetic on the same (a, b, o, d),
        // on the beams that can be hit only, so the scan keeps its bits (the extent is as conservative as the A2 cull itself).
        // Measured (tools/time_step.py, same box): BASELINE configs[3]'s shard (4096 envs, stage_4, 36 beams) 15.8 -> 12.5 us per
        // launch, tape form 12.7 -> 9.8 us per step.  At 10 beams the expansion costs more than it saves -- a wave of a configs[2]
        // workgroup queues only ~40 segments per step, one partial pass either way: 12.96 -> 13.19 us -- so it stays off there.
        constexpr bool kPairB = NB > 16;
        // (Rejected, profiles/r04_stage_b_pool_ablation.txt: pooling what the 16 waves' stage-B queues hold at the end of a step --
        // a wave queues ~40 segments per step, so its one pass runs with 60 % of its lanes -- removes 6.5 of 16 passes per
        // workgroup and step and is 14-17 % SLOWER: the passes land on the tail of the slowest wave, behind dependent LDS atomics.)
        constexpr int kBeamsPerEntry = (NB > 16) ? 4 : 2;
        auto exact_tests = [&](const bool on, const float4 g, const unsigned el, const int b_first, const int n_b, auto n_max)
                               __attribute__((always_inline)) {
            constexpr int kMaxB = decltype(n_max)::value;
            const float2 o = sm.org[el];
            const float rx = g.x - o.x, ry = g.y - o.y;
            const float ex = g.z - g.x, ey = g.w - g.y;
            const float k = fmaf(rx, ey, -(ry * ex));
            // div_pos needs |den| in [2^-60, 2^20]: true unless the segment is degenerate-small
            // (coordinates are documented to be < 2^19); such passes take the plain IEEE divide.
            const bool tiny = fmaxf(fabsf(ex), fabsf(ey)) < 0x1p-10f;
            if (__builtin_expect(__any(tiny), 0)) {
                // the empty volatile asm keeps this a real (wave-uniform) branch the compiler cannot speculate
                asm volatile("; degenerate-segment pass" ::: "memory");
                for (int u = 0; u < kMaxB; ++u) {
                    const int b = min(b_first + u, NB - 1);
                    const float2 d = sm.dir[b * EPB + el];
                    const unsigned bits = __float_as_uint(ray_seg(rx, ry, ex, ey, k, d.x, d.y)) & 0x7fffffffu;
                    if (on && u < n_b && bits < kInfBits) atomicMin(&sm.rng[b * EPB + el], bits);
                }
            } else {
#pragma unroll
                for (int u = 0; u < kMaxB; ++u) {
                    const int b = (kMaxB == NB) ? u : min(b_first + u, NB - 1);
                    const float2 d = sm.dir[b * EPB + el];
                    const unsigned bits = ray_seg_bits(rx, ry, ex, ey, k, d.x, d.y);
                    // hits only: same-address LDS atomics of a wave serialise, and most lanes miss
                    if (on && u < n_b && bits < kInfBits) atomicMin(&sm.rng[b * EPB + el], bits);
                }
            }
        };
        // up to 64 queued segments against every beam of their env (lane = segment): the tail of a small map, whose whole cast
        // is one pass -- there the two extra queue hops of expand / testB would only add latency
        auto flushB = [&](int n) __attribute__((always_inline)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool on = lane < n;
            const int slot = (rhead + lane) & 127;
            const float4 g = on ? r4[slot] : make_float4(0.f, 0.f, 1.f, 0.f);
            const unsigned el = on ? (kPairB ? (re[slot] & 255u) : re[slot]) : 0u;
            exact_tests(on, g, el, 0, NB, std::integral_constant<int, NB>{});
            rhead += n;
        };
        // (Measured without gain, round 4: splitting a wave's LAST pass -- ~40 segments at configs[2], 15-30 on a small shared map --
        // over two or four lanes per segment, each testing half / a quarter of the beams: 117 / 75 instead of 206 instructions on the
        // tail that barrier B waits for, and configs[2] 12.6-12.7 vs 12.6-12.7 us, the 16-env rollout 5.11-5.18 vs 5.06-5.10.)
        // up to 64 entries (lane = entry): kBeamsPerEntry consecutive beams of one queued segment
        auto testB = [&](int n) __attribute__((always_inline)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool on = lane < n;
            const unsigned e = on ? pe[(phead + lane) & 127] : 0u;
            const int slot = (int)(e & 127u);
            const float4 g = on ? r4[slot] : make_float4(0.f, 0.f, 1.f, 0.f);
            const unsigned el = on ? (re[slot] & 255u) : 0u;
            exact_tests(on, g, el, (int)((e >> 8) & 255u), (int)(e >> 16), std::integral_constant<int, kBeamsPerEntry>{});
            phead += n;
        };
        // the oldest n queued segments -> entries.  Lane = segment writes its j-th entry in round j (ballot compaction per round:
        // no prefix sum; a pillar facet is done after round 0, a wall across the whole fan after NB / kBeamsPerEntry rounds).
        auto expand = [&](int n) __attribute__((always_inline)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int slot = (rhead + lane) & 127;
            const unsigned w = (lane < n) ? re[slot] : 0u;
            const int b0 = (int)((w >> 8) & 255u), c = (int)(w >> 16);
            const int ne = (c + kBeamsPerEntry - 1) / kBeamsPerEntry;   // 0 on the lanes beyond n
            rhead += n;
            for (int j = 0;; ++j) {   // wave-uniform
                const bool mine = ne > j;
                const unsigned long long bal = __ballot(mine);
                if (!bal) break;
                const int off = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                if (mine)
                    pe[(ptail + off) & 127] = (unsigned)slot | ((unsigned)(b0 + kBeamsPerEntry * j) << 8) |
                                              ((unsigned)min(kBeamsPerEntry, c - kBeamsPerEntry * j) << 16);
                ptail += __popcll(bal);
                if (ptail - phead >= 64) testB(64);
            }
            // ... and the rest: an entry names its segment by the r4 slot, and the next A2 pass may rewrite the slots of the
            // segments just expanded (the ring only protects the ones still queued), so no entry outlives its expansion
            if (ptail > phead) testB(ptail - phead);
        };
        // stage A2, dense over up to 64 survivors of the range / behind cull: does the arc the segment subtends, as seen
        // from the sensor, hold a beam at all -- and which?  (Beams are 20 degrees apart, a pillar facet at 2 m subtends 2.)
        constexpr float kInvDelta = (float)((NB - 1) / (2.0 * kAngleMax)), kBeamMargin = 0.02f;
        auto flushA2 = [&](int n) __attribute__((always_inline)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool on = lane < n;
            const int slot = (qhead + lane) & 127;
            const float4 g = on ? q4[slot] : make_float4(0.f, 0.f, 1.f, 0.f);
            const unsigned el = on ? qe[slot] : 0u;
            const float2 o = sm.org[el], h = sm.hd[el];
            const float ax = g.x - o.x, ay = g.y - o.y, bx = g.z - o.x, by = g.w - o.y;
            const float xa = fmaf(ax, h.x, ay * h.y), ya = fmaf(ay, h.x, -(ax * h.y));
            const float xb = fmaf(bx, h.x, by * h.y), yb = fmaf(by, h.x, -(bx * h.y));
            bool bad_a, bad_b;
            const float fa = beam_coord(xa, ya, 2.f * kInvDelta, (float)kAngleMax * kInvDelta, &bad_a);
            const float fb = beam_coord(xb, yb, 2.f * kInvDelta, (float)kAngleMax * kInvDelta, &bad_b);
            const float lo = fminf(fa, fb), hi = fmaxf(fa, fb);
            // the arc runs through straight-behind iff the segment crosses the negative X axis: Y changes sign and the
            // crossing X = cross(a, b) / (yb - ya) is negative
            const float kl = fmaf(xa, yb, -(xb * ya)), ey = yb - ya, ex = xb - xa;
            const bool wrap = (ya * yb < 0.f) && (kl * ey < 0.f);
            const bool through = kl * kl <= 1e-6f * fmaf(ex, ex, ey * ey);   // the line passes within 1 mm of the sensor
            const float ce = ceilf(lo - kBeamMargin), fl = floorf(hi + kBeamMargin);
            const bool inside = (fl >= ce) && (fl >= 0.f) && (ce <= (float)(NB - 1));
            const bool outside = (lo + kBeamMargin >= 0.f) || (hi - kBeamMargin <= (float)(NB - 1));
            // an arc whose beam extent cannot be trusted (an endpoint next to the sensor or straight behind it, a line through the
            // sensor, an arc through straight-behind) keeps the segment for every beam
            const bool all = bad_a || bad_b || through || wrap;
            const bool keep = on && (all ? (bad_a || bad_b || through || outside) : inside);
            unsigned ext = 0u;
            if constexpr (kPairB) {   // beams ce .. fl of the fan (v_med3 clamps; a NaN coordinate has set `bad`)
                const float b_lo = __builtin_amdgcn_fmed3f(ce, 0.f, (float)(NB - 1)), b_hi = __builtin_amdgcn_fmed3f(fl, 0.f, (float)(NB - 1));
                ext = all ? ((unsigned)NB << 16) : (((unsigned)(int)b_lo << 8) | ((unsigned)((int)(b_hi - b_lo) + 1) << 16));
            }
            qhead += n;
            const unsigned long long bal = __ballot(keep);
            if (bal) {
                const int off = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                if (keep) {
                    const int s2 = (rtail + off) & 127;
                    r4[s2] = g;
                    re[s2] = el | ext;
                }
                rtail += __popcll(bal);
                if (rtail - rhead >= 64) {
                    if constexpr (kPairB) expand(64);
                    else flushB(64);
                }
            }
        };
        // stage A: cull + compaction of one tile (64 lanes = 64 segments, or 64 / epp segments of epp envs)
        // (Round 4 tried a leaner form -- the distance test in the world frame, "behind" as the half plane X < -1 mm: 12 fewer
        // vector instructions per 128-segment pass, SQ_INSTS_VALU -2.4 % -- and measured it SLOWER on the same box, configs[2]
        // 13.18 vs 13.08 us, the house map 34.6 vs 33.3 us: the cast is not bound by the instruction count alone.)
        auto keep_of = [&](const float4 g, const float2 o, const float2 h) __attribute__((always_inline)) -> bool {
            // endpoints in the robot frame (X ahead, Y left); float32 throughout: this decides only what is tested exactly
            const float ax = g.x - o.x, ay = g.y - o.y, bx = g.z - o.x, by = g.w - o.y;
            const float xa = fmaf(ax, h.x, ay * h.y), ya = fmaf(ay, h.x, -(ax * h.y));
            const float xb = fmaf(bx, h.x, by * h.y), yb = fmaf(by, h.x, -(bx * h.y));
            // behind: both endpoints inside the convex cone X < -1e-3 |Y|; the beams span +-(pi/2 + 1.2e-6)
            const bool behind = (fmaf(1e-3f, fabsf(ya), xa) < 0.f) && (fmaf(1e-3f, fabsf(yb), xb) < 0.f);
            // far: distance from the sensor to the segment above 3.5 m (threshold 12.3 = (3.5 * 1.002)^2).  The closest point
            // is a + t e with t = clamp(-a.e / e.e, 0, 1); the error of the hardware reciprocal moves it by < 1e-7 |e|, far
            // inside the 7 mm margin for any segment shorter than 10 km; a zero-length segment gives t = 0 (0 x inf = NaN
            // takes v_med3's minimum).  So the test stays conservative.
            const float ex = xb - xa, ey = yb - ya;
            const float e2 = fmaf(ex, ex, ey * ey), ae = fmaf(xa, ex, ya * ey);
            const float t = __builtin_amdgcn_fmed3f(-ae * __builtin_amdgcn_rcpf(e2), 0.f, 1.f);
            const float cx = fmaf(t, ex, xa), cy = fmaf(t, ey, ya);
            const float d2 = fmaf(cx, cx, cy * cy);
            // d2 < 1e-6: the segment passes within 1 mm of the sensor, where the angles are noise: keep
            return (d2 < 1e-6f) || !(behind || (d2 > 12.3f));
        };
        auto push = [&](const float4 g, const unsigned el, const bool keep) __attribute__((always_inline)) {
            const unsigned long long bal = __ballot(keep);
            if (bal) {
                const int off = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                if (keep) {
                    const int slot = (qtail + off) & 127;
                    q4[slot] = g;
                    qe[slot] = el;
                }
                qtail += __popcll(bal);
                if (qtail - qhead >= 64) flushA2(64);
            }
        };
        auto cull = [&](const Pos p, const float4 g, const bool v) __attribute__((always_inline)) {
            const int el = min(p.it * epp + sub, nloc - 1);
            const float2 o = sm.org[el], h = sm.hd[el];
            push(g, (unsigned)el, keep_of(g, o, h) & v);   // evaluated unconditionally, see cull2
        };
        auto cull2 = [&](const Pos p, const float4 ga, const bool va, const float4 gb, const bool vb) __attribute__((always_inline)) {
            const int el = min(p.it, nloc - 1);
            const float2 o = sm.org[el], h = sm.hd[el];
            // both tests unconditionally (an invalid lane holds a clamped, valid segment): `va && keep_of(...)` made each an exec-masked
            // branch of its own, one dependent chain after the other; as two independent chains the compiler interleaves them
            // (configs[2] 12.87 -> 12.62 us per launch, tape 9.30 -> 9.18, same box)
            const bool ka = keep_of(ga, o, h) & va, kb = keep_of(gb, o, h) & vb;
            push(ga, (unsigned)el, ka);   // at most 63 + 64 queued before either flush: the ring holds 128
            push(gb, (unsigned)el, kb);
        };
        auto consume = [&](const Pos p, const float4 ga, const bool va, const float4 gb, const bool vb) __attribute__((always_inline)) {
            if constexpr (PAIR) cull2(p, ga, va, gb, vb);
            else cull(p, ga, va);
        };
        // BOXES: bit t of item_live(env) = tile t can matter for this env's pose: lane = tile, the same conservative range /
        // behind-the-fan tests as stage A applied to the tile's bounding box (all four corners inside the behind cone)
        auto item_live = [&](const int it) __attribute__((always_inline)) -> unsigned long long {
            const int el = min(it, nloc - 1);
            const float2 o = sm.org[el], h = sm.hd[el];
            const float4 bx = sm.tbox[lane];
            const float dx = fmaxf(fmaxf(bx.x - o.x, o.x - bx.z), 0.f), dy = fmaxf(fmaxf(bx.y - o.y, o.y - bx.w), 0.f);
            const bool far = fmaf(dx, dx, dy * dy) > 12.3f;
            const float x0 = bx.x - o.x, x1 = bx.z - o.x, y0 = bx.y - o.y, y1 = bx.w - o.y;
            auto behind = [&](float cx, float cy) {
                const float X = fmaf(cx, h.x, cy * h.y), Y = fmaf(cy, h.x, -(cx * h.y));
                return fmaf(1e-3f, fabsf(Y), X) < 0.f;
            };
            const bool back = behind(x0, y0) && behind(x1, y0) && behind(x0, y1) && behind(x1, y1);
            return __ballot((lane < ntiles) && !(far || back));
        };
        unsigned long long cur_live = 0ull;   // BOXES: live tiles of the item the last assigned position belongs to
        auto next_item = [&]() __attribute__((always_inline)) -> Pos {   // BOXES: first live tile of the next item that has one
            for (;;) {
                const int it = grab();
                if (it >= n_items) return Pos{it, 0};
                cur_live = item_live(it);
                if (cur_live) return Pos{it, (int)__builtin_ctzll(cur_live)};
            }
        };
        auto advance = [&](const Pos p) __attribute__((always_inline)) -> Pos {
            if (p.it >= n_items) return p;
            if (BOXES) {
                const unsigned long long rest = (p.t >= 63) ? 0ull : (cur_live & (~0ull << (p.t + 1)));
                if (rest) return Pos{p.it, (int)__builtin_ctzll(rest)};
                return next_item();
            }
            if (p.t + 1 < ntl) return Pos{p.it, p.t + 1};
            return Pos{grab(), 0};
        };
        if (wave < PW) {   // pose waves join when part 2 is done
            p0 = BOXES ? next_item() : Pos{grab(), 0};
            request(p0, g0, v0, g0b, v0b);
            p1 = advance(p0);
            request(p1, g1, v1, g1b, v1b);
            p2 = advance(p1);
            request(p2, g2, v2, g2b, v2b);
        } else if (BOXES) {   // ray waves: the live tiles of the item their last pre-assigned tile belongs to
            cur_live = (p2.it < n_items) ? item_live(p2.it) : 0ull;
        }
        // three tiles in flight; the slots take turns (a register rotation would have to wait for the load it moves)
        for (;;) {   // wave-uniform
            if (p0.it >= n_items) break;
            consume(p0, g0, v0, g0b, v0b);
            p0 = advance(p2);
            request(p0, g0, v0, g0b, v0b);
            if (p1.it >= n_items) break;
            consume(p1, g1, v1, g1b, v1b);
            p1 = advance(p0);
            request(p1, g1, v1, g1b, v1b);
            if (p2.it >= n_items) break;
            consume(p2, g2, v2, g2b, v2b);
            p2 = advance(p1);
            request(p2, g2, v2, g2b, v2b);
        }
        // The tail.  10 beams: stage A2 only filters, so when everything that is left fits one stage-B pass it is skipped: the
        // stage-A survivors join the stage-B queue directly (one pass of 206 VALU instead of 75 + 206; a small map never needs A2).
        // 36 beams: a pass over all beams is the expensive thing, A2 and the expansion always run.
        if (!kPairB && qtail - qhead + rtail - rhead <= 64) {
            const int n = qtail - qhead;
            if (lane < n) {
                const int s1 = (qhead + lane) & 127, s2 = (rtail + lane) & 127;
                r4[s2] = q4[s1];
                re[s2] = qe[s1];
            }
            rtail += n;
            qhead += n;
        }
        if (qtail > qhead) flushA2(qtail - qhead);
        if (rtail > rhead) {
            if constexpr (kPairB) expand(rtail - rhead);
            else flushB(rtail - rhead);
        }
    }
    __syncthreads();  // barrier B: nearest hits complete

    // ---------------- scan -> observation entries, lane = (beam, env): sanitise (environment_new.py:192-198), / 3.5 (:289),
    // and min(scan) for the collision rule (:200).  Nearest hits are uint bit patterns of non-negative floats (inf = none).
    for (int idx = tid; idx < NB * EPB; idx += kThreads) {
        const int b = idx / EPB, e = idx % EPB;
        if (e < nloc) {
            float r = sensor_value(__uint_as_float(sm.rng[idx]), sigma, (sigma > 0.f) ? sm.noise[idx] : 0.f, below_min);
            if (r == INFINITY) r = 3.5f;  // :193-194 (+inf only: -inf passes through like in the reference; NaN cannot occur)
            // (float)((double)r / 3.5) == r / 3.5f : double rounding is innocuous for a quotient of
            // two float32 values (53 >= 2*24+2), so the float32 divide gives the reference's bits.
            sm.obs[e * DP + b] = r / 3.5f;
            if (SENS && r < 0.f)
                atomicOr(&sm.neg[e], 1u);
            else
                atomicMin(&sm.mn_bits[e], __float_as_uint(r));
        }
    }
    __syncthreads();  // barrier B2: lidar entries and their minima complete
    hook(wave, lane);

    // ---------------- part 3 (wave 0, lane = env): rules of getState / step / setReward + episode logic
    if (own) {
        const int e = lane;
        const int i = base + e;
        x = sm.sv_d[0][e]; y = sm.sv_d[1][e]; th = sm.sv_d[2][e]; gx = sm.sv_d[3][e]; gy = sm.sv_d[4][e];
        pdist = sm.sv_d[5][e]; dist = sm.sv_d[6][e]; ret0 = sm.sv_d[10][e];
        double path = sm.sv_d[12][e];
        act = sm.sv_act[e]; pact = sm.sv_pact[e]; ctr = sm.sv_ctr[e];
        const uint32_t step0 = sm.sv_step[e] & kStepMask;
        float* row = sm.obs + e * DP;
        const float* noise = (sigma > 0.f) ? sm.noise + e : nullptr;
        const float mn = (SENS && sm.neg[e]) ? -INFINITY : __uint_as_float(sm.mn_bits[e]);
        // row[B .. B + 5] (past action, dist / diag, yaw / 360, rel_theta / 360, diff / 180) were written in part 2
        const bool d = (0.2 > (double)mn) && ((double)mn > 0);  // environment_new.py:200
        const bool a = dist <= P.thr;                             // :204
        // setReward, :209-222
        double r = 500. * (pdist - dist);
        pdist = dist;
        if (d) r = -100.;
        if (a) r = 120.;
        bool moved_stream = false;
        if (a && P.respawn) {  // :245-267 (drawn ahead by the spec lane of record 1)
            gx = sm.sp_rg[0][e]; gy = sm.sp_rg[1][e];
            ctr = sm.sp_rctr[e];
            moved_stream = true;
            if (!P.auto_reset) pdist = hypot(gx - x, gy - y);  // with auto_reset the arrival ends the episode: pdist is re-based below
        }
        uint32_t step = step0 + 1;
        double ret = ret0 + r;
        const bool timeout = (P.max_ep_steps > 0) && ((int)step >= P.max_ep_steps);  // ppo.py:552
        const bool end = d || a || timeout;
        io.reward[row0 + i] = (float)r;
        io.done[row0 + i] = d ? 1 : 0;
        io.arrive[row0 + i] = a ? 1 : 0;
        if (io.ended) io.ended[row0 + i] = end ? 1 : 0;
        if (end) {
            if (io.ep_return) io.ep_return[row0 + i] = (float)ret;
            if (io.ep_length) io.ep_length[row0 + i] = (int32_t)step;
            if (io.ep_path_out) io.ep_path_out[row0 + i] = (float)path;   // ppo.py:533-537: the final step's displacement is never added
        }
        path += sm.sv_d[11][e];
        float2 next_pact = act;  // ppo.py:543
        if (end && P.auto_reset) {  // ppo.py:582-593 + Env.reset, environment_new.py:312-382: the record read / prepared in part 2
            const int c = (a && P.respawn) ? 1 : 0;
            x = sm.sp_d[c][0][e]; y = sm.sp_d[c][1][e]; th = sm.sp_d[c][2][e]; gx = sm.sp_d[c][3][e]; gy = sm.sp_d[c][4][e];
            dist = sm.sp_d[c][5][e];
            ctr = sm.sp_ctr[c][e];
            step = 0;
            ret = 0;
            path = 0;
            moved_stream = true;
            next_pact = make_float2(0.f, 0.f);
            pdist = dist;  // getGoalDistace, :116-120,:359
            if (SENS) {
                write_lidar(row, sm.sp_scan[c][e], 1, noise, EPB, sigma, below_min, B);
            } else {
#pragma unroll 2
                for (int b = 0; b < NB; ++b) row[b] = sm.sp_scan[c][e][b];
            }
            const float4 tl = sm.sp_tail[c][e];
            row[B + 0] = 0.f; row[B + 1] = 0.f;        // :372-373
            row[B + 2] = tl.x; row[B + 3] = tl.y; row[B + 4] = tl.z; row[B + 5] = tl.w;
        }
        // the records written or read in part 2 stay current until the env's goal stream moves
        const bool rec_ok = (P.auto_reset || P.respawn) && !moved_stream;
        const uint32_t stepw_out = step | (rec_ok ? kRecValid : 0u);
        if (PERSIST) {
            sm.st_d[0][e] = x; sm.st_d[1][e] = y; sm.st_d[2][e] = th; sm.st_d[3][e] = gx; sm.st_d[4][e] = gy;
            sm.st_d[5][e] = pdist; sm.st_d[6][e] = ret; sm.st_d[7][e] = path;
            sm.st_pact[e] = next_pact;
            sm.st_step[e] = stepw_out;
            sm.st_ctr[e] = ctr;
        }
        if (!PERSIST || last_step) {
            P.x[i] = x; P.y[i] = y; P.th[i] = th;
            P.gx[i] = gx; P.gy[i] = gy; P.past_dist[i] = pdist;
            P.past_action[i] = next_pact;
            P.ep_step[i] = (int32_t)stepw_out;
            P.ep_ret[i] = ret;
            P.ep_path[i] = path;
            P.rng_ctr[i] = ctr;
        }
    }
    __syncthreads();  // barrier C: observation tile complete in LDS

    // ---------------- coalesced store of the block's observation tile
    const int n_out = nloc * D;
    if (P.obs_f16) {
        __half* o = reinterpret_cast<__half*>(io.obs_out) + (row0 + (size_t)base) * D;
        for (int k = tid; k < n_out; k += kThreads) o[k] = __float2half_rn(sm.obs[(k / D) * DP + (k % D)]);
    } else {
        float* o = reinterpret_cast<float*>(io.obs_out) + (row0 + (size_t)base) * D;
        for (int k = tid; k < n_out; k += kThreads) o[k] = sm.obs[(k / D) * DP + (k % D)];
    }
}

// Entries f0 .. f0 + KS - 1 of the observation env `e` (local) will hold when this step is over (entries past the row: 0), readable
// from barrier B2 on -- i.e. BEFORE the rules lane of part 3 has run: the row in sm.obs as it stands, unless the step ends the
// episode and auto_reset replaces it by the reset observation (ppo.py:582-593).  Whether it does follows from three values that are
// final at barrier B2 -- the minimum of the scan (collision, environment_new.py:200), the goal distance (arrival, :204), the step
// count (time-out, ppo.py:552) -- and the reset observation is the record the spec lanes staged in part 2.  The same expressions as
// part 3 on the same operands, so the policy that reads its input through here sees exactly the row part 3 leaves in sm.obs.
template <int NB, int EPB, int NW, bool SENS, int KS, class PRef>
__device__ __forceinline__ void next_obs_n(PRef P, const StepSmem<NB, EPB, NW>& sm, const int e, const int f0, const float sigma,
                                           const int below_min, float (&x)[KS]) {
    constexpr int D = NB + 6, DP = NB + 7;
    const float* row = sm.obs + e * DP;
#pragma unroll
    for (int j = 0; j < KS; ++j) x[j] = (f0 + j < D) ? row[min(f0 + j, D - 1)] : 0.f;
    const float mn = (SENS && sm.neg[e]) ? -INFINITY : __uint_as_float(sm.mn_bits[e]);
    const bool d = (0.2 > (double)mn) && ((double)mn > 0);                                        // environment_new.py:200
    const bool a = sm.sv_d[6][e] <= P.thr;                                                       // :204
    const uint32_t step = (sm.sv_step[e] & kStepMask) + 1;
    const bool timeout = (P.max_ep_steps > 0) && ((int)step >= P.max_ep_steps);                  // ppo.py:552
    if ((d || a || timeout) && P.auto_reset) {
        const int c = (a && P.respawn) ? 1 : 0;
        const float4 tl = sm.sp_tail[c][e];
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const int idx = f0 + j;
            float val;
            if (idx < NB) {
                val = sm.sp_scan[c][e][min(idx, NB - 1)];
                if (SENS) {   // write_lidar's entry: the start pose's nearest hit through this step's range noise
                    float r = sensor_value(val, sigma, (sigma > 0.f) ? sm.noise[min(idx, NB - 1) * EPB + e] : 0.f, below_min);
                    if (r == INFINITY) r = 3.5f;
                    val = r / 3.5f;
                }
            } else if (idx < NB + 2) {
                val = 0.f;                                                                       // environment_new.py:372-373
            } else if (idx < D) {
                const int q = idx - NB - 2;
                val = q == 0 ? tl.x : q == 1 ? tl.y : q == 2 ? tl.z : tl.w;
            } else {
                val = 0.f;
            }
            x[j] = val;
        }
    }
}

// The kernarg segment of step_kernel: the kernel takes this struct as its ONLY by-value parameter, so the segment IS the struct
// (members at their natural alignment, in order) and the reads through the segment pointer below cannot drift from the signature.
struct StepKArgs {
    Params P;
    StepIO io;
};
typedef const StepKArgs __attribute__((address_space(4))) * StepKArgsPtr;
static_assert(std::is_trivially_copyable<StepKArgs>::value && offsetof(StepKArgs, P) == 0 && sizeof(StepIO) == 10 * sizeof(void*),
              "kernarg mirror of step_kernel");

// Eight envs on eight waves: two workgroups per CU overlap their chains only if both fit the register file, i.e. four waves per
// SIMD -- the second launch bound caps the allocation at 128 VGPRs there (the tape kernel took 133-153 without it, one workgroup
// per CU, 4096 envs in two rounds).  Sixteen envs on eight waves carry the bound only where the instantiation needs more than 128
// VGPRs AND the rule runs it with two workgroups per CU (pick_epb, 8192-env shards): a launch per step at 10 beams without the
// 128-segment passes (131 VGPRs; house map 8192 envs 38.8 -> 25.8 us, stage_1 4096 envs 7.6 -> 7.2) and the tape form at 36 beams
// or with the 128-segment passes (131-172; 36 beams 8192 envs 13.2 -> 8.2 us per step, per-env maps 11.6 -> 6.9).  On the other
// 16-env instantiations the bound changes nothing it should (they fit) and cost 2-5 % (the scheduler's occupancy target).  0 = none.
constexpr int min_waves_per_simd(int nb, int epb, int nw, int pair, bool tape) {
    if (nw != 8) return 0;
    if (epb == 8) return 4;
    if (epb == 16) return (tape ? (nb > 16 || pair != 0) : (nb == 10 && pair == 0)) ? 4 : 0;
    return 0;
}

template <int NB, int EPB, bool SENS, int NW = 4, bool BOXES = false, int PAIR = 0>
__global__ __launch_bounds__(64 * NW, min_waves_per_simd(NB, EPB, NW, PAIR, false)) void step_kernel(StepKArgs) {
    __shared__ StepSmem<NB, EPB, NW> sm;
    __shared__ int next_env;
    // the parameter is read through the kernarg segment pointer (see step_body): scalar loads at each use
    StepKArgsPtr A = (StepKArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(A));
    step_body<NB, EPB, SENS, false, NW, BOXES, PAIR, const Params __attribute__((address_space(4)))&,
              const StepIO __attribute__((address_space(4)))&>(A->P, sm, next_env, A->io);
}

// ---------------------------------------------------------------- the persistent rollout kernel (PPO.rollout, ppo.py:463-641)
// All T steps of the rollout in ONE launch for the 16-64-64 policy: a workgroup owns its EPB envs for the whole rollout and
// alternates   policy phase (wave 0: PPO.get_action of mlp64_policy.h on the observation tile in LDS -> action, log-prob)
//              step phases  (step_body above: motion, cast, rules, auto-reset; the next observation tile lands in LDS)
// with workgroup barriers only: no kernel boundary, no launch ramp and no observation round trip through HBM between the
// 2 T phases.  Every row of the [T, N, .] buffers is written exactly as the per-step entry points would write it (same
// device functions, same Philox keys), so the two paths produce the same bits.
struct RolloutArgs {
    const float* params;      // actor parameters (mlp64 layout)
    void* obs_buf;            // [T + 1, N, B + 6] f32 (f16: navsim_cfg.obs_f16): row 0 = the reset observations (input), rows 1..T written here
    float* act_buf;           // [T, N, 2]
    float* logp_buf;          // [T, N]
    float* reward;            // [T, N]
    uint8_t *done, *arrive, *ended;   // [T, N]
    float* ep_return;         // [T, N], nullable
    int32_t* ep_length;       // [T, N], nullable
    float* ep_path;           // [T, N], nullable
    const float* var_ptr;     // device scalar: exploration variance
    const uint32_t* step_base;   // device scalar, nullable: rollout steps taken before this launch (noise counter)
    uint64_t seed;
    int T;
};

template <int NB, int EPB, bool SENS, int NW, bool BOXES = false>
__global__ __launch_bounds__(64 * NW) void rollout_kernel(Params P, RolloutArgs R) {
    constexpr int D = NB + 6, DP = D + 1, kThreads = 64 * NW;
    using PL = mlp64::Layout<D>;   // the (B + 6)-64-64 policy: 16-wide rows with 10 beams, 42-wide with 36
    constexpr int KS = PL::KS;
    __shared__ StepSmem<NB, EPB, NW> sm;
    __shared__ int next_env;
    __shared__ __attribute__((aligned(16))) float wts[PL::P_ACTOR + 2];   // the actor, staged once for all T steps
    __shared__ float2 pol_z[4][16];   // policy phase: per-tile partial sums of the two output units
    __shared__ float2 pol_eps[16];    // ... and the step's action noise
    __shared__ unsigned pol_cnt;      // arrivals of the in-step policy's five waves (the last one finishes)
    static_assert(NW >= 6 && EPB <= 16, "policy phase: four tile waves + the noise wave beside wave 0's rules; one 16-env policy tile per workgroup");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int base = blockIdx.x * EPB;
    const int nloc = min(EPB, P.N - base);
    const size_t N = (size_t)P.N;
    for (int k = tid; k < PL::P_ACTOR; k += kThreads) wts[k] = R.params[k];
    if (tid < nloc) {   // the envs' state: HBM -> LDS for the whole rollout
        const int e = tid, i = base + e;
        sm.st_d[0][e] = P.x[i]; sm.st_d[1][e] = P.y[i]; sm.st_d[2][e] = P.th[i]; sm.st_d[3][e] = P.gx[i]; sm.st_d[4][e] = P.gy[i];
        sm.st_d[5][e] = P.past_dist[i]; sm.st_d[6][e] = P.ep_ret[i]; sm.st_d[7][e] = P.ep_path[i];
        sm.st_pact[e] = P.past_action[i];
        sm.st_step[e] = (uint32_t)P.ep_step[i];
        sm.st_ctr[e] = P.rng_ctr[i];
    }
    const bool half_rows = P.obs_f16 != 0;   // float16 observation buffers: the policy reads what a reader of the buffers would
    for (int k = tid; k < nloc * D; k += kThreads)
        sm.obs[(k / D) * DP + (k % D)] = half_rows ? __half2float(reinterpret_cast<const __half*>(R.obs_buf)[(size_t)base * D + k])
                                                   : reinterpret_cast<const float*>(R.obs_buf)[(size_t)base * D + k];
    for (int k = tid; k < 2 * NB; k += kThreads) sm.beam[k] = P.beam_cs[k];
    for (int k = tid; k < (int)(sizeof(Rects) / 8); k += kThreads)   // the goal rejection rectangles, for all T steps
        reinterpret_cast<uint64_t*>(&sm.rects)[k] = reinterpret_cast<const uint64_t*>(P.rects)[k];
    const uint32_t step0 = R.step_base ? *R.step_base : 0u;
    const float var = *R.var_ptr;
    if (tid == 0) pol_cnt = 0u;
    __syncthreads();
    // The policy (mlp64_policy.h) as the kernel runs it: four "tile waves" each compute layer 1 and one 16-row tile of layer 2 with
    // its share of the two output units (tile_part), one wave draws the step's action noise meanwhile, and -- behind a barrier --
    // wave 0 finishes (sigmoid / tanh, clamp, log-prob) and publishes the action (finish).
    const float sigma = SENS ? P.sigma : 0.f;
    const int below_min = SENS ? P.below_min_mode : 0;
    auto tile_part = [&](const int t2, auto in_step) __attribute__((always_inline)) {
        const int e = lane & 15, kk = lane >> 4;   // lane (env, kk) feeds obs[env][KS kk .. KS kk + KS - 1]
        const bool valid = e < nloc;
        float xs[KS];
        if constexpr (decltype(in_step)::value) {   // inside a step, behind barrier B2: the rows the step will leave
            next_obs_n<NB, EPB, NW, SENS, KS, const Params&>(P, sm, min(e, nloc - 1), KS * kk, sigma, below_min, xs);
            if (half_rows) {
#pragma unroll
                for (int j = 0; j < KS; ++j) xs[j] = __half2float(__float2half_rn(xs[j]));
            }
        } else {
            mlp64::policy_row<PL>(sm.obs + min(e, nloc - 1) * DP, kk, half_rows, xs);
        }
        if (!valid) {
#pragma unroll
            for (int j = 0; j < KS; ++j) xs[j] = 0.f;
        }
        mlp64::f32x4 c1[4];
        mlp64::policy_hidden1<PL>(wts, xs, lane, c1);
        float pz3, pz4;
        mlp64::policy_tile2<PL>(wts, c1, lane, t2, pz3, pz4);
        if (kk == 0) pol_z[t2][e] = make_float2(pz3, pz4);
    };
    auto draw_noise = [&](const uint32_t step) __attribute__((always_inline)) {
        float e0, e1;
        mlp64::policy_noise(step, R.seed, P.env_id_base + (uint64_t)(base + lane), e0, e1);
        pol_eps[lane] = make_float2(e0, e1);
    };
    const float sd = sqrtf(var), log_var = logf(var);   // policy_finish's sqrtf(var) / logf(var), once per launch: same bits
    auto finish = [&](const size_t tn) __attribute__((always_inline)) {   // any wave, lane = env: row tn / N of act_buf / logp_buf
        const int e = lane;
        const float pz3[4] = {pol_z[0][e].x, pol_z[1][e].x, pol_z[2][e].x, pol_z[3][e].x};
        const float pz4[4] = {pol_z[0][e].y, pol_z[1][e].y, pol_z[2][e].y, pol_z[3][e].y};
        const float2 eps = pol_eps[e];
        const mlp64::PolicyOut o = mlp64::policy_finish_pre<PL>(wts, pz3, pz4, var, sd, log_var, eps.x, eps.y);
        sm.act_l[e] = make_float2(o.a0, o.a1);
        reinterpret_cast<float2*>(R.act_buf)[tn + base + e] = make_float2(o.a0, o.a1);
        R.logp_buf[tn + base + e] = o.logp;
    };
    // the action of step 0, from the reset observations
    if (wave < 4) tile_part(wave, std::false_type{});
    else if (wave == 4 && lane < nloc) draw_noise(step0);
    __syncthreads();
    if (wave == 0 && lane < nloc) finish(0);
    __syncthreads();
    for (int t = 0; t < R.T; ++t) {
        const size_t tn = (size_t)t * N;
        asm volatile("" ::: "memory");   // keeps the weight reads of the policy phase inside the loop (registers are scarce)
        void* const obs_row = half_rows ? (void*)(reinterpret_cast<__half*>(R.obs_buf) + (tn + N) * D)
                                        : (void*)(reinterpret_cast<float*>(R.obs_buf) + (tn + N) * D);
        const StepIO io = {nullptr, nullptr, obs_row, R.reward + tn, R.done + tn, R.arrive + tn, R.ended + tn,
                           R.ep_return ? R.ep_return + tn : nullptr, R.ep_length ? R.ep_length + tn : nullptr,
                           R.ep_path ? R.ep_path + tn : nullptr};
        // The policy of step t + 1 runs INSIDE step t, behind barrier B2.  Wave 0 works through the rules there (a float64 latency
        // chain, 0.85 us) and every other wave would only wait for it at barrier C; instead waves 1-4 run the tile parts, wave 5
        // draws the noise, and whichever of the five arrives last (an LDS counter: a wave's LDS operations are performed in order,
        // so the one that reads 4 sees what the other four wrote) finishes and publishes the action -- all before barrier C,
        // so the next step starts right behind this one.  What the rules still have to decide about the observation tile --
        // whether an env's row is replaced by its reset observation -- the tile waves work out themselves from the values that
        // are final at B2 (next_obs_n), so they read exactly the rows the step leaves: same device functions on the same inputs,
        // the buffers keep their bits.  Round 3 ran the phase between the steps with the whole workgroup waiting (5.2 us per
        // step); stamps of this form in profiles/r04_rollout16_phase_stamps.txt.
        const bool more = t + 1 < R.T;
        auto hook = [&](const int wv, const int ln) __attribute__((always_inline)) {
            // (measured: the parts on waves 1, 2, 3, 5 -- keeping them off wave 0's SIMD if waves land round robin -- 5.34 instead
            // of 5.01 us per step)
            if (more && wv >= 1 && wv <= 5) {
                if (wv <= 4) tile_part(wv - 1, std::true_type{});
                else if (ln < nloc) draw_noise(step0 + (uint32_t)(t + 1));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                unsigned old = 0u;
                if (ln == 0) old = atomicAdd(&pol_cnt, 1u);
                if (__builtin_amdgcn_readfirstlane(old) == 4u) {   // the last of the five
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    if (ln == 0) pol_cnt = 0u;
                    if (ln < nloc) finish(tn + N);
                }
            }
        };
        step_body<NB, EPB, SENS, true, NW, BOXES, 0, const Params&, const StepIO&, decltype(hook)>(P, sm, next_env, io,
                                                                                                 t == R.T - 1, 0, hook);
        // the observation tile of step t + 1 is in sm.obs (its store only reads it), its action in sm.act_l
    }
}

// ---------------------------------------------------------------- the persistent rollout for the reference's ACTIVE actor
// PPO.rollout (ppo.py:463-641) with NetActor (net_actor.py:56-144: two residual blocks of 512 hidden units) choosing every action
// in the kernel: rollout_kernel's workgroup (16 envs on 8 waves, env state and observation tile in LDS for all T steps) with the
// policy step of csrc/resmlp_policy.h -- the one navppo_resmlp512_act launches per step -- in front of every env step: all eight
// waves (each owns 64 hidden units), two workgroup sums, wave 0 finishes (clamp, log-prob of the clamped action); the Philox /
// Box-Muller noise of step t + 1 is drawn inside step t (wave 5, behind barrier B2, where it would only wait for the rules).
// Same device functions and Philox keys as the per-step entry points: every row of the [T, N, .] buffers keeps its bits.
//
// Where the weights live decides the time.  A CU pulls L2 hits at ~37 GB/s (measured here: 197 KB in 5 us, the same with 64 and
// with 256 workgroups -- outstanding misses x latency, not a shared limit), so the 197 KB of weights streamed per step were the
// longest phase of the first version.  Block 1 (66 KB) and the 114 small tail parameters now stay in LDS for the whole launch
// (145 KB per workgroup with the step body's 79 KB: one workgroup per CU); block 2 (130 KB) still streams, requested first so
// that it is in flight while block 1 runs.  Per 512-step rollout at 4096 envs: 9.1 ms as a hipGraph of 512 x (policy launch +
// step launch), 6.4 ms persistent with everything streamed, 5.9 ms now (bench.py resmlp512.rollout_ms).  Requesting the
// weights a phase early, inside the env step, was measured and is not done: the 128 registers per lane do not fit beside the
// step body (24-58 spilled VGPRs, 6.9-7.4 ms).  The phases of one step (tools/resmlp_rollout_phases.py; profiles/
// r05_rollout_resmlp_phases.txt): block 1 2.1 us, sum 1.9 (it waits for block 2's weights), block 2 2.3 (the f32 MFMA floor of
// both blocks on one CU is 2.6), sum + finish 1.3, env step 4.0.
template <bool SENS>
__global__ __launch_bounds__(64 * 8) void rollout_resmlp_kernel(Params P, RolloutArgs R) {
    constexpr int NB = 10, EPB = 16, NW = 8, D = NB + 6, DP = D + 1, kThreads = 64 * NW;
    static_assert(D == resmlp::rp::D && EPB == resmlp::kPolEnvs && NW == resmlp::kPolWaves, "the policy step's workgroup");
    __shared__ StepSmem<NB, EPB, NW> sm;
    __shared__ int next_env;
    __shared__ __attribute__((aligned(16))) resmlp::PolicySmem ps;
    __shared__ __attribute__((aligned(16))) resmlp::Block1Smem bs;   // block 1's weights + the small tail of the parameters
    __shared__ float2 pol_eps[EPB];   // the action noise of the next policy step (drawn inside the env step in front of it)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int base = blockIdx.x * EPB;
    const int nloc = min(EPB, P.N - base);
    const size_t N = (size_t)P.N;
    if (tid < nloc) {   // the envs' state: HBM -> LDS for the whole rollout
        const int e = tid, i = base + e;
        sm.st_d[0][e] = P.x[i]; sm.st_d[1][e] = P.y[i]; sm.st_d[2][e] = P.th[i]; sm.st_d[3][e] = P.gx[i]; sm.st_d[4][e] = P.gy[i];
        sm.st_d[5][e] = P.past_dist[i]; sm.st_d[6][e] = P.ep_ret[i]; sm.st_d[7][e] = P.ep_path[i];
        sm.st_pact[e] = P.past_action[i];
        sm.st_step[e] = (uint32_t)P.ep_step[i];
        sm.st_ctr[e] = P.rng_ctr[i];
    }
    const bool half_rows = P.obs_f16 != 0;   // float16 observation buffers (BASELINE configs[4]): the policy reads what a reader of the buffers would
    for (int k = tid; k < nloc * D; k += kThreads)
        sm.obs[(k / D) * DP + (k % D)] = half_rows ? __half2float(reinterpret_cast<const __half*>(R.obs_buf)[(size_t)base * D + k])
                                                   : reinterpret_cast<const float*>(R.obs_buf)[(size_t)base * D + k];
    for (int k = tid; k < 2 * NB; k += kThreads) sm.beam[k] = P.beam_cs[k];
    for (int k = tid; k < (int)(sizeof(Rects) / 8); k += kThreads)
        reinterpret_cast<uint64_t*>(&sm.rects)[k] = reinterpret_cast<const uint64_t*>(P.rects)[k];
    const uint32_t step0 = R.step_base ? *R.step_base : 0u;
    const float var = *R.var_ptr;
    const int l15 = lane & 15, q = lane >> 4;
    const bool valid = l15 < nloc;
    auto draw_noise = [&](const uint32_t step) __attribute__((always_inline)) {   // lane = env
        float e0, e1;
        mlp64::policy_noise(step, R.seed, P.env_id_base + (uint64_t)(base + lane), e0, e1);
        pol_eps[lane] = make_float2(e0, e1);
    };
    // The weights of a policy step (this wave's 64 hidden units: 128 registers) are REQUESTED inside the env step in front of it --
    // behind barrier B2, where every wave but wave 0 only waits for the rules -- and arrive during the step's tail: the L2 round
    // trip of 197 KB per workgroup and step is off the critical path.  (They cannot simply stay in registers: the step body needs them.)
    resmlp::Weights W;
    resmlp::stage_block1(R.params, lane, wave, bs);
    if (wave == 5 && lane < nloc) draw_noise(step0);
    __syncthreads();
    for (int t = 0; t < R.T; ++t) {
        const size_t tn = (size_t)t * N;
        // ---- PPO.get_action (ppo.py:673-706) on the observation tile the previous step (or the reset) left in LDS
        resmlp::f32x4 xq = resmlp::zero4();
        if (valid) {
            const float* row = sm.obs + l15 * DP + 4 * q;
            xq = resmlp::f32x4{row[0], row[1], row[2], row[3]};
            if (half_rows) {   // (the tile in LDS is float32; the row in the buffer -- what navppo_resmlp512_act would read -- is its rounding to half)
#pragma unroll
                for (int j = 0; j < 4; ++j) xq[j] = __half2float(__float2half_rn(xq[j]));
            }
        }
        float z3, z4;
        resmlp::load_weights_b(R.params, lane, wave, W);   // requested first: in flight while block 1 runs out of LDS
        resmlp::load_weights_a_lds(bs, lane, wave, W);
        resmlp::policy_preact_w(W, bs.b2a, bs.tail, xq, lane, wave, ps, z3, z4);
        if (wave == 0 && q == 0 && valid) {
            const float2 eps = pol_eps[l15];
            const resmlp::Action o = resmlp::policy_finish(bs.tail, z3, z4, var, eps.x, eps.y);
            sm.act_l[l15] = make_float2(o.a0, o.a1);
            reinterpret_cast<float2*>(R.act_buf)[tn + base + l15] = make_float2(o.a0, o.a1);
            R.logp_buf[tn + base + l15] = o.logp;
        }
        __syncthreads();
        // ---- the env step: the same step body as every other entry point; behind its barrier B2 the next policy step's weights are
        // requested and its noise is drawn
        void* const obs_row = half_rows ? (void*)(reinterpret_cast<__half*>(R.obs_buf) + (tn + N) * D)
                                        : (void*)(reinterpret_cast<float*>(R.obs_buf) + (tn + N) * D);
        const StepIO io = {nullptr, nullptr, obs_row, R.reward + tn, R.done + tn, R.arrive + tn,
                           R.ended + tn, R.ep_return ? R.ep_return + tn : nullptr, R.ep_length ? R.ep_length + tn : nullptr,
                           R.ep_path ? R.ep_path + tn : nullptr};
        const bool more = t + 1 < R.T;
        auto hook = [&](const int wv, const int ln) __attribute__((always_inline)) {
            if (more) {
                if (wv == 5 && ln < nloc) draw_noise(step0 + (uint32_t)(t + 1));
            }
        };
        step_body<NB, EPB, SENS, true, NW, false, 0, const Params&, const StepIO&, decltype(hook)>(P, sm, next_env, io, t == R.T - 1, 0, hook);
        __syncthreads();   // the observation tile of step t + 1 is complete in sm.obs (the tile store only reads it)
    }
}

// ---------------------------------------------------------------- n steps of an action tape in ONE launch (navsim_step_seq)
// The step loop of PPO.rollout / the evaluation loop (ppo.py:505-594, main.py:176-235) when the actions are already known -- a
// recorded tape, a scripted or random policy: a workgroup keeps its envs for all n steps, their state lives in LDS between the
// steps (HBM sees it before the first and after the last), and the steps follow each other with workgroup barriers only.  What a
// step kernel launch pays around its cast -- the launch ramp of 256 x 16 waves, the kernarg and state round trips ahead of the
// pose, the boundary between two graph nodes -- is paid once per tape; the segment stream of step t + 1 is requested while the
// pose of step t + 1 is computed.  Same step_body, same workgroup shapes as step_kernel: the rows are bit-identical to n
// navsim_step calls.  (Built with MachineLICM off, navbot_ppo_amd/build.py: the pass hoists the float64 constants of the whole
// step body out of the step loop and holds them in 60+ registers across it -- spills in the 16-wave shape, 11.3 instead of 9.4 us
// per step at configs[2].)
struct SeqArgs {
    StepIO io;   // the [T, N, .] buffers: action = the tape, obs_out [T, N, B + 6] (f32 or f16), reward / flags / episode statistics
                 // [T, N] (nullable as in navsim_step); step_body addresses row t through its row0 argument
    int T;
};
struct SeqKArgs {
    Params P;
    SeqArgs R;
};
typedef const SeqKArgs __attribute__((address_space(4))) * SeqKArgsPtr;

static_assert(std::is_trivially_copyable<SeqKArgs>::value && offsetof(SeqKArgs, P) == 0, "kernarg mirror of steps_kernel");

template <int NB, int EPB, bool SENS, int NW = 4, bool BOXES = false, int PAIR = 0>
__global__ __launch_bounds__(64 * NW, min_waves_per_simd(NB, EPB, NW, PAIR, true)) void steps_kernel(SeqKArgs) {   // the segment IS the struct (see step_kernel)
    __shared__ StepSmem<NB, EPB, NW> sm;
    __shared__ int next_env;
    // parameters through the kernarg segment pointer, as in step_kernel (scalar loads at each use, no spilled SGPRs)
    SeqKArgsPtr A = (SeqKArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(A));
    const Params __attribute__((address_space(4)))& P = A->P;
    const SeqArgs __attribute__((address_space(4)))& R = A->R;
    const int tid = threadIdx.x;
    const int base = blockIdx.x * EPB;
    const int nloc = min(EPB, P.N - base);
    const int T = R.T;
    float2 a_next = make_float2(0.f, 0.f);
    if (tid < nloc) {   // the envs' state: HBM -> LDS for the whole tape; the first action
        const int e = tid, i = base + e;
        a_next = R.io.action[i];
        sm.st_d[0][e] = P.x[i]; sm.st_d[1][e] = P.y[i]; sm.st_d[2][e] = P.th[i]; sm.st_d[3][e] = P.gx[i]; sm.st_d[4][e] = P.gy[i];
        sm.st_d[5][e] = P.past_dist[i]; sm.st_d[6][e] = P.ep_ret[i]; sm.st_d[7][e] = P.ep_path[i];
        sm.st_pact[e] = P.past_action[i];
        sm.st_step[e] = (uint32_t)P.ep_step[i];
        sm.st_ctr[e] = P.rng_ctr[i];
    }
    for (int k = tid; k < 2 * NB; k += 64 * NW) sm.beam[k] = P.beam_cs[k];
    for (int k = tid; k < (int)(sizeof(Rects) / 8); k += 64 * NW)
        reinterpret_cast<uint64_t*>(&sm.rects)[k] = reinterpret_cast<const uint64_t*>(P.rects)[k];
    for (int t = 0; t < T; ++t) {
        const size_t N = (size_t)P.N;
        const size_t tn = (size_t)t * N;
        if (tid < nloc) {
            sm.act_l[tid] = a_next;
            if (t + 1 < T) a_next = R.io.action[tn + N + base + tid];   // lands under this step
        }
        __syncthreads();
        step_body<NB, EPB, SENS, true, NW, BOXES, PAIR, const Params __attribute__((address_space(4)))&,
                  const StepIO __attribute__((address_space(4)))&>(P, sm, next_env, R.io, t == T - 1, tn);
    }
}

// ---------------------------------------------------------------- the persistent rollout at the big shard shapes
// navsim_rollout_mlp64 for shards beyond 4096 envs: steps_kernel's workgroup (64 envs on 16 waves, one per CU, the cast at its
// vector-issue bound, the cast variants of launch_step) with a policy phase in front of every step -- the closed-loop form of the
// tape kernel: the action of step t is PPO.get_action (ppo.py:673-706) of the observation tile step t - 1 left in LDS.
// The phase is bound by the SIMDs' MFMA pipes (64 envs x 10.2 kFLOP at the f32 MFMA rate of 64 FLOP per cycle and SIMD = 2560
// cycles), so each of the EPB / 16 policy tiles of 16 envs belongs to ONE wave -- the waves of a workgroup land on the SIMDs round
// robin, one tile per SIMD -- which runs policy_wave16's sequence: layer 1, the four 16-row tiles of layer 2 in order, the finish
// on its lanes 0-15 (no partial sums through LDS, no redundant layer 1: two waves per tile measured 2.05 us per step over the tape
// kernel, this 1.6).  The action noise does not depend on the observation: wave EPB / 16 draws the noise of step t + 1 during the
// policy phase of step t (two LDS rows that take turns).  Same device functions, same order of the partial sums and same Philox
// keys as rollout_kernel / navppo_mlp64_act: the rows are bit-identical to the per-step entry points.
struct BigKArgs {
    Params P;
    RolloutArgs R;
    StepIO io;   // rows of step t at element offset t N: obs_out = obs_buf + N D (row t + 1 of obs_buf), reward / flags / statistics
};
typedef const BigKArgs __attribute__((address_space(4))) * BigKArgsPtr;

static_assert(std::is_trivially_copyable<BigKArgs>::value && offsetof(BigKArgs, P) == 0, "kernarg mirror of rollout_big_kernel");

template <int EPB, bool SENS, int NW, bool BOXES = false, int PAIR = 0>
__global__ __launch_bounds__(64 * NW) void rollout_big_kernel(BigKArgs) {   // the segment IS the struct (see step_kernel)
    constexpr int NB = 10, D = NB + 6, DP = D + 1, kThreads = 64 * NW;
    constexpr int TW = EPB / 16;   // policy tiles = MFMA waves of the policy phase
    using PL = mlp64::Layout<D>;
    constexpr int KS = PL::KS;
    static_assert(D == mlp64::IN, "64-env workgroups: the 10-beam tile (36 beams do not fit the LDS)");
    static_assert(EPB % 16 == 0 && EPB <= 64 && TW < NW, "policy phase: EPB / 16 tile waves + the noise wave");
    __shared__ StepSmem<NB, EPB, NW> sm;
    __shared__ int next_env;
    __shared__ __attribute__((aligned(16))) float wts[mlp64::P_ACTOR + 2];   // the actor, staged once for all T steps
    __shared__ float2 pol_eps[2][EPB];   // action noise of this step and of the next one
    BigKArgsPtr A = (BigKArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(A));
    const Params __attribute__((address_space(4)))& P = A->P;
    const RolloutArgs __attribute__((address_space(4)))& R = A->R;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int base = blockIdx.x * EPB;
    const int nloc = min(EPB, P.N - base);
    const int T = R.T;
    {
        const float* const prm = R.params;
        for (int k = tid; k < mlp64::P_ACTOR; k += kThreads) wts[k] = prm[k];
    }
    if (tid < nloc) {   // the envs' state: HBM -> LDS for the whole rollout
        const int e = tid, i = base + e;
        sm.st_d[0][e] = P.x[i]; sm.st_d[1][e] = P.y[i]; sm.st_d[2][e] = P.th[i]; sm.st_d[3][e] = P.gx[i]; sm.st_d[4][e] = P.gy[i];
        sm.st_d[5][e] = P.past_dist[i]; sm.st_d[6][e] = P.ep_ret[i]; sm.st_d[7][e] = P.ep_path[i];
        sm.st_pact[e] = P.past_action[i];
        sm.st_step[e] = (uint32_t)P.ep_step[i];
        sm.st_ctr[e] = P.rng_ctr[i];
    }
    const bool half_rows = P.obs_f16 != 0;   // float16 observation buffers: the policy reads what a reader of the buffers would
    {
        const void* const ob = R.obs_buf;
        for (int k = tid; k < nloc * D; k += kThreads)
            sm.obs[(k / D) * DP + (k % D)] = half_rows ? __half2float(reinterpret_cast<const __half*>(ob)[(size_t)base * D + k])
                                                       : reinterpret_cast<const float*>(ob)[(size_t)base * D + k];
    }
    for (int k = tid; k < 2 * NB; k += kThreads) sm.beam[k] = P.beam_cs[k];
    for (int k = tid; k < (int)(sizeof(Rects) / 8); k += kThreads)
        reinterpret_cast<uint64_t*>(&sm.rects)[k] = reinterpret_cast<const uint64_t*>(P.rects)[k];
    const uint32_t step0 = R.step_base ? *R.step_base : 0u;
    const float var = *R.var_ptr;
    const uint64_t seed = R.seed, gid = P.env_id_base + (uint64_t)(base + lane);
    // policy tile k (envs 16 k .. 16 k + 15) belongs to wave TW + k -- the waves of a workgroup land on the SIMDs round robin, one
    // tile per SIMD -- and the noise of a row to wave 2 TW; wave 0 stays free for the rules (see the hook below)
    constexpr int kTileWave0 = TW, kNoiseWave = 2 * TW;
    static_assert(kNoiseWave < NW, "tile waves and the noise wave beside wave 0");
    // get_action (ppo.py:673-706) of policy tile `k` on the observation tile in LDS -> action / log-prob of row tn / N
    const float sigma = SENS ? P.sigma : 0.f;
    const int below_min = SENS ? P.below_min_mode : 0;
    const float sd = sqrtf(var), log_var = logf(var);   // policy_finish's sqrtf(var) / logf(var), once per launch: same bits
    auto tile_policy = [&](const int k, const size_t tn, const int eps_row, auto in_step) __attribute__((always_inline)) {
        const int e = 16 * k + (lane & 15), kk = lane >> 4;   // lane (env, kk) feeds obs[env][4 kk .. 4 kk + 3]
        const bool valid = e < nloc;
        float xs[KS];
        if constexpr (decltype(in_step)::value) {   // inside a step, behind barrier B2: the rows the step will leave
            next_obs_n<NB, EPB, NW, SENS, KS, const Params __attribute__((address_space(4)))&>(P, sm, min(e, nloc - 1), KS * kk, sigma,
                                                                                               below_min, xs);
            if (half_rows) {
#pragma unroll
                for (int j = 0; j < KS; ++j) xs[j] = __half2float(__float2half_rn(xs[j]));
            }
        } else {
            mlp64::policy_row<PL>(sm.obs + min(e, nloc - 1) * DP, kk, half_rows, xs);
        }
        if (!valid) {
#pragma unroll
            for (int j = 0; j < KS; ++j) xs[j] = 0.f;
        }
        const float2 eps = pol_eps[eps_row][min(e, nloc - 1)];   // requested ahead of the MFMA chain, used behind it
        mlp64::f32x4 c1[4];
        mlp64::policy_hidden1<PL>(wts, xs, lane, c1);
        float pz3[4], pz4[4];
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) mlp64::policy_tile2<PL>(wts, c1, lane, t2, pz3[t2], pz4[t2]);
        if (kk == 0 && valid) {
            const mlp64::PolicyOut o = mlp64::policy_finish_pre<PL>(wts, pz3, pz4, var, sd, log_var, eps.x, eps.y);
            sm.act_l[e] = make_float2(o.a0, o.a1);
            reinterpret_cast<float2*>(R.act_buf)[tn + base + e] = make_float2(o.a0, o.a1);
            R.logp_buf[tn + base + e] = o.logp;
        }
    };
    auto draw_noise = [&](const int t) __attribute__((always_inline)) {   // the noise of rollout row t: rows take turns in two LDS rows
        float e0, e1;
        mlp64::policy_noise(step0 + (uint32_t)t, seed, gid, e0, e1);
        pol_eps[t & 1][lane] = make_float2(e0, e1);
    };
    if (wave == kNoiseWave && lane < nloc) {
        draw_noise(0);
        if (T > 1) draw_noise(1);
    }
    __syncthreads();
    // the action of step 0, from the reset observations
    if (wave >= kTileWave0 && wave < kTileWave0 + TW) tile_policy(wave - kTileWave0, 0, 0, std::false_type{});
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const size_t N = (size_t)P.N;
        const size_t tn = (size_t)t * N;
        asm volatile("" ::: "memory");   // keeps the weight reads of the policy phase inside the loop
        // The policy of step t + 1 runs INSIDE step t, behind barrier B2, while wave 0 works through the rules (a float64 latency
        // chain) and the other waves would only wait for it at barrier C.  Whether the rules replace an env's observation row by its
        // reset observation (ppo.py:582-593) the tile waves work out themselves from the values that are final at B2 (next_obs_n):
        // they read exactly the rows the step leaves -- same device functions on the same inputs, the buffers keep their bits --
        // and the action is in LDS before barrier C, so the next step starts without a policy phase in front of it.  Round 3 ran
        // the phase between the steps, the whole workgroup waiting for it: + 1.6 us per step over the tape kernel.
        const bool more = t + 1 < T;
        auto hook = [&](const int wv, const int ln) __attribute__((always_inline)) {
            if (more) {
                if (wv >= kTileWave0 && wv < kTileWave0 + TW) tile_policy(wv - kTileWave0, tn + N, (t + 1) & 1, std::true_type{});
                else if (wv == kNoiseWave && ln < nloc && t + 2 < T) draw_noise(t + 2);   // row (t + 2) & 1: not the one being read
            }
        };
        step_body<NB, EPB, SENS, true, NW, BOXES, PAIR, const Params __attribute__((address_space(4)))&,
                  const StepIO __attribute__((address_space(4)))&, decltype(hook)>(P, sm, next_env, A->io, t == T - 1, tn, hook);
        // the observation tile of step t + 1 is in sm.obs (its store only reads it) and its action in sm.act_l
    }
}

// Env.getOdometry (environment_new.py:138-181) for n independent (position, orientation quaternion, goal) triples
__global__ void odometry_kernel(int n, const double* __restrict__ px, const double* __restrict__ py, const double* __restrict__ q,
                                const double* __restrict__ goal, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double yaw = yaw_from_quat4(q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]);
    double rel, diff;
    goal_rel(px[i], py[i], goal[2 * i], goal[2 * i + 1], yaw, rel, diff);
    out[3 * i] = yaw; out[3 * i + 1] = rel; out[3 * i + 2] = diff;
}

// clears kRecValid of every env: the cached next-episode records no longer match (tables / rectangles / counters changed)
__global__ void invalidate_records_kernel(int32_t* __restrict__ ep_step, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) ep_step[i] = (int32_t)((uint32_t)ep_step[i] & kStepMask);
}

// The yaw of Env.getOdometry (environment_new.py:142-147) at every start pose, tabulated with the device's own sin / cos /
// atan2 through the same device function the reset kernel calls, so that the step kernel's speculative lanes reproduce
// goal_angles(x, y, yaw, ...) bit for bit without the quaternion round trip (an atan2 off the record path)
__global__ void starts_sc_kernel(const double* __restrict__ starts, int K, double* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const double th = starts[3 * k + 2];
    out[k] = yaw_from_quat(sin(th / 2), cos(th / 2));
}

// ---------------------------------------------------------------- reset (Env.reset, masked)
__global__ void reset_kernel(Params P, const uint8_t* __restrict__ mask, void* __restrict__ obs_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.N) return;
    if (mask && !mask[i]) return;
    const int B = P.B, D = B + 6;
    uint32_t ctr = P.rng_ctr[i];
    double gx, gy, yaw, rel_theta, diff;
    int k0;
    sample_episode<const Params&>(P, *P.rects, i, ctr, k0, gx, gy);
    const double x = P.starts[3 * k0], y = P.starts[3 * k0 + 1], th = P.starts[3 * k0 + 2];
    goal_angles(x, y, th, gx, gy, yaw, rel_theta, diff);
    const double dist = hypot(gx - x, gy - y);
    const float* sp = P.spawn_scan + ((P.per_env ? (size_t)i * P.K : 0) + k0) * B;  // nearest hits at the start pose
    const uint64_t gid = P.env_id_base + (uint64_t)i;
    auto lidar = [&](int b) {
        const float n = (P.sigma > 0.f) ? lidar_noise(P.key0, P.key1, gid, ctr, 0xFFFFu, b) : 0.f;
        float r = sensor_value(sp[b], P.sigma, n, P.below_min_mode);
        if (r == INFINITY) r = 3.5f;  // environment_new.py:193-194
        return r / 3.5f;              // :362
    };
    const float tail[6] = {0.f, 0.f, (float)(dist / P.diag), (float)(yaw / 360), (float)(rel_theta / 360), (float)(diff / 180)};
    if (P.obs_f16) {
        __half* o = reinterpret_cast<__half*>(obs_out) + (size_t)i * D;
        for (int b = 0; b < B; ++b) o[b] = __float2half_rn(lidar(b));
        for (int k = 0; k < 6; ++k) o[B + k] = __float2half_rn(tail[k]);
    } else {
        float* o = reinterpret_cast<float*>(obs_out) + (size_t)i * D;
        for (int b = 0; b < B; ++b) o[b] = lidar(b);
        for (int k = 0; k < 6; ++k) o[B + k] = tail[k];
    }
    P.x[i] = x; P.y[i] = y; P.th[i] = th;
    P.gx[i] = gx; P.gy[i] = gy; P.past_dist[i] = dist;
    P.past_action[i] = make_float2(0.f, 0.f);
    P.ep_step[i] = 0;   // also clears kRecValid: the goal stream moved
    P.ep_ret[i] = 0;
    P.ep_path[i] = 0;
    P.rng_ctr[i] = ctr;
}

// ---------------------------------------------------------------- LiDAR only (spawn scans, tests, tooling)
// thread = (env, pose): thread t casts pose[(t % n_poses_per_env) or t] against env (t / n_poses_per_env)'s map.
//   spawn scans:  n_poses_per_env = K, shared_poses = 1 : pose table [K][3] reused by every env
//   navsim_raycast: n_poses_per_env = 1, shared_poses = 0 : pose[t]
__global__ void raycast_kernel(Params P, const double* __restrict__ pose, int n_poses_per_env, int shared_poses, int n,
                               int raw, float* __restrict__ ranges) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int env = t / n_poses_per_env;
    const size_t pi = shared_poses ? (size_t)(t % n_poses_per_env) : (size_t)t;
    const double x = pose[pi * 3 + 0], y = pose[pi * 3 + 1], th = pose[pi * 3 + 2];
    const double cth = cos(th), sth = sin(th);
    const float ox = (float)(x + kLidarX * cth), oy = (float)(y + kLidarX * sth);
    const float4* __restrict__ sp = P.seg + (P.per_env ? (size_t)env * P.S : 0);
    for (int b = 0; b < P.B; ++b) {
        const double bc = P.beam_cs[b], bs = P.beam_cs[P.B + b];
        const float c = (float)(cth * bc - sth * bs);
        const float s = (float)(sth * bc + cth * bs);
        float best = INFINITY;
        for (int j = 0; j < P.S; ++j) {
            const float4 g = sp[j];
            const float rx = g.x - ox, ry = g.y - oy;
            const float ex = g.z - g.x, ey = g.w - g.y;
            const float k = fmaf(rx, ey, -(ry * ex));
            const float tt = ray_seg(rx, ry, ex, ey, k, c, s);
            best = tt < best ? tt : best;
        }
        // raw: the nearest hit itself (spawn-scan table, noise is applied when it is used); else the noise-free scan value
        ranges[(size_t)t * P.B + b] = raw ? best : sensor_value(best, 0.f, 0.f, P.below_min_mode);
    }
}

// noise-free lidar entries of a reset observation from the nearest hits at the start pose (environment_new.py:192-198,:362)
__global__ void spawn_obs_kernel(const float* __restrict__ best, int n, int below_min_mode, float* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    float r = sensor_value(best[k], 0.f, 0.f, below_min_mode);
    if (r == INFINITY) r = 3.5f;
    out[k] = r / 3.5f;
}

// ---------------------------------------------------------------- return scan (PPO.compute_rtgs)
// R[t] = r[t] + gamma R[t+1], restarting at 0 behind every episode end and at the batch end (ppo.py:658-666), float64
// accumulate, float32 store.  Per env column that is a serial recurrence of T dependent float64 multiply -> add pairs
// (~35 cycles each: 7.5 us for T = 512 however the memory is fed -- rounds 1-2: 40 us, then 16.5 us with an LDS pipeline).
// This kernel splits the column over T instead: a workgroup owns kRtgCols adjacent columns and kRtgChunks chunks of
// kRtgRows rows (thread = column x chunk, 1024 waves at N = 4096 with 64 loads in flight per lane):
//   pass 1   every chunk scans its rows from registers with carry-in 0 -> B (the return at its first row) and
//            A = gamma^kRtgRows, or 0 when an episode ends inside the chunk: R_first = B + A R_in
//   carry    R_in of chunk c = R_first of chunk c + 1, composed serially from the (A, B) pairs in LDS (<= 15 steps)
//   pass 2   the chunk re-runs the reference's recurrence from its true carry-in and stores
// Every row behind (earlier than) an episode end inside its chunk carries the reference's bits by construction; the others see
// a carry-in whose float64 rounding differs from the serial chain's (relative 1e-15), which moves the float32 store by one ulp
// with probability ~2e-8 per element.  Contract: <= 1 float32 ulp from ppo.py:660-669, flags untouched (tests state it);
// `exact` != 0 (argument of navsim_rtg_scan / navsim_gae_scan) selects the serial kernel below, which is bit-identical.  "if ended: disc = 0" is folded into the
// factor: disc * 0 is (+-)0 and r + (+-)0 == r, so the value is that of ppo.py:660-665 while the select leaves the chain.
// T > 512: super-chunks of 512 rows from the batch end, the carry between them is the stored float64 of pass 2.
constexpr int kRtgCols = 16, kRtgChunks = 16, kRtgRows = 32, kRtgSuper = kRtgChunks * kRtgRows;

//
// GAE (north-star extension, off by default; navsim_gae_scan): the lambda-return obeys the same kind of recurrence,
//   R[t] = (r[t] + gamma (1 - lambda) V[t+1]) + gamma lambda R[t+1]      behind an episode end / the batch end: R[t] = r[t]
// (episode ends and the batch end are terminal as in ppo.py:552-553,658-666; last_value bootstraps the batch end only when
// given), advantage A[t] = float32(R[t]) - V[t].  lambda = 1 without bootstrap: the additive term is r[t] + 0 and the factor
// gamma * 1, i.e. the arithmetic of the return scan bit for bit, and A = rtg - V (ppo.py:277).
template <bool GAE>
__global__ __launch_bounds__(256) void rtg_kernel_split(const float* __restrict__ rew, const uint8_t* __restrict__ ended, int T, int N,
                                                        double gamma, float* __restrict__ out, const float* __restrict__ value,
                                                        const float* __restrict__ last_value, double lam, float* __restrict__ adv) {
    __shared__ double s_b[kRtgChunks][kRtgCols], s_a[kRtgChunks][kRtgCols], s_sup[kRtgCols];
    const int tid = threadIdx.x, col = tid & (kRtgCols - 1), c = tid >> 4;
    // Workgroups go round-robin over the 8 XCDs (one L2 each) while a 128-byte line holds the rewards of 2 column blocks and
    // the flags of 8: give each XCD runs of 8 adjacent column blocks so that a line is fetched into one L2, not into eight
    // (measured 33.6 MB instead of 10.5 MB over the fabric otherwise).  Needs the block count to be a multiple of 64.
    int q = blockIdx.x;
    if ((gridDim.x & 63) == 0) {
        const int x = q & 7, y = q >> 3;
        q = ((y >> 3) * 8 + x) * 8 + (y & 7);
    }
    const size_t n = (size_t)q * kRtgCols + col;   // N % 16 == 0: all 16 columns exist
    const double gl = GAE ? gamma * lam : gamma, gv = GAE ? gamma * (1.0 - lam) : 0.0;
    double gpow = 1.0;
#pragma unroll
    for (int u = 0; u < kRtgRows; ++u) gpow *= gl;
    double rsup = (GAE && last_value) ? (double)last_value[n] : 0.0;  // ppo.py:660 (GAE: R beyond the batch end = the bootstrap value)
    for (int t_hi = T; t_hi > 0; t_hi -= kRtgSuper) {
        // rows t0 .. t0 + 31 of this chunk; t < 0 only below the first row of the batch (the lowest super-chunk): loaded from
        // row 0 (any valid address), never stored, and nothing valid depends on them (the scan runs towards them)
        const int t0 = t_hi - kRtgSuper + c * kRtgRows;
        float r[kRtgRows];
        uint8_t e[kRtgRows];
        float v[GAE ? kRtgRows + 1 : 1];   // GAE: V of rows t0 .. t0 + 32 (the row after the chunk; beyond the batch: last_value or 0)
        if (GAE) {
            const int tn = t0 + kRtgRows;
            v[kRtgRows] = tn < T ? value[(size_t)max(tn, 0) * N + n] : (last_value ? last_value[n] : 0.f);
        }
#pragma unroll
        for (int u = kRtgRows - 1; u >= 0; --u) {   // requested in the order pass 1 consumes them
            const size_t k = (size_t)max(t0 + u, 0) * N + n;
            r[u] = rew[k];
            e[u] = ended[k];
            if (GAE) v[u] = value[k];
        }
        // The batch end is terminal like an episode end (ppo.py:601,658) unless last_value bootstraps it: then the last row's
        // factor stays gamma lambda and R beyond it is last_value (rsup above), so R[T-1] = r + gamma (1-lambda) V_last + gamma lambda V_last
        double b = 0;
        bool any = false;
#pragma unroll
        for (int u = kRtgRows - 1; u >= 0; --u) {
            const double g = e[u] ? 0.0 : gl;
            const double add = GAE ? (double)r[u] + (e[u] ? 0.0 : gv * (double)v[u + 1]) : (double)r[u];
            b = add + b * g;
            any |= e[u] != 0;
        }
        s_b[c][col] = b;
        s_a[c][col] = any ? 0.0 : gpow;
        __syncthreads();
        double ak[kRtgChunks], bk[kRtgChunks];
#pragma unroll
        for (int k = 1; k < kRtgChunks; ++k) {
            ak[k] = s_a[k][col];
            bk[k] = s_b[k][col];
        }
        double R = rsup;
#pragma unroll
        for (int k = kRtgChunks - 1; k >= 1; --k)
            if (k > c) R = bk[k] + ak[k] * R;
#pragma unroll
        for (int u = kRtgRows - 1; u >= 0; --u) {
            const double g = e[u] ? 0.0 : gl;
            const double add = GAE ? (double)r[u] + (e[u] ? 0.0 : gv * (double)v[u + 1]) : (double)r[u];
            R = add + R * g;   // ppo.py:665
            if (t0 + u >= 0) {
                const size_t k = (size_t)(t0 + u) * N + n;
                if (out) out[k] = (float)R;   // ppo.py:669
                if (GAE) adv[k] = (float)R - v[u];
            }
        }
        if (t_hi > kRtgSuper) {   // uniform: another super-chunk follows
            if (c == 0) s_sup[col] = R;
            __syncthreads();
            rsup = s_sup[col];
        }
    }
}

// any N / alignment, and the bit-exact mode: thread = env column, reverse over T
__global__ void rtg_kernel_generic(const float* __restrict__ rew, const uint8_t* __restrict__ ended, int T, int N, double gamma,
                                   float* __restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double disc = 0;  // ppo.py:660
#pragma unroll 8
    for (int t = T - 1; t >= 0; --t) {
        const size_t k = (size_t)t * N + n;
        const float r = rew[k];
        if (ended[k]) disc = 0;
        disc = (double)r + disc * gamma;  // ppo.py:665
        out[k] = (float)disc;             // ppo.py:669
    }
}

// GAE, any N: thread = env column, reverse over T (the recurrence of rtg_kernel_split<true>, serially)
__global__ void gae_kernel_generic(const float* __restrict__ rew, const uint8_t* __restrict__ ended, const float* __restrict__ value,
                                   const float* __restrict__ last_value, int T, int N, double gamma, double lam,
                                   float* __restrict__ adv, float* __restrict__ ret) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double gl = gamma * lam, gv = gamma * (1.0 - lam);
    double R = last_value ? (double)last_value[n] : 0.0;
    float vnext = last_value ? last_value[n] : 0.f;
    for (int t = T - 1; t >= 0; --t) {
        const size_t k = (size_t)t * N + n;
        const bool e = ended[k] != 0;
        const float vt = value[k];
        const double g = e ? 0.0 : gl;
        const double add = (double)rew[k] + (e ? 0.0 : gv * (double)vnext);
        R = add + R * g;
        if (ret) ret[k] = (float)R;
        adv[k] = (float)R - vt;
        vnext = vt;
    }
}

}  // namespace

// ==================================================================== host side / C ABI
struct navsim {
    navsim_cfg cfg;
    Params P;
    void* state_block = nullptr;   // one allocation for all per-env state
    double* beam_cs_dev = nullptr;
    Rects* rects_dev = nullptr;
    Rects rects_host;
    float4* seg_sorted_dev = nullptr;   // shared maps of 65..4096 segments: the handle's own Morton-ordered copy
    float4* tile_box_dev = nullptr;
    float* spawn_scan_dev = nullptr;
    float* spawn_obs_dev = nullptr;
    double* starts_dev = nullptr;   // [K][3]
    double* starts_sc_dev = nullptr;  // [K]
    double* goals_dev = nullptr;    // [G][2]
    const float* seg_dev = nullptr;
    bool has_map = false;
    // kernel-shape overrides of THIS handle (navsim_set_shape; tests and A/B timing): envs per workgroup, 0 = the rule of pick_epb
    // below; pair_cast false = 64-segment passes for every map
    int force_epb = 0;
    bool pair_cast = true;
    int pair_cast_request = -1;   // what navsim_set_shape was given (-1 = the rule): reported by navsim_get_info
};

// Envs per workgroup.  Bigger workgroups make the float64 lanes of the geometry / rules phases denser (a wave instruction
// costs the same with 16 or 64 active lanes) and start fewer workgroups, but need N / EPB >= the number of CUs to fill the
// chip: measured (tools/time_step.py) 16384 envs: 18.5 / 16.7 / 15.9 us with 16 / 32 / 64; 8192: 69.8 / 64.8 / 73.9; 4096: 8.9 / 8.6 / 9.8.
// Round 4 swept shard size x shape x beam count on shared maps (tools/epb_sweep.py, profiles/r04_epb_sweep.txt), us per step,
// shapes 8 / 16 / 32:
//   36 beams, tape form    1024 envs 6.3 / 10.0 / 9.9   2048: 6.5 / 10.1 / 9.8   4096: 7.0 / 10.2 / 9.9   8192: 13.0 / 11.0 / 10.0
//   36 beams, one launch   1024: 10.5 / 12.9 / 12.7     2048: 11.0 / 13.1 / 12.9  4096: 11.9 / 13.5 / 13.1  8192: 13.9 / 14.6 / 13.5
//   10 beams, tape form    1024:  4.45 / 5.06 / 4.72    2048: 4.49 / 5.09 / 4.70  4096: 4.94 / 5.27 / 4.81
//   10 beams, one launch   1024:  8.6 / 7.6 / 7.2       2048: 8.8 / 7.8 / 7.3     4096: 9.2 / 8.0 / 7.6
// With 36 beams a workgroup's chain is long (36 beam directions per env in the pose phase, stage B per beam group) and eight envs
// on four waves, two to four workgroups per CU, overlap their chains: up to 4096 envs the 8-env shape; small 10-beam shards take it
// in the tape form only (a launch per step pays for 4 x as many workgroups to start) and the 32-env shape otherwise.
// The 36-beam 8-env shape then went from four to eight waves (one env's beam groups per wave; min_waves_per_simd keeps two
// workgroups on a CU): tape form 1024 / 2048 / 4096 envs 6.3 / 6.5 / 7.0 -> 5.2 / 5.4 / 6.3 us per step, a launch per step
// unchanged (10.4 / 11.0 / 12.0); with 10 beams eight waves gain 2-3 % at 1024-2048 envs and nothing at 4096: four stay.
// The same on the shared 2048-segment house map (tile boxes; EPB_MAP=house): 8 / 32 envs per workgroup, tape form 1024-4096 envs
// 18.5-20.2 / 29.1-29.8 us per step, 8192: 25.0 / 30.7; one launch per step 1024-4096: 20.8-22.6 / 31.3-32.2, 8192: 38.2 / 32.6;
// 16384 envs: the 64-env shape (35 / 37 us) ahead of every smaller one.
// Per-env stage_2 maps (EPB_MAP=per_env), 8 / 32: tape 1024-2048 envs 5.6 / 7.2, 4096: 6.2 / 7.3, 8192: 11.5 / 7.7; a launch per step 9.2-10.2 / 9.2-9.4.
// Waves of the 8-env workgroup.  Eight (one env per wave) halve each wave's cast chain: on the house map (tile boxes) the tape runs
// 18.4 / 18.7 / 20.2 -> 12.1 / 12.6 / 15.0 us per step at 1024 / 2048 / 4096 envs and a launch per step 20.7 / 21.2 / 22.4 -> 14.8 /
// 15.4 / 18.0; per-env stage_2 maps as a tape 5.5 / 5.6 / 6.1 -> 4.6 / 4.7 / 5.6; 36 beams see the comment above.  Two such
// workgroups fit a CU (128 VGPRs each: min_waves_per_simd), so beyond 4096 envs the shape runs in rounds and is not chosen.
// The 16-env workgroup went from four to eight waves the same way (two envs per wave; us per step, the rule before / 16 envs on eight waves):
//   stage_1, 10 beams, tape form   1024: 4.30 / 3.90   2048: 4.36 / 3.91   4096: 4.80 / 4.09   8192: 4.97 / 4.97   (a launch per step: equal)
//   per-env maps, a launch per step 1024: 9.13 / 7.86   2048: 9.24 / 8.18   4096: 9.42 / 8.47   8192: 10.5 / 10.0   (tape: the 8-env shape stays)
//   36 beams, a launch per step    1024: 10.3 / 9.3    2048: 11.0 / 9.7    4096: 12.1 / 10.1   8192: 13.6 / 11.8   16384: 25.0 / 21.3
//   house map, tape form           8192: 24.6 (8 envs on four waves) / 23.6 -- the last user of a four-wave shape, which is gone
// With the 128-VGPR bound on the 16-env instantiations that need it (min_waves_per_simd), the rule before / 16 envs: 36 beams, tape
// 8192: 9.95 / 8.16   16384: 19.7 / 16.0; house map, a launch per step 8192: 32.7 / 25.8; per-env maps, tape 8192: 7.58 / 6.89;
// stage_1, a launch per step 1024: 7.17 / 6.77   4096: 7.59 / 7.23   8192: 7.96 / 8.00.
static int pick_epb(const navsim* h, int n_envs, int n_beams, bool tape, bool boxes, bool per_env) {
    if (h->force_epb >= 8) return h->force_epb;
    if (n_beams > 16) return (tape && n_envs <= 4096) ? 8 : (n_envs <= 16384 ? 16 : 32);
    if (boxes && n_envs <= 8192) return n_envs <= 4096 ? 8 : 16;
    if (per_env && tape && n_envs <= 8192) return n_envs <= 4096 ? 8 : 16;
    if (per_env && !tape && n_envs <= 8192) return 16;
    if (n_envs >= 16384) return 64;
    if (!boxes && !per_env && n_envs <= 4096) return 16;
    return 32;
}

// The instantiation navsim_step (tape = false) / navsim_step_seq (tape = true) launches for this handle: envs and waves per
// workgroup and the cast variant -- 0 = 64-segment passes, 1 = 128-segment passes (maps of 65+ segments without tile boxes:
// per-env maps, shared maps beyond 4096 segments), 2 = the same with non-temporal loads (the stream does not fit the caches,
// navsim_set_map; exists for the big 10-beam shapes), 3 = tile bounding boxes (shared maps of 65..4096 segments).
struct ShapePick {
    int epb, waves, cast;
};
static ShapePick pick_shape(const navsim* h, bool tape) {
    const int NB = h->P.B;
    const bool boxes = h->P.tile_box != nullptr;
    const int want = pick_epb(h, h->P.N, NB, tape, boxes, h->P.per_env != 0);
    ShapePick sp;
    if (want == 8) sp.epb = 8, sp.waves = 8;                                    // eight waves: one env per wave
    else if (want == 32 || (want == 64 && NB > 10)) sp.epb = 32, sp.waves = 8;  // float64 geometry / rules lanes twice as dense
    else if (want == 64) sp.epb = 64, sp.waves = 16;                            // (10 beams: the 36-beam tile does not fit the LDS)
    else sp.epb = 16, sp.waves = 8;                                             // two envs per wave
    const bool pair = h->pair_cast && !boxes && h->P.S > 64 && h->P.seg_pack_log2 == 6;
    const bool nt = (h->P.per_env & (tape ? 2 : 4)) != 0;   // navsim_set_map: the per-env stream does not fit the L2s / the Infinity Cache
    sp.cast = boxes ? 3 : (pair ? ((nt && NB == 10 && sp.epb >= 32) ? 2 : 1) : 0);
    return sp;
}

// KERNEL: step_kernel | steps_kernel; `go(kernel, epb, waves)` launches
#define NAVSIM_DISPATCH_CAST(KERNEL, EPB_, NW_)                                                            \
    do {                                                                                                   \
        if (sp.cast == 3) {                                                                                \
            if (sens) go(KERNEL<NB, EPB_, true, NW_, true>, EPB_, NW_);                                    \
            else go(KERNEL<NB, EPB_, false, NW_, true>, EPB_, NW_);                                        \
        } else if (sp.cast == 2) {                                                                         \
            constexpr int kNT = (NB == 10 && EPB_ >= 32) ? 2 : 1;   /* the streaming variant exists for the big shapes */ \
            if (sens) go(KERNEL<NB, EPB_, true, NW_, false, kNT>, EPB_, NW_);                              \
            else go(KERNEL<NB, EPB_, false, NW_, false, kNT>, EPB_, NW_);                                  \
        } else if (sp.cast == 1) {                                                                         \
            if (sens) go(KERNEL<NB, EPB_, true, NW_, false, 1>, EPB_, NW_);                                \
            else go(KERNEL<NB, EPB_, false, NW_, false, 1>, EPB_, NW_);                                    \
        } else {                                                                                           \
            if (sens) go(KERNEL<NB, EPB_, true, NW_, false>, EPB_, NW_);                                   \
            else go(KERNEL<NB, EPB_, false, NW_, false>, EPB_, NW_);                                       \
        }                                                                                                  \
    } while (0)
#define NAVSIM_DISPATCH_SHAPE(KERNEL)                                              \
    do {                                                                           \
        if (sp.epb == 8) NAVSIM_DISPATCH_CAST(KERNEL, 8, 8);                       \
        else if (sp.epb == 32) NAVSIM_DISPATCH_CAST(KERNEL, 32, 8);                \
        else if (sp.epb == 64) {                                                   \
            if constexpr (NB == 10) NAVSIM_DISPATCH_CAST(KERNEL, 64, 16);          \
        } else NAVSIM_DISPATCH_CAST(KERNEL, 16, 8);                                \
    } while (0)

// `query` non-null: the selected instantiation is not launched, its resources (registers, scratch = spilled registers, LDS) are read
// into *query (navsim_get_info: the run-time check that what the rule picks fits the way the rule assumes)
template <int NB>
static void launch_step(const navsim* h, const float* action, const float* past, void* obs, float* reward, uint8_t* done,
                        uint8_t* arrive, uint8_t* ended, float* ep_ret, int32_t* ep_len, float* ep_path, hipStream_t st,
                        hipFuncAttributes* query = nullptr) {
    const bool sens = h->P.sigma > 0.f || h->P.below_min_mode != 0;
    const StepKArgs ka = {h->P, StepIO{(const float2*)action, (const float2*)past, obs, reward, done, arrive, ended, ep_ret, ep_len,
                                       ep_path}};
    auto go = [&](auto kernel, int epb, int nw) {
        if (query) (void)hipFuncGetAttributes(query, reinterpret_cast<const void*>(kernel));
        else hipLaunchKernelGGL(kernel, dim3((h->P.N + epb - 1) / epb), dim3(64 * nw), 0, st, ka);
    };
    const ShapePick sp = pick_shape(h, false);
    NAVSIM_DISPATCH_SHAPE(step_kernel);
}

// navsim_step_seq: the same shapes and cast variants, all steps of the tape in one launch
template <int NB>
static void launch_steps(const navsim* h, const SeqArgs& R, hipStream_t st, hipFuncAttributes* query = nullptr) {
    const bool sens = h->P.sigma > 0.f || h->P.below_min_mode != 0;
    const SeqKArgs ka = {h->P, R};
    auto go = [&](auto kernel, int epb, int nw) {
        if (query) (void)hipFuncGetAttributes(query, reinterpret_cast<const void*>(kernel));
        else hipLaunchKernelGGL(kernel, dim3((h->P.N + epb - 1) / epb), dim3(64 * nw), 0, st, ka);
    };
    const ShapePick sp = pick_shape(h, true);
    NAVSIM_DISPATCH_SHAPE(steps_kernel);
}
#undef NAVSIM_DISPATCH_SHAPE
#undef NAVSIM_DISPATCH_CAST

// navsim_rollout_mlp64's kernel for this handle: kind 0 = unavailable, 1 = rollout_kernel (EPB envs on 8 waves, any beam
// count), 2 = rollout_big_kernel (64 envs on 16 waves, 10 beams, the cast variants of the step kernel).
// A 16-env workgroup is ONE latency chain per step whatever its size: measured (tools/time_rollout.py, 512 steps) 4096 / 2048 /
// 1024 / 512 envs take 2.92 / 2.90 / 2.89 / 2.97 ms with 16 envs per workgroup, 3.05-3.06 ms with 8 and 2.96 ms with 4 wherever the
// grid still fits one round of 256 CUs (a second round doubles the time: 256 registers x 8 waves fill a CU) -- spreading a small
// shard over more CUs buys nothing.  Beyond 4096 envs (10 beams): the tape kernel's 64-env workgroup with the policy phase inside
// every step -- the 16-env shape needs a second round of workgroups there (measured, us per step, 16-env / 64-env shape: 4608 envs
// 10.6 / 7.3, 8192 10.5 / 7.4, 12288 15.5 / 7.4, 16384 20.5 / 7.6; 4096: 5.3 / 7.5), while a 64-env workgroup costs the same
// 7.3-7.6 us whether 72 or 256 CUs hold one.  36 beams: the 16-env shape at every size (the 64-env tile does not fit the LDS).
struct RolloutPick {
    int kind, epb, waves, cast;
};
static RolloutPick pick_rollout(const navsim* h) {
    RolloutPick rp = {0, 0, 0, 0};
    if (h->P.B != 10 && h->P.B != 36) return rp;
    const bool boxes = h->P.tile_box != nullptr;
    const bool pair = h->pair_cast && !boxes && h->P.S > 64 && h->P.seg_pack_log2 == 6;
    if (h->P.B == 10 && (h->force_epb == 64 || h->force_epb == 32 || (h->force_epb == 0 && h->P.N > 16 * 256))) {
        rp.kind = 2; rp.epb = 64; rp.waves = 16;
        // tile-box maps (the 2048-segment house map: a vector-issue-bound cast) on 4097..8192 envs: 64-env workgroups would leave half
        // the CUs empty (8192 envs: 35.4 us per step) -- 32 envs on 8 waves, one workgroup on every CU
        if (h->force_epb == 32 || (h->force_epb == 0 && boxes && h->P.N <= 32 * 256)) rp.epb = 32, rp.waves = 8;
        rp.cast = boxes ? 3 : (pair ? ((h->P.per_env & 2) ? 2 : 1) : 0);   // 2: one step's per-env stream exceeds 1.25 x the Infinity Cache
        if (rp.epb == 32 && rp.cast != 3) rp.cast = 0;   // (the forced 32-env shape exists with 64-segment passes and with tile boxes)
        return rp;
    }
    rp.kind = 1; rp.waves = 8;
    rp.epb = (h->P.B == 10 && !boxes && (h->force_epb == 4 || h->force_epb == 8)) ? h->force_epb : 16;
    rp.cast = boxes ? 3 : 0;   // (round 5: the 16-env shape has the tile-box cast too -- the house map on shards up to 4096 envs)
    return rp;
}

static int invalidate_records(navsim* h, hipStream_t st) {
    hipLaunchKernelGGL(invalidate_records_kernel, dim3((h->P.N + 255) / 256), dim3(256), 0, st, h->P.ep_step, h->P.N);
    HIP_TRY(hipGetLastError());
    return NAVSIM_OK;
}

// Uploads the start-pose table [K][3] and tabulates its half-angle (cos, sin) on the device.
static int upload_starts(navsim* h, const double* starts_host, int K, hipStream_t st) {
    (void)hipFree(h->starts_dev);
    (void)hipFree(h->starts_sc_dev);
    h->starts_dev = h->starts_sc_dev = nullptr;
    HIP_TRY(hipMalloc(&h->starts_dev, sizeof(double) * 3 * K));
    HIP_TRY(hipMalloc(&h->starts_sc_dev, sizeof(double) * K));
    HIP_TRY(hipMemcpy(h->starts_dev, starts_host, sizeof(double) * 3 * K, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(starts_sc_kernel, dim3((K + 63) / 64), dim3(64), 0, st, h->starts_dev, K, h->starts_sc_dev);
    HIP_TRY(hipGetLastError());
    h->P.starts = h->starts_dev;
    h->P.starts_sc = h->starts_sc_dev;
    h->P.K = K;
    return NAVSIM_OK;
}

static const double kResetRects[4][4] = {  // environment_new.py:340-343
    {1.7, 2.3, -1.2, 1.2}, {-2.3, -1.7, -1.2, 1.2}, {-1.2, 1.2, 1.7, 2.3}, {-1.2, 1.2, -2.3, -1.7}};
static const double kRespawnRects[4][4] = {  // environment_new.py:248-251
    {1.6, 2.4, -1.4, 1.4}, {-2.4, -1.6, -1.4, 1.4}, {-1.4, 1.4, 1.6, 2.4}, {-1.4, 1.4, -2.4, -1.6}};

// Allocations and tables of a new handle (navsim_create frees the handle if any step fails).
static int init_handle(navsim* h, const navsim_cfg* cfg) {
    h->cfg = *cfg;
    const size_t N = (size_t)cfg->n_envs;
    const int B = cfg->n_beams;
    // one block: 8 f64 state arrays + 6 + 2 f64 record arrays, 2 float4, float2 past_action, 2 uint2, i32 ep_step, 2 u32
    const size_t bytes = N * (16 * sizeof(double) + 2 * sizeof(float4) + sizeof(float2) + 2 * sizeof(uint2) +
                              sizeof(int32_t) + 2 * sizeof(uint32_t));
    HIP_TRY(hipMalloc(&h->state_block, bytes));
    HIP_TRY(hipMemset(h->state_block, 0, bytes));
    Params& P = h->P;
    std::memset(&P, 0, sizeof(P));
    double* d = reinterpret_cast<double*>(h->state_block);
    P.x = d; P.y = d + N; P.th = d + 2 * N; P.gx = d + 3 * N; P.gy = d + 4 * N;
    P.past_dist = d + 5 * N; P.ep_ret = d + 6 * N; P.ep_path = d + 7 * N;
    P.rec_g = d + 8 * N; P.rsp_g = d + 14 * N;
    P.rec_tail = reinterpret_cast<float4*>(d + 16 * N);   // 16-byte aligned: hipMalloc base + a multiple of 128 N bytes
    P.rec_ck = reinterpret_cast<uint2*>(P.rec_tail + 2 * N);
    P.past_action = reinterpret_cast<float2*>(P.rec_ck + 2 * N);
    P.ep_step = reinterpret_cast<int32_t*>(P.past_action + N);
    P.rng_ctr = reinterpret_cast<uint32_t*>(P.ep_step + N);
    P.rsp_ctr = P.rng_ctr + N;
    P.seg_pack_log2 = 6;
    P.N = cfg->n_envs;
    P.B = B;
    P.max_ep_steps = cfg->max_episode_steps;
    P.auto_reset = cfg->auto_reset;
    P.respawn = cfg->respawn_on_arrive;
    P.obs_f16 = cfg->obs_f16;
    P.below_min_mode = cfg->lidar_below_min ? 1 : 0;
    P.sigma = cfg->lidar_noise_sigma;
    P.key0 = (uint32_t)cfg->seed;
    P.key1 = (uint32_t)(cfg->seed >> 32);
    P.env_id_base = cfg->env_id_base;
    P.thr = cfg->threshold_arrive;
    P.spawn_x = cfg->spawn_x; P.spawn_y = cfg->spawn_y; P.spawn_yaw = cfg->spawn_yaw;
    P.goal_lo = cfg->goal_lo; P.goal_hi = cfg->goal_hi;
    P.diag = std::sqrt(2.0) * (3.8 + 3.8);  // environment_new.py:21

    // beam table: samples evenly over [min_angle, max_angle] (gazebo.xacro:110-115)
    double cs[2 * 64];
    for (int i = 0; i < B; ++i) {
        const double phi = kAngleMin + (double)i * ((kAngleMax - kAngleMin) / (double)(B - 1));
        cs[i] = std::cos(phi);
        cs[B + i] = std::sin(phi);
    }
    HIP_TRY(hipMalloc(&h->beam_cs_dev, sizeof(double) * 2 * B));
    HIP_TRY(hipMemcpy(h->beam_cs_dev, cs, sizeof(double) * 2 * B, hipMemcpyHostToDevice));
    P.beam_cs = h->beam_cs_dev;

    std::memset(&h->rects_host, 0, sizeof(Rects));
    std::memcpy(h->rects_host.r[0], kResetRects, sizeof(kResetRects));
    std::memcpy(h->rects_host.r[1], kRespawnRects, sizeof(kRespawnRects));
    h->rects_host.n[0] = h->rects_host.n[1] = 4;
    HIP_TRY(hipMalloc(&h->rects_dev, sizeof(Rects)));
    HIP_TRY(hipMemcpy(h->rects_dev, &h->rects_host, sizeof(Rects), hipMemcpyHostToDevice));
    P.rects = h->rects_dev;

    // start-pose table: K = 1, the cfg spawn pose (turtlebot3_stage_1.launch:3-5) until navsim_set_spawn_sampler
    {
        const double sp[3] = {cfg->spawn_x, cfg->spawn_y, cfg->spawn_yaw};
        const int rc = upload_starts(h, sp, 1, nullptr);
        if (rc != NAVSIM_OK) return rc;
        P.G = 0;
        P.goals = nullptr;
    }
    return NAVSIM_OK;
}

#pragma GCC visibility push(default)
extern "C" {

int navsim_version(void) { return NAVSIM_ABI_VERSION; }

const char* navsim_last_error(void) { return g_err.c_str(); }

void navsim_default_cfg(navsim_cfg* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->n_envs = 1;
    c->n_beams = 10;            // gazebo.xacro:111
    c->threshold_arrive = 0.2;  // environment_new.py:45
    c->goal_lo = -3.6;          // environment_new.py:337
    c->goal_hi = 3.6;
}

int navsim_create(const navsim_cfg* cfg, navsim_t** out) {
    if (!cfg || !out) return fail(NAVSIM_E_ARG, "navsim_create: null argument");
    *out = nullptr;
    if (cfg->n_envs < 1) return fail(NAVSIM_E_ARG, "navsim_create: n_envs must be >= 1");
    if (cfg->n_beams != 10 && cfg->n_beams != 36)
        return fail(NAVSIM_E_ARG, "navsim_create: n_beams must be 10 or 36");
    if (!(cfg->goal_hi > cfg->goal_lo)) return fail(NAVSIM_E_ARG, "navsim_create: empty goal box");
    if (!(cfg->lidar_noise_sigma >= 0.f)) return fail(NAVSIM_E_ARG, "navsim_create: negative lidar_noise_sigma");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(NAVSIM_E_HIP, "navsim_create: no HIP device (there is no CPU path)");

    navsim* h = new navsim();
    const int rc = init_handle(h, cfg);
    if (rc != NAVSIM_OK) {
        navsim_destroy(h);   // frees whatever was allocated before the failure
        return rc;
    }
    *out = h;
    return NAVSIM_OK;
}

void navsim_destroy(navsim_t* h) {
    if (!h) return;
    (void)hipFree(h->state_block);
    (void)hipFree(h->beam_cs_dev);
    (void)hipFree(h->rects_dev);
    (void)hipFree(h->spawn_scan_dev);
    (void)hipFree(h->spawn_obs_dev);
    (void)hipFree(h->seg_sorted_dev);
    (void)hipFree(h->tile_box_dev);
    (void)hipFree(h->starts_dev);
    (void)hipFree(h->starts_sc_dev);
    (void)hipFree(h->goals_dev);
    delete h;
}

int navsim_set_shape(navsim_t* h, int32_t envs_per_workgroup, int32_t pair_cast) {
    if (!h) return fail(NAVSIM_E_ARG, "navsim_set_shape: null handle");
    const int v = envs_per_workgroup;
    if (!(v == 0 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) || pair_cast < -1 || pair_cast > 1)
        return fail(NAVSIM_E_ARG, "navsim_set_shape: envs_per_workgroup is 0 | 4 | 8 | 16 | 32 | 64, pair_cast -1 | 0 | 1");
    h->force_epb = v;
    h->pair_cast = pair_cast != 0;
    h->pair_cast_request = pair_cast;
    return NAVSIM_OK;
}

int navsim_get_info(navsim_t* h, navsim_info* out) {
    if (!h || !out) return fail(NAVSIM_E_ARG, "navsim_get_info: null argument");
    std::memset(out, 0, sizeof(*out));
    out->abi_version = NAVSIM_ABI_VERSION;
    out->n_envs = h->P.N; out->n_beams = h->P.B; out->obs_f16 = h->P.obs_f16;
    out->n_segments = h->P.S; out->per_env_map = h->P.per_env & 1; out->tile_boxes = h->P.tile_box != nullptr;
    out->has_map = h->has_map;
    out->forced_epb = h->force_epb; out->forced_pair_cast = h->pair_cast_request;
    if (h->has_map) {
        const ShapePick a = pick_shape(h, false), b = pick_shape(h, true);
        out->step_epb = a.epb; out->step_waves = a.waves; out->step_cast = a.cast;
        out->seq_epb = b.epb; out->seq_waves = b.waves; out->seq_cast = b.cast;
        const RolloutPick r = pick_rollout(h);
        out->rollout_kind = r.kind; out->rollout_epb = r.epb; out->rollout_waves = r.waves; out->rollout_cast = r.cast;
        hipFuncAttributes fa;
        std::memset(&fa, 0, sizeof(fa));
        if (h->P.B == 10) launch_step<10>(h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &fa);
        else launch_step<36>(h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &fa);
        out->step_vgprs = fa.numRegs; out->step_scratch_bytes = (int32_t)fa.localSizeBytes; out->step_lds_bytes = (int32_t)fa.sharedSizeBytes;
        std::memset(&fa, 0, sizeof(fa));
        SeqArgs sr;
        std::memset(&sr, 0, sizeof(sr));
        if (h->P.B == 10) launch_steps<10>(h, sr, nullptr, &fa);
        else launch_steps<36>(h, sr, nullptr, &fa);
        out->seq_vgprs = fa.numRegs; out->seq_scratch_bytes = (int32_t)fa.localSizeBytes; out->seq_lds_bytes = (int32_t)fa.sharedSizeBytes;
    }
    return NAVSIM_OK;
}

int navsim_set_goal_rects(navsim_t* h, int32_t which, const double* rects_host, int32_t n_rects) {
    if (!h || (which != 0 && which != 1) || n_rects < 0 || n_rects > kMaxRects || (n_rects && !rects_host))
        return fail(NAVSIM_E_ARG, "navsim_set_goal_rects: bad argument");
    HIP_TRY(hipDeviceSynchronize());
    std::memcpy(h->rects_host.r[which], rects_host, sizeof(double) * 4 * n_rects);
    h->rects_host.n[which] = n_rects;
    HIP_TRY(hipMemcpy(h->rects_dev, &h->rects_host, sizeof(Rects), hipMemcpyHostToDevice));
    const int rc = invalidate_records(h, nullptr);
    if (rc != NAVSIM_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return NAVSIM_OK;
}

static int rebuild_spawn_scans(navsim* h, hipStream_t st) {
    Params& P = h->P;
    const size_t n_poses = (size_t)(P.per_env ? P.N : 1) * P.K;
    if (h->spawn_scan_dev) {
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipFree(h->spawn_scan_dev));
        HIP_TRY(hipFree(h->spawn_obs_dev));
        h->spawn_scan_dev = h->spawn_obs_dev = nullptr;
    }
    HIP_TRY(hipMalloc(&h->spawn_scan_dev, n_poses * P.B * sizeof(float)));
    HIP_TRY(hipMalloc(&h->spawn_obs_dev, n_poses * P.B * sizeof(float)));
    P.spawn_scan = h->spawn_scan_dev;
    P.spawn_obs = h->spawn_obs_dev;
    hipLaunchKernelGGL(raycast_kernel, dim3((unsigned)((n_poses + 63) / 64)), dim3(64), 0, st, P, h->starts_dev, P.K, 1,
                       (int)n_poses, 1, h->spawn_scan_dev);
    hipLaunchKernelGGL(spawn_obs_kernel, dim3((unsigned)((n_poses * P.B + 255) / 256)), dim3(256), 0, st, h->spawn_scan_dev,
                       (int)(n_poses * P.B), P.below_min_mode, h->spawn_obs_dev);
    HIP_TRY(hipGetLastError());
    return NAVSIM_OK;
}

int navsim_set_map(navsim_t* h, const float* seg_dev, int32_t n_segments, int32_t per_env, void* stream) {
    if (!h || !seg_dev || n_segments < 1) return fail(NAVSIM_E_ARG, "navsim_set_map: bad argument");
    Params& P = h->P;
    P.seg = reinterpret_cast<const float4*>(seg_dev);
    P.S = n_segments;
    P.per_env = per_env ? 1 : 0;
    // bit 1: the per-env segment stream of one step (N x S x 16 B) exceeds 1.25 x the 256 MiB Infinity Cache -> the persistent
    // kernels load it non-temporally too (the PAIR == 2 instantiations)
    if (per_env && (size_t)P.N * (size_t)n_segments * sizeof(float4) >= (size_t)320 << 20) P.per_env |= 2;
    // bit 2: it does not fit the 8 x 4 MiB of L2 -> the one-launch-per-step kernel loads it non-temporally
    if (per_env && (size_t)P.N * (size_t)n_segments * sizeof(float4) >= (size_t)32 << 20) P.per_env |= 4;
    P.seg_pack_log2 = 6;
    P.tile_box = nullptr;
    if (h->seg_sorted_dev || h->tile_box_dev) {
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        (void)hipFree(h->seg_sorted_dev);
        (void)hipFree(h->tile_box_dev);
        h->seg_sorted_dev = h->tile_box_dev = nullptr;
    }
    if (!per_env && n_segments > 64 && n_segments <= 4096) {
        // A shared map big enough to tile: keep an own copy in Morton order of the segment midpoints, so that a 64-segment
        // tile is a compact patch of the map, with the bounding box of every tile.  The order of the segments does not
        // change the scan (the nearest hit is a min).
        const int S = n_segments, nt = (S + 63) / 64;
        std::vector<float4> seg((size_t)S);
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        HIP_TRY(hipMemcpy(seg.data(), seg_dev, sizeof(float4) * S, hipMemcpyDeviceToHost));
        float lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY};
        for (const float4& g : seg) {
            const float mx = 0.5f * (g.x + g.z), my = 0.5f * (g.y + g.w);
            if (std::isfinite(mx) && std::isfinite(my)) {
                lo[0] = std::fmin(lo[0], mx); hi[0] = std::fmax(hi[0], mx);
                lo[1] = std::fmin(lo[1], my); hi[1] = std::fmax(hi[1], my);
            }
        }
        auto spread = [](uint32_t v) {   // 16 bits -> every other bit
            v &= 0xFFFFu;
            v = (v | (v << 8)) & 0x00FF00FFu; v = (v | (v << 4)) & 0x0F0F0F0Fu;
            v = (v | (v << 2)) & 0x33333333u; v = (v | (v << 1)) & 0x55555555u;
            return v;
        };
        std::vector<std::pair<uint32_t, int>> key((size_t)S);
        for (int j = 0; j < S; ++j) {
            const float mx = 0.5f * (seg[j].x + seg[j].z), my = 0.5f * (seg[j].y + seg[j].w);
            uint32_t code = 0xFFFFFFFFu;   // non-finite segments go last
            if (std::isfinite(mx) && std::isfinite(my)) {
                const float sx = (hi[0] > lo[0]) ? (mx - lo[0]) / (hi[0] - lo[0]) : 0.f;
                const float sy = (hi[1] > lo[1]) ? (my - lo[1]) / (hi[1] - lo[1]) : 0.f;
                code = spread((uint32_t)(sx * 65535.f)) | (spread((uint32_t)(sy * 65535.f)) << 1);
            }
            key[j] = {code, j};
        }
        std::sort(key.begin(), key.end());
        std::vector<float4> sorted((size_t)S), box((size_t)nt);
        for (int j = 0; j < S; ++j) sorted[j] = seg[key[j].second];
        for (int t = 0; t < nt; ++t) {
            // NaN coordinates make the box NaN-free but that segment can never be hit either (every test on it fails)
            float4 b = make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
            for (int j = 64 * t; j < std::min(S, 64 * t + 64); ++j) {
                const float4 g = sorted[j];
                b.x = std::fmin(b.x, std::fmin(g.x, g.z)); b.y = std::fmin(b.y, std::fmin(g.y, g.w));
                b.z = std::fmax(b.z, std::fmax(g.x, g.z)); b.w = std::fmax(b.w, std::fmax(g.y, g.w));
            }
            box[t] = b;
        }
        HIP_TRY(hipMalloc(&h->seg_sorted_dev, sizeof(float4) * S));
        HIP_TRY(hipMalloc(&h->tile_box_dev, sizeof(float4) * nt));
        HIP_TRY(hipMemcpy(h->seg_sorted_dev, sorted.data(), sizeof(float4) * S, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->tile_box_dev, box.data(), sizeof(float4) * nt, hipMemcpyHostToDevice));
        P.seg = h->seg_sorted_dev;
        P.tile_box = h->tile_box_dev;
    }
    if (n_segments <= 32) {   // several envs share one 64-lane pass of the cast
        int lg = 0;
        while ((1 << lg) < n_segments) ++lg;
        P.seg_pack_log2 = lg;
    }
    const int rc = rebuild_spawn_scans(h, (hipStream_t)stream);
    if (rc != NAVSIM_OK) return rc;
    h->has_map = true;
    return NAVSIM_OK;
}

int navsim_set_spawn_sampler(navsim_t* h, const double* starts_host, int32_t n_starts, const double* goals_host,
                             int32_t n_goals, double min_dist, double max_dist, void* stream) {
    if (!h || !starts_host || n_starts < 1 || n_goals < 0 || (n_goals > 0 && !goals_host) || !(max_dist >= min_dist))
        return fail(NAVSIM_E_ARG, "navsim_set_spawn_sampler: bad argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipStreamSynchronize(st));
    Params& P = h->P;
    (void)hipFree(h->goals_dev);
    h->goals_dev = nullptr;
    {
        const int rc = upload_starts(h, starts_host, n_starts, st);
        if (rc != NAVSIM_OK) return rc;
    }
    if (n_goals > 0) {
        HIP_TRY(hipMalloc(&h->goals_dev, sizeof(double) * 2 * n_goals));
        HIP_TRY(hipMemcpy(h->goals_dev, goals_host, sizeof(double) * 2 * n_goals, hipMemcpyHostToDevice));
    }
    P.goals = h->goals_dev;
    P.G = n_goals;
    P.min_dist = min_dist;
    P.max_dist = max_dist;
    {
        const int rc = invalidate_records(h, st);
        if (rc != NAVSIM_OK) return rc;
    }
    if (h->has_map) return rebuild_spawn_scans(h, st);
    return NAVSIM_OK;
}

int navsim_reset(navsim_t* h, const uint8_t* mask_dev, void* obs_dev, void* stream) {
    if (!h || !obs_dev) return fail(NAVSIM_E_ARG, "navsim_reset: bad argument");
    if (!h->has_map) return fail(NAVSIM_E_STATE, "navsim_reset: call navsim_set_map first");
    hipLaunchKernelGGL(reset_kernel, dim3((h->P.N + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->P, mask_dev,
                       obs_dev);
    HIP_TRY(hipGetLastError());
    return NAVSIM_OK;
}

int navsim_step(navsim_t* h, const float* action_dev, const float* past_action_dev, void* obs_dev,
                float* reward_dev, uint8_t* done_dev, uint8_t* arrive_dev, uint8_t* ended_dev,
                float* ep_return_dev, int32_t* ep_length_dev, float* ep_path_dev, void* stream) {
    if (!h || !action_dev || !obs_dev || !reward_dev || !done_dev || !arrive_dev)
        return fail(NAVSIM_E_ARG, "navsim_step: null required buffer");
    if (!h->has_map) return fail(NAVSIM_E_STATE, "navsim_step: call navsim_set_map first");
    hipStream_t st = (hipStream_t)stream;
    if (h->P.B == 10)
        launch_step<10>(h, action_dev, past_action_dev, obs_dev, reward_dev, done_dev, arrive_dev, ended_dev,
                        ep_return_dev, ep_length_dev, ep_path_dev, st);
    else
        launch_step<36>(h, action_dev, past_action_dev, obs_dev, reward_dev, done_dev, arrive_dev, ended_dev,
                        ep_return_dev, ep_length_dev, ep_path_dev, st);
    HIP_TRY(hipGetLastError());
    return NAVSIM_OK;
}

int navsim_rollout_mlp64(navsim_t* h, const float* actor_params_dev, void* obs_buf_dev, float* act_buf_dev,
                         float* logp_buf_dev, float* reward_dev, uint8_t* done_dev, uint8_t* arrive_dev, uint8_t* ended_dev,
                         float* ep_return_dev, int32_t* ep_length_dev, float* ep_path_dev, const float* var_dev,
                         uint64_t act_seed, const uint32_t* step_base_dev, int32_t n_steps, void* stream) {
    if (!h || !actor_params_dev || !obs_buf_dev || !act_buf_dev || !logp_buf_dev || !reward_dev || !done_dev || !arrive_dev ||
        !ended_dev || !var_dev || n_steps < 0)
        return fail(NAVSIM_E_ARG, "navsim_rollout_mlp64: bad argument");
    if (!h->has_map) return fail(NAVSIM_E_STATE, "navsim_rollout_mlp64: call navsim_set_map first");
    const RolloutPick rp = pick_rollout(h);
    if (rp.kind == 0) return fail(NAVSIM_E_ARG, "navsim_rollout_mlp64: the (B + 6)-64-64 policy needs 10 or 36 beams");
    if (((uintptr_t)actor_params_dev & 15) || ((uintptr_t)act_buf_dev & 7) || ((uintptr_t)obs_buf_dev & 15))
        return fail(NAVSIM_E_ARG, "navsim_rollout_mlp64: params and obs must be 16-byte, act 8-byte aligned");
    if (n_steps == 0) return NAVSIM_OK;
    RolloutArgs R;
    R.params = actor_params_dev; R.obs_buf = obs_buf_dev; R.act_buf = act_buf_dev; R.logp_buf = logp_buf_dev;
    R.reward = reward_dev; R.done = done_dev; R.arrive = arrive_dev; R.ended = ended_dev; R.ep_return = ep_return_dev;
    R.ep_length = ep_length_dev; R.ep_path = ep_path_dev; R.var_ptr = var_dev; R.step_base = step_base_dev;
    R.seed = act_seed; R.T = n_steps;
    const bool sens = h->P.sigma > 0.f || h->P.below_min_mode != 0;
    hipStream_t st = (hipStream_t)stream;
    if (rp.kind == 2) {   // 64 envs on 16 waves, the cast variants of launch_step (see pick_rollout)
        const size_t D = (size_t)h->P.B + 6;
        StepIO io;
        io.action = nullptr; io.past_override = nullptr;
        io.obs_out = h->P.obs_f16 ? (void*)(reinterpret_cast<__half*>(obs_buf_dev) + (size_t)h->P.N * D)
                                  : (void*)(reinterpret_cast<float*>(obs_buf_dev) + (size_t)h->P.N * D);
        io.reward = reward_dev; io.done = done_dev; io.arrive = arrive_dev; io.ended = ended_dev;
        io.ep_return = ep_return_dev; io.ep_length = ep_length_dev; io.ep_path_out = ep_path_dev;
        const dim3 grid((h->P.N + rp.epb - 1) / rp.epb), block(64 * rp.waves);
        const BigKArgs ka = {h->P, R, io};
#define NAVSIM_BIG(BOXES_, PAIR_)                                                                                         \
    do {                                                                                                                  \
        if (sens) hipLaunchKernelGGL((rollout_big_kernel<64, true, 16, BOXES_, PAIR_>), grid, block, 0, st, ka);   \
        else hipLaunchKernelGGL((rollout_big_kernel<64, false, 16, BOXES_, PAIR_>), grid, block, 0, st, ka);       \
    } while (0)
        if (rp.epb == 32) {   // (tile boxes; forced: any map without the 128-segment passes)
            if (rp.cast == 3) {
                if (sens) hipLaunchKernelGGL((rollout_big_kernel<32, true, 8, true, 0>), grid, block, 0, st, ka);
                else hipLaunchKernelGGL((rollout_big_kernel<32, false, 8, true, 0>), grid, block, 0, st, ka);
            } else {
                if (sens) hipLaunchKernelGGL((rollout_big_kernel<32, true, 8, false, 0>), grid, block, 0, st, ka);
                else hipLaunchKernelGGL((rollout_big_kernel<32, false, 8, false, 0>), grid, block, 0, st, ka);
            }
        } else if (rp.cast == 3) NAVSIM_BIG(true, 0);
        else if (rp.cast == 2) NAVSIM_BIG(false, 2);
        else if (rp.cast == 1) NAVSIM_BIG(false, 1);
        else NAVSIM_BIG(false, 0);
#undef NAVSIM_BIG
        HIP_TRY(hipGetLastError());
        return NAVSIM_OK;
    }
    // 8 waves per workgroup: more ray waves shorten the cast
    constexpr int kRollWaves = 8;
    const dim3 grid((h->P.N + rp.epb - 1) / rp.epb), block(64 * kRollWaves);
    if (h->P.B == 36) {
        if (rp.cast == 3) {
            if (sens) hipLaunchKernelGGL((rollout_kernel<36, 16, true, kRollWaves, true>), grid, block, 0, st, h->P, R);
            else hipLaunchKernelGGL((rollout_kernel<36, 16, false, kRollWaves, true>), grid, block, 0, st, h->P, R);
        } else {
            if (sens) hipLaunchKernelGGL((rollout_kernel<36, 16, true, kRollWaves>), grid, block, 0, st, h->P, R);
            else hipLaunchKernelGGL((rollout_kernel<36, 16, false, kRollWaves>), grid, block, 0, st, h->P, R);
        }
    } else if (rp.cast == 3) {
        if (sens) hipLaunchKernelGGL((rollout_kernel<10, 16, true, kRollWaves, true>), grid, block, 0, st, h->P, R);
        else hipLaunchKernelGGL((rollout_kernel<10, 16, false, kRollWaves, true>), grid, block, 0, st, h->P, R);
    } else if (rp.epb == 4) {
        if (sens) hipLaunchKernelGGL((rollout_kernel<10, 4, true, kRollWaves>), grid, block, 0, st, h->P, R);
        else hipLaunchKernelGGL((rollout_kernel<10, 4, false, kRollWaves>), grid, block, 0, st, h->P, R);
    } else if (rp.epb == 8) {
        if (sens) hipLaunchKernelGGL((rollout_kernel<10, 8, true, kRollWaves>), grid, block, 0, st, h->P, R);
        else hipLaunchKernelGGL((rollout_kernel<10, 8, false, kRollWaves>), grid, block, 0, st, h->P, R);
    } else {
        if (sens) hipLaunchKernelGGL((rollout_kernel<10, 16, true, kRollWaves>), grid, block, 0, st, h->P, R);
        else hipLaunchKernelGGL((rollout_kernel<10, 16, false, kRollWaves>), grid, block, 0, st, h->P, R);
    }
    HIP_TRY(hipGetLastError());
    return NAVSIM_OK;
}

int navsim_rollout_resmlp512(navsim_t* h, const float* actor_params_dev, void* obs_buf_dev, float* act_buf_dev, float* logp_buf_dev,
                             float* reward_dev, uint8_t* done_dev, uint8_t* arrive_dev, uint8_t* ended_dev, float* ep_return_dev,
                             int32_t* ep_length_dev, float* ep_path_dev, const float* var_dev, uint64_t act_seed,
                             const uint32_t* step_base_dev, int32_t n_steps, void* stream) {
    if (!h || !actor_params_dev || !obs_buf_dev || !act_buf_dev || !logp_buf_dev || !reward_dev || !done_dev || !arrive_dev ||
        !ended_dev || !var_dev || n_steps < 0)
        return fail(NAVSIM_E_ARG, "navsim_rollout_resmlp512: bad argument");
    if (!h->has_map) return fail(NAVSIM_E_STATE, "navsim_rollout_resmlp512: call navsim_set_map first");
    if (h->P.B != 10)
        return fail(NAVSIM_E_ARG, "navsim_rollout_resmlp512: the reference's nets read 16-wide observations (10 beams)");
    if (((uintptr_t)actor_params_dev & 15) || ((uintptr_t)act_buf_dev & 7) || ((uintptr_t)obs_buf_dev & 15))
        return fail(NAVSIM_E_ARG, "navsim_rollout_resmlp512: params and obs must be 16-byte, act 8-byte aligned");
    if (n_steps == 0) return NAVSIM_OK;
    RolloutArgs R;
    R.params = actor_params_dev; R.obs_buf = obs_buf_dev; R.act_buf = act_buf_dev; R.logp_buf = logp_buf_dev;
    R.reward = reward_dev; R.done = done_dev; R.arrive = arrive_dev; R.ended = ended_dev; R.ep_return = ep_return_dev;
    R.ep_length = ep_length_dev; R.ep_path = ep_path_dev; R.var_ptr = var_dev; R.step_base = step_base_dev;
    R.seed = act_seed; R.T = n_steps;
    const bool sens = h->P.sigma > 0.f || h->P.below_min_mode != 0;
    const dim3 grid((h->P.N + 15) / 16), block(64 * 8);
    if (sens) hipLaunchKernelGGL(rollout_resmlp_kernel<true>, grid, block, 0, (hipStream_t)stream, h->P, R);
    else hipLaunchKernelGGL(rollout_resmlp_kernel<false>, grid, block, 0, (hipStream_t)stream, h->P, R);
    HIP_TRY(hipGetLastError());
    return NAVSIM_OK;
}

int navsim_step_seq(navsim_t* h, const float* actions_dev, int32_t n_steps, void* obs_dev, float* reward_dev, uint8_t* done_dev,
                    uint8_t* arrive_dev, uint8_t* ended_dev, float* ep_return_dev, int32_t* ep_length_dev, float* ep_path_dev,
                    void* stream) {
    if (!h || !actions_dev || !obs_dev || !reward_dev || !done_dev || !arrive_dev || n_steps < 0)
        return fail(NAVSIM_E_ARG, "navsim_step_seq: bad argument");
    if (!h->has_map) return fail(NAVSIM_E_STATE, "navsim_step_seq: call navsim_set_map first");
    if ((uintptr_t)actions_dev & 7) return fail(NAVSIM_E_ARG, "navsim_step_seq: actions must be 8-byte aligned");
    if (n_steps == 0) return NAVSIM_OK;
    SeqArgs R;
    R.io = StepIO{reinterpret_cast<const float2*>(actions_dev), nullptr, obs_dev, reward_dev, done_dev, arrive_dev, ended_dev,
                  ep_return_dev, ep_length_dev, ep_path_dev};
    R.T = n_steps;
    if (h->P.B == 10) launch_steps<10>(h, R, (hipStream_t)stream);
    else launch_steps<36>(h, R, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return NAVSIM_OK;
}

int navsim_odometry(int32_t n, const double* x_dev, const double* y_dev, const double* quat_dev, const double* goal_dev,
                    double* out_dev, void* stream) {
    if (n < 0 || (n > 0 && (!x_dev || !y_dev || !quat_dev || !goal_dev || !out_dev)))
        return fail(NAVSIM_E_ARG, "navsim_odometry: bad argument");
    if (n == 0) return NAVSIM_OK;
    hipLaunchKernelGGL(odometry_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, x_dev, y_dev, quat_dev,
                       goal_dev, out_dev);
    HIP_TRY(hipGetLastError());
    return NAVSIM_OK;
}

int navsim_raycast(navsim_t* h, const double* pose_dev, float* ranges_dev, void* stream) {
    if (!h || !pose_dev || !ranges_dev) return fail(NAVSIM_E_ARG, "navsim_raycast: bad argument");
    if (!h->has_map) return fail(NAVSIM_E_STATE, "navsim_raycast: call navsim_set_map first");
    hipLaunchKernelGGL(raycast_kernel, dim3((h->P.N + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->P, pose_dev, 1, 0,
                       h->P.N, 0, ranges_dev);
    HIP_TRY(hipGetLastError());
    return NAVSIM_OK;
}

int navsim_get_state(navsim_t* h, double* pose, double* goal, double* past_dist, float* past_action,
                     int32_t* ep_step, uint32_t* rng_ctr, void* stream) {
    if (!h) return fail(NAVSIM_E_ARG, "navsim_get_state: null handle");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    const size_t N = (size_t)h->P.N;
    const Params& P = h->P;
    if (pose) {
        double* tmp = new double[3 * N];
        hipError_t e = hipMemcpy(tmp, P.x, sizeof(double) * 3 * N, hipMemcpyDeviceToHost);  // x,y,th are adjacent
        if (e == hipSuccess)
            for (size_t i = 0; i < N; ++i) {
                pose[3 * i] = tmp[i];
                pose[3 * i + 1] = tmp[N + i];
                pose[3 * i + 2] = tmp[2 * N + i];
            }
        delete[] tmp;
        HIP_TRY(e);
    }
    if (goal) {
        double* tmp = new double[2 * N];
        hipError_t e = hipMemcpy(tmp, P.gx, sizeof(double) * 2 * N, hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            for (size_t i = 0; i < N; ++i) {
                goal[2 * i] = tmp[i];
                goal[2 * i + 1] = tmp[N + i];
            }
        delete[] tmp;
        HIP_TRY(e);
    }
    if (past_dist) HIP_TRY(hipMemcpy(past_dist, P.past_dist, sizeof(double) * N, hipMemcpyDeviceToHost));
    if (past_action) HIP_TRY(hipMemcpy(past_action, P.past_action, sizeof(float2) * N, hipMemcpyDeviceToHost));
    if (ep_step) {
        HIP_TRY(hipMemcpy(ep_step, P.ep_step, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < N; ++i) ep_step[i] = (int32_t)((uint32_t)ep_step[i] & kStepMask);
    }
    if (rng_ctr) HIP_TRY(hipMemcpy(rng_ctr, P.rng_ctr, sizeof(uint32_t) * N, hipMemcpyDeviceToHost));
    return NAVSIM_OK;
}

int navsim_set_state(navsim_t* h, const double* pose, const double* goal, const double* past_dist,
                     const float* past_action, const int32_t* ep_step, const uint32_t* rng_ctr, void* stream) {
    if (!h) return fail(NAVSIM_E_ARG, "navsim_set_state: null handle");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    const size_t N = (size_t)h->P.N;
    const Params& P = h->P;
    if (pose) {
        double* tmp = new double[3 * N];
        for (size_t i = 0; i < N; ++i) {
            tmp[i] = pose[3 * i];
            tmp[N + i] = pose[3 * i + 1];
            tmp[2 * N + i] = pose[3 * i + 2];
        }
        hipError_t e = hipMemcpy(P.x, tmp, sizeof(double) * 3 * N, hipMemcpyHostToDevice);
        delete[] tmp;
        HIP_TRY(e);
    }
    if (goal) {
        double* tmp = new double[2 * N];
        for (size_t i = 0; i < N; ++i) {
            tmp[i] = goal[2 * i];
            tmp[N + i] = goal[2 * i + 1];
        }
        hipError_t e = hipMemcpy(P.gx, tmp, sizeof(double) * 2 * N, hipMemcpyHostToDevice);
        delete[] tmp;
        HIP_TRY(e);
    }
    if (past_dist) HIP_TRY(hipMemcpy(P.past_dist, past_dist, sizeof(double) * N, hipMemcpyHostToDevice));
    if (past_action) HIP_TRY(hipMemcpy(P.past_action, past_action, sizeof(float2) * N, hipMemcpyHostToDevice));
    if (ep_step) HIP_TRY(hipMemcpy(P.ep_step, ep_step, sizeof(int32_t) * N, hipMemcpyHostToDevice));
    if (rng_ctr) HIP_TRY(hipMemcpy(P.rng_ctr, rng_ctr, sizeof(uint32_t) * N, hipMemcpyHostToDevice));
    if (rng_ctr && !ep_step) {   // the goal stream moved: the cached next-episode records are stale (ep_step clears the flag itself)
        const int rc = invalidate_records(h, (hipStream_t)stream);
        if (rc != NAVSIM_OK) return rc;
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    }
    return NAVSIM_OK;
}

int navsim_rtg_scan(const float* rew_dev, const uint8_t* ended_dev, int32_t T, int32_t N, double gamma,
                    float* out_dev, int32_t exact, void* stream) {
    if (T < 0 || N < 0) return fail(NAVSIM_E_ARG, "navsim_rtg_scan: negative size");
    if (T == 0 || N == 0) return NAVSIM_OK;  // empty batch: nothing to scan (pointers may be null)
    if (!rew_dev || !ended_dev || !out_dev) return fail(NAVSIM_E_ARG, "navsim_rtg_scan: null buffer");
    if (N % kRtgCols == 0 && !exact)
        hipLaunchKernelGGL(rtg_kernel_split<false>, dim3(N / kRtgCols), dim3(256), 0, (hipStream_t)stream, rew_dev, ended_dev, T, N, gamma,
                           out_dev, (const float*)nullptr, (const float*)nullptr, 1.0, (float*)nullptr);
    else
        hipLaunchKernelGGL(rtg_kernel_generic, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, rew_dev, ended_dev, T, N,
                           gamma, out_dev);
    HIP_TRY(hipGetLastError());
    return NAVSIM_OK;
}

int navsim_gae_scan(const float* rew_dev, const uint8_t* ended_dev, const float* value_dev, const float* last_value_dev, int32_t T,
                    int32_t N, double gamma, double lam, float* adv_dev, float* ret_dev, int32_t exact, void* stream) {
    if (T < 0 || N < 0) return fail(NAVSIM_E_ARG, "navsim_gae_scan: negative size");
    if (T == 0 || N == 0) return NAVSIM_OK;
    if (!rew_dev || !ended_dev || !value_dev || !adv_dev) return fail(NAVSIM_E_ARG, "navsim_gae_scan: null buffer");
    if (!(lam >= 0.0 && lam <= 1.0)) return fail(NAVSIM_E_ARG, "navsim_gae_scan: lambda outside [0, 1]");
    if (N % kRtgCols == 0 && !exact)
        hipLaunchKernelGGL(rtg_kernel_split<true>, dim3(N / kRtgCols), dim3(256), 0, (hipStream_t)stream, rew_dev, ended_dev, T, N, gamma,
                           ret_dev, value_dev, last_value_dev, lam, adv_dev);
    else
        hipLaunchKernelGGL(gae_kernel_generic, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, rew_dev, ended_dev, value_dev,
                           last_value_dev, T, N, gamma, lam, adv_dev, ret_dev);
    HIP_TRY(hipGetLastError());
    return NAVSIM_OK;
}

}  // extern "C"
#pragma GCC visibility pop
