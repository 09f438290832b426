/*
 * navsim.h -- C ABI of libnavsim.so: the MI355X (gfx950) batched LiDAR-navigation
 * simulator and return scan that replace the reference's Gazebo/ROS-backed
 * `Env.reset()/Env.step()` and `PPO.compute_rtgs()`.
 *
 * The reference exposes NO plugin / FFI / operator ABI for this path: its boundary is
 * a duck-typed Python class.  Each entry point below therefore cites
 * the reference *Python* interface it replaces; navbot_ppo_amd/env.py binds these with
 * ctypes and re-creates that Python surface (same names, argument meaning, error
 * behaviour).  INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types; every function returns 0 on
 *     success or a negative NAVSIM_E_* code and never throws; navsim_last_error()
 *     gives the message for the calling thread.
 *   - `*_dev` pointers are DEVICE (HIP) pointers owned by the caller (e.g.
 *     torch.Tensor.data_ptr()); `*_host` pointers are host memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are
 *     asynchronous and stream-ordered unless stated otherwise; no internal threads; a
 *     handle must not be used from two threads at once; distinct handles are independent.
 *   - there is no CPU fallback: without a HIP device every call that touches the
 *     device fails with NAVSIM_E_HIP.
 */
#ifndef NAVSIM_H
#define NAVSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NAVSIM_ABI_VERSION 6

#define NAVSIM_OK 0
#define NAVSIM_E_ARG (-1)   /* bad argument / unsupported configuration */
#define NAVSIM_E_HIP (-2)   /* HIP runtime error (message has hipGetErrorString) */
#define NAVSIM_E_STATE (-3) /* call order: e.g. step before set_map */

typedef struct navsim navsim_t;

/*
 * Replaces the constructor arguments / module constants of the reference env:
 *   Env(is_training, ...)                    project_ppo/src/environment_new.py:27
 *   threshold_arrive 0.2 (train) / 0.4       environment_new.py:44-47
 *   max_timesteps_per_episode                project_ppo/src/ppo.py:552 (the caller-side timeout)
 *   spawn pose (0,0,0)                       turtlebot3_gazebo/launch/turtlebot3_stage_1.launch:3-5
 *   goal box U(-3.6,3.6)^2                   environment_new.py:337-338
 */
typedef struct navsim_cfg {
    int32_t n_envs;            /* N: envs simulated by this handle (one GPU's shard) */
    int32_t n_beams;           /* B: LiDAR samples (10 = reference, gazebo.xacro:111; 36 supported) */
    int32_t max_episode_steps; /* timeout in env steps; 0 = none */
    int32_t auto_reset;        /* 1: envs that end are reset inside step (ppo.py:582-593) and obs is the post-reset obs */
    int32_t respawn_on_arrive; /* 1: on arrival draw a new goal + re-base past_distance (environment_new.py:245-267) */
    int32_t obs_f16;           /* 1: obs buffers are IEEE half instead of float */
    int32_t lidar_below_min;   /* readings under range_min 0.12 m: 0 = clamp to 0.12 (default), 1 = -inf as Gazebo's ray sensor
                                  reports them; the reference passes -inf through unsanitised, which also suppresses the
                                  collision flag (environment_new.py:192-201) */
    float lidar_noise_sigma;   /* Gaussian range noise of the sensor (turtlebot3_burger.gazebo.xacro:122-126: 0.01); 0 = off
                                  (default, needed for bit-parity runs).  Applied to in-range readings, then clamped to
                                  [0.12, 3.5]; Philox keyed by (seed, env id, goal draws, episode step, beam) + Box-Muller */
    uint64_t seed;             /* Philox key */
    uint64_t env_id_base;      /* global id of env 0: RNG streams are keyed by (seed, env_id_base + i) */
    double threshold_arrive;
    double spawn_x, spawn_y, spawn_yaw;
    double goal_lo, goal_hi;
} navsim_cfg;

/* ABI / build info.  navsim_version() == NAVSIM_ABI_VERSION. */
int navsim_version(void);
const char* navsim_last_error(void);

/* Fills *cfg with the reference defaults (N=1, B=10, train threshold, stage_1 spawn and goal box). */
void navsim_default_cfg(navsim_cfg* cfg);

/* Env.__init__  (environment_new.py:27-47).  Allocates per-env state in HBM on the current device.  No entry point of the
 * library reads the process environment: everything a handle does follows from cfg, its map and navsim_set_shape, everything
 * a free function does from its arguments. */
int navsim_create(const navsim_cfg* cfg, navsim_t** out);
void navsim_destroy(navsim_t* h);

/*
 * Kernel-shape overrides of ONE handle (tests, A/B timing; no reference counterpart -- results never depend on them, every
 * shape writes the same rows).  envs_per_workgroup: 0 = the built-in rule (by shard size, beam count, map kind and entry
 * point), or 4 (navsim_rollout_mlp64 only) | 8 | 16 | 32 | 64.  pair_cast: -1 = the rule, 0 = 64-segment passes for every
 * map, 1 = 128-segment passes where the rule allows them.  Takes effect from the next launch.
 */
int navsim_set_shape(navsim_t* h, int32_t envs_per_workgroup, int32_t pair_cast);

/*
 * What the handle is and which kernel instantiation each entry point would launch right now (a profile can then name the
 * kernel it timed).  `*_epb` = envs per workgroup, `*_waves` = waves per workgroup, `*_cast`: 0 = 64-segment passes,
 * 1 = 128-segment passes, 2 = 128-segment passes with non-temporal loads, 3 = tile bounding boxes.
 * rollout_kind: 0 = navsim_rollout_mlp64 unavailable for this handle, 1 = rollout_kernel (rollout_epb = 4 | 8 | 16 envs on 8
 * waves), 2 = rollout_big_kernel (rollout_epb = 64 envs on 16 waves, or 32 envs on 8 waves: tile-box maps on 4097..8192 envs and
 * the forced 32-env shape).  forced_epb / forced_pair_cast: what navsim_set_shape was last given (0 / -1 = the rule).
 */
typedef struct navsim_info {
    int32_t abi_version, n_envs, n_beams, obs_f16;
    int32_t n_segments, per_env_map, tile_boxes, has_map;
    int32_t forced_epb, forced_pair_cast;
    int32_t step_epb, step_waves, step_cast;       /* navsim_step */
    int32_t seq_epb, seq_waves, seq_cast;          /* navsim_step_seq */
    int32_t rollout_kind, rollout_epb, rollout_waves, rollout_cast;   /* navsim_rollout_mlp64 */
    /* resources of the selected step / tape instantiations as the loaded code object reports them (hipFuncGetAttributes):
       vector registers per lane, scratch bytes per lane (= spilled registers x 4; 0 = none), static LDS bytes per workgroup */
    int32_t step_vgprs, step_scratch_bytes, step_lds_bytes;
    int32_t seq_vgprs, seq_scratch_bytes, seq_lds_bytes;
    int32_t reserved[6];
} navsim_info;
int navsim_get_info(navsim_t* h, navsim_info* out);

/*
 * The static world the reference gets from Gazebo (worlds/train_world_new.world:85-416):
 * S line segments (ax,ay,bx,by) float32.  per_env=0: seg_dev is [S,4], shared by all envs;
 * per_env=1: seg_dev is [N,S,4], env i reads its own S segments.  The buffer is NOT copied:
 * it must stay alive and unchanged while the handle uses it.  (Exception: a shared map of 65..4096 segments is copied,
 * synchronously, into a handle-owned buffer in Morton order of the segment midpoints together with the bounding box of
 * every 64-segment tile, which lets the step kernel skip whole tiles; the order of segments does not affect any result.)  Also ray-casts the spawn pose
 * (the scan every reset observes) on `stream`.
 */
int navsim_set_map(navsim_t* h, const float* seg_dev, int32_t n_segments, int32_t per_env, void* stream);

/*
 * Goal-rejection rectangles (xmin,xmax,ymin,ymax; inclusive), host pointer, copied.
 * which=0: used by reset (environment_new.py:340-343); which=1: used by the arrival
 * re-spawn (environment_new.py:248-251).  Defaults are the reference's stage_1 values.
 * n_rects <= 16.  Synchronous.
 * Bound: the reference redraws until a goal is accepted (environment_new.py:340-345 loops forever on rectangles that
 * cover the goal box); the kernels draw at most 64 times and then keep the last draw.  With the stock rectangles
 * (18 % / 25 % of the box rejected) the chance of exhausting 64 draws is < 1e-38; rectangles covering more than ~70 %
 * of [goal_lo, goal_hi]^2 make accepted-after-64 goals inside a rectangle plausible (0.7^64 = 1e-10 per reset) -- keep
 * the rejected share below that or enlarge the goal box.
 */
int navsim_set_goal_rects(navsim_t* h, int32_t which, const double* rects_host, int32_t n_rects);

/*
 * GoalSpawnSampler  (project_ppo/src/spawn_goal_sampler.py:37-62; parsed as --use_external_sampler at arguments.py:41 but
 * never wired into the reference's Env): curated start poses [n_starts,3] (x,y,yaw) and goal points [n_goals,2], HOST
 * pointers, copied.  With n_goals > 0 every reset (and in-step auto-reset) picks a start pose and a goal uniformly from the
 * tables until min_dist <= |start - goal| <= max_dist (at most 100 attempts, then one unconditional pick) instead of the
 * fixed spawn pose + uniform goal box; n_goals == 0 keeps the uniform goal rule but resets to starts[0].  Re-casts the
 * scans of all start poses on `stream`.  Synchronous on `stream`.
 */
int navsim_set_spawn_sampler(navsim_t* h, const double* starts_host, int32_t n_starts, const double* goals_host,
                             int32_t n_goals, double min_dist, double max_dist, void* stream);

/*
 * Env.reset()  (environment_new.py:312-382), masked and batched: for every env i with
 * mask_dev[i] != 0 (all envs if mask_dev is NULL): pose <- spawn, goal <- sampled with the
 * reset rejection rule, past_distance <- distance to goal, episode counters <- 0,
 * past_action <- (0,0), and row i of obs_dev ([N, B+6]) <- the reset observation.
 * Rows of unmasked envs are not written.
 */
int navsim_reset(navsim_t* h, const uint8_t* mask_dev, void* obs_dev, void* stream);

/*
 * Env.step(action, past_action)  (environment_new.py:272-310) for all N envs:
 * action scaling (:273-278), 0.2 s of diff-drive motion, B-beam ray-cast, getOdometry
 * (:138-181), getState (:183-207), observation assembly (:289-301), setReward (:209-270),
 * plus the caller-side episode logic of PPO.rollout (ppo.py:543-593): episode step count,
 * timeout, and -- if cfg.auto_reset -- the masked reset.
 *
 *   action_dev       [N,2] f32  in   (a0 in [0,1] -> v=a0/4 m/s ; a1 in [-1,1] -> w rad/s)
 *   past_action_dev  [N,2] f32  in   nullable: NULL = the action of the previous step, (0,0)
 *                                    after a reset (ppo.py:543,591); non-NULL = Env.step's argument
 *   obs_dev          [N,B+6]    out  f32 (f16 if cfg.obs_f16): B lidar/3.5, past_action, dist/diag,
 *                                    yaw/360, rel_theta/360, diff_angle/180
 *   reward_dev       [N] f32    out
 *   done_dev         [N] u8     out  collision flag (0 < min(scan) < 0.2)
 *   arrive_dev       [N] u8     out  distance <= threshold_arrive
 *   ended_dev        [N] u8     out  nullable: done | arrive | timeout (the RTG episode-end flag)
 *   ep_return_dev    [N] f32    out  nullable: written only where ended: sum of the episode's rewards
 *   ep_length_dev    [N] i32    out  nullable: written only where ended: episode length in steps
 *   ep_path_dev      [N] f32    out  nullable: written only where ended: the episode's path length as PPO.rollout
 *                                    accumulates it (ppo.py:533-537: distances between the positions read BEFORE each
 *                                    step, so the last step's displacement is not part of it)
 *
 * Precision of the pose (what "identical" means for the float64 state): the motion model of turtlebot3_fake.cpp:154-163
 * integrates six 30 Hz sub-steps with one library sincos each.  The kernel evaluates the library sincos of the FINAL heading
 * (which also gives the sensor origin and the beam directions, i.e. everything the scan's bits depend on) and obtains the six
 * sub-step headings from it by angle addition, sums the sub-step displacements last-first and takes the step's path length
 * from the summed displacement.  x, y and the path length therefore agree with the serial integration to ~1e-16 per step
 * (relative 1e-15), not bit for bit; the heading itself (six additions of delta_theta) is bit-identical.  The tests hold the pose
 * to 1e-11 over hundreds of steps; observations and flags are compared separately (flags exactly, observations 1e-6, > 99 %
 * of the rows bit-identical -- a 1e-16 difference moves a float32 rounding or a decimal round() tie with probability ~1e-9).
 */
int navsim_step(navsim_t* h, const float* action_dev, const float* past_action_dev, void* obs_dev,
                float* reward_dev, uint8_t* done_dev, uint8_t* arrive_dev, uint8_t* ended_dev,
                float* ep_return_dev, int32_t* ep_length_dev, float* ep_path_dev, void* stream);

/*
 * Env attributes the reference's callers read or that tests / checkpoints need
 * (position.x/.y: ppo.py:535; goal_position, past_distance: environment_new.py:29-41).
 * HOST pointers, any of them may be NULL; SYNCHRONOUS (waits for `stream`).
 *   pose [N,3] f64 (x,y,theta) ; goal [N,2] f64 ; past_dist [N] f64 ; past_action [N,2] f32 ;
 *   ep_step [N] i32 ; rng_ctr [N] u32 (Philox draws consumed by env i)
 */
int navsim_get_state(navsim_t* h, double* pose_host, double* goal_host, double* past_dist_host,
                     float* past_action_host, int32_t* ep_step_host, uint32_t* rng_ctr_host, void* stream);
int navsim_set_state(navsim_t* h, const double* pose_host, const double* goal_host,
                     const double* past_dist_host, const float* past_action_host,
                     const int32_t* ep_step_host, const uint32_t* rng_ctr_host, void* stream);

/*
 * PPO.compute_rtgs()  (ppo.py:643-671) on the vectorised layout: rew_dev [T,N] f32,
 * ended_dev [T,N] u8 (last step of an episode), out_dev [T,N] f32:
 *   R[t,n] = rew[t,n] + gamma * (ended[t,n] ? 0 : R[t+1,n]),  R[T,n] = 0 (no bootstrap, ppo.py:601)
 * accumulated in float64 and stored as float32 like the reference (:665,:669).
 * N % 16 == 0 runs the scan split over T (chunk-local scans composed through float64 carries): every stored float32 is
 * within ONE ulp of the reference's serial recurrence (identical behind an episode end inside its 32-row chunk; elsewhere
 * a store differs with probability ~2e-8).  Any other N, or exact != 0, runs the serial recurrence: bit-identical, about 4x
 * slower.  (ABI v6: `exact` is an argument; up to v5 it was NAVSIM_RTG_EXACT in the process environment -- the library reads
 * nothing from the environment any more.)
 */
int navsim_rtg_scan(const float* rew_dev, const uint8_t* ended_dev, int32_t T, int32_t N, double gamma,
                    float* out_dev, int32_t exact, void* stream);

/*
 * Generalised advantage estimation -- the "GAE / return scan" BASELINE.json's north_star names; an EXTENSION: the reference
 * has only compute_rtgs (ppo.py:643-671) and A = rtgs - V (ppo.py:277), which is this scan at lambda = 1.  Off by default
 * (PPOConfig.gae_lambda = None).
 *   value_dev [T,N] f32 = V(s_t) of the stored observations; last_value_dev [N] f32, nullable: V of the state after the last
 *   row (NULL = the batch end is terminal, as in the reference, ppo.py:601,658); episode ends are always terminal (:552-553).
 *     R[t] = rew[t] + (ended[t] ? 0 : gamma (1 - lam) V[t+1] + gamma lam R[t+1])          float64 accumulate
 *     ret_dev [T,N] (nullable) = float32(R[t]) -- the lambda-return, the critic's target;   adv_dev [T,N] = float32(R[t]) - V[t]
 * lam = 1 and last_value_dev = NULL: ret_dev is bit-identical to navsim_rtg_scan's output and adv_dev to rtgs - V.  Same
 * kernel, same <= 1 ulp contract and the same `exact` / N % 16 rule as navsim_rtg_scan.
 */
int navsim_gae_scan(const float* rew_dev, const uint8_t* ended_dev, const float* value_dev, const float* last_value_dev, int32_t T,
                    int32_t N, double gamma, double lam, float* adv_dev, float* ret_dev, int32_t exact, void* stream);

/*
 * The hot loop of PPO.rollout (project_ppo/src/ppo.py:505-594) for the (B + 6)-64-64 policy, all n_steps steps in ONE launch:
 * per step PPO.get_action (ppo.py:673-706; what navppo_mlp64_act computes, include/navppo.h) followed by what navsim_step
 * computes, for every env, with the rows of step t written at offset t * N of each [n_steps, N, .] buffer.  A workgroup
 * keeps its envs for the whole rollout, so there is no kernel boundary and no observation round trip between steps; the
 * results are bit-identical to n_steps pairs of navppo_mlp64_act / navsim_step calls with the same seeds.
 *   actor_params_dev [NAVPPO_MLP64_ACTOR_PARAMS_D(B + 6)] f32  the actor in the layout of navppo.h (10 beams: 5378, 36: 7042)
 *   obs_buf_dev  [n_steps + 1, N, B + 6] f32 (f16 if cfg.obs_f16)   row 0 in: the observations the rollout starts from
 *                (navsim_reset); rows 1.. out.  With f16 buffers the policy reads every row as a reader of the buffer would:
 *                rounded to half (so the rows still equal the per-step path's bit for bit).
 *   act_buf_dev  [n_steps, N, 2]   logp_buf_dev [n_steps, N]   reward_dev [n_steps, N]   done / arrive / ended [n_steps, N] u8
 *   ep_return_dev / ep_length_dev / ep_path_dev  [n_steps, N], nullable, written where ended (as in navsim_step)
 *   var_dev  device scalar: exploration variance (ppo.py:123-124)
 *   act_seed, step_base_dev (device scalar, nullable = 0): action noise = Philox(act_seed, env id, *step_base_dev + t)
 * Workgroup shapes, same rows bit for bit: 16 envs on 8 waves (a latency chain per workgroup, one round of workgroups up to
 * 4096 envs; 64-segment passes, or tile boxes on shared 65..4096-segment maps; the only shape with 36 beams) and, with 10 beams
 * beyond 4096 envs per GPU, 64 envs on 16 waves with the cast variants of navsim_step (tile boxes, 128-segment passes of per-env
 * maps) -- 32 envs on 8 waves for tile-box maps on 4097..8192 envs: the closed-loop form of navsim_step_seq (navsim_set_shape:
 * 4 | 8 | 16 | 32 | 64 forces a shape; navsim_get_info reports the one a launch would take).  params_dev and obs_buf_dev must be
 * 16-byte aligned, act_buf_dev 8-byte (NAVSIM_E_ARG otherwise).
 */
int navsim_rollout_mlp64(navsim_t* h, const float* actor_params_dev, void* obs_buf_dev, float* act_buf_dev,
                         float* logp_buf_dev, float* reward_dev, uint8_t* done_dev, uint8_t* arrive_dev, uint8_t* ended_dev,
                         float* ep_return_dev, int32_t* ep_length_dev, float* ep_path_dev, const float* var_dev,
                         uint64_t act_seed, const uint32_t* step_base_dev, int32_t n_steps, void* stream);

/*
 * The same hot loop (ppo.py:505-594) with the reference's ACTIVE actor choosing every action in the kernel: NetActor
 * (project_ppo/src/net_actor.py:56-144 -- two residual blocks of 512 hidden units, LeakyReLU(0.2), heads sigmoid / tanh), i.e. per
 * step what navppo_resmlp512_act computes (include/navppo.h) followed by what navsim_step computes, all n_steps steps in ONE launch;
 * buffers, noise keys and bit-for-bit equality with the per-step entry points as for navsim_rollout_mlp64.
 *   actor_params_dev [NAVPPO_RESMLP512_ACTOR_PARAMS = 50290] f32, the layout of navppo.h (no BatchNorm entries)
 *   obs_buf_dev [n_steps + 1, N, 16] f32 (f16 if cfg.obs_f16: the policy then reads every row rounded to half, as a reader of the
 *               buffer would -- the rows still equal the per-step path's bit for bit).   Needs n_beams == 10.
 * 16 envs on 8 waves per workgroup at every shard size (each wave owns 64 hidden units; the 197 KB of weights stream from L2 every
 * step), 64-segment passes for every map (no tile boxes: correct on every map, the house map just tests all its tiles).
 */
int navsim_rollout_resmlp512(navsim_t* h, const float* actor_params_dev, void* obs_buf_dev, float* act_buf_dev, float* logp_buf_dev,
                             float* reward_dev, uint8_t* done_dev, uint8_t* arrive_dev, uint8_t* ended_dev, float* ep_return_dev,
                             int32_t* ep_length_dev, float* ep_path_dev, const float* var_dev, uint64_t act_seed,
                             const uint32_t* step_base_dev, int32_t n_steps, void* stream);

/*
 * n_steps calls of navsim_step with the actions of a tape, in ONE launch: the step loop of PPO.rollout (ppo.py:505-594) or of the
 * evaluation loop (main.py:176-235) when the actions do not depend on the observations being produced -- a recorded tape, a
 * scripted or random policy, an open-loop controller.  A workgroup keeps its envs for the whole tape (their state stays on chip
 * between the steps), so the launch ramp, the state round trips and the kernel boundaries of n_steps launches are paid once.
 *   actions_dev [n_steps, N, 2] f32 (8-byte aligned); row t is the `action_dev` of step t (past_action = the action executed before)
 *   obs_dev     [n_steps, N, B+6] f32 (f16 if cfg.obs_f16): the observation AFTER step t, as navsim_step writes it
 *   reward_dev / done_dev / arrive_dev / ended_dev (nullable)  [n_steps, N]
 *   ep_return_dev / ep_length_dev / ep_path_dev  [n_steps, N], nullable, written where ended (as in navsim_step)
 * Every row is bit-identical to what n_steps navsim_step calls write; the handle's state afterwards is the same too.
 */
int navsim_step_seq(navsim_t* h, const float* actions_dev, int32_t n_steps, void* obs_dev, float* reward_dev, uint8_t* done_dev,
                    uint8_t* arrive_dev, uint8_t* ended_dev, float* ep_return_dev, int32_t* ep_length_dev, float* ep_path_dev,
                    void* stream);

/*
 * Env.getOdometry()  (environment_new.py:138-181) on its own, for n independent samples -- the odometry callback's arithmetic
 * exactly as the reference runs it, general quaternion included (the step kernel applies the same device functions to the
 * yaw-only quaternion of its planar pose):
 *   yaw        = round(degrees(atan2(2 (qx qy + qw qz), 1 - 2 (qy^2 + qz^2)))), + 360 if negative        (:142-147)
 *   rel_theta  = round(degrees(8-case quadrant angle of round(goal - pos, 1)), 2)                           (:149-169)
 *   diff_angle = round(wrap(yaw - rel_theta) to [-180, 180], 2)                                              (:170-176)
 * x_dev, y_dev [n] f64; quat_dev [n,4] f64 (qx, qy, qz, qw); goal_dev [n,2] f64; out_dev [n,3] f64 (yaw, rel_theta,
 * diff_angle).  Python round() semantics (half to even on the decimal value) are reproduced exactly.
 */
int navsim_odometry(int32_t n, const double* x_dev, const double* y_dev, const double* quat_dev, const double* goal_dev,
                    double* out_dev, void* stream);

/* LiDAR only (no state change): ranges_dev [N,B] f32 raw scan (inf = no return) for poses
 * pose_dev [N,3] f64.  Used by tests and by map tooling (spawn_goal_sampler-style validation). */
int navsim_raycast(navsim_t* h, const double* pose_dev, float* ranges_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NAVSIM_H */
