/*
 * navsim_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call this file.  The product (navbot_ppo_amd/, libnavsim.so)
 * never links or imports it and has no CPU fallback.
 *
 * Two layers:
 *
 *  (A) Restatement of the reference's arithmetic rules, each function citing
 *      the reference file:line it follows.  PINNED: checked against golden
 *      vectors recorded from the reference itself (tests/golden/ JSON files, made by
 *      tests/golden/generate_golden.py which imports /root/reference with
 *      stubbed ROS modules).
 *        orc_py_round_nd      Python round(x, n)  (used at environment_new.py:149-176)
 *        orc_get_odometry     environment_new.py:138-181
 *        orc_get_state        environment_new.py:183-207
 *        orc_assemble_obs     environment_new.py:288-301 / 361-374
 *        orc_set_reward       environment_new.py:209-270 (arithmetic part)
 *        orc_goal_rejected    environment_new.py:248-251 / 340-343
 *        orc_compute_rtgs_*   ppo.py:643-671
 *
 *  (B) The simulator the reference delegates to Gazebo (diff-drive motion and
 *      the LiDAR ray-cast).  The plugins are not in the reference tree, so this
 *      part is AUTHORED by the build from the in-tree specifications:
 *        motion   turtlebot3_simulations/turtlebot3_fake/src/turtlebot3_fake.cpp:110-119,124-180
 *                 (+ turtlebot3_fake.h:39, gazebo.xacro:62-66)
 *        LiDAR    turtlebot3/turtlebot3_description/urdf/turtlebot3_burger.gazebo.xacro:104-127
 *                 mount: turtlebot3_burger.urdf.xacro:134-138
 *      PARITY UNPINNED for (B): the reference holds no test or golden vector
 *      for motion or ray-cast; they are pinned only by the analytic known-answer
 *      tests in tests/test_oracle_sim.py.
 *
 * Build: gcc -O2 -ffp-contract=off -mfma -shared -fPIC (see oracle/Makefile).
 * All fused multiply-adds are written explicitly (fma/fmaf); everything else
 * is plain IEEE double/float so the HIP kernels can reproduce it bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* constants                                   */
/* ------------------------------------------------------------------ */
#define WHEEL_RADIUS 0.033   /* turtlebot3_fake.h:39 */
#define WHEEL_SEP 0.160      /* turtlebot3_fake.cpp:44, gazebo.xacro:65 */
#define SUBSTEPS 6           /* 30 Hz drive updates (gazebo.xacro:62) per 5 Hz scan (gazebo.xacro:107) */
#define LIDAR_X (-0.032)     /* urdf.xacro:137 */
#define ANGLE_MIN (-1.5707975) /* gazebo.xacro:113 */
#define ANGLE_MAX (1.5707975)  /* gazebo.xacro:114 */
#define RANGE_MIN 0.12f      /* gazebo.xacro:118 */
#define RANGE_MAX 3.5f       /* gazebo.xacro:119 */
#define MAX_GOAL_TRIES 64

/* ================================================================== */
/* (A) reference rules                                                */
/* ================================================================== */

/* Python round(x, nd) for nd in {1,2}: correctly rounded decimal rounding,
 * ties-to-even ON THE EXACT BINARY VALUE (CPython float.__round__ goes through
 * dtoa, not x*10**n).  Exactness comes from the fma residual of x*10^nd. */
ORC_API double orc_py_round_nd(double x, int nd) {
    const double s = (nd == 1) ? 10.0 : (nd == 2 ? 100.0 : 1.0);
    if (!isfinite(x)) return x;
    double p = x * s;
    double err = fma(x, s, -p); /* exact: x*s == p + err */
    double fl = floor(p);
    double frac = p - fl; /* exact for |p| < 2^52 */
    double q;
    if (frac < 0.5)
        q = fl;
    else if (frac > 0.5)
        q = fl + 1.0;
    else { /* p is exactly k + 0.5: the residual decides, then half-even */
        if (err > 0.0)
            q = fl + 1.0;
        else if (err < 0.0)
            q = fl;
        else
            q = (fmod(fl, 2.0) == 0.0) ? fl : fl + 1.0;
    }
    /* frac==0 with negative residual can never cross a .5 boundary: |err| <= ulp(p)/2 */
    double r = q / s; /* correctly rounded quotient == strtod of the decimal string */
    if (r == 0.0) r = copysign(0.0, x); /* round(-0.04, 1) == -0.0 */
    return r;
}

/* environment_new.py:138-181 */
ORC_API void orc_get_odometry(double px, double py, double qx, double qy, double qz, double qw,
                              double gx, double gy, int32_t* yaw_out, double* rel_theta_out,
                              double* diff_angle_out) {
    const double rad2deg = 180.0 / M_PI; /* math.degrees */
    /* :142  round() with no digits = half-to-even = rint */
    double yaw = rint(atan2(2 * (qx * qy + qw * qz), 1 - 2 * (qy * qy + qz * qz)) * rad2deg);
    if (!(yaw >= 0)) yaw = yaw + 360; /* :144-147 */
    double dx = orc_py_round_nd(gx - px, 1); /* :149 */
    double dy = orc_py_round_nd(gy - py, 1); /* :150 */
    double theta;
    if (dx > 0 && dy > 0) /* :153-168 */
        theta = atan(dy / dx);
    else if (dx > 0 && dy < 0)
        theta = 2 * M_PI + atan(dy / dx);
    else if (dx < 0 && dy < 0)
        theta = M_PI + atan(dy / dx);
    else if (dx < 0 && dy > 0)
        theta = M_PI + atan(dy / dx);
    else if (dx == 0 && dy > 0)
        theta = 1.0 / 2 * M_PI;
    else if (dx == 0 && dy < 0)
        theta = 3.0 / 2 * M_PI;
    else if (dy == 0 && dx > 0)
        theta = 0;
    else
        theta = M_PI;
    double rel_theta = orc_py_round_nd(theta * rad2deg, 2); /* :169 */
    double diff = yaw - rel_theta;                          /* :170 */
    if ((0 <= diff && diff <= 180) || (-180 <= diff && diff < 0)) /* :171-176 */
        diff = orc_py_round_nd(diff, 2);
    else if (diff < -180)
        diff = orc_py_round_nd(360 + diff, 2);
    else
        diff = orc_py_round_nd(-360 + diff, 2);
    *yaw_out = (int32_t)yaw;
    *rel_theta_out = rel_theta;
    *diff_angle_out = diff;
}

/* environment_new.py:183-207.  ranges are the LaserScan float32 values widened to double. */
ORC_API void orc_get_state(const double* ranges, int L, double px, double py, double gx, double gy,
                           double threshold_arrive, double* scan_out, double* dist_out,
                           int32_t* done_out, int32_t* arrive_out) {
    double mn = INFINITY;
    for (int i = 0; i < L; ++i) {
        double r = ranges[i];
        if (r == INFINITY) /* :193 (+inf only; -inf passes through) */
            r = 3.5;
        else if (isnan(r)) /* :195 */
            r = 0;
        scan_out[i] = r;
        /* Python min(): first minimal element, NaN cannot occur after sanitising */
        if (r < mn) mn = r;
    }
    *done_out = (0.2 > mn && mn > 0) ? 1 : 0; /* :200 */
    double dist = hypot(gx - px, gy - py);    /* :203 */
    *dist_out = dist;
    *arrive_out = (dist <= threshold_arrive) ? 1 : 0; /* :204 */
}

/* environment_new.py:289-301 (step) and :362-374 (reset, past_action = 0,0).
 * n_feat lidar features are picked at indices int(i*L/n_feat); the reference
 * fixes n_feat = 10 (:293).  obs has n_feat + 6 entries. */
ORC_API void orc_assemble_obs(const double* scan, int L, int n_feat, const double* past_action,
                              double dist, double yaw, double rel_theta, double diff_angle,
                              double* obs) {
    const double diagonal_dis = sqrt(2.0) * (3.8 + 3.8); /* :21 */
    for (int i = 0; i < n_feat; ++i) {
        int idx = (int)((double)i * (double)L / (double)n_feat); /* int(i * L / 10), true division */
        obs[i] = scan[idx] / 3.5;                                 /* :289 */
    }
    obs[n_feat + 0] = past_action[0]; /* :299-300 */
    obs[n_feat + 1] = past_action[1];
    obs[n_feat + 2] = dist / diagonal_dis; /* :301 */
    obs[n_feat + 3] = yaw / 360;
    obs[n_feat + 4] = rel_theta / 360;
    obs[n_feat + 5] = diff_angle / 180;
}

/* environment_new.py:209-222: returns reward, updates *past_distance (:214).
 * The goal re-spawn of :245-267 is done by the caller (needs the RNG). */
ORC_API double orc_set_reward(double* past_distance, double current_distance, int done, int arrive) {
    double distance_rate = (*past_distance - current_distance); /* :211 */
    double reward = 500. * distance_rate;                       /* :213 */
    *past_distance = current_distance;                          /* :214 */
    if (done) reward = -100.;                                   /* :216-217 */
    if (arrive) reward = 120.;                                  /* :220-221 */
    return reward;
}

/* Goal rejection rectangles.  which=0: reset (:340-343), which=1: respawn after arrival (:248-251).
 * rects: [R][4] = xmin,xmax,ymin,ymax, inclusive on all sides as in the reference. */
static const double kResetRects[4][4] = {{1.7, 2.3, -1.2, 1.2},
                                         {-2.3, -1.7, -1.2, 1.2},
                                         {-1.2, 1.2, 1.7, 2.3},
                                         {-1.2, 1.2, -2.3, -1.7}};
static const double kRespawnRects[4][4] = {{1.6, 2.4, -1.4, 1.4},
                                           {-2.4, -1.6, -1.4, 1.4},
                                           {-1.4, 1.4, 1.6, 2.4},
                                           {-1.4, 1.4, -2.4, -1.6}};

static int rejected(const double* rects, int R, double x, double y) {
    for (int r = 0; r < R; ++r) {
        const double* q = rects + 4 * r;
        if (q[0] <= x && x <= q[1] && q[2] <= y && y <= q[3]) return 1;
    }
    return 0;
}

ORC_API int orc_goal_rejected(int which, double x, double y) {
    return rejected(which ? &kRespawnRects[0][0] : &kResetRects[0][0], 4, x, y);
}

/* ppo.py:643-671 on the reference's ragged list-of-episodes layout.
 * rews: concatenated episode rewards (python floats), lens[n_eps]; out: float32[sum lens]. */
ORC_API void orc_compute_rtgs_ragged(const double* rews, const int32_t* lens, int n_eps,
                                     double gamma, float* out) {
    int64_t off = 0;
    for (int e = 0; e < n_eps; ++e) {
        double disc = 0; /* :660 */
        for (int t = lens[e] - 1; t >= 0; --t) {
            disc = rews[off + t] + disc * gamma; /* :665 */
            out[off + t] = (float)disc;          /* :669 torch.tensor(..., dtype=torch.float) */
        }
        off += lens[e];
    }
}

/* Same recurrence on the vectorised [T,N] layout: column n is env n, ended[t,n]!=0 marks the
 * last step of an episode (collision | arrival | timeout, ppo.py:552-553); the batch end is an
 * episode end too (:601, no bootstrap). */
ORC_API void orc_compute_rtgs_tn(const float* rew, const uint8_t* ended, int T, int N,
                                 double gamma, float* out) {
    for (int n = 0; n < N; ++n) {
        double disc = 0;
        for (int t = T - 1; t >= 0; --t) {
            if (ended[(int64_t)t * N + n]) disc = 0;
            disc = (double)rew[(int64_t)t * N + n] + disc * gamma;
            out[(int64_t)t * N + n] = (float)disc;
        }
    }
}

/* ================================================================== */
/* (B) authored simulator                                             */
/* ================================================================== */

/* Philox4x32-10 (Salmon et al. 2011), counter-based: one call = 4 x u32. */
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

ORC_API void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    memcpy(out, c, sizeof c);
}

typedef struct orc_cfg {
    int32_t n_envs;
    int32_t n_beams;
    int32_t max_episode_steps; /* 0 = no timeout (ppo.py:552) */
    int32_t auto_reset;        /* masked in-step reset, ppo.py:591-593 */
    int32_t respawn_on_arrive; /* environment_new.py:245-267 */
    int32_t obs_f16;           /* unused by the oracle (obs are float32); keeps the layout of navsim_cfg */
    int32_t lidar_below_min;   /* 0 clamp to range_min, 1 -inf (Gazebo) */
    float lidar_noise_sigma;   /* 0 = off */
    uint64_t seed;
    uint64_t env_id_base; /* global id of env 0 (multi-GPU shards) */
    double threshold_arrive; /* environment_new.py:44-47 */
    double spawn_x, spawn_y, spawn_yaw; /* turtlebot3_stage_1.launch:3-5 */
    double goal_lo, goal_hi;            /* environment_new.py:337-338 */
} orc_cfg;

typedef struct orc_sim {
    orc_cfg cfg;
    int S, per_env;
    float* seg; /* [S][4] or [N][S][4]: ax,ay,bx,by */
    double *x, *y, *th, *gx, *gy, *past_dist;
    float* past_action; /* [N][2] */
    int32_t* ep_step;
    uint32_t* rng_ctr;
    double *ep_ret; /* running episode return */
    double *ep_path; /* running episode path length, ppo.py:533-537 */
    double* beam_cos; /* cos/sin of the beam angle relative to heading */
    double* beam_sin;
    double reset_rects[16][4];
    double respawn_rects[16][4];
    int n_reset_rects, n_respawn_rects;
    /* GoalSpawnSampler tables (spawn_goal_sampler.py:37-62); K >= 1 start poses, G == 0: uniform goal rule */
    double* starts; /* [K][3] */
    double* goals;  /* [G][2] */
    int K, G;
    double min_dist, max_dist;
} orc_sim;

ORC_API orc_sim* orc_sim_create(const orc_cfg* cfg) {
    orc_sim* s = (orc_sim*)calloc(1, sizeof(orc_sim));
    s->cfg = *cfg;
    int N = cfg->n_envs, B = cfg->n_beams;
    s->x = calloc(N, sizeof(double));
    s->y = calloc(N, sizeof(double));
    s->th = calloc(N, sizeof(double));
    s->gx = calloc(N, sizeof(double));
    s->gy = calloc(N, sizeof(double));
    s->past_dist = calloc(N, sizeof(double));
    s->ep_ret = calloc(N, sizeof(double));
    s->ep_path = calloc(N, sizeof(double));
    s->past_action = calloc(2 * (size_t)N, sizeof(float));
    s->ep_step = calloc(N, sizeof(int32_t));
    s->rng_ctr = calloc(N, sizeof(uint32_t));
    s->beam_cos = calloc(B, sizeof(double));
    s->beam_sin = calloc(B, sizeof(double));
    for (int i = 0; i < B; ++i) {
        /* gazebo.xacro:110-115: samples evenly spread over [min_angle, max_angle] */
        double phi = (B > 1) ? ANGLE_MIN + (double)i * ((ANGLE_MAX - ANGLE_MIN) / (double)(B - 1)) : 0.0;
        s->beam_cos[i] = cos(phi);
        s->beam_sin[i] = sin(phi);
    }
    memcpy(s->reset_rects, kResetRects, sizeof kResetRects);
    memcpy(s->respawn_rects, kRespawnRects, sizeof kRespawnRects);
    s->n_reset_rects = 4;
    s->n_respawn_rects = 4;
    for (int i = 0; i < N; ++i) {
        s->x[i] = cfg->spawn_x;
        s->y[i] = cfg->spawn_y;
        s->th[i] = cfg->spawn_yaw;
    }
    s->starts = calloc(3, sizeof(double));
    s->starts[0] = cfg->spawn_x; s->starts[1] = cfg->spawn_y; s->starts[2] = cfg->spawn_yaw;
    s->K = 1;
    s->G = 0;
    return s;
}

ORC_API int orc_sim_set_spawn_sampler(orc_sim* s, const double* starts, int K, const double* goals, int G,
                                      double min_dist, double max_dist) {
    if (K < 1 || G < 0) return -1;
    free(s->starts); free(s->goals);
    s->starts = malloc(sizeof(double) * 3 * K);
    memcpy(s->starts, starts, sizeof(double) * 3 * K);
    s->goals = G ? malloc(sizeof(double) * 2 * G) : NULL;
    if (G) memcpy(s->goals, goals, sizeof(double) * 2 * G);
    s->K = K; s->G = G; s->min_dist = min_dist; s->max_dist = max_dist;
    return 0;
}

ORC_API void orc_sim_destroy(orc_sim* s) {
    if (!s) return;
    free(s->seg); free(s->x); free(s->y); free(s->th); free(s->gx); free(s->gy);
    free(s->past_dist); free(s->ep_ret); free(s->ep_path); free(s->past_action); free(s->ep_step);
    free(s->rng_ctr); free(s->beam_cos); free(s->beam_sin); free(s->starts); free(s->goals);
    free(s);
}

ORC_API int orc_sim_set_map(orc_sim* s, const float* seg, int S, int per_env) {
    size_t n = (size_t)S * 4 * (per_env ? (size_t)s->cfg.n_envs : 1);
    free(s->seg);
    s->seg = malloc(n * sizeof(float));
    memcpy(s->seg, seg, n * sizeof(float));
    s->S = S;
    s->per_env = per_env;
    return 0;
}

ORC_API int orc_sim_set_goal_rects(orc_sim* s, int which, const double* rects, int R) {
    if (R > 16) return -1;
    if (which) { memcpy(s->respawn_rects, rects, sizeof(double) * 4 * R); s->n_respawn_rects = R; }
    else { memcpy(s->reset_rects, rects, sizeof(double) * 4 * R); s->n_reset_rects = R; }
    return 0;
}

/* host-pointer state access (same field order as navsim_get_state / navsim_set_state) */
ORC_API void orc_sim_get_state(const orc_sim* s, double* pose, double* goal, double* past_dist,
                               float* past_action, int32_t* ep_step, uint32_t* rng_ctr) {
    int N = s->cfg.n_envs;
    for (int i = 0; i < N; ++i) {
        if (pose) { pose[3 * i] = s->x[i]; pose[3 * i + 1] = s->y[i]; pose[3 * i + 2] = s->th[i]; }
        if (goal) { goal[2 * i] = s->gx[i]; goal[2 * i + 1] = s->gy[i]; }
        if (past_dist) past_dist[i] = s->past_dist[i];
        if (past_action) { past_action[2 * i] = s->past_action[2 * i]; past_action[2 * i + 1] = s->past_action[2 * i + 1]; }
        if (ep_step) ep_step[i] = s->ep_step[i];
        if (rng_ctr) rng_ctr[i] = s->rng_ctr[i];
    }
}

ORC_API void orc_sim_set_state(orc_sim* s, const double* pose, const double* goal,
                               const double* past_dist, const float* past_action,
                               const int32_t* ep_step, const uint32_t* rng_ctr) {
    int N = s->cfg.n_envs;
    for (int i = 0; i < N; ++i) {
        if (pose) { s->x[i] = pose[3 * i]; s->y[i] = pose[3 * i + 1]; s->th[i] = pose[3 * i + 2]; }
        if (goal) { s->gx[i] = goal[2 * i]; s->gy[i] = goal[2 * i + 1]; }
        if (past_dist) s->past_dist[i] = past_dist[i];
        if (past_action) { s->past_action[2 * i] = past_action[2 * i]; s->past_action[2 * i + 1] = past_action[2 * i + 1]; }
        if (ep_step) s->ep_step[i] = ep_step[i];
        if (rng_ctr) s->rng_ctr[i] = rng_ctr[i];
    }
}

/* Uniform goal in [lo,hi]^2 (random.uniform = a + (b-a)*random(), environment_new.py:337-338)
 * from one Philox call per attempt, rejection per the rectangle list. */
static void sample_goal(orc_sim* s, int i, int which) {
    const orc_cfg* c = &s->cfg;
    const double* rects = which ? &s->respawn_rects[0][0] : &s->reset_rects[0][0];
    int R = which ? s->n_respawn_rects : s->n_reset_rects;
    uint64_t gid = c->env_id_base + (uint64_t)i;
    uint32_t key[2] = {(uint32_t)c->seed, (uint32_t)(c->seed >> 32)};
    double gx = 0, gy = 0;
    for (int tries = 0; tries < MAX_GOAL_TRIES; ++tries) {
        uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), s->rng_ctr[i], 0x6e617673u};
        uint32_t r[4];
        orc_philox4x32_10(ctr, key, r);
        s->rng_ctr[i] += 1;
        double ux = (double)((((uint64_t)r[0] << 32) | r[1]) >> 11) * 0x1.0p-53;
        double uy = (double)((((uint64_t)r[2] << 32) | r[3]) >> 11) * 0x1.0p-53;
        gx = c->goal_lo + (c->goal_hi - c->goal_lo) * ux;
        gy = c->goal_lo + (c->goal_hi - c->goal_lo) * uy;
        if (!rejected(rects, R, gx, gy)) break;
    }
    s->gx[i] = gx;
    s->gy[i] = gy;
}

/* LiDAR: B rays from the sensor origin, nearest hit against the segment list.
 * f64 pose -> f32 origin and directions, then f32 tests (explicit fmaf), t = k/den
 * correctly rounded, range = min over hits.  Out-of-range handling per
 * gazebo.xacro:117-120: >= max -> +inf ; < min -> clamped to min ("clamp" mode). */
static void raycast_best(const float* seg, int S, double x, double y, double th, const double* beam_cos,
                         const double* beam_sin, int B, float* best_out) {
    double cth = cos(th), sth = sin(th);
    double ox = x + LIDAR_X * cth;
    double oy = y + LIDAR_X * sth;
    float oxf = (float)ox, oyf = (float)oy;
    for (int b = 0; b < B; ++b) {
        double c = cth * beam_cos[b] - sth * beam_sin[b];
        double sn = sth * beam_cos[b] + cth * beam_sin[b];
        float cf = (float)c, sf = (float)sn;
        float best = INFINITY;
        for (int j = 0; j < S; ++j) {
            float ax = seg[4 * j], ay = seg[4 * j + 1], bx = seg[4 * j + 2], by = seg[4 * j + 3];
            float rx = ax - oxf, ry = ay - oyf;
            float ex = bx - ax, ey = by - ay;
            float k = fmaf(rx, ey, -(ry * ex));   /* cross(a - o, e) */
            float den = fmaf(cf, ey, -(sf * ex)); /* cross(d, e)     */
            float un = fmaf(rx, sf, -(ry * cf));  /* cross(a - o, d) */
            int valid;
            if (den > 0.0f)
                valid = (k >= 0.0f) && (un >= 0.0f) && (un <= den);
            else if (den < 0.0f)
                valid = (k <= 0.0f) && (un <= 0.0f) && (un >= den);
            else
                valid = 0;
            if (valid) {
                float t = k / den;
                if (t < best) best = t;
            }
        }
        best_out[b] = best;
    }
}

/* gazebo.xacro:117-126: >= max -> +inf ; < min -> min (mode 0) or -inf (mode 1) ; else best + sigma*n clamped to [min,max] */
static float sensor_value(float best, float sigma, float n, int below_min_mode) {
    if (!(best < RANGE_MAX)) return INFINITY;
    if (best < RANGE_MIN) return below_min_mode ? -INFINITY : RANGE_MIN;
    float r = fmaf(sigma, n, best);
    return fminf(fmaxf(r, RANGE_MIN), RANGE_MAX);
}

/* standard normal for beam b: Philox per four beams + Box-Muller in float32 (same counter layout as the kernel) */
static float lidar_noise(const orc_cfg* c, uint64_t gid, uint32_t ctr, uint32_t ep_step, int b) {
    uint32_t key[2] = {(uint32_t)c->seed, (uint32_t)(c->seed >> 32)};
    uint32_t cw[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), ctr, 0x4C000000u | ((ep_step & 0xFFFFu) << 8) | (uint32_t)(b >> 2)};
    uint32_t r[4];
    orc_philox4x32_10(cw, key, r);
    int pr = (b >> 1) & 1;
    float u1 = ((float)(r[2 * pr] >> 8) + 1.0f) * 0x1.0p-24f;
    float u2 = (float)(r[2 * pr + 1] >> 8) * 0x1.0p-24f;
    float rad = sqrtf(-2.0f * logf(u1));
    float ang = 6.283185307179586f * u2;
    return (b & 1) ? rad * sinf(ang) : rad * cosf(ang);
}

/* LiDAR: B rays from the sensor origin, nearest hit against the segment list.
 * f64 pose -> f32 origin and directions, then f32 tests (explicit fmaf), t = k/den
 * correctly rounded, range = min over hits.  Noise-free, clamp mode (the default sensor). */
ORC_API void orc_raycast(const float* seg, int S, double x, double y, double th,
                         const double* beam_cos, const double* beam_sin, int B, float* ranges) {
    raycast_best(seg, S, x, y, th, beam_cos, beam_sin, B, ranges);
    for (int b = 0; b < B; ++b) ranges[b] = sensor_value(ranges[b], 0.f, 0.f, 0);
}

static const float* env_seg(const orc_sim* s, int i) {
    return s->per_env ? s->seg + (size_t)i * s->S * 4 : s->seg;
}

/* observation for env i at its current pose; returns flags through pointers */
static void observe(orc_sim* s, int i, const double past_action[2], uint32_t noise_ctr, uint32_t noise_step,
                    float* obs_row, double* dist, int32_t* done, int32_t* arrive) {
    int B = s->cfg.n_beams;
    const orc_cfg* c = &s->cfg;
    float rf[256];
    double rd[256], scan[256], obs[256 + 6];
    raycast_best(env_seg(s, i), s->S, s->x[i], s->y[i], s->th[i], s->beam_cos, s->beam_sin, B, rf);
    for (int b = 0; b < B; ++b) {
        float n = (c->lidar_noise_sigma > 0.f) ? lidar_noise(c, c->env_id_base + (uint64_t)i, noise_ctr, noise_step, b) : 0.f;
        rf[b] = sensor_value(rf[b], c->lidar_noise_sigma, n, c->lidar_below_min);
    }
    for (int b = 0; b < B; ++b) rd[b] = (double)rf[b];
    /* odom message as Gazebo would publish it: yaw-only quaternion */
    double qz = sin(s->th[i] / 2), qw = cos(s->th[i] / 2);
    int32_t yaw;
    double rel_theta, diff;
    orc_get_odometry(s->x[i], s->y[i], 0.0, 0.0, qz, qw, s->gx[i], s->gy[i], &yaw, &rel_theta, &diff);
    orc_get_state(rd, B, s->x[i], s->y[i], s->gx[i], s->gy[i], s->cfg.threshold_arrive, scan, dist,
                  done, arrive);
    orc_assemble_obs(scan, B, B, past_action, *dist, (double)yaw, rel_theta, diff, obs);
    for (int k = 0; k < B + 6; ++k) obs_row[k] = (float)obs[k]; /* ppo.py:616 float64 -> float32 */
}

/* spawn_goal_sampler.py:52-62: uniform table picks until min_dist <= |start-goal| <= max_dist, <= 100 attempts, then
 * one unconditional pick; one Philox call per attempt (r[0] -> start index, r[1] -> goal index). */
static int sample_tables(orc_sim* s, int i) {
    const orc_cfg* c = &s->cfg;
    uint64_t gid = c->env_id_base + (uint64_t)i;
    uint32_t key[2] = {(uint32_t)c->seed, (uint32_t)(c->seed >> 32)};
    int k = 0;
    for (int tries = 0; tries <= 100; ++tries) {
        uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), s->rng_ctr[i], 0x6e617673u};
        uint32_t r[4];
        orc_philox4x32_10(ctr, key, r);
        s->rng_ctr[i] += 1;
        k = (int)(((uint64_t)r[0] * (uint64_t)s->K) >> 32);
        int g = (int)(((uint64_t)r[1] * (uint64_t)s->G) >> 32);
        s->gx[i] = s->goals[2 * g];
        s->gy[i] = s->goals[2 * g + 1];
        double dx = s->starts[3 * k] - s->gx[i], dy = s->starts[3 * k + 1] - s->gy[i];
        double dist = sqrt(dx * dx + dy * dy); /* np.linalg.norm, :57 */
        if (tries == 100 || (s->min_dist <= dist && dist <= s->max_dist)) break;
    }
    return k;
}

/* use_key: the in-step auto-reset re-uses the step's noise draws (noise_ctr, noise_step); an explicit reset keys its
 * noise by (goal draws after sampling, 0xFFFF) */
static void reset_env(orc_sim* s, int i, float* obs_row, int use_key, uint32_t noise_ctr, uint32_t noise_step) {
    int k = 0;
    if (s->G > 0)
        k = sample_tables(s, i);
    else
        sample_goal(s, i, 0); /* environment_new.py:337-345 */
    s->x[i] = s->starts[3 * k]; /* /gazebo/reset_world, environment_new.py:323-325 */
    s->y[i] = s->starts[3 * k + 1];
    s->th[i] = s->starts[3 * k + 2];
    s->ep_step[i] = 0;
    s->ep_ret[i] = 0;
    s->ep_path[i] = 0;
    s->past_action[2 * i] = 0;
    s->past_action[2 * i + 1] = 0;
    double zero[2] = {0, 0};
    double dist;
    int32_t d, a;
    if (!use_key) { noise_ctr = s->rng_ctr[i]; noise_step = 0xFFFFu; }
    observe(s, i, zero, noise_ctr, noise_step, obs_row, &dist, &d, &a);
    s->past_dist[i] = dist; /* getGoalDistace :116-120, :359 */
}

/* mask: nullable [N] u8; obs: [N][B+6] f32, rows of unmasked envs untouched */
ORC_API void orc_sim_reset(orc_sim* s, const uint8_t* mask, float* obs) {
    int N = s->cfg.n_envs, D = s->cfg.n_beams + 6;
    for (int i = 0; i < N; ++i)
        if (!mask || mask[i]) reset_env(s, i, obs + (size_t)i * D, 0, 0, 0);
}

/* one env step for all envs.  past_action_override: nullable [N][2] (Env.step(action, past_action)).
 * ended / ep_return / ep_length nullable. */
ORC_API void orc_sim_step(orc_sim* s, const float* action, const float* past_action_override,
                          float* obs, float* reward, uint8_t* done, uint8_t* arrive, uint8_t* ended,
                          float* ep_return, int32_t* ep_length, float* ep_path) {
    const orc_cfg* c = &s->cfg;
    int N = c->n_envs, D = c->n_beams + 6;
    for (int i = 0; i < N; ++i) {
        double a0 = (double)action[2 * i], a1 = (double)action[2 * i + 1];
        /* environment_new.py:273-278 */
        double v = a0 / 4;
        double w = a1;
        /* turtlebot3_fake.cpp:117-118 */
        double vl = v - (w * WHEEL_SEP / 2);
        double vr = v + (w * WHEEL_SEP / 2);
        double dt = 1.0 / 30.0;
        /* turtlebot3_fake.cpp:133-146 */
        double wl = vl / WHEEL_RADIUS, wr = vr / WHEEL_RADIUS;
        double wheel_l = wl * dt, wheel_r = wr * dt;
        /* :154-155 */
        double delta_s = WHEEL_RADIUS * (wheel_r + wheel_l) / 2.0;
        double delta_theta = WHEEL_RADIUS * (wheel_r - wheel_l) / WHEEL_SEP;
        const double x_old = s->x[i], y_old = s->y[i];
        for (int k = 0; k < SUBSTEPS; ++k) { /* :158-160 */
            s->x[i] += delta_s * cos(s->th[i] + (delta_theta / 2.0));
            s->y[i] += delta_s * sin(s->th[i] + (delta_theta / 2.0));
            s->th[i] += delta_theta;
        }
        double pa[2];
        if (past_action_override) {
            pa[0] = (double)past_action_override[2 * i];
            pa[1] = (double)past_action_override[2 * i + 1];
        } else {
            pa[0] = (double)s->past_action[2 * i];
            pa[1] = (double)s->past_action[2 * i + 1];
        }
        double dist;
        int32_t d, a;
        const uint32_t nz_ctr = s->rng_ctr[i], nz_step = (uint32_t)s->ep_step[i];
        observe(s, i, pa, nz_ctr, nz_step, obs + (size_t)i * D, &dist, &d, &a);
        double r = orc_set_reward(&s->past_dist[i], dist, d, a);
        if (a && c->respawn_on_arrive) { /* environment_new.py:245-267 */
            sample_goal(s, i, 1);
            s->past_dist[i] = hypot(s->gx[i] - s->x[i], s->gy[i] - s->y[i]);
        }
        s->ep_step[i] += 1;
        s->ep_ret[i] += r;
        int timeout = (c->max_episode_steps > 0) && (s->ep_step[i] >= c->max_episode_steps); /* ppo.py:552 */
        int end = d || a || timeout;
        reward[i] = (float)r;
        done[i] = (uint8_t)d;
        arrive[i] = (uint8_t)a;
        if (ended) ended[i] = (uint8_t)end;
        if (end) {
            if (ep_return) ep_return[i] = (float)s->ep_ret[i];
            if (ep_length) ep_length[i] = s->ep_step[i];
            /* ppo.py:533-537: the path grows by |curr_pos - prev_pos| with positions read BEFORE each step, so the
             * displacement of the episode's last step is never added */
            if (ep_path) ep_path[i] = (float)s->ep_path[i];
        }
        {
            const double mx = s->x[i] - x_old, my = s->y[i] - y_old;
            s->ep_path[i] += sqrt(mx * mx + my * my); /* np.linalg.norm */
        }
        s->past_action[2 * i] = action[2 * i]; /* ppo.py:543 */
        s->past_action[2 * i + 1] = action[2 * i + 1];
        if (end && c->auto_reset) reset_env(s, i, obs + (size_t)i * D, 1, nz_ctr, nz_step); /* ppo.py:582-593 */
    }
}
