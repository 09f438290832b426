/*
 * navppo.h -- C ABI of the fused PPO loss + gradient kernel for the D-64-64 heads (libnavsim.so).
 *
 * Replaces one pass of the reference's update loop body -- PPO.evaluate() (project_ppo/src/ppo.py:708-737),
 * the ratio / clipped-surrogate / MSE losses (ppo.py:316-343) and both backward() calls (ppo.py:349,386) --
 * for the "mlp64x2" policy (actor: Linear(D,64)-ReLU-Linear(64,64)-ReLU-{sigmoid(Linear(64,1)), tanh(Linear(64,1))},
 * critic: ...-Linear(64,1); net_actor.py:147-189, graph_code/ppo_for_beginners/network.py:11-50).
 * The optimiser step (Adam, ppo.py:381,392) and the gradient all-reduce stay with the caller.
 *
 * Observation rows (every `obs_dev` below): `obs_dim` = D = n_beams + 6 columns -- 16 (10 beams: main.py:18, BASELINE
 * configs[0..2] and [4]) or 42 (36 beams: configs[3]) -- of float32 (`obs_f16` = 0) or float16 (`obs_f16` = 1, configs[4]'s
 * "fp16 obs buffers", navsim_cfg.obs_f16), row-major and dense.  Half rows are widened to float32 as they are loaded; all
 * arithmetic behind the load is float32 either way.
 *
 * All pointers are DEVICE pointers owned by the caller; calls are asynchronous on `stream` (hipStream_t as void*);
 * return 0 or a negative code, message in navppo_last_error().  float32 arithmetic throughout (navppo_mlp64_*: f32-input MFMA, exact
 * f32 fma chains; navppo_mlp64_bf16x3_* and navppo_resmlp512_*: float32 products from three bf16 pieces per operand, see there);
 * no CPU fallback.  Alignment: params_dev / actor_params_dev 16 bytes, act_dev 8 bytes, obs_dev 16 bytes for
 * 16 columns, 8 bytes for 42 float32 columns, 4 bytes for 42 float16 columns (torch allocations are 256-byte aligned),
 * checked at the call.
 */
#ifndef NAVPPO_H
#define NAVPPO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NAVPPO_MLP64_ACTOR_PARAMS_D(d) ((d) * 64 + 64 + 64 * 64 + 64 + 64 + 1 + 64 + 1)  /* D = 16: 5378, D = 42: 7042 */
#define NAVPPO_MLP64_CRITIC_PARAMS_D(d) ((d) * 64 + 64 + 64 * 64 + 64 + 64 + 1)          /* D = 16: 5313, D = 42: 6977 */
#define NAVPPO_MLP64_ACTOR_PARAMS NAVPPO_MLP64_ACTOR_PARAMS_D(16)
#define NAVPPO_MLP64_CRITIC_PARAMS NAVPPO_MLP64_CRITIC_PARAMS_D(16)
#define NAVPPO_MLP64_MAX_BLOCKS 512     /* rows of the workspace (the kernel launches one persistent workgroup per CU) */

const char* navppo_last_error(void);

/* bytes of scratch `workspace_dev` must provide (per-workgroup partial gradients) for rows of obs_dim columns; 0 = bad obs_dim */
size_t navppo_mlp64_workspace_bytes(int32_t obs_dim);

/*
 *   params_dev   [PA + PC] f32       actor then critic (PA = NAVPPO_MLP64_ACTOR_PARAMS_D(D), PC = ..._CRITIC_PARAMS_D(D); D = 16:
 *                                    5378 + 5313), each in nn.Module.named_parameters() order:
 *                                    layer1.weight[64,D], layer1.bias[64], layer2.weight[64,64], layer2.bias[64],
 *                                    layer3.weight[1,64], layer3.bias[1] (, layer4.weight[1,64], layer4.bias[1])
 *   obs_dev      [n,D]   batch_obs (f32 | f16, see above)      act_dev [n,2] batch_acts (the clamped actions, ppo.py:546)
 *   logp_old_dev [n]     batch_log_probs  rtg_dev [n]   batch_rtgs      adv_dev [n] normalised advantages A_k (ppo.py:284)
 *   var          diagonal of cov_mat (ppo.py:123-124)   clip   PPO clip (ppo.py:770)
 *   grad_dev     [PA + PC] f32      out: d(actor_loss)/d(actor params), d(critic_loss)/d(critic params), losses being
 *                                    MEANS over the n samples (ppo.py:342-343) -- overwritten, not accumulated
 *   stats_dev    [8] f32            out: [0] actor_loss [1] approx_kl [2] clip_frac (ppo.py:326,335) [4] critic_loss
 */
int navppo_mlp64_loss_grad(const float* params_dev, const void* obs_dev, int32_t obs_dim, int32_t obs_f16, const float* act_dev,
                           const float* logp_old_dev, const float* rtg_dev, const float* adv_dev, int64_t n_samples,
                           float var, float clip, float* grad_dev, float* stats_dev, void* workspace_dev, void* stream);

/*
 * One net of navppo_mlp64_loss_grad (net 0 = actor: ppo.py:316-342,349; net 1 = critic: :343,386): writes that net's slice of
 * grad_dev and its statistics only.  The multi-GPU epoch launches the actor's pass, starts the all-reduce of the actor's
 * gradient slice, and runs the critic's pass while that all-reduce is in flight (same arguments as navppo_mlp64_loss_grad).
 */
int navppo_mlp64_loss_grad_net(int32_t net, const float* params_dev, const void* obs_dev, int32_t obs_dim, int32_t obs_f16,
                               const float* act_dev, const float* logp_old_dev, const float* rtg_dev, const float* adv_dev,
                               int64_t n_samples, float var, float clip, float* grad_dev, float* stats_dev, void* workspace_dev,
                               void* stream);

/*
 * torch.optim.Adam's step (defaults: no weight decay, no amsgrad) on a flat buffer with the gradient scaled first: the
 * multi-GPU epoch is navppo_mlp64_loss_grad -> all-reduce(sum) of grad_dev over RCCL -> navppo_adam_step(grad_scale = 1 / world).
 */
int navppo_adam_step(float* params_dev, const float* grad_dev, float* adam_m_dev, float* adam_v_dev, int64_t n, float grad_scale,
                     float lr, float beta1, float beta2, float eps, int32_t step, void* stream);

/*
 * The episode sums behind one iteration's log line (ppo.py:552-560, :833) over the [T, N] buffers a rollout fills (n = T N
 * entries each): sums_dev [6] f64 = episodes (ended), successes (arrive), collisions (done and not arrive), timeouts (ended,
 * neither), sum of ep_length (all entries; the buffer is zero where no episode ended), sum of ep_return over ended entries.
 * workspace_dev: NAVPPO_EPISODE_SUMS_WS_BYTES.  Per-block partials are added in a fixed order: the result is deterministic.
 */
#define NAVPPO_EPISODE_SUMS_WS_BYTES (256 * 6 * 8)
int navppo_episode_sums(const uint8_t* ended_dev, const uint8_t* arrive_dev, const uint8_t* done_dev, const int32_t* ep_length_dev,
                        const float* ep_return_dev, int64_t n, double* sums_dev, void* workspace_dev, void* stream);

/* V = critic(obs).squeeze() (ppo.py:275, :724) for n rows: the forward half of the critic's fused pass.  value_dev [n] f32. */
int navppo_mlp64_value(const float* critic_params_dev, const void* obs_dev, int32_t obs_dim, int32_t obs_f16, int64_t n_samples,
                       float* value_dev, void* stream);

/*
 * One whole update epoch of ppo.py:305-392 on one GPU: navppo_mlp64_loss_grad followed by the two Adam steps of
 * ppo.py:381,392 (torch.optim.Adam defaults: no weight decay, no amsgrad) applied in place by the kernel that sums the
 * workgroups' partial gradients.  step = 1, 2, ... (Adam's bias correction); adam_m_dev / adam_v_dev [PA + PC] f32 are
 * the optimiser's moments (zero before the first step).  grad_dev and stats_dev are filled as by navppo_mlp64_loss_grad; in addition
 * stats_dev[3] / [7] receive the SQUARED gradient norms of the actor / the critic of the update_epoch call BEFORE this one on the same
 * workspace (calls with consecutive `step`; the value of the first call is meaningless): with the last epoch's norms taken from grad_dev
 * a caller has every epoch's norms for the means the reference logs (ppo.py:351-352, 389-390) at no extra launch.  The same holds for
 * navppo_mlp64_bf16x3_update_epoch and navppo_resmlp512_update_epoch.
 * Multi-GPU runs use navppo_mlp64_loss_grad + an all-reduce + their own optimiser step instead.
 */
int navppo_mlp64_update_epoch(float* params_dev, const void* obs_dev, int32_t obs_dim, int32_t obs_f16, const float* act_dev,
                              const float* logp_old_dev, const float* rtg_dev, const float* adv_dev, int64_t n_samples, float var,
                              float clip, float lr, float beta1, float beta2, float eps, int32_t step, float* adam_m_dev,
                              float* adam_v_dev, float* grad_dev, float* stats_dev, void* workspace_dev, void* stream);

/* ------------------------------------------------------------------------------------------------------------------------
 * The same update (ppo.py:305-397; 16- or 42-column rows, float32 or float16) with every matrix product evaluated on the bf16 MFMA from operands split into
 * three bf16 pieces -- "bf16x3": a = a0 + a1 + a2 exactly (8 + 8 + 8 significand bits), a b ~ the six leading piece products,
 * each exact in float32, float32 accumulation, small terms first.  float32-equivalent by measurement (against float64 the
 * error is not above the f32-MFMA path's on any contraction of the kernel: tests/test_gpu_bf16x3.py), 16 x the
 * MFMA rate per piece product; inputs, outputs, statistics and the Adam step are those of the float32 entry points above,
 * which stay available (PPOConfig.update_arith = "f32").
 *
 * The observations are split ONCE per update -- they do not change over the epochs -- into prep_dev
 * (navppo_mlp64_bf16x3_prep_bytes(n, obs_dim) bytes -- 192 per sample at 16 columns, 576 at 42 (48 on chip): row-major pieces
 * for the forward products, tile-transposed pieces for the weight gradient of layer 1); the epoch entry points then take
 * (prep_dev, obs_dim) in place of (obs_dev, obs_dim, obs_f16).
 */
size_t navppo_mlp64_bf16x3_prep_bytes(int64_t n_samples, int32_t obs_dim /* 16 | 42; anything else: 0 */);
int navppo_mlp64_bf16x3_prepare(const void* obs_dev, int32_t obs_dim /* 16 | 42 */, int32_t obs_f16, int64_t n_samples, void* prep_dev,
                                void* stream);
/* as navppo_mlp64_loss_grad / _loss_grad_net / _update_epoch, same workspace (navppo_mlp64_workspace_bytes(obs_dim)) */
int navppo_mlp64_bf16x3_loss_grad(const float* params_dev, const void* prep_dev, int32_t obs_dim, const float* act_dev,
                                  const float* logp_old_dev, const float* rtg_dev, const float* adv_dev, int64_t n_samples, float var, float clip,
                                  float* grad_dev, float* stats_dev, void* workspace_dev, void* stream);
int navppo_mlp64_bf16x3_loss_grad_net(int32_t net, const float* params_dev, const void* prep_dev, int32_t obs_dim, const float* act_dev,
                                      const float* logp_old_dev, const float* rtg_dev, const float* adv_dev, int64_t n_samples, float var,
                                      float clip, float* grad_dev, float* stats_dev, void* workspace_dev, void* stream);
int navppo_mlp64_bf16x3_update_epoch(float* params_dev, const void* prep_dev, int32_t obs_dim, const float* act_dev, const float* logp_old_dev,
                                     const float* rtg_dev, const float* adv_dev, int64_t n_samples, float var, float clip, float lr,
                                     float beta1, float beta2, float eps, int32_t step, float* adam_m_dev, float* adam_v_dev,
                                     float* grad_dev, float* stats_dev, void* workspace_dev, void* stream);

/*
 * PPO.get_action() (ppo.py:673-706) for all envs of a shard in one launch: mean = actor(obs) (net_actor forward),
 * sample MVN(mean, var*I), clamp a0 to [0,1] and a1 to [-1,1] (ppo.py:700-703), log-prob of the CLAMPED action (:704).
 *   actor_params_dev [PA]   obs_dev [n,D] (f32 | f16)   act_dev [n,2] out   logp_dev [n] out   mean_dev [n,2] out, nullable
 *   noise_dev [n,2] standard normal draws, nullable: NULL = Philox4x32-10 keyed by (seed, env_id_base + i, step) +
 *   Box-Muller inside the kernel (the reference uses torch's global generator, unseeded by default: ppo.py:805-811);
 *   step = *step_base_dev (nullable = 0) + step_offset.  var_dev and step_base_dev are DEVICE scalars so that a
 *   captured hipGraph of T launches sees the current variance and a fresh noise stream on every replay.
 * The exploration-covariance decay of ppo.py:694-695 is the caller's (it changes `var` between launches).
 */
int navppo_mlp64_act(const float* actor_params_dev, const void* obs_dev, int32_t obs_dim, int32_t obs_f16, const float* noise_dev,
                     int64_t n_envs, const float* var_dev, uint64_t seed, uint64_t env_id_base, const uint32_t* step_base_dev,
                     uint32_t step_offset, float* act_dev, float* logp_dev, float* mean_dev, void* stream);

/* ------------------------------------------------------------------------------------------------------------------------
 * The reference's ACTIVE nets ("resmlp512"): NetActor / NetCritic (project_ppo/src/net_actor.py:56-144, net_critic.py:50-130),
 * two residual blocks ResBlock(16, 16) and ResBlock(32, 32) with 512 hidden units and LeakyReLU(0.2)
 * (net_actor.py:16-53), actor heads sigmoid(out1) / tanh(out2), critic head out.  Fused MFMA kernels (csrc/ppo_resmlp512.hip):
 * float32 in, float32 out, float32 accumulate; every matrix product except the weight gradients of the first block runs as float32
 * products rebuilt from three exact bf16 pieces per operand ("bf16x3", as navppo_mlp64_bf16x3_*; where a product contracts over 16
 * values only, two piece products share one MFMA), those run on the f32-input MFMA -- ONE arithmetic, also
 * for navppo_resmlp512_value (a caller who needs native float32 products throughout uses PyTorch: PPOConfig.update_arith = "f32").
 * Same contracts as the mlp64 entry points above unless stated.
 *
 *   params_dev  [50290 + 50257] f32   actor then critic, each in nn.Module.named_parameters() order WITHOUT the BatchNorm
 *               entries the reference's forward never uses (net_actor.py:44,48,137):
 *               rb1.fc1.weight[512,16], rb1.fc1.bias[512], rb1.fc2.weight[16,512], rb1.fc2.bias[16],
 *               rb2.fc1.weight[512,32], rb2.fc1.bias[512], rb2.fc2.weight[32,512], rb2.fc2.bias[32],
 *               out1.weight[1,32], out1.bias[1], out2.weight[1,32], out2.bias[1]      (critic: out.weight[1,32], out.bias[1])
 *   workspace_dev  navppo_resmlp512_workspace_bytes(n_samples) bytes (partial block outputs per hidden slice, 2.5 KB per
 *               sample, + partial-gradient rows); contents are scratch.
 */
#define NAVPPO_RESMLP512_ACTOR_PARAMS 50290
#define NAVPPO_RESMLP512_CRITIC_PARAMS 50257

size_t navppo_resmlp512_workspace_bytes(int64_t n_samples);

/* evaluate() + losses + both backward() calls of ppo.py:307-386; grad_dev [50290 + 50257], stats_dev [8] as navppo_mlp64_loss_grad.
 * obs_dev: [n, 16] rows, float32 (obs_f16 = 0) or float16 (obs_f16 = 1: BASELINE configs[4], navsim_cfg.obs_f16; ABI v6), 16-byte
 * aligned; half rows are widened as they are loaded, in every entry point below. */
int navppo_resmlp512_loss_grad(const float* params_dev, const void* obs_dev, int32_t obs_f16, const float* act_dev, const float* logp_old_dev,
                               const float* rtg_dev, const float* adv_dev, int64_t n_samples, float var, float clip,
                               float* grad_dev, float* stats_dev, void* workspace_dev, void* stream);

/* one whole epoch of ppo.py:305-392 on one GPU (losses, gradients, both Adam steps); as navppo_mlp64_update_epoch */
int navppo_resmlp512_update_epoch(float* params_dev, const void* obs_dev, int32_t obs_f16, const float* act_dev, const float* logp_old_dev,
                                  const float* rtg_dev, const float* adv_dev, int64_t n_samples, float var, float clip, float lr,
                                  float beta1, float beta2, float eps, int32_t step, float* adam_m_dev, float* adam_v_dev,
                                  float* grad_dev, float* stats_dev, void* workspace_dev, void* stream);

/* V = critic(obs).squeeze() (ppo.py:275, :724); critic_params_dev [50257] (8-byte aligned suffices), value_dev [n] */
int navppo_resmlp512_value(const float* critic_params_dev, const void* obs_dev, int32_t obs_f16, int64_t n_samples, float* value_dev,
                           void* workspace_dev, void* stream);

/* PPO.get_action() (ppo.py:673-706) for all envs of a shard in one launch; arguments as navppo_mlp64_act, actor_params_dev [50290] */
int navppo_resmlp512_act(const float* actor_params_dev, const void* obs_dev, int32_t obs_f16, const float* noise_dev, int64_t n_envs,
                         const float* var_dev, uint64_t seed, uint64_t env_id_base, const uint32_t* step_base_dev,
                         uint32_t step_offset, float* act_dev, float* logp_dev, float* mean_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NAVPPO_H */
