/*
 * navppo.h -- C ABI of the fused PPO loss + gradient kernel for the 16-64-64 heads (libnavsim.so).
 *
 * Replaces one pass of the reference's update loop body -- PPO.evaluate() (project_ppo/src/ppo.py:708-737),
 * the ratio / clipped-surrogate / MSE losses (ppo.py:316-343) and both backward() calls (ppo.py:349,386) --
 * for the "mlp64x2" policy (actor: Linear(16,64)-ReLU-Linear(64,64)-ReLU-{sigmoid(Linear(64,1)), tanh(Linear(64,1))},
 * critic: ...-Linear(64,1); net_actor.py:147-189, graph_code/ppo_for_beginners/network.py:11-50).
 * The optimiser step (Adam, ppo.py:381,392) and the gradient all-reduce stay with the caller.
 *
 * All pointers are DEVICE pointers owned by the caller; calls are asynchronous on `stream` (hipStream_t as void*);
 * return 0 or a negative code, message in navppo_last_error().  float32 throughout (f32-input MFMA: exact f32 fma
 * chains); no CPU fallback.
 */
#ifndef NAVPPO_H
#define NAVPPO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NAVPPO_MLP64_ACTOR_PARAMS 5378  /* 16*64+64 + 64*64+64 + 64+1 + 64+1 */
#define NAVPPO_MLP64_CRITIC_PARAMS 5313 /* 16*64+64 + 64*64+64 + 64+1 */
#define NAVPPO_MLP64_MAX_BLOCKS 256     /* persistent workgroups: one per CU */

const char* navppo_last_error(void);

/* bytes of scratch `workspace_dev` must provide (per-workgroup partial gradients) */
size_t navppo_mlp64_workspace_bytes(void);

/*
 *   params_dev   [5378 + 5313] f32   actor then critic, each in nn.Module.named_parameters() order:
 *                                    layer1.weight[64,16], layer1.bias[64], layer2.weight[64,64], layer2.bias[64],
 *                                    layer3.weight[1,64], layer3.bias[1] (, layer4.weight[1,64], layer4.bias[1])
 *   obs_dev      [n,16]  batch_obs        act_dev [n,2] batch_acts (the clamped actions, ppo.py:546)
 *   logp_old_dev [n]     batch_log_probs  rtg_dev [n]   batch_rtgs      adv_dev [n] normalised advantages A_k (ppo.py:284)
 *   var          diagonal of cov_mat (ppo.py:123-124)   clip   PPO clip (ppo.py:770)
 *   grad_dev     [5378 + 5313] f32  out: d(actor_loss)/d(actor params), d(critic_loss)/d(critic params), losses being
 *                                    MEANS over the n samples (ppo.py:342-343) -- overwritten, not accumulated
 *   stats_dev    [8] f32            out: [0] actor_loss [1] approx_kl [2] clip_frac (ppo.py:326,335) [4] critic_loss
 */
int navppo_mlp64_loss_grad(const float* params_dev, const float* obs_dev, const float* act_dev,
                           const float* logp_old_dev, const float* rtg_dev, const float* adv_dev, int64_t n_samples,
                           float var, float clip, float* grad_dev, float* stats_dev, void* workspace_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NAVPPO_H */
