/*
 * cat_ppo.h -- C ABI of libcat_learn.so, part 3: the PPO loss (with its gradient) and the optimiser step of the
 * self-play learner, each as one or two launches instead of dozens of elementwise
 * kernels.  What is restated: skrl's MAPPO update as configured by the reference (src/configs/mappo_config.py:5-50 --
 * clipped surrogate, entropy bonus, scaled MSE value loss, KL early stop, gradient-norm clip, Adam); [SKRL-RECALL], as
 * the torch formulation in as_cops_and_thieves_amd/selfplay/mappo.py which these kernels are tested against.
 *
 * Conventions as in cat_sim.h (status codes, device buffers, stream as void*, no CPU fallback).
 */
#ifndef CAT_PPO_H
#define CAT_PPO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAT_PPO_ABI_VERSION 2
#define CAT_PPO_ACTIONS 4            /* the four impulse actions of every agent (cat_sim.h, cat_step) */
#define CAT_PPO_MAX_CHUNKS 256

enum { CAT_PPO_OK = 0, CAT_PPO_ERR_BAD_ARG = -1, CAT_PPO_ERR_HIP = -2 };

/* For each of the G agents, over its M samples:
 *   surrogate = sum min(adv ratio, adv clamp(ratio, 1 - clip, 1 + clip)),  ratio = exp(logp(action) - old_logp)
 *   sq_error  = sum (value - ret)^2
 *   entropy   = sum -sum_j p_j log p_j
 *   kl        = sum (ratio - 1) - (logp - old_logp)
 * partial[g][chunk] = {surrogate, sq_error, entropy, kl} of one chunk of the samples (the caller adds the chunks up), and
 * the gradient of   L = sum_g ( -surrogate_g - entropy_scale entropy_g + value_scale sq_error_g ) / M
 * w.r.t. the logits and the values. */
typedef struct cat_ppo_loss {
    int32_t G, M, chunks, pad;
    const void *logits;         /* bf16 [G][M][4] */
    const void *values;         /* bf16 [G][M] */
    const int64_t *actions;     /* [G][M], 0..3 */
    const float *old_logp, *adv, *ret;   /* [G][M] */
    float ratio_clip, value_scale, entropy_scale, pad2;
    void *d_logits;             /* bf16 [G][M][4] */
    void *d_values;             /* bf16 [G][M] */
    float *partial;             /* [G][chunks][4] */
} cat_ppo_loss;

/* One optimiser step on the flat [G][P] parameter buffers of a role:
 *   active_g  = epoch_active_g * (kl_g <= kl_threshold)           (kl_threshold <= 0: no gate); written back
 *   grad      = ar[g][0..P) * col_train;  clipped per agent to grad_norm_clip (torch.nn.utils.clip_grad_norm_)
 *   Adam (bias-corrected, per-entry step counts) on the entries with gate = active_g * col_train = 1
 *   lp        = bf16(master)                                       (when lp != NULL)
 * ar is [G][P + 1]: column P carries the agent's KL statistic (copied to kl_out). */
typedef struct cat_ppo_adam {
    int32_t G, P, chunks, pad;
    const float *ar;            /* [G][P + 1] */
    const float *col_train;     /* [G][P] */
    float *epoch_active;        /* [G] */
    float *m, *v, *steps;       /* [G][P] */
    float *master;              /* [G][P] */
    void *lp;                   /* bf16 [G][P] or NULL */
    float *kl_out;              /* [G] */
    float *norm_partial;        /* [G][chunks] scratch */
    float lr, beta1, beta2, eps, grad_norm_clip, kl_threshold;
} cat_ppo_adam;

/* Generalised advantage estimation over a stored rollout (skrl's compute_gae, reverse scan over the T ticks; one thread per
 * (agent, env) column):  delta_t = r_t + gamma * V_{t+1} * (1 - done_t) - V_t,  A_t = delta_t + gamma * lambda * (1 - done_t) * A_{t+1},
 * V_T = last_values, A_T = 0;  adv = A, ret = A + V.  The advantage normalisation stays with the caller. */
typedef struct cat_ppo_gae {
    int32_t G, T, N, pad;
    const float *rewards, *values;      /* [G][T][N] */
    const uint8_t *dones;               /* [T][N]: 1 = the episode ended with tick t (no bootstrap across it) */
    const float *last_values;           /* [G][N] */
    float gamma, lambda;
    float *adv, *ret;                   /* [G][T][N] */
} cat_ppo_gae;
int cat_ppo_gae_scan(const cat_ppo_gae *a, void *stream);

int cat_ppo_abi_version(void);
const char *cat_ppo_last_error(void);
int cat_ppo_loss_grad(const cat_ppo_loss *a, void *stream);
int cat_ppo_adam_step(const cat_ppo_adam *a, void *stream);

#ifdef __cplusplus
}
#endif
#endif
