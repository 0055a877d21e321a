/*
 * cat_rollout.h -- C ABI of libcat_learn.so, part 5: the glue of a rollout tick between the env core and the networks
 * (observation -> model-input packing; rank 2: action sampling), one launch each instead
 * of ~30 / ~20 small tensor kernels per tick.
 *
 *  - cat_rollout_pack: libcat_sim.so's observation buffers -> the networks' input rows, in the layouts skrl's flattening
 *    of the reference's Dict spaces produces (src/models/lstm_policy_net.py:101-103: policy row = [distance(R) |
 *    object_type(R)]; src/models/lstm_value_net.py:122-137: critic row = the first 4R entries of an agent's shared state
 *    in sorted-key order = [distance_shared | object_type_shared | own_distances | own_obj_types]; SURVEY quirk Q11: the
 *    reference feeds every critic the FIRST agent's state), scaled and stored as bf16.
 *  - cat_rollout_sample: one categorical draw per (agent, env) from the policy logits by inverse CDF on a supplied
 *    uniform number, with the log-probability of the drawn action, written where the rollout keeps them, and the action
 *    into the env's [N][A] action matrix.
 *
 * Conventions as in cat_sim.h.
 */
#ifndef CAT_ROLLOUT_H
#define CAT_ROLLOUT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAT_ROLLOUT_ABI_VERSION 1
#define CAT_ROLLOUT_MAX_AGENTS 8

enum { CAT_ROLLOUT_OK = 0, CAT_ROLLOUT_ERR_BAD_ARG = -1, CAT_ROLLOUT_ERR_HIP = -2 };

typedef struct cat_rollout_pack_args {
    int32_t N, A, R, G;                         /* envs, agents in the env, rays, agents packed by this call */
    int32_t agent[CAT_ROLLOUT_MAX_AGENTS];      /* env agent index of packed agent g */
    int32_t n_cops;                             /* agents [0, n_cops) are team 0 */
    int32_t first_agent_state;                  /* 1: every critic row is built from env agent 0 (reference quirk Q11) */
    float distance_scale, type_scale;
    const void *obs_distance;                   /* f16 [N][A][R]   (cat_outputs.obs_distance) */
    const void *obs_type;                       /* u8  [N][A][R] */
    const void *shared_distance;                /* f16 [N][2][R] */
    const void *shared_type;                    /* u8  [N][2][R] */
    void *policy_in;                            /* bf16 [G][N][2R], strides below (elements) */
    int64_t sp_g, sp_n;
    void *value_in;                             /* bf16 [G][N][4R] */
    int64_t sv_g, sv_n;
} cat_rollout_pack_args;

typedef struct cat_rollout_sample_args {
    int32_t N, A, G, pad;
    int32_t agent[CAT_ROLLOUT_MAX_AGENTS];      /* column of packed agent g in the action matrix */
    const void *logits;                         /* bf16 [G][N][4] contiguous */
    const float *uniform;                       /* [G][N] in [0, 1) */
    const void *values;                         /* bf16 [G][N] contiguous or NULL */
    int64_t *act_out;                           /* [G][N], row stride sa_g */
    float *logp_out;                            /* [G][N], row stride sl_g */
    float *value_out;                           /* [G][N], row stride sl_g; or NULL */
    int64_t sa_g, sl_g;
    int32_t *actions;                           /* [N][A] */
} cat_rollout_sample_args;

/* After the env tick: the agents' rewards into the rollout's [G][N] rows, and the episode-end flags (cat_outputs.terminated,
   u8 [N]) as the rollout keeps them: done_out / start_out (bool = 1 byte) and keep_out = 1 - terminated as fp32 (the recurrent
   networks' "carry the state" mask of the NEXT tick).  Any of the three flag outputs may be NULL. */
typedef struct cat_rollout_post_args {
    int32_t N, A, G, pad;
    int32_t agent[CAT_ROLLOUT_MAX_AGENTS];
    const float *reward;                        /* [N][A] (cat_outputs.reward) */
    const uint8_t *terminated;                  /* [N] */
    float *reward_out;                          /* [G][N], row stride sr_g */
    int64_t sr_g;
    uint8_t *done_out, *start_out;              /* [N] */
    float *keep_out;                            /* [N] */
} cat_rollout_post_args;
int cat_rollout_post(const cat_rollout_post_args *a, void *stream);

int cat_rollout_abi_version(void);
const char *cat_rollout_last_error(void);
int cat_rollout_pack(const cat_rollout_pack_args *a, void *stream);
int cat_rollout_sample(const cat_rollout_sample_args *a, void *stream);

#ifdef __cplusplus
}
#endif
#endif
