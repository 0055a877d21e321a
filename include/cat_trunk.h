/*
 * cat_trunk.h -- C ABI of libcat_learn.so, part 2: the convolutional trunk of the self-play learner's
 * networks as one MI355X (gfx950) kernel per direction.
 *
 * Replaces, for the G stacked networks of a role, the first four layers of the reference's models
 * (src/models/lstm_policy_net.py:28-40, src/models/lstm_value_net.py:46-58):
 *     Conv1d(C, 64, kernel 5, stride 2) -> ReLU -> Conv1d(64, 32, kernel 5, stride 3) -> ReLU
 * on observation rows [C channels][R rays] (policy C = 2, value C = 4).  The 64-channel intermediate (30 positions x 64
 * channels = 1920 values per sample at R = 64) lives in LDS only: forward writes the [L2 x 32] result, backward
 * recomputes the intermediate, and returns the four parameter gradients as per-workgroup partial sums.
 *
 * Conventions as in cat_sim.h / cat_lstm.h (status codes, device buffers, stream as void*, bf16 = upper half of a
 * binary32, no CPU fallback).  Strides are in elements; innermost dimensions are contiguous.
 */
#ifndef CAT_TRUNK_H
#define CAT_TRUNK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAT_TRUNK_ABI_VERSION 2
#define CAT_TRUNK_C1 64          /* channels after the first convolution */
#define CAT_TRUNK_C2 32          /* channels after the second */
#define CAT_TRUNK_TILE 16        /* samples per workgroup pass */
#define CAT_TRUNK_BWD_WAVES 8    /* waves of a backward workgroup */

enum { CAT_TRUNK_OK = 0, CAT_TRUNK_ERR_BAD_ARG = -1, CAT_TRUNK_ERR_HIP = -2, CAT_TRUNK_ERR_TOO_LARGE = -3 };

typedef struct cat_trunk_dims {
    int32_t G;      /* stacked networks */
    int32_t N;      /* samples per network */
    int32_t C;      /* input channels: 2 or 4 */
    int32_t R;      /* rays (input positions), even */
} cat_trunk_dims;

/* L1 = (R - 5) / 2 + 1 positions after the first convolution, L2 = (L1 - 5) / 3 + 1 after the second. */
int cat_trunk_out_positions(const cat_trunk_dims *d);
/* 1 if the backward pass of these dimensions fits the 160 KB of LDS (R <= 102 for C = 4), else 0. */
int cat_trunk_supported(const cat_trunk_dims *d);
/* Number of workgroups per network the backward launch uses = the leading extent of the partial-sum buffers. */
int cat_trunk_backward_blocks(const cat_trunk_dims *d);

typedef struct cat_trunk_params {
    const void *w1;     /* bf16 [G][64][C][5] */
    const void *b1;     /* bf16 [G][64] */
    const void *w2;     /* bf16 [G][32][64][5] */
    const void *b2;     /* bf16 [G][32] */
    int64_t sw1_g, sb1_g, sw2_g, sb2_g;
} cat_trunk_params;

/* Optional row selection of the input (a PPO minibatch read straight out of the rollout buffer, without a gathered copy):
   with rows != NULL, sample n reads row (n / sel) * block + rows[n % sel] of x -- sel sequences picked out of block, step-major.
   Every rows[j] must lie in [0, block); the kernels do not check it. */
typedef struct cat_trunk_rows {
    const int64_t *rows;    /* DEVICE [sel], or NULL: sample n reads row n */
    int32_t sel, block;
} cat_trunk_rows;

typedef struct cat_trunk_fwd {
    cat_trunk_dims d;
    cat_trunk_params p;
    const void *x;      /* bf16 [G][N][C * R] (or the larger buffer x_rows selects from), (channel, ray) order */
    int64_t sx_g, sx_n;
    cat_trunk_rows x_rows;
    void *out;          /* bf16 [G][N][L2 * 32], (position, channel) order, after the second ReLU */
    int64_t so_g, so_n;
} cat_trunk_fwd;

typedef struct cat_trunk_bwd {
    cat_trunk_dims d;
    cat_trunk_params p;
    const void *x;
    int64_t sx_g, sx_n;
    cat_trunk_rows x_rows;   /* as in forward */
    const void *out;    /* what forward wrote (the ReLU mask of the second layer) */
    const void *d_out;  /* bf16, gradient w.r.t. out; same strides as out */
    int64_t so_g, so_n;
    /* fp32 partial sums, one slab per workgroup: the caller adds the B = cat_trunk_backward_blocks() slabs up */
    float *part_dw1;    /* [G][B][64][32]: column kk * C + c  (columns >= 5 C are not weights) */
    float *part_db1;    /* [G][B][64] */
    float *part_dw2;    /* [G][B][32][320]: column kk * 64 + c_in */
    float *part_db2;    /* [G][B][32] */
} cat_trunk_bwd;

/* Second stage of cat_trunk_backward: adds the B per-workgroup slabs up, puts the columns back into the parameters' own
   order and stores bf16 -- dst (+)= sum.  Each dst is [G][...] with the parameter's shape per network and row stride s*_g. */
typedef struct cat_trunk_finish_args {
    cat_trunk_dims d;
    int32_t blocks, accumulate;
    const float *part_dw1, *part_db1, *part_dw2, *part_db2;
    void *dw1, *db1, *dw2, *db2;        /* bf16 [G][64][C][5], [G][64], [G][32][64][5], [G][32] */
    int64_t sw1_g, sb1_g, sw2_g, sb2_g;
} cat_trunk_finish_args;
int cat_trunk_grad_finish(const cat_trunk_finish_args *a, void *stream);

int cat_trunk_abi_version(void);
const char *cat_trunk_last_error(void);
int cat_trunk_forward(const cat_trunk_fwd *a, void *stream);
int cat_trunk_backward(const cat_trunk_bwd *a, void *stream);

#ifdef __cplusplus
}
#endif
#endif
