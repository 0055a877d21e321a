/*
 * cat_dense.h -- C ABI of libcat_learn.so, part 4: the epilogues of the learner's dense layers (SURVEY.md section 8(f),
 * rank 2).  A layer of the G stacked networks is y = act(x W^T + b) (torch.nn.Linear + ReLU / Tanh in the reference's
 * models, src/models/lstm_policy_net.py:36-53, src/models/lstm_value_net.py:54-75).  The product stays a library GEMM; what
 * surrounds it -- the bias broadcast, the activation, the activation's derivative and the bias gradient (a column sum
 * over all rows) -- is one in-place pass forward and one pass backward instead of four library kernels and a GEMM with a
 * row of ones.  Conventions as in cat_sim.h.
 */
#ifndef CAT_DENSE_H
#define CAT_DENSE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAT_DENSE_ABI_VERSION 2
#define CAT_DENSE_MAX_CHUNKS 256
#define CAT_DENSE_MAX_OUT 1024

enum { CAT_DENSE_OK = 0, CAT_DENSE_ERR_BAD_ARG = -1, CAT_DENSE_ERR_HIP = -2 };
enum { CAT_ACT_NONE = 0, CAT_ACT_RELU = 1, CAT_ACT_TANH = 2 };

typedef struct cat_dense_dims {
    int32_t G;      /* stacked networks */
    int32_t M;      /* rows per network */
    int32_t out;    /* columns: 1, or a multiple of 4 up to CAT_DENSE_MAX_OUT */
    int32_t act;    /* CAT_ACT_* */
} cat_dense_dims;

/* y[g][m][:] = act(y[g][m][:] + bias[g][:]) in place; y bf16 [G][M][out] contiguous, bias bf16 with row stride sb_g */
int cat_dense_bias_act(const cat_dense_dims *d, void *y, const void *bias, int64_t sb_g, void *stream);

/* g_out = d_y * act'(y) (from the layer's OUTPUT y; g_out may be NULL when act is CAT_ACT_NONE) and
   partial[g][chunk][:] = column sums of that over chunk `chunk` of the rows: the caller adds the chunks up for the bias
   gradient.  d_y, y, g_out bf16 [G][M][out] contiguous; partial fp32 [G][chunks][out]. */
int cat_dense_act_grad(const cat_dense_dims *d, const void *d_y, const void *y, void *g_out, float *partial, int32_t chunks,
                       void *stream);

/* dst[g][j] (+)= sum over c < chunks of partial[g][c][j], j < n, stored as bf16: the second stage of the column sums above
   and of the other per-workgroup partial sums of this library.  dst1 may be NULL; accumulate != 0 adds to what dst holds. */
int cat_dense_sum_chunks(const float *partial, int32_t G, int32_t chunks, int32_t n, void *dst0, int64_t sd0_g, void *dst1,
                         int64_t sd1_g, int32_t accumulate, void *stream);
/* the same for two independent buffers in ONE launch (a layer's weight and bias gradients); b may be NULL */
typedef struct cat_dense_sum_job {
    const float *partial;
    int32_t chunks, n;
    void *dst0;
    int64_t sd0_g;
    void *dst1;
    int64_t sd1_g;
    int32_t accumulate, pad;
} cat_dense_sum_job;
int cat_dense_sum_chunks2(const cat_dense_sum_job *a, const cat_dense_sum_job *b, int32_t G, void *stream);

/* The weight gradient of a dense layer: partial[g][s][m][n] = sum over the rows k of split s of a[g][k][m] * b[g][k][n]
   (a = the gradient w.r.t. the layer's pre-activations [G][K][M], b = the layer's input [G][K][N], both bf16 contiguous;
   16-byte aligned when their width is a multiple of 8, else read element by element); fp32 slabs, one per split of the K rows, which cat_dense_sum_chunks adds up (into the bf16
   gradient).  splits = cat_dense_wgrad_splits(...) (enough workgroups to fill the device), at most CAT_DENSE_MAX_CHUNKS. */
typedef struct cat_dense_wgrad_args {
    int32_t G, K, M, N;
    const void *a, *b;
    float *partial;             /* [G][splits][M][N] */
    int32_t splits, pad;
    /* optional second input that shares the gradient a (an LSTM layer: W_ih's input and W_hh's h_in share d_xproj, which
       is then read once): b1 bf16 [G][K][N1] or NULL, its slabs partial1 [G][splits][M][N1].  splits: take
       cat_dense_wgrad_splits(G, K, M, 128 * (ceil(N / 128) + ceil(N1 / 128))) -- the tile count of both together. */
    const void *b1;
    float *partial1;
    int32_t N1, pad1;
} cat_dense_wgrad_args;
int cat_dense_wgrad_splits(int32_t G, int32_t K, int32_t M, int32_t N);
int cat_dense_wgrad(const cat_dense_wgrad_args *a, void *stream);

/* The layer itself and its input gradient for the G stacked networks (bf16, fp32 accumulation), so that a training step's
   products do not depend on a BLAS library's kernel selection:
     forward : y[g][m][n] = act( sum_k x[g][m][k] w[g][n][k] + bias[g][n] )      x [G][M][K], w [G][N][K] (row stride K, network
               stride sw_g), bias [G][N] (stride sb_g) or NULL, y [G][M][N]; K % 8 == 0
     dgrad   : dx[g][m][k] = sum_n gr[g][m][n] w[g][n][k]                        gr [G][M][N], dx [G][M][K]; K % 8 == 0
   Everything contiguous in its last dimension and 16-byte aligned (gr: when N % 8 == 0, else read element by element). */
typedef struct cat_dense_gemm_args {
    int32_t G, M, N, K;
    const void *x_or_gr;        /* forward: x; dgrad: gr */
    const void *w;
    int64_t sw_g;
    const void *bias;           /* forward only */
    int64_t sb_g;
    int32_t act, pad;           /* forward only: CAT_ACT_* */
    void *out;                  /* forward: y; dgrad: dx */
} cat_dense_gemm_args;
int cat_dense_forward(const cat_dense_gemm_args *a, void *stream);
int cat_dense_dgrad(const cat_dense_gemm_args *a, void *stream);

int cat_dense_abi_version(void);
const char *cat_dense_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
