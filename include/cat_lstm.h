/*
 * cat_lstm.h -- C ABI of libcat_learn.so: the LSTM recurrence of the self-play learner (SURVEY.md
 * section 8(f), rank 2: the caller of the env hot path) as two MI355X (gfx950) kernels.
 *
 * Replaces, for the role-stacked networks of as_cops_and_thieves_amd/selfplay/stacked.py, what the
 * reference gets from torch.nn.LSTM inside its skrl models (src/models/lstm_policy_net.py:28-53,
 * src/models/lstm_value_net.py:46-75: one nn.LSTM(256 -> 128) per network, sequences of 16 ticks,
 * states zeroed where an episode starts): the whole window of one layer, for the G stacked
 * networks of a role, in ONE launch forward and ONE launch backward (bf16 operands on the matrix
 * cores, fp32 accumulation and cell arithmetic).  The input-side projection, the weight gradients
 * and everything around the recurrence stay ordinary batched GEMMs of the caller.
 *
 * Conventions as in cat_sim.h: int status (0 ok, <0 error), caller-owned DEVICE buffers, explicit
 * hipStream_t passed as void* (NULL = default stream), no CPU fallback.  bf16 = the upper 16 bits of
 * an IEEE binary32.  H = 128 (hidden size, CAT_LSTM_HIDDEN) is compiled in; gate order i, f, g, o.
 */
#ifndef CAT_LSTM_H
#define CAT_LSTM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAT_LSTM_ABI_VERSION 1
#define CAT_LSTM_HIDDEN 128
#define CAT_LSTM_ROWS_PER_BLOCK 16   /* sequences per workgroup */
#define CAT_LSTM_MAX_T 1024

enum { CAT_LSTM_OK = 0, CAT_LSTM_ERR_BAD_ARG = -1, CAT_LSTM_ERR_HIP = -2 };

/* Element strides are in bf16 elements; the innermost dimension is contiguous everywhere.  Addresses of rows
   must be 8-byte aligned, those of w_hh rows 16-byte aligned. */
typedef struct cat_lstm_dims {
    int32_t G;      /* stacked networks */
    int32_t T;      /* steps of the window */
    int32_t B;      /* sequences per network */
    int32_t pad;
} cat_lstm_dims;

/* Bytes of the three "saved for backward" buffers cat_lstm_seq_forward fills when they are non-NULL (a layout only
   cat_lstm_seq_backward reads).  acts: the four activated gates; cell: c entering each step and tanh(c) leaving it. */
int cat_lstm_blocks(const cat_lstm_dims *d);       /* workgroups per network = ceil(B / CAT_LSTM_ROWS_PER_BLOCK) */
size_t cat_lstm_saved_acts_bytes(const cat_lstm_dims *d);
size_t cat_lstm_saved_cell_bytes(const cat_lstm_dims *d);

typedef struct cat_lstm_fwd {
    cat_lstm_dims d;
    const void *xproj;          /* bf16 [G][T][B][4H]: x_t W_ih^T (+ b_ih + b_hh when bias is NULL) */
    int64_t sx_g, sx_t, sx_b;
    const void *bias;           /* bf16 [G][4H], added to every step's pre-activations (b_ih, or b_ih + b_hh); or NULL */
    int64_t sb_g;
    const void *bias2;          /* a second such vector (b_hh), or NULL: the kernel adds the two */
    int64_t sb2_g;
    const void *w_hh;           /* bf16 [G][4H][H] */
    int64_t sw_g;
    const void *h0, *c0;        /* bf16 [G][B][H], contiguous */
    const float *keep;          /* fp32 [T][B]: 1 = carry the state into step t, 0 = an episode starts there; NULL = all 1 */
    void *out;                  /* bf16 [G][T][B][H]: h_t */
    int64_t so_g, so_t, so_b;
    void *h_last, *c_last;      /* bf16 [G][B][H], contiguous: h_T, c_T (not masked) */
    void *h_in;                 /* bf16 [G][T][B][H] contiguous or NULL: keep_t * h_{t-1}, the operand of the weight gradient */
    void *saved_acts;           /* or NULL (inference: nothing is kept) */
    void *saved_cell;           /* or NULL; both or neither */
} cat_lstm_fwd;

typedef struct cat_lstm_bwd {
    cat_lstm_dims d;
    const void *d_out;          /* bf16 [G][T][B][H] or NULL (= 0) */
    int64_t so_g, so_t, so_b;
    const void *d_h_last, *d_c_last;   /* bf16 [G][B][H] contiguous, or NULL (= 0) */
    const void *w_hh;
    int64_t sw_g;
    const float *keep;
    const void *saved_acts, *saved_cell;
    void *d_xproj;              /* bf16 [G][T][B][4H]: gradient of the summed gate pre-activations */
    int64_t sx_g, sx_t, sx_b;
    void *d_h0, *d_c0;          /* bf16 [G][B][H] contiguous, or NULL (not wanted) */
    float *part_dbias;          /* fp32 [G][cat_lstm_blocks()][4H] or NULL: per-workgroup sums of d_xproj over its rows and all
                                   steps (the bias gradient once the caller has added the workgroups up) */
} cat_lstm_bwd;

int cat_lstm_abi_version(void);
const char *cat_lstm_last_error(void);
/* h_t, c_t = cell(xproj[t] + (keep_t h_{t-1}) W_hh^T, keep_t c_{t-1}), t = 0..T-1 */
int cat_lstm_seq_forward(const cat_lstm_fwd *a, void *stream);
/* back-propagation through the same window; the weight gradient is d_xproj^T h_in, left to the caller */
int cat_lstm_seq_backward(const cat_lstm_bwd *a, void *stream);

#ifdef __cplusplus
}
#endif
#endif
