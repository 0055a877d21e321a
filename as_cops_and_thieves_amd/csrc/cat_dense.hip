// cat_dense.hip -- libcat_learn.so, part 4: bias + activation in place, and activation derivative + bias gradient
// (include/cat_dense.h).  Bandwidth-bound passes over [G][M][out] bf16: 8-byte accesses, consecutive threads on
// consecutive column groups; the column sums are two-stage (per-chunk partial sums in registers, LDS across the row
// lanes of a block, chunks added up by the caller).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "cat_dense.h"

namespace {

constexpr int BLOCK = 256;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float act_fwd(float x, int act)
{
    if (act == CAT_ACT_RELU) return fmaxf(x, 0.0f);
    if (act == CAT_ACT_TANH) return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
    return x;
}
__device__ __forceinline__ float act_der(float y, int act)   // from the activation's output
{
    if (act == CAT_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
    if (act == CAT_ACT_TANH) return 1.0f - y * y;
    return 1.0f;
}

__global__ __launch_bounds__(BLOCK) void bias_act_kernel(cat_dense_dims d, __bf16 *y, const __bf16 *bias, int64_t sb_g)
{
    const int g = blockIdx.y;
    const size_t per = (size_t)d.M * d.out;
    __bf16 *yg = y + (size_t)g * per;
    const __bf16 *bg = bias + (size_t)g * sb_g;
    if (d.out % 4 == 0) {
        const size_t n4 = per / 4;
        for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (size_t)gridDim.x * BLOCK) {
            const int col = (int)((4 * i) % d.out);
            f32x4 v = __builtin_convertvector(*(bf16x4 *)(yg + 4 * i), f32x4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e] + (float)bg[col + e], d.act);
            *(bf16x4 *)(yg + 4 * i) = __builtin_convertvector(v, bf16x4);
        }
    } else {   // out == 1
        const float b = (float)bg[0];
        for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < per; i += (size_t)gridDim.x * BLOCK)
            yg[i] = (__bf16)act_fwd((float)yg[i] + b, d.act);
    }
}

// block = (row lanes) x (column groups of 4); thread (rl, cg) walks the rows rl, rl + lanes, ... of its chunk
__global__ __launch_bounds__(BLOCK) void act_grad_kernel(cat_dense_dims d, const __bf16 *dy, const __bf16 *y, __bf16 *gout, float *partial)
{
    __shared__ float lds[BLOCK * 4];
    const int g = blockIdx.z, chunk = blockIdx.x, chunks = gridDim.x;
    const int width = d.out % 4 == 0 ? 4 : 1;
    const int groups = d.out / width;                 // column groups in a row
    const int gpb = groups < BLOCK ? groups : BLOCK;  // column groups per block pass
    const int lanes = BLOCK / gpb;                    // row lanes
    const int cgl = threadIdx.x % gpb, rl = threadIdx.x / gpb;
    const int rows_per = (d.M + chunks - 1) / chunks, r0 = chunk * rows_per, r1 = min(d.M, r0 + rows_per);
    const size_t base = (size_t)g * d.M * d.out;
    for (int cg0 = blockIdx.y * gpb; cg0 < groups; cg0 += gridDim.y * gpb) {
        const int cg = cg0 + cgl;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (cg < groups && rl < lanes) {
            for (int r = r0 + rl; r < r1; r += lanes) {
                const size_t o = base + (size_t)r * d.out + (size_t)cg * width;
                if (width == 4) {
                    f32x4 v = __builtin_convertvector(*(const bf16x4 *)(dy + o), f32x4);
                    if (d.act != CAT_ACT_NONE) {
                        const f32x4 yy = __builtin_convertvector(*(const bf16x4 *)(y + o), f32x4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= act_der(yy[e], d.act);
                        const bf16x4 vb = __builtin_convertvector(v, bf16x4);
                        *(bf16x4 *)(gout + o) = vb;
                        v = __builtin_convertvector(vb, f32x4);      // the sum of what the GEMMs will see
                    }
                    acc += v;
                } else {
                    float v = (float)dy[o];
                    if (d.act != CAT_ACT_NONE) {
                        v *= act_der((float)y[o], d.act);
                        const __bf16 vb = (__bf16)v;
                        gout[o] = vb;
                        v = (float)vb;
                    }
                    acc[0] += v;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[threadIdx.x * 4 + e] = acc[e];
        __syncthreads();
        if (rl == 0 && cg < groups) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < lanes; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += lds[(k * gpb + cgl) * 4 + e];
            float *out = partial + ((size_t)g * chunks + chunk) * d.out + (size_t)cg * width;
            for (int e = 0; e < width; ++e) out[e] = s[e];
        }
    }
}

// block = 32 columns x 8 chunk lanes: a column's chunks are read by eight threads in parallel (32 loads each at 256
// chunks) and added up through LDS.  One launch serves up to two independent jobs (a layer's weight and bias gradients).
struct SumJob {
    const float *partial;
    int chunks, n;
    __bf16 *dst0, *dst1;
    long long sd0, sd1;
    int accumulate, blocks;
};

__global__ __launch_bounds__(BLOCK) void sum_chunks_kernel(SumJob j0, SumJob j1)
{
    __shared__ float lds[BLOCK];
    const bool second = (int)blockIdx.x >= j0.blocks;
    const SumJob &jb = second ? j1 : j0;
    const int bx = second ? (int)blockIdx.x - j0.blocks : (int)blockIdx.x;
    const int g = blockIdx.y, cl = threadIdx.x & 31, lane = threadIdx.x >> 5, j = bx * 32 + cl;
    float s = 0.0f;
    if (j < jb.n) {
        const float *p = jb.partial + (size_t)g * jb.chunks * jb.n + j;
        for (int c = lane; c < jb.chunks; c += 8) s += p[(size_t)c * jb.n];
    }
    lds[threadIdx.x] = s;
    __syncthreads();
    if (lane == 0 && j < jb.n) {
#pragma unroll
        for (int k = 1; k < 8; ++k) s += lds[32 * k + cl];
        if (jb.accumulate) {
            jb.dst0[(size_t)g * jb.sd0 + j] = (__bf16)(s + (float)jb.dst0[(size_t)g * jb.sd0 + j]);
            if (jb.dst1) jb.dst1[(size_t)g * jb.sd1 + j] = (__bf16)(s + (float)jb.dst1[(size_t)g * jb.sd1 + j]);
        } else {
            jb.dst0[(size_t)g * jb.sd0 + j] = (__bf16)s;
            if (jb.dst1) jb.dst1[(size_t)g * jb.sd1 + j] = (__bf16)s;
        }
    }
}

// ---- weight gradient: C[m][n] = sum_k A[k][m] B[k][n], both operands stored with the reduction index k as the ROW.
// 128 x 128 output tile per workgroup, 4 waves of 64 x 64 (16 accumulator tiles), the K rows split over blockIdx.y.
// The operands go global -> LDS as they lie (coalesced 16-byte runs of a row) and come back as MFMA fragments through
// ds_read_b64_tr_b16, the hardware transposed read: a 16-lane group reads a 4-row x 16-column block and each lane
// receives one COLUMN of it -- the four k values of its m (or n).  A fragment's eight k's are rows {4q..4q+3} and
// {16+4q..16+4q+3} of the 32-row step (the same permutation for A and B, so the product is unchanged): with a row
// stride of 288 B the two groups of a 32-lane half then sit in disjoint banks.
constexpr int WG_BM = 128, WG_BN = 128, WG_BK = 32, WG_LD = 144;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 tr_frag(const __bf16 *tile, int row0, int col0)
{
    // this lane supplies the address of row (row0 + (i >> 2)), columns col0 + 4 (i & 3), i = lane % 16
    const int i = threadIdx.x & 15;
    const __bf16 *p0 = tile + (row0 + (i >> 2)) * WG_LD + col0 + 4 * (i & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p0 + 16 * WG_LD));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ __launch_bounds__(BLOCK) void wgrad_kernel(const cat_dense_wgrad_args a)
{
    __shared__ __attribute__((aligned(16))) __bf16 sa[2][WG_BK * WG_LD], sb[2][WG_BK * WG_LD];
    // column tiles: those of b, then those of the optional second input b1 (its own width, slabs and row stride)
    const int tiles_n0 = (a.N + WG_BN - 1) / WG_BN, tiles_n = tiles_n0 + (a.b1 ? (a.N1 + WG_BN - 1) / WG_BN : 0);
    const int tn = blockIdx.x % tiles_n;
    const bool second = tn >= tiles_n0;
    const int N = second ? a.N1 : a.N;
    const int m0 = (blockIdx.x / tiles_n) * WG_BM, n0 = (second ? tn - tiles_n0 : tn) * WG_BN, split = blockIdx.y, g = blockIdx.z;
    const int rows_per = ((a.K + a.splits - 1) / a.splits + WG_BK - 1) / WG_BK * WG_BK;
    const int k0 = split * rows_per, k1 = min(a.K, k0 + rows_per);
    const int lr = threadIdx.x >> 4, lc = threadIdx.x & 15;
    const __bf16 *A = (const __bf16 *)a.a + (size_t)g * a.K * a.M, *B = (const __bf16 *)(second ? a.b1 : a.b) + (size_t)g * a.K * N;
    const bf16x8 z8 = {};
    bf16x8 ra[2], rb[2];
    const bool vec_a = a.M % 8 == 0, vec_b = N % 8 == 0;       // rows of 16-byte runs; else (tiny heads) element by element
    auto row_run = [&](const __bf16 *base, int width, int col, bool vec) -> bf16x8 {
        if (vec) return col < width ? *(const bf16x8 *)(base + col) : z8;
        bf16x8 v = z8;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (col + j < width) v[j] = base[col + j];
        return v;
    };
    auto gload = [&](int kb) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = kb + lr + 16 * h;
            ra[h] = row < k1 ? row_run(A + (size_t)row * a.M, a.M, m0 + 8 * lc, vec_a) : z8;
            rb[h] = row < k1 ? row_run(B + (size_t)row * N, N, n0 + 8 * lc, vec_b) : z8;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *(bf16x8 *)&sa[buf][(lr + 16 * h) * WG_LD + 8 * lc] = ra[h];
            *(bf16x8 *)&sb[buf][(lr + 16 * h) * WG_LD + 8 * lc] = rb[h];
        }
    };
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, q = l >> 4, r = l & 15, wm = w >> 1, wn = w & 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = (k1 - k0 + WG_BK - 1) / WG_BK;        // <= 0 for an empty split: the slab is written as zeros
    if (nk > 0) {
        gload(k0);
        lstore(0);
    }
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
        const int buf = it & 1;
        if (it + 1 < nk) gload(k0 + WG_BK * (it + 1));
        bf16x8 af[4], bf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            af[t] = tr_frag(sa[buf], 4 * q, wm * 64 + 16 * t);
            bf[t] = tr_frag(sb[buf], 4 * q, wn * 64 + 16 * t);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        if (it + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    float *out = (second ? a.partial1 : a.partial) + ((size_t)g * a.splits + split) * a.M * N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m0 + wm * 64 + 16 * i + 4 * q + e;
            if (m < a.M)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 64 + 16 * j + r;
                    if (n < N) out[(size_t)m * N + n] = acc[i][j][e];
                }
        }
}

// ---- forward product with the bias / activation epilogue, and the input gradient ------------------------------------------
// Both: 128 x 128 output tile per workgroup, 4 waves of 64 x 64, 32-deep steps through double-buffered LDS, computed
// transposed (the weights are the A operand) so that a lane's four accumulator registers are four consecutive output
// columns of one row: 8-byte stores.  Forward reduces over k, which is the contiguous index of x AND of w: plain 16-byte
// fragment reads (80-byte LDS rows: 16 lanes on 16 rows cover all banks).  The input gradient reduces over n, the ROW
// index of w: its fragments come through the transposed LDS read as in wgrad_kernel, the gradient's through two 8-byte
// reads that follow the same k permutation.
constexpr int GM_LD = 40;         // LDS row of a [128][32] tile, in elements (80 B)

__device__ __forceinline__ bf16x8 run8(const __bf16 *base, int width, int col, bool vec)
{
    const bf16x8 z8 = {};
    if (vec) return col < width ? *(const bf16x8 *)(base + col) : z8;
    bf16x8 v = z8;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (col + j < width) v[j] = base[col + j];
    return v;
}

__global__ __launch_bounds__(BLOCK) void dense_fwd_kernel(const cat_dense_gemm_args a)
{
    __shared__ __attribute__((aligned(16))) __bf16 sx[2][WG_BM * GM_LD], sw[2][WG_BN * GM_LD];
    const int tiles_n = (a.N + WG_BN - 1) / WG_BN;
    const int m0 = (blockIdx.x / tiles_n) * WG_BM, n0 = (blockIdx.x % tiles_n) * WG_BN, g = blockIdx.z;
    const __bf16 *X = (const __bf16 *)a.x_or_gr + (size_t)g * a.M * a.K, *W = (const __bf16 *)a.w + (size_t)g * a.sw_g;
    const int lrow = threadIdx.x >> 2, lch = threadIdx.x & 3;          // thread: rows lrow, lrow + 64; 16-byte chunk lch of the 32 k
    const bf16x8 z8 = {};
    bf16x8 rx[2], rw[2];
    auto gload = [&](int kb) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + lrow + 64 * h, n = n0 + lrow + 64 * h, k = kb + 8 * lch;
            rx[h] = (m < a.M && k < a.K) ? *(const bf16x8 *)(X + (size_t)m * a.K + k) : z8;
            rw[h] = (n < a.N && k < a.K) ? *(const bf16x8 *)(W + (size_t)n * a.K + k) : z8;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *(bf16x8 *)&sx[buf][(lrow + 64 * h) * GM_LD + 8 * lch] = rx[h];
            *(bf16x8 *)&sw[buf][(lrow + 64 * h) * GM_LD + 8 * lch] = rw[h];
        }
    };
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, q = l >> 4, r = l & 15, wn = w >> 1, wm = w & 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = (a.K + WG_BK - 1) / WG_BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
        const int buf = it & 1;
        if (it + 1 < nk) gload(WG_BK * (it + 1));
        bf16x8 wf[4], xf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            wf[t] = *(const bf16x8 *)&sw[buf][(wn * 64 + 16 * t + r) * GM_LD + 8 * q];
            xf[t] = *(const bf16x8 *)&sx[buf][(wm * 64 + 16 * t + r) * GM_LD + 8 * q];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        if (it + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    const __bf16 *bg = a.bias ? (const __bf16 *)a.bias + (size_t)g * a.sb_g : nullptr;
    __bf16 *Y = (__bf16 *)a.out + (size_t)g * a.M * a.N;
    const bool vec_n = a.N % 4 == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + 16 * i + 4 * q;
        float b[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = (bg && n + e < a.N) ? (float)bg[n + e] : 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + 16 * j + r;
            if (m >= a.M || n >= a.N) continue;
            f32x4 v = acc[i][j];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e] + b[e], a.act);
            __bf16 *dst = Y + (size_t)m * a.N + n;
            if (vec_n) *(bf16x4 *)dst = __builtin_convertvector(v, bf16x4);       // n + 3 < N follows from N % 4 == 0
            else
                for (int e = 0; e < 4 && n + e < a.N; ++e) dst[e] = (__bf16)v[e];
        }
    }
}

__global__ __launch_bounds__(BLOCK) void dense_dgrad_kernel(const cat_dense_gemm_args a)
{
    __shared__ __attribute__((aligned(16))) __bf16 sg[2][WG_BM * GM_LD], swt[2][WG_BK * WG_LD];
    const int tiles_k = (a.K + 127) / 128;
    const int m0 = (blockIdx.x / tiles_k) * WG_BM, k0 = (blockIdx.x % tiles_k) * 128, g = blockIdx.z;
    const __bf16 *GR = (const __bf16 *)a.x_or_gr + (size_t)g * a.M * a.N, *W = (const __bf16 *)a.w + (size_t)g * a.sw_g;
    const int lrow = threadIdx.x >> 2, lch = threadIdx.x & 3;          // gradient tile [128 m][32 n]
    const int wrow = threadIdx.x >> 4, wch = threadIdx.x & 15;         // weight tile [32 n][128 k]: rows wrow, wrow + 16
    const bool vec_g = a.N % 8 == 0;
    const bf16x8 z8 = {};
    bf16x8 rg[2], rw[2];
    auto gload = [&](int nb) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + lrow + 64 * h;
            rg[h] = m < a.M ? run8(GR + (size_t)m * a.N, a.N, nb + 8 * lch, vec_g) : z8;
            const int n = nb + wrow + 16 * h, k = k0 + 8 * wch;
            rw[h] = (n < a.N && k < a.K) ? *(const bf16x8 *)(W + (size_t)n * a.K + k) : z8;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *(bf16x8 *)&sg[buf][(lrow + 64 * h) * GM_LD + 8 * lch] = rg[h];
            *(bf16x8 *)&swt[buf][(wrow + 16 * h) * WG_LD + 8 * wch] = rw[h];
        }
    };
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, q = l >> 4, r = l & 15, wk = w >> 1, wm = w & 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nn = (a.N + WG_BK - 1) / WG_BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int it = 0; it < nn; ++it) {
        const int buf = it & 1;
        if (it + 1 < nn) gload(WG_BK * (it + 1));
        bf16x8 wf[4], gf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            wf[t] = tr_frag(swt[buf], 4 * q, wk * 64 + 16 * t);                       // A[row = k-out][n in {4q.., 16+4q..}]
            const __bf16 *row = &sg[buf][(wm * 64 + 16 * t + r) * GM_LD];
            const bf16x4 lo = *(const bf16x4 *)(row + 4 * q), hi = *(const bf16x4 *)(row + 16 + 4 * q);
            gf[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);          // B[the same n's][col = m]
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], gf[j], acc[i][j], 0, 0, 0);
        if (it + 1 < nn) lstore(buf ^ 1);
        __syncthreads();
    }
    __bf16 *DX = (__bf16 *)a.out + (size_t)g * a.M * a.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + wk * 64 + 16 * i + 4 * q;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + 16 * j + r;
            if (m < a.M && k < a.K) *(bf16x4 *)(DX + (size_t)m * a.K + k) = __builtin_convertvector(acc[i][j], bf16x4);   // K % 8 == 0
        }
    }
}

thread_local char g_err[256] = "";
int fail(int code, const char *msg)
{
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}
bool dims_ok(const cat_dense_dims *d)
{
    return d && d->G > 0 && d->G <= 65535 && d->M > 0 && (d->out == 1 || (d->out % 4 == 0 && d->out > 0 && d->out <= CAT_DENSE_MAX_OUT)) &&
           d->act >= CAT_ACT_NONE && d->act <= CAT_ACT_TANH;
}

}   // namespace

extern "C" int cat_dense_abi_version(void) { return CAT_DENSE_ABI_VERSION; }
extern "C" const char *cat_dense_last_error(void) { return g_err; }

extern "C" int cat_dense_bias_act(const cat_dense_dims *d, void *y, const void *bias, int64_t sb_g, void *stream)
{
    if (!dims_ok(d) || !y || !bias) return fail(CAT_DENSE_ERR_BAD_ARG, "cat_dense_bias_act: bad dimensions or NULL buffer");
    if ((uintptr_t)y % 8) return fail(CAT_DENSE_ERR_BAD_ARG, "cat_dense_bias_act: y must be 8-byte aligned");
    const size_t work = ((size_t)d->M * d->out + 3) / 4;
    size_t blocks = (work + BLOCK - 1) / BLOCK;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bias_act_kernel, dim3((unsigned)blocks, d->G), dim3(BLOCK), 0, (hipStream_t)stream, *d, (__bf16 *)y, (const __bf16 *)bias, sb_g);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_DENSE_OK : fail(CAT_DENSE_ERR_HIP, hipGetErrorString(e));
}

extern "C" int cat_dense_act_grad(const cat_dense_dims *d, const void *d_y, const void *y, void *g_out, float *partial, int32_t chunks,
                                  void *stream)
{
    if (!dims_ok(d) || !d_y || !partial || chunks <= 0 || chunks > CAT_DENSE_MAX_CHUNKS)
        return fail(CAT_DENSE_ERR_BAD_ARG, "cat_dense_act_grad: bad dimensions or NULL buffer");
    if (d->act != CAT_ACT_NONE && (!y || !g_out)) return fail(CAT_DENSE_ERR_BAD_ARG, "cat_dense_act_grad: y and g_out are required with an activation");
    if (((uintptr_t)d_y % 8) || ((uintptr_t)y % 8) || ((uintptr_t)g_out % 8)) return fail(CAT_DENSE_ERR_BAD_ARG, "cat_dense_act_grad: misaligned buffer");
    const int groups = d->out % 4 == 0 ? d->out / 4 : d->out;
    const int gy = (groups + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(act_grad_kernel, dim3(chunks, gy, d->G), dim3(BLOCK), 0, (hipStream_t)stream, *d, (const __bf16 *)d_y,
                       (const __bf16 *)y, (__bf16 *)g_out, partial);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_DENSE_OK : fail(CAT_DENSE_ERR_HIP, hipGetErrorString(e));
}

extern "C" int cat_dense_sum_chunks(const float *partial, int32_t G, int32_t chunks, int32_t n, void *dst0, int64_t sd0_g, void *dst1,
                                    int64_t sd1_g, int32_t accumulate, void *stream)
{
    cat_dense_sum_job j = {partial, chunks, n, dst0, sd0_g, dst1, sd1_g, accumulate, 0};
    return cat_dense_sum_chunks2(&j, nullptr, G, stream);
}

extern "C" int cat_dense_sum_chunks2(const cat_dense_sum_job *a, const cat_dense_sum_job *b, int32_t G, void *stream)
{
    if (!a || !a->partial || !a->dst0 || G <= 0 || G > 65535 || a->chunks <= 0 || a->n <= 0 ||
        (b && (!b->partial || !b->dst0 || b->chunks <= 0 || b->n <= 0)))
        return fail(CAT_DENSE_ERR_BAD_ARG, "cat_dense_sum_chunks: bad dimensions or NULL buffer");
    SumJob j0 = {a->partial, a->chunks, a->n, (__bf16 *)a->dst0, (__bf16 *)a->dst1, a->sd0_g, a->sd1_g, a->accumulate, (a->n + 31) / 32};
    SumJob j1 = j0;
    j1.blocks = 0;
    if (b) j1 = SumJob{b->partial, b->chunks, b->n, (__bf16 *)b->dst0, (__bf16 *)b->dst1, b->sd0_g, b->sd1_g, b->accumulate, (b->n + 31) / 32};
    hipLaunchKernelGGL(sum_chunks_kernel, dim3(j0.blocks + (b ? j1.blocks : 0), G), dim3(BLOCK), 0, (hipStream_t)stream, j0, j1);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_DENSE_OK : fail(CAT_DENSE_ERR_HIP, hipGetErrorString(e));
}

extern "C" int cat_dense_wgrad_splits(int32_t G, int32_t K, int32_t M, int32_t N)
{
    if (G <= 0 || K <= 0 || M <= 0 || N <= 0) return CAT_DENSE_ERR_BAD_ARG;
    const int tiles = G * ((M + WG_BM - 1) / WG_BM) * ((N + WG_BN - 1) / WG_BN);
    int s = (512 + tiles - 1) / tiles;                     // about two workgroups per CU
    const int by_rows = K / 128 > 0 ? K / 128 : 1;         // at least four 32-row steps per split
    if (s > by_rows) s = by_rows;
    if (s > CAT_DENSE_MAX_CHUNKS) s = CAT_DENSE_MAX_CHUNKS;
    return s < 1 ? 1 : s;
}

extern "C" int cat_dense_wgrad(const cat_dense_wgrad_args *a, void *stream)
{
    if (!a || a->G <= 0 || a->G > 65535 || a->K <= 0 || a->M <= 0 || a->N <= 0 || a->splits <= 0 || a->splits > CAT_DENSE_MAX_CHUNKS)
        return fail(CAT_DENSE_ERR_BAD_ARG, "cat_dense_wgrad: bad dimensions");
    if (!a->a || !a->b || !a->partial || ((a->M % 8 == 0) && ((uintptr_t)a->a % 16)) || ((a->N % 8 == 0) && ((uintptr_t)a->b % 16)))
        return fail(CAT_DENSE_ERR_BAD_ARG, "cat_dense_wgrad: NULL or misaligned buffer");
    if (a->b1 && (a->N1 <= 0 || !a->partial1 || ((a->N1 % 8 == 0) && ((uintptr_t)a->b1 % 16))))
        return fail(CAT_DENSE_ERR_BAD_ARG, "cat_dense_wgrad: the second input needs N1 > 0, its slabs and an aligned buffer");
    const int tiles = ((a->M + WG_BM - 1) / WG_BM) * ((a->N + WG_BN - 1) / WG_BN + (a->b1 ? (a->N1 + WG_BN - 1) / WG_BN : 0));
    hipLaunchKernelGGL(wgrad_kernel, dim3(tiles, a->splits, a->G), dim3(BLOCK), 0, (hipStream_t)stream, *a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_DENSE_OK : fail(CAT_DENSE_ERR_HIP, hipGetErrorString(e));
}

static int gemm_args_ok(const cat_dense_gemm_args *a)
{
    return a && a->G > 0 && a->G <= 65535 && a->M > 0 && a->N > 0 && a->K > 0 && a->K % 8 == 0 && a->x_or_gr && a->w && a->out &&
           !((uintptr_t)a->w % 16) && !(a->sw_g % 8) && !((uintptr_t)a->out % 16);
}

extern "C" int cat_dense_forward(const cat_dense_gemm_args *a, void *stream)
{
    if (!gemm_args_ok(a) || ((uintptr_t)a->x_or_gr % 16) || a->act < CAT_ACT_NONE || a->act > CAT_ACT_TANH)
        return fail(CAT_DENSE_ERR_BAD_ARG, "cat_dense_forward: bad dimensions (K % 8), NULL or misaligned buffer");
    const int tiles = ((a->M + WG_BM - 1) / WG_BM) * ((a->N + WG_BN - 1) / WG_BN);
    hipLaunchKernelGGL(dense_fwd_kernel, dim3(tiles, 1, a->G), dim3(BLOCK), 0, (hipStream_t)stream, *a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_DENSE_OK : fail(CAT_DENSE_ERR_HIP, hipGetErrorString(e));
}

extern "C" int cat_dense_dgrad(const cat_dense_gemm_args *a, void *stream)
{
    if (!gemm_args_ok(a) || ((a->N % 8 == 0) && ((uintptr_t)a->x_or_gr % 16)))
        return fail(CAT_DENSE_ERR_BAD_ARG, "cat_dense_dgrad: bad dimensions (K % 8), NULL or misaligned buffer");
    const int tiles = ((a->M + WG_BM - 1) / WG_BM) * ((a->K + 127) / 128);
    hipLaunchKernelGGL(dense_dgrad_kernel, dim3(tiles, 1, a->G), dim3(BLOCK), 0, (hipStream_t)stream, *a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_DENSE_OK : fail(CAT_DENSE_ERR_HIP, hipGetErrorString(e));
}
