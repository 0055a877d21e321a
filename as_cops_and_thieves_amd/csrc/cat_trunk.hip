// cat_trunk.hip -- libcat_learn.so, part 2: Conv1d(C,64,5,s2) -> ReLU -> Conv1d(64,32,5,s3) -> ReLU of the learner's
// role-stacked networks, forward and backward, on MI355X (gfx950).  include/cat_trunk.h has the interface.
//
// Why a kernel: as library GEMMs the two convolutions are either im2col copies of hundreds of MB or dense GEMMs on the
// Toeplitz expansion of the weights (6-13x the flops), and either way the 1920-wide intermediate of every sample
// crosses HBM about ten times per training step (GEMM out, bias, ReLU, GEMM in; mask, dgrad out, wgrad in, bias grad).
// Here a workgroup takes 16 samples at a time and the intermediate exists in LDS only:
//
//   * every product is computed TRANSPOSED on v_mfma_f32_16x16x32_bf16 with the weights as the A operand, held in
//     registers for the life of the (persistent) workgroup, and the 16 samples on the accumulator's lanes, so a lane's
//     four accumulator registers are four consecutive channels of one sample;
//   * with the input re-laid (ray, channel) and the intermediate (position, channel), the window of either convolution
//     is a CONTIGUOUS run of the row: the B operand of a window is plain 8/16-byte LDS reads, no im2col;
//   * backward recomputes the intermediate (cheaper than a 1920-wide round trip), computes its gradient per POSITION
//     (each position collects the one or two (output position, tap) pairs that cover it, so nothing is scattered), and
//     forms the weight gradients as products whose inner index is (sample, position): the operands it needs sample-
//     contiguous are written to LDS in [column][sample] order by the passes that produce them.  Weight and bias
//     gradients accumulate in registers over all tiles of the workgroup and leave as one fp32 slab per workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "cat_trunk.h"

namespace {

constexpr int C1 = CAT_TRUNK_C1, C2 = CAT_TRUNK_C2, KW = 5, TS = CAT_TRUNK_TILE, NW = 4, LANES = 64;
constexpr int WIN2 = KW * C1;                 // 320: the second convolution's window in the (position, channel) row
constexpr int LDS_LIMIT = 160 * 1024;

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ bf16x4 narrow(f32x4 v) { return __builtin_convertvector(v, bf16x4); }
__device__ __forceinline__ f32x4 relu4(f32x4 v)
{
    return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
}
__device__ __forceinline__ bf16x8 ld8(const __bf16 *p)   // 8-byte aligned
{
    const bf16x4 a = *(const bf16x4 *)p, b = *(const bf16x4 *)(p + 4);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 splat8(float v)
{
    const __bf16 b = (__bf16)v;
    return bf16x8{b, b, b, b, b, b, b, b};
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

struct Geo {
    int C, R, CR, L1, L2, XS, A1S, D2, D2S;
    __device__ __host__ explicit Geo(const cat_trunk_dims &d)
    {
        C = d.C; R = d.R; CR = C * R;
        L1 = (R - KW) / 2 + 1; L2 = (L1 - KW) / 3 + 1;
        XS = (CR + 40 + 7) & ~7;      // input row (ray, channel) + zeroed tail (the last windows read up to 31 past 5 C), 16-byte rows
        A1S = L1 * C1 + 8;            // forward's intermediate row (position, channel), 16-byte aligned, bank-skewed
        D2 = L2 * C2; D2S = D2 + 8;
    }
    __device__ __host__ size_t fwd_lds() const { return 2 * ((size_t)TS * XS + (size_t)TS * A1S); }
    __device__ __host__ size_t bwd_lds() const
    {
        const size_t tiles = 2 * ((size_t)TS * XS * 2 + (size_t)L1 * C1 * TS + (size_t)TS * D2S + (size_t)D2 * TS);
        const size_t reduce = (size_t)CAT_TRUNK_BWD_WAVES * C1 * 32 * sizeof(float);   // the waves' dW1 sums at the end of the kernel
        return tiles > reduce ? tiles : reduce;
    }
};

// A fragments of the first convolution: A[row = channel 16 mt + r][k = kk * C + c] = w1[channel][c][kk], 0 for k >= 5 C
__device__ __forceinline__ void load_w1(const cat_trunk_params &p, int g, int C, int q, int r, bf16x8 (&w1f)[4], f32x4 (&bias1)[4])
{
    const __bf16 *w1 = (const __bf16 *)p.w1 + (size_t)g * p.sw1_g, *b1 = (const __bf16 *)p.b1 + (size_t)g * p.sb1_g;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * q + j, kk = k / C, c = k % C;
            w1f[mt][j] = (k < KW * C) ? w1[((16 * mt + r) * C + c) * KW + kk] : (__bf16)0.0f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) bias1[mt][e] = (float)b1[16 * mt + 4 * q + e];
    }
}

// four rays of one channel; the last run of a row whose ray count is not a multiple of 4 is read ray by ray
__device__ __forceinline__ bf16x4 load_rays(const __bf16 *chan, int ray0, int R)
{
    if (ray0 + 4 <= R) return *(const bf16x4 *)(chan + ray0);
    bf16x4 v = narrow(f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (ray0 + i < R) v[i] = chan[ray0 + i];
    return v;
}

// input tile -> LDS in (ray, channel) order (and optionally [column][sample]).  Thread (sample s = t / 16, u = t % 16)
// takes runs of 4 rays: one 8-byte load per channel, interleaved in registers, written as one run of 4 C elements.
// Rows past N read as zero; the row tails (columns >= C R) are zeroed once per launch by clear_tails.
// the row of x that sample n reads (cat_trunk_rows): a minibatch picked out of the rollout buffer, step-major
__device__ __forceinline__ size_t source_row(const cat_trunk_rows &m, int n)
{
    if (!m.rows) return (size_t)n;
    const int t = n / m.sel;
    return (size_t)t * m.block + (size_t)m.rows[n - t * m.sel];
}

__device__ __forceinline__ void stage_x(const __bf16 *xg, int64_t sx_n, const cat_trunk_rows &xr, int n0, int N, const Geo &ge, __bf16 *xs,
                                        __bf16 *xT)
{
    const int s = threadIdx.x >> 4, u = threadIdx.x & 15;
    const bool ok = n0 + s < N;
    const __bf16 *row = xg + (ok ? source_row(xr, n0 + s) : 0) * sx_n;
    const bf16x4 z4 = narrow(f32x4{0.f, 0.f, 0.f, 0.f});
    for (int ch = u; 4 * ch < ge.R; ch += 16) {
        bf16x4 v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = (ok && c < ge.C) ? load_rays(row + c * ge.R, 4 * ch, ge.R) : z4;
        __bf16 *dst = xs + s * ge.XS + 4 * ch * ge.C;
        if (ge.C == 2) {
            *(bf16x8 *)dst = bf16x8{v[0][0], v[1][0], v[0][1], v[1][1], v[0][2], v[1][2], v[0][3], v[1][3]};
        } else {
            *(bf16x8 *)dst = bf16x8{v[0][0], v[1][0], v[2][0], v[3][0], v[0][1], v[1][1], v[2][1], v[3][1]};
            *(bf16x8 *)(dst + 8) = bf16x8{v[0][2], v[1][2], v[2][2], v[3][2], v[0][3], v[1][3], v[2][3], v[3][3]};
        }
        if (xT) {
            __bf16 *t = xT + (4 * ch * ge.C) * TS + s;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < ge.C) t[(i * ge.C + c) * TS] = v[c][i];
        }
    }
}

__device__ __forceinline__ void clear_tails(const Geo &ge, __bf16 *xs, __bf16 *xT)
{
    const int tail = ge.XS - ge.CR;
    for (int e = threadIdx.x; e < TS * tail; e += NW * LANES) {
        const int s = e / tail, j = ge.CR + e - s * tail;
        xs[s * ge.XS + j] = (__bf16)0.0f;
        if (xT) xT[j * TS + s] = (__bf16)0.0f;
    }
}

__global__ __launch_bounds__(NW *LANES) void trunk_fwd_kernel(const cat_trunk_fwd a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Geo ge(a.d);
    __bf16 *xs = (__bf16 *)smem, *a1 = xs + TS * ge.XS;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, q = l >> 4, r = l & 15;
    const int g = blockIdx.y, N = a.d.N, ntiles = (N + TS - 1) / TS;

    bf16x8 w1f[4], w2f[2][10];
    f32x4 bias1[4], bias2[2];
    load_w1(a.p, g, ge.C, q, r, w1f, bias1);
    {   // A[row = out channel 16 mt + r][k = kk * 64 + ci] = w2[out][ci][kk]
        const __bf16 *w2 = (const __bf16 *)a.p.w2 + (size_t)g * a.p.sw2_g, *b2 = (const __bf16 *)a.p.b2 + (size_t)g * a.p.sb2_g;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int ks = 0; ks < 10; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 32 * ks + 8 * q + j, kk = k >> 6, ci = k & 63;
                    w2f[mt][ks][j] = w2[((16 * mt + r) * C1 + ci) * KW + kk];
                }
#pragma unroll
            for (int e = 0; e < 4; ++e) bias2[mt][e] = (float)b2[16 * mt + 4 * q + e];
        }
    }
    const __bf16 *xg = (const __bf16 *)a.x + (size_t)g * a.sx_g;
    __bf16 *og = (__bf16 *)a.out + (size_t)g * a.so_g;
    clear_tails(ge, xs, nullptr);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n0 = tile * TS;
        stage_x(xg, a.sx_n, a.x_rows, n0, N, ge, xs, nullptr);
        __syncthreads();
        for (int p = w; p < ge.L1; p += NW) {
            const bf16x8 xb = ld8(xs + r * ge.XS + 2 * p * ge.C + 8 * q);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                *(bf16x4 *)(a1 + r * ge.A1S + p * C1 + 16 * mt + 4 * q) = narrow(relu4(MFMA(w1f[mt], xb, bias1[mt])));
        }
        __syncthreads();
        for (int l2 = w; l2 < ge.L2; l2 += NW) {
            f32x4 acc[2] = {bias2[0], bias2[1]};
#pragma unroll
            for (int ks = 0; ks < 10; ++ks) {
                const bf16x8 ab = *(const bf16x8 *)(a1 + r * ge.A1S + 3 * C1 * l2 + 32 * ks + 8 * q);
                acc[0] = MFMA(w2f[0][ks], ab, acc[0]);
                acc[1] = MFMA(w2f[1][ks], ab, acc[1]);
            }
            if (n0 + r < N) {
                __bf16 *o = og + (size_t)(n0 + r) * a.so_n + l2 * C2 + 4 * q;
                *(bf16x4 *)(o) = narrow(relu4(acc[0]));
                *(bf16x4 *)(o + 16) = narrow(relu4(acc[1]));
            }
        }
        __syncthreads();
    }
}

// ---- backward: 8 waves (two per SIMD: the workgroup's LDS fills the CU, so latency is hidden inside the workgroup) ----
constexpr int NWB = CAT_TRUNK_BWD_WAVES;

struct Raw {            // one tile's global data in flight: fetched a tile ahead, committed to LDS at the top of the loop
    bf16x4 x[4];        // thread (sample t / 32, u = t % 32): rays 4 u .. 4 u + 3 of every channel
    bf16x4 y[4], d[4];  // runs 4 (u + 32 i) .. of the layer's output and of its gradient
};

__device__ __forceinline__ void fetch_raw(Raw &rw, const __bf16 *xg, int64_t sx_n, const cat_trunk_rows &xr, const __bf16 *og,
                                          const __bf16 *dg, int64_t so_n, int n0, int N, const Geo &ge)
{
    const int s = threadIdx.x >> 5, u = threadIdx.x & 31;
    const bool ok = n0 + s < N;
    const bf16x4 z4 = narrow(f32x4{0.f, 0.f, 0.f, 0.f});
    const __bf16 *row = xg + (ok ? source_row(xr, n0 + s) : 0) * sx_n;
#pragma unroll
    for (int c = 0; c < 4; ++c) rw.x[c] = (ok && c < ge.C && 4 * u < ge.R) ? load_rays(row + c * ge.R, 4 * u, ge.R) : z4;
    const size_t o = (size_t)(n0 + s) * so_n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = u + 32 * i;
        const bool in = ok && 4 * ch < ge.D2;
        rw.y[i] = in ? *(const bf16x4 *)(og + o + 4 * ch) : z4;
        rw.d[i] = in ? *(const bf16x4 *)(dg + o + 4 * ch) : z4;
    }
}

__device__ __forceinline__ void commit_raw(const Raw &rw, const Geo &ge, __bf16 *xs, __bf16 *xT, __bf16 *dp2, __bf16 *dp2T)
{
    const int s = threadIdx.x >> 5, u = threadIdx.x & 31;
    if (4 * u < ge.R) {
        const bf16x4 *v = rw.x;
        __bf16 *dst = xs + s * ge.XS + 4 * u * ge.C;
        if (ge.C == 2) {
            *(bf16x8 *)dst = bf16x8{v[0][0], v[1][0], v[0][1], v[1][1], v[0][2], v[1][2], v[0][3], v[1][3]};
        } else {
            *(bf16x8 *)dst = bf16x8{v[0][0], v[1][0], v[2][0], v[3][0], v[0][1], v[1][1], v[2][1], v[3][1]};
            *(bf16x8 *)(dst + 8) = bf16x8{v[0][2], v[1][2], v[2][2], v[3][2], v[0][3], v[1][3], v[2][3], v[3][3]};
        }
        __bf16 *t = xT + (4 * u * ge.C) * TS + s;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < ge.C) t[(i * ge.C + c) * TS] = v[c][i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = u + 32 * i;
        if (4 * ch < ge.D2) {                       // gradient before the second ReLU, row-major and sample-contiguous
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((float)rw.y[i][e] > 0.0f) ? rw.d[i][e] : (__bf16)0.0f;
            *(bf16x4 *)(dp2 + s * ge.D2S + 4 * ch) = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) dp2T[(4 * ch + e) * TS + s] = v[e];
        }
    }
}

__global__ __launch_bounds__(NWB *LANES) void trunk_bwd_kernel(const cat_trunk_bwd a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Geo ge(a.d);
    __bf16 *xs = (__bf16 *)smem;                    // [16][XS]          input rows, (ray, channel)
    __bf16 *xT = xs + TS * ge.XS;                   // [XS][16]          the same, sample-contiguous
    __bf16 *a1T = xT + TS * ge.XS;                  // [L1 * 64][16]     intermediate after ReLU, sample-contiguous
    __bf16 *dp2 = a1T + ge.L1 * C1 * TS;            // [16][D2S]         gradient before the second ReLU
    __bf16 *dp2T = dp2 + TS * ge.D2S;               // [L2 * 32][16]     the same, sample-contiguous
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, q = l >> 4, r = l & 15;
    const int g = blockIdx.y, N = a.d.N, ntiles = (N + TS - 1) / TS;
    const int L1 = ge.L1, L2 = ge.L2;

    // The intermediate and its gradient are computed with the SAMPLES on the accumulator's rows here (operands swapped
    // w.r.t. forward): a lane then holds four consecutive samples of one channel = one 8-byte run of the [column][sample]
    // images the weight-gradient products read.  The bias becomes a per-lane (per-column) constant.
    bf16x8 w1f[4], w2t[KW][4];
    f32x4 bias1n[4];
    {
        f32x4 unused[4];
        load_w1(a.p, g, ge.C, q, r, w1f, unused);
        const __bf16 *b1 = (const __bf16 *)a.p.b1 + (size_t)g * a.p.sb1_g;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const float b = (float)b1[16 * mt + r];
            bias1n[mt] = f32x4{b, b, b, b};
        }
        // B[k = out channel 8 q + j][n = ci 16 mt + r] = w2[out][ci][kk]
        const __bf16 *w2 = (const __bf16 *)a.p.w2 + (size_t)g * a.p.sw2_g;
#pragma unroll
        for (int kk = 0; kk < KW; ++kk)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int j = 0; j < 8; ++j) w2t[kk][mt][j] = w2[((8 * q + j) * C1 + 16 * mt + r) * KW + kk];
    }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const bf16x8 zero8 = splat8(0.0f), ones8 = splat8(1.0f);
    // weight-gradient tiles of this wave: dW2 row tile (w & 1), column tiles (w >> 1) + 4 i.  dW1 (and db1) is accumulated
    // by every wave over ITS positions, all 4 x 2 tiles, straight from the accumulators that hold the gradient of the
    // intermediate (below); the waves' sums are added up at the end of the kernel.
    const int mt2 = w & 1, ct2 = w >> 1;
    f32x4 dw2[5] = {zero, zero, zero, zero, zero}, db2 = zero, dw1[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) dw1[mt][0] = dw1[mt][1] = zero;
    // window column 31 is never a real tap (5 C <= 20): it carries ones, so column 31 of dW1 is the bias gradient
    const s16x4 ones4 = {0x3F80, 0x3F80, 0x3F80, 0x3F80};       // bf16 1.0

    const __bf16 *xg = (const __bf16 *)a.x + (size_t)g * a.sx_g;
    const __bf16 *og = (const __bf16 *)a.out + (size_t)g * a.so_g, *dg = (const __bf16 *)a.d_out + (size_t)g * a.so_g;
    const int sh = 8 * (q & 1), half = q >> 1;      // a K step of the weight-gradient products = 2 positions x 16 samples
    for (int e = threadIdx.x; e < TS * (ge.XS - ge.CR); e += NWB * LANES) {      // the zero tails, once
        const int tail = ge.XS - ge.CR, s = e / tail, j = ge.CR + e - s * tail;
        xs[s * ge.XS + j] = (__bf16)0.0f;
        xT[j * TS + s] = (__bf16)0.0f;
    }
    Raw raw;
    fetch_raw(raw, xg, a.sx_n, a.x_rows, og, dg, a.so_n, blockIdx.x * TS, N, ge);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        commit_raw(raw, ge, xs, xT, dp2, dp2T);
        fetch_raw(raw, xg, a.sx_n, a.x_rows, og, dg, a.so_n, (tile + (int)gridDim.x) * TS, N, ge);      // past the end: zeros, unused
        __syncthreads();
        for (int p = w; p < L1; p += NWB) {          // the intermediate again: D[sample 4 q + e][channel 16 mt + r]
            const bf16x8 xb = ld8(xs + r * ge.XS + 2 * p * ge.C + 8 * q);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                *(bf16x4 *)(a1T + (p * C1 + 16 * mt + r) * TS + 4 * q) = narrow(relu4(MFMA(xb, w1f[mt], bias1n[mt])));
        }
        __syncthreads();
        for (int p = w; p < L1; p += NWB) {          // gradient of position p: the (output position, tap) pairs covering it
            f32x4 acc[4] = {zero, zero, zero, zero};
#pragma unroll
            for (int kk = 0; kk < KW; ++kk) {
                const int t = p - kk, l2 = t / 3;
                if (t >= 0 && t == 3 * l2 && l2 < L2) {
                    const bf16x8 db = *(const bf16x8 *)(dp2 + r * ge.D2S + l2 * C2 + 8 * q);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[mt] = MFMA(db, w2t[kk][mt], acc[mt]);
                }
            }
            // dW1[channel][kk * C + c] += sum over the 16 samples of (gradient before the first ReLU) x (input window of p):
            // the accumulator holds samples 4 q .. 4 q + 3 of channel 16 mt + r -- exactly the A fragment of the K = 16
            // MFMA (v_mfma_f32_16x16x16_bf16), so it goes from the accumulator into the product with no LDS image
            s16x4 xw[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) xw[nt] = *(const s16x4 *)(xT + (2 * p * ge.C + 16 * nt + r) * TS + 4 * q);
            if (r == 15) xw[1] = ones4;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const bf16x4 y = *(const bf16x4 *)(a1T + (p * C1 + 16 * mt + r) * TS + 4 * q);
                bf16x4 v = narrow(acc[mt]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ((float)y[e] > 0.0f) ? v[e] : (__bf16)0.0f;
                const s16x4 va = __builtin_bit_cast(s16x4, v);
                dw1[mt][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va, xw[0], dw1[mt][0], 0, 0, 0);
                dw1[mt][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va, xw[1], dw1[mt][1], 0, 0, 0);
            }
        }
        for (int l0 = 0; l0 < L2; l0 += 2) {         // dW2[out][window column] += sum over (sample, position)
            const int lq = l0 + half;
            const bool ok = lq < L2;
            const bf16x8 af = ok ? *(const bf16x8 *)(dp2T + (lq * C2 + 16 * mt2 + r) * TS + sh) : zero8;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const bf16x8 bf = ok ? *(const bf16x8 *)(a1T + (3 * C1 * lq + 16 * (ct2 + 4 * i) + r) * TS + sh) : zero8;
                dw2[i] = MFMA(af, bf, dw2[i]);
            }
            if (ct2 == 0) db2 = MFMA(af, ones8, db2);
        }
        __syncthreads();
    }

    const size_t slab = (size_t)g * gridDim.x + blockIdx.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < 5; ++i) a.part_dw2[(slab * C2 + 16 * mt2 + 4 * q + e) * WIN2 + 16 * (ct2 + 4 * i) + r] = dw2[i][e];
        if (r == 0 && ct2 == 0) a.part_db2[slab * C2 + 16 * mt2 + 4 * q + e] = db2[e];
    }
    // dW1: the eight waves' [64][32] accumulators are added up through LDS (free by now) into the workgroup's slab
    float *red = (float *)smem;                                            // [8][64][32] fp32 = 64 KB
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[(w * C1 + 16 * mt + 4 * q + e) * 32 + r] = dw1[mt][0][e];
            red[(w * C1 + 16 * mt + 4 * q + e) * 32 + 16 + r] = dw1[mt][1][e];
        }
    __syncthreads();
    for (int j = threadIdx.x; j < C1 * 32; j += NWB * LANES) {
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < NWB; ++k) sum += red[k * C1 * 32 + j];
        a.part_dw1[slab * C1 * 32 + j] = sum;
        if ((j & 31) == 31) a.part_db1[slab * C1 + (j >> 5)] = sum;          // column 31 = the bias gradient
    }
}

// second stage of the parameter gradients: one thread per slab element, summed over the slabs, stored where the parameter
// keeps that element (w1: column kk * C + c -> [ch][c][kk]; w2: column kk * 64 + ci -> [co][ci][kk])
__global__ __launch_bounds__(256) void trunk_finish_kernel(const cat_trunk_finish_args a)
{
    const int g = blockIdx.y, C = a.d.C, B = a.blocks;
    const int n1 = C1 * KW * C, nb1 = C1, n2 = C2 * WIN2, nb2 = C2;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const float *src;
    __bf16 *dst;
    size_t slab;
    if (j < n1) {
        const int ch = j / (KW * C), col = j - ch * (KW * C), kk = col / C, c = col - kk * C;
        src = a.part_dw1 + (size_t)g * B * C1 * 32 + ch * 32 + col; slab = (size_t)C1 * 32;
        dst = (__bf16 *)a.dw1 + (size_t)g * a.sw1_g + (ch * C + c) * KW + kk;
    } else if (j < n1 + nb1) {
        const int ch = j - n1;
        src = a.part_db1 + (size_t)g * B * C1 + ch; slab = C1;
        dst = (__bf16 *)a.db1 + (size_t)g * a.sb1_g + ch;
    } else if (j < n1 + nb1 + n2) {
        const int k = j - n1 - nb1, co = k / WIN2, col = k - co * WIN2, kk = col >> 6, ci = col & 63;
        src = a.part_dw2 + (size_t)g * B * C2 * WIN2 + k; slab = (size_t)C2 * WIN2;
        dst = (__bf16 *)a.dw2 + (size_t)g * a.sw2_g + (co * C1 + ci) * KW + kk;
    } else if (j < n1 + nb1 + n2 + nb2) {
        const int co = j - n1 - nb1 - n2;
        src = a.part_db2 + (size_t)g * B * C2 + co; slab = C2;
        dst = (__bf16 *)a.db2 + (size_t)g * a.sb2_g + co;
    } else {
        return;
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 4 <= B; b += 4) {
        s0 += src[(size_t)b * slab]; s1 += src[(size_t)(b + 1) * slab]; s2 += src[(size_t)(b + 2) * slab]; s3 += src[(size_t)(b + 3) * slab];
    }
    for (; b < B; ++b) s0 += src[(size_t)b * slab];
    const float s = (s0 + s1) + (s2 + s3);
    *dst = (__bf16)(a.accumulate ? s + (float)*dst : s);
}

thread_local char g_err[256] = "";
int fail(int code, const char *msg)
{
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

int g_cus = 0;
int compute_units()
{
    if (g_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_cus = prop.multiProcessorCount;
        if (g_cus <= 0) g_cus = 256;          // MI355X; also what a build host without a device assumes
    }
    return g_cus;
}

bool dims_ok(const cat_trunk_dims &d)
{
    return d.G > 0 && d.G <= 65535 && d.N > 0 && (d.C == 2 || d.C == 4) && d.R >= 20 && d.R <= 512 && d.R % 2 == 0;
}
bool aligned(const void *p, size_t a) { return ((uintptr_t)p % a) == 0; }
bool rows_ok(const cat_trunk_rows &r, int N) { return !r.rows || (r.sel > 0 && r.block >= r.sel && N % r.sel == 0); }
int tiles_of(const cat_trunk_dims &d) { return (d.N + TS - 1) / TS; }
int bwd_blocks(const cat_trunk_dims &d)
{
    const int per_net = compute_units() / d.G;          // backward: one workgroup per CU (its LDS fills the CU)
    const int b = per_net < 1 ? 1 : per_net;
    return b < tiles_of(d) ? b : tiles_of(d);
}
int fwd_blocks(const cat_trunk_dims &d)
{
    const int per_net = 2 * compute_units() / d.G;
    const int b = per_net < 1 ? 1 : per_net;
    return b < tiles_of(d) ? b : tiles_of(d);
}
bool params_ok(const cat_trunk_params &p)
{
    return p.w1 && p.b1 && p.w2 && p.b2 && aligned(p.w1, 2) && aligned(p.b1, 2) && aligned(p.w2, 2) && aligned(p.b2, 2);
}

}   // namespace

extern "C" int cat_trunk_abi_version(void) { return CAT_TRUNK_ABI_VERSION; }
extern "C" const char *cat_trunk_last_error(void) { return g_err; }
extern "C" int cat_trunk_out_positions(const cat_trunk_dims *d) { return d && dims_ok(*d) ? Geo(*d).L2 : CAT_TRUNK_ERR_BAD_ARG; }
extern "C" int cat_trunk_supported(const cat_trunk_dims *d)
{
    return d && dims_ok(*d) && Geo(*d).L2 >= 1 && Geo(*d).bwd_lds() <= (size_t)LDS_LIMIT ? 1 : 0;
}
extern "C" int cat_trunk_backward_blocks(const cat_trunk_dims *d) { return d && dims_ok(*d) ? bwd_blocks(*d) : CAT_TRUNK_ERR_BAD_ARG; }

extern "C" int cat_trunk_forward(const cat_trunk_fwd *a, void *stream)
{
    if (!a || !dims_ok(a->d)) return fail(CAT_TRUNK_ERR_BAD_ARG, "cat_trunk_forward: bad dimensions");
    if (!cat_trunk_supported(&a->d)) return fail(CAT_TRUNK_ERR_TOO_LARGE, "cat_trunk_forward: these dimensions do not fit the LDS");
    if (!params_ok(a->p) || !a->x || !a->out) return fail(CAT_TRUNK_ERR_BAD_ARG, "cat_trunk_forward: a required buffer is NULL or misaligned");
    if (!aligned(a->x, 8) || (a->sx_g % 4) || (a->sx_n % 4) || !aligned(a->out, 8) || (a->so_g % 4) || (a->so_n % 4))
        return fail(CAT_TRUNK_ERR_BAD_ARG, "cat_trunk_forward: misaligned buffer or stride");
    if (!rows_ok(a->x_rows, a->d.N)) return fail(CAT_TRUNK_ERR_BAD_ARG, "cat_trunk_forward: x_rows needs 0 < sel <= block and N a multiple of sel");
    const Geo ge(a->d);
    const int lds = (int)ge.fwd_lds();
    static int lds_set = 0;
    if (lds > lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(trunk_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return fail(CAT_TRUNK_ERR_HIP, "cat_trunk_forward: hipFuncSetAttribute failed");
        lds_set = lds;
    }
    hipLaunchKernelGGL(trunk_fwd_kernel, dim3(fwd_blocks(a->d), a->d.G), dim3(NW * LANES), lds, (hipStream_t)stream, *a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_TRUNK_OK : fail(CAT_TRUNK_ERR_HIP, hipGetErrorString(e));
}

extern "C" int cat_trunk_backward(const cat_trunk_bwd *a, void *stream)
{
    if (!a || !dims_ok(a->d)) return fail(CAT_TRUNK_ERR_BAD_ARG, "cat_trunk_backward: bad dimensions");
    if (!cat_trunk_supported(&a->d)) return fail(CAT_TRUNK_ERR_TOO_LARGE, "cat_trunk_backward: these dimensions do not fit the LDS");
    if (!params_ok(a->p) || !a->x || !a->out || !a->d_out || !a->part_dw1 || !a->part_db1 || !a->part_dw2 || !a->part_db2)
        return fail(CAT_TRUNK_ERR_BAD_ARG, "cat_trunk_backward: a required buffer is NULL or misaligned");
    if (!aligned(a->x, 8) || (a->sx_g % 4) || (a->sx_n % 4) || !aligned(a->out, 8) || !aligned(a->d_out, 8) || (a->so_g % 4) ||
        (a->so_n % 4) || !aligned(a->part_dw1, 4) || !aligned(a->part_dw2, 4))
        return fail(CAT_TRUNK_ERR_BAD_ARG, "cat_trunk_backward: misaligned buffer or stride");
    if (!rows_ok(a->x_rows, a->d.N)) return fail(CAT_TRUNK_ERR_BAD_ARG, "cat_trunk_backward: x_rows needs 0 < sel <= block and N a multiple of sel");
    const Geo ge(a->d);
    const int lds = (int)ge.bwd_lds();
    static int lds_set = 0;
    if (lds > lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(trunk_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return fail(CAT_TRUNK_ERR_HIP, "cat_trunk_backward: hipFuncSetAttribute failed");
        lds_set = lds;
    }
    hipLaunchKernelGGL(trunk_bwd_kernel, dim3(bwd_blocks(a->d), a->d.G), dim3(NWB * LANES), lds, (hipStream_t)stream, *a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_TRUNK_OK : fail(CAT_TRUNK_ERR_HIP, hipGetErrorString(e));
}

extern "C" int cat_trunk_grad_finish(const cat_trunk_finish_args *a, void *stream)
{
    if (!a || !dims_ok(a->d) || a->blocks <= 0) return fail(CAT_TRUNK_ERR_BAD_ARG, "cat_trunk_grad_finish: bad dimensions");
    if (!a->part_dw1 || !a->part_db1 || !a->part_dw2 || !a->part_db2 || !a->dw1 || !a->db1 || !a->dw2 || !a->db2)
        return fail(CAT_TRUNK_ERR_BAD_ARG, "cat_trunk_grad_finish: a required buffer is NULL");
    const int n = C1 * KW * a->d.C + C1 + C2 * WIN2 + C2;
    hipLaunchKernelGGL(trunk_finish_kernel, dim3((n + 255) / 256, a->d.G), dim3(256), 0, (hipStream_t)stream, *a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_TRUNK_OK : fail(CAT_TRUNK_ERR_HIP, hipGetErrorString(e));
}
