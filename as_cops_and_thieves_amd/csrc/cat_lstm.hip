// cat_lstm.hip -- libcat_learn.so: the LSTM recurrence of the role-stacked self-play learner on MI355X (gfx950).
//
// What it replaces (include/cat_lstm.h): per BPTT step a batched GEMM + the fused gate kernel + the episode-start
// masks forward, the gate gradient + a batched GEMM + masks backward -- about forty ~5 us launches per step and
// layer.  Here the window of one layer is ONE launch per direction:
//
//   * a workgroup = 16 sequences of one network, 8 waves; wave w owns hidden units [16w, 16w+16) with all four gates
//     (the recurrence is a chain of T dependent steps with 16 sequences as the matrix cores' N: what shortens a step is more
//     waves sharing its gate arithmetic -- two per SIMD, whose matrix-core and transcendental work overlap);
//   * the products are computed TRANSPOSED on v_mfma_f32_16x16x32_bf16: gates^T [4H x 16] = W_hh [4H x H] * h^T, so the
//     A operand is rows of W_hh (register resident for the whole window: 16 fragments = 64 VGPRs per lane), the lanes
//     of an accumulator are sequences, and the four registers of a lane are four CONSECUTIVE hidden units: every
//     global access of the recurrence is an 8-byte (4 x bf16) access, the cell arithmetic happens in the accumulator
//     registers in fp32, and the cell state c never leaves the registers during the window;
//   * the only exchange between the waves is h_t (forward) / the gate gradients (backward), through a double-buffered
//     padded LDS tile: one barrier per step;
//   * what backward needs (activated gates, c entering the step, tanh c leaving it) is written in the accumulator's own
//     lane order -- fully coalesced, read back by the same lanes.
//
// Backward: d h_{t-1}^T [H x 16] = W_hh^T [H x 4H] * d gates^T; the A operand is W_hh^T gathered once per launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "cat_lstm.h"

namespace {

#ifndef CAT_LSTM_WAVES
#define CAT_LSTM_WAVES 8
#endif
constexpr int H = CAT_LSTM_HIDDEN, H4 = 4 * H, BM = CAT_LSTM_ROWS_PER_BLOCK, NW = CAT_LSTM_WAVES, LANES = 64;
constexpr int HH = H / (16 * NW);                    // 16-unit groups of hidden units per wave (8 waves: 1, 4 waves: 2)
static_assert(HH * 16 * NW == H && HH >= 1, "the waves of a workgroup split the hidden units in groups of 16");
constexpr int HPAD = H + 8, GPAD = H4 + 8;          // LDS row strides (bf16): 272 B / 1040 B -> 16 rows hit 64 distinct banks

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x4 widen(bf16x4 v) { return __builtin_convertvector(v, f32x4); }
__device__ __forceinline__ bf16x4 narrow(f32x4 v) { return __builtin_convertvector(v, bf16x4); }
__device__ __forceinline__ bf16x4 zero4() { return narrow(f32x4{0.f, 0.f, 0.f, 0.f}); }

__device__ __forceinline__ float sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// index (in bf16 elements) of a lane's 4-vector inside the saved buffers
__device__ __forceinline__ size_t acts_index(int t, int g, int G, int nblk, int blk, int w, int tile, int lane)
{
    return ((((size_t)(t * G + g) * nblk + blk) * NW + w) * (4 * HH) + tile) * (LANES * 4) + lane * 4;
}
__device__ __forceinline__ size_t cell_index(int t, int g, int G, int nblk, int blk, int w, int which, int hh, int lane)
{
    return (((((size_t)(t * G + g) * nblk + blk) * NW + w) * 2 + which) * HH + hh) * (LANES * 4) + lane * 4;
}

// SAVE = training (keeps what backward needs; 8 waves of 16 hidden units, one workgroup per CU); !SAVE = inference (rollout
// ticks, T = 1: thousands of sequences, nothing kept; 4 waves of 32 hidden units and two workgroups per CU, which hide each
// other's latencies -- with 8 waves and one workgroup per CU a tick's launch took 27.9 us instead of 22.8)
template <bool SAVE, int NWT>
__global__ __launch_bounds__(NWT *LANES, 2) void lstm_seq_fwd_kernel(const cat_lstm_fwd a)
{
    constexpr int HHT = H / (16 * NWT);                  // 16-unit groups of hidden units per wave of THIS instantiation
    static_assert(!SAVE || NWT == NW, "the saved buffers are laid out for the backward kernel's wave count");
    __shared__ __attribute__((aligned(16))) __bf16 hbuf[2][BM][HPAD];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, q = l >> 4, r = l & 15;
    const int g = blockIdx.y, G = a.d.G, T = a.d.T, B = a.d.B, nblk = (B + BM - 1) / BM;
    const int hid0 = 16 * HHT * w + 4 * q;                       // + 16 * hh: this lane's four hidden units of half hh

    // W_hh fragments: A[row = gate column n0 + r][k = 32 ks + 8 q + j]
    bf16x8 wf[4][HHT][4];
    {
        const __bf16 *wg = (const __bf16 *)a.w_hh + (size_t)g * a.sw_g;
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int hh = 0; hh < HHT; ++hh)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    wf[gt][hh][ks] = *(const bf16x8 *)(wg + (size_t)(gt * H + 16 * HHT * w + 16 * hh + r) * H + 32 * ks + 8 * q);
    }
    constexpr bool save = SAVE;
    // The summed gate biases: in registers in the training instantiation; in LDS in the inference one (a rollout tick), whose 128
    // registers of W_hh fragments + accumulators left 8 values spilled to scratch at the 256 registers two workgroups per CU allow.
    bf16x4 bias[SAVE ? 4 : 1][SAVE ? HHT : 1];
    __shared__ __attribute__((aligned(16))) __bf16 sbias[SAVE ? 4 : 4 * H];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
#pragma unroll
        for (int hh = 0; hh < HHT; ++hh)
        {
            f32x4 bsum = a.bias ? widen(*(const bf16x4 *)((const __bf16 *)a.bias + (size_t)g * a.sb_g + gt * H + hid0 + 16 * hh))
                                : f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.bias2) bsum += widen(*(const bf16x4 *)((const __bf16 *)a.bias2 + (size_t)g * a.sb2_g + gt * H + hid0 + 16 * hh));
            // one rounding of the sum, as the bf16 addition of the two vectors gives
            if constexpr (SAVE) bias[gt][hh] = narrow(bsum);
            else if (r == 0) *(bf16x4 *)&sbias[gt * H + hid0 + 16 * hh] = narrow(bsum);   // the 16 sequence rows of a lane group hold the same units
        }
    // a workgroup takes the 16-sequence blocks blk, blk + gridDim.x, ...: one each in training (the grid covers them), several
    // when there are more blocks than the device holds at once (a rollout tick of thousands of envs) -- W_hh is loaded once
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int b = blk * BM + r;
    const bool row_ok = b < B;
    const __bf16 *xg = (const __bf16 *)a.xproj + (size_t)g * a.sx_g + (size_t)b * a.sx_b;
    __bf16 *og = (__bf16 *)a.out + (size_t)g * a.so_g + (size_t)b * a.so_b;
    const size_t state_row = ((size_t)g * B + b) * H;
    f32x4 c[HHT];
    {
        const float k0 = (a.keep && row_ok) ? a.keep[b] : 1.0f;
#pragma unroll
        for (int hh = 0; hh < HHT; ++hh) {
            const int hid = hid0 + 16 * hh;
            f32x4 h = widen(row_ok ? *(const bf16x4 *)((const __bf16 *)a.h0 + state_row + hid) : zero4()) * k0;
            c[hh] = widen(row_ok ? *(const bf16x4 *)((const __bf16 *)a.c0 + state_row + hid) : zero4()) * k0;
            const bf16x4 hb = narrow(h);
            *(bf16x4 *)&hbuf[0][r][hid] = hb;
            if (SAVE && a.h_in && row_ok) *(bf16x4 *)((__bf16 *)a.h_in + (((size_t)g * T + 0) * B + b) * H + hid) = hb;
        }
    }
    __syncthreads();

    // the input-side pre-activations and the keep flag of step t + 1 are fetched while step t computes: the recurrence
    // is a chain of T dependent steps, and a global-memory round trip at the head of each would be most of its length
    bf16x4 xn[4][HHT];
    float kn_next = 1.0f;
    auto fetch = [&](int t) {
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int hh = 0; hh < HHT; ++hh)
                xn[gt][hh] = (row_ok && t < T) ? *(const bf16x4 *)(xg + (size_t)t * a.sx_t + gt * H + hid0 + 16 * hh) : zero4();
        kn_next = (a.keep && row_ok && t + 1 < T) ? a.keep[(size_t)(t + 1) * B + b] : 1.0f;
    };
    // What a step leaves in global memory is stored DURING the next step, right after that step's wait: the wait at the head
    // of a step is one in-order counter that reaches zero only when everything issued before it has completed, stores
    // included -- issued at the end of a step, their acknowledgements were what every step of the chain then sat out.
    bf16x4 p_gi[HHT], p_gf[HHT], p_gg[HHT], p_go[HHT], p_c[HHT], p_tc[HHT], p_hy[HHT], p_hb[HHT];
    auto emit = [&](int tp) {
#pragma unroll
        for (int hh = 0; hh < HHT; ++hh) {
            const int hid = hid0 + 16 * hh;
            if (save) {
                __bf16 *sa = (__bf16 *)a.saved_acts, *sc = (__bf16 *)a.saved_cell;
                *(bf16x4 *)(sa + acts_index(tp, g, G, nblk, blk, w, 0 * HHT + hh, l)) = p_gi[hh];
                *(bf16x4 *)(sa + acts_index(tp, g, G, nblk, blk, w, 1 * HHT + hh, l)) = p_gf[hh];
                *(bf16x4 *)(sa + acts_index(tp, g, G, nblk, blk, w, 2 * HHT + hh, l)) = p_gg[hh];
                *(bf16x4 *)(sa + acts_index(tp, g, G, nblk, blk, w, 3 * HHT + hh, l)) = p_go[hh];
                *(bf16x4 *)(sc + cell_index(tp, g, G, nblk, blk, w, 0, hh, l)) = p_c[hh];
                *(bf16x4 *)(sc + cell_index(tp, g, G, nblk, blk, w, 1, hh, l)) = p_tc[hh];
            }
            if (row_ok) *(bf16x4 *)(og + (size_t)tp * a.so_t + hid) = p_hy[hh];
            if (SAVE && a.h_in && row_ok && tp + 1 < T) *(bf16x4 *)((__bf16 *)a.h_in + (((size_t)g * T + tp + 1) * B + b) * H + hid) = p_hb[hh];
        }
    };
    fetch(0);
    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        f32x4 acc[4][HHT];
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int hh = 0; hh < HHT; ++hh) {
                if constexpr (SAVE) acc[gt][hh] = widen(xn[gt][hh]) + widen(bias[gt][hh]);
                else acc[gt][hh] = widen(xn[gt][hh]) + widen(*(const bf16x4 *)&sbias[gt * H + hid0 + 16 * hh]);
            }
        const float kn = kn_next;
        // The next step's loads must be ISSUED after this step's values have been waited for: hoisted above the lines before
        // (as the scheduler does), the wait for xn also waits for the loads just issued, and every step of the chain sits out
        // a global round trip (5 us per step at 8192 sequences).
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int hh = 0; hh < HHT; ++hh) asm volatile("" : "+v"(acc[gt][hh]) : : "memory");
        fetch(t + 1);
        if (SAVE && t > 0) emit(t - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 hb = *(const bf16x8 *)&hbuf[cur][r][32 * ks + 8 * q];
#pragma unroll
            for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                for (int hh = 0; hh < HHT; ++hh)
                    acc[gt][hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[gt][hh][ks], hb, acc[gt][hh], 0, 0, 0);
        }
#pragma unroll
        for (int hh = 0; hh < HHT; ++hh) {
            const int hid = hid0 + 16 * hh;
            f32x4 gi, gf, gg, go, cy, tc, hy;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                gi[i] = sigmoidf(acc[0][hh][i]);
                gf[i] = sigmoidf(acc[1][hh][i]);
                gg[i] = tanh_fast(acc[2][hh][i]);
                go[i] = sigmoidf(acc[3][hh][i]);
                cy[i] = gf[i] * c[hh][i] + gi[i] * gg[i];
                tc[i] = tanh_fast(cy[i]);
                hy[i] = go[i] * tc[i];
            }
            if (save) {
                p_gi[hh] = narrow(gi); p_gf[hh] = narrow(gf); p_gg[hh] = narrow(gg); p_go[hh] = narrow(go);
                p_c[hh] = narrow(c[hh]); p_tc[hh] = narrow(tc);
            }
            p_hy[hh] = narrow(hy);
            if (t + 1 < T) {
                const bf16x4 hb = narrow(hy * kn);
                *(bf16x4 *)&hbuf[cur ^ 1][r][hid] = hb;
                p_hb[hh] = hb;
                c[hh] = cy * kn;
            } else if (row_ok) {
                *(bf16x4 *)((__bf16 *)a.h_last + state_row + hid) = narrow(hy);
                *(bf16x4 *)((__bf16 *)a.c_last + state_row + hid) = narrow(cy);
            }
        }
        if (!SAVE) emit(t);        // a rollout tick (T = 1): nothing to overlap with, and no registers to spare
        __syncthreads();
    }
    if (SAVE) emit(T - 1);
    }   // blocks of this workgroup
}

__global__ __launch_bounds__(NW *LANES) void lstm_seq_bwd_kernel(const cat_lstm_bwd a)
{
    __shared__ __attribute__((aligned(16))) __bf16 dgbuf[2][BM][GPAD];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, q = l >> 4, r = l & 15;
    const int g = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x, G = a.d.G, T = a.d.T, B = a.d.B;
    const int b = blk * BM + r;
    const bool row_ok = b < B;
    const int hid0 = 16 * HH * w + 4 * q;

    // W_hh^T fragments: A[row = hidden m0 + r][k = gate column 32 ks + 8 q + j] = W_hh[k][m0 + r]
    bf16x8 wt[HH][16];
    {
        const __bf16 *wg = (const __bf16 *)a.w_hh + (size_t)g * a.sw_g;
#pragma unroll
        for (int hh = 0; hh < HH; ++hh)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    wt[hh][ks][j] = wg[(size_t)(32 * ks + 8 * q + j) * H + 16 * HH * w + 16 * hh + r];
    }
    const __bf16 *og = a.d_out ? (const __bf16 *)a.d_out + (size_t)g * a.so_g + (size_t)b * a.so_b : nullptr;
    __bf16 *xg = (__bf16 *)a.d_xproj + (size_t)g * a.sx_g + (size_t)b * a.sx_b;
    const size_t state_row = ((size_t)g * B + b) * H;
    const __bf16 *sa = (const __bf16 *)a.saved_acts, *sc = (const __bf16 *)a.saved_cell;

    f32x4 dbs[4][HH];                                  // sums over the steps of this lane's gate gradients (bias gradient)
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
#pragma unroll
        for (int hh = 0; hh < HH; ++hh) dbs[gt][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 dh[HH], dc[HH];
#pragma unroll
    for (int hh = 0; hh < HH; ++hh) {
        const int hid = hid0 + 16 * hh;
        dh[hh] = widen((a.d_h_last && row_ok) ? *(const bf16x4 *)((const __bf16 *)a.d_h_last + state_row + hid) : zero4());
        dc[hh] = widen((a.d_c_last && row_ok) ? *(const bf16x4 *)((const __bf16 *)a.d_c_last + state_row + hid) : zero4());
    }

    // what step t - 1 reads from global memory is fetched while step t computes (see the forward kernel)
    bf16x4 pd[HH], pa[4][HH], pc[2][HH];
    float kt_next = 1.0f;
    auto fetch = [&](int t) {
        const bool in = t >= 0;
#pragma unroll
        for (int hh = 0; hh < HH; ++hh) {
            pd[hh] = (in && og && row_ok) ? *(const bf16x4 *)(og + (size_t)t * a.so_t + hid0 + 16 * hh) : zero4();
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) pa[gt][hh] = in ? *(const bf16x4 *)(sa + acts_index(t, g, G, nblk, blk, w, gt * HH + hh, l)) : zero4();
            pc[0][hh] = in ? *(const bf16x4 *)(sc + cell_index(t, g, G, nblk, blk, w, 0, hh, l)) : zero4();
            pc[1][hh] = in ? *(const bf16x4 *)(sc + cell_index(t, g, G, nblk, blk, w, 1, hh, l)) : zero4();
        }
        kt_next = (in && a.keep && row_ok) ? a.keep[(size_t)t * B + b] : 1.0f;
    };
    fetch(T - 1);
    for (int t = T - 1; t >= 0; --t) {
        const int buf = t & 1;
        const float kt = kt_next;
        f32x4 d2[HH], gi2[HH], gf2[HH], gg2[HH], go2[HH], cin2[HH], tc2[HH];
#pragma unroll
        for (int hh = 0; hh < HH; ++hh) {
            d2[hh] = dh[hh] + widen(pd[hh]);
            gi2[hh] = widen(pa[0][hh]); gf2[hh] = widen(pa[1][hh]); gg2[hh] = widen(pa[2][hh]); go2[hh] = widen(pa[3][hh]);
            cin2[hh] = widen(pc[0][hh]); tc2[hh] = widen(pc[1][hh]);
        }
        fetch(t - 1);
#pragma unroll
        for (int hh = 0; hh < HH; ++hh) {
            const int hid = hid0 + 16 * hh;
            const f32x4 d = d2[hh], gi = gi2[hh], gf = gf2[hh], gg = gg2[hh], go = go2[hh], cin = cin2[hh], tc = tc2[hh];
            f32x4 di, df, dg, d_o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dct = d[i] * go[i] * (1.0f - tc[i] * tc[i]) + dc[hh][i];
                d_o[i] = d[i] * tc[i] * go[i] * (1.0f - go[i]);
                di[i] = dct * gg[i] * gi[i] * (1.0f - gi[i]);
                df[i] = dct * cin[i] * gf[i] * (1.0f - gf[i]);
                dg[i] = dct * gi[i] * (1.0f - gg[i] * gg[i]);
                dc[hh][i] = dct * gf[i] * kt;
            }
            const bf16x4 v0 = narrow(di), v1 = narrow(df), v2 = narrow(dg), v3 = narrow(d_o);
            if (row_ok) { dbs[0][hh] += widen(v0); dbs[1][hh] += widen(v1); dbs[2][hh] += widen(v2); dbs[3][hh] += widen(v3); }
            *(bf16x4 *)&dgbuf[buf][r][0 * H + hid] = v0;
            *(bf16x4 *)&dgbuf[buf][r][1 * H + hid] = v1;
            *(bf16x4 *)&dgbuf[buf][r][2 * H + hid] = v2;
            *(bf16x4 *)&dgbuf[buf][r][3 * H + hid] = v3;
            if (row_ok) {
                __bf16 *x = xg + (size_t)t * a.sx_t + hid;
                *(bf16x4 *)(x + 0 * H) = v0;
                *(bf16x4 *)(x + 1 * H) = v1;
                *(bf16x4 *)(x + 2 * H) = v2;
                *(bf16x4 *)(x + 3 * H) = v3;
            }
        }
        __syncthreads();
        if (t > 0 || a.d_h0) {
            f32x4 acc[HH];
#pragma unroll
            for (int hh = 0; hh < HH; ++hh) acc[hh] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const bf16x8 db = *(const bf16x8 *)&dgbuf[buf][r][32 * ks + 8 * q];
#pragma unroll
                for (int hh = 0; hh < HH; ++hh)
                    acc[hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[hh][ks], db, acc[hh], 0, 0, 0);
            }
#pragma unroll
            for (int hh = 0; hh < HH; ++hh) dh[hh] = acc[hh] * kt;
        }
    }
    if (a.part_dbias) {   // add the 16 rows (lanes r) up; lane r == 0 of every group of 16 writes four columns
        float *out = a.part_dbias + ((size_t)g * nblk + blk) * H4;
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int hh = 0; hh < HH; ++hh) {
                f32x4 v = dbs[gt][hh];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int off = 8; off > 0; off >>= 1) v[e] += __shfl_down(v[e], off, 16);
                if (r == 0) *(f32x4 *)(out + gt * H + hid0 + 16 * hh) = v;
            }
    }
    if (row_ok) {
#pragma unroll
        for (int hh = 0; hh < HH; ++hh) {
            const int hid = hid0 + 16 * hh;
            if (a.d_h0) *(bf16x4 *)((__bf16 *)a.d_h0 + state_row + hid) = narrow(dh[hh]);
            if (a.d_c0) *(bf16x4 *)((__bf16 *)a.d_c0 + state_row + hid) = narrow(dc[hh]);
        }
    }
}

thread_local char g_err[256] = "";

int fail(int code, const char *msg)
{
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

bool dims_ok(const cat_lstm_dims &d) { return d.G > 0 && d.G <= 65535 && d.T > 0 && d.T <= CAT_LSTM_MAX_T && d.B > 0; }
bool aligned(const void *p, size_t a) { return ((uintptr_t)p % a) == 0; }
int blocks_of(const cat_lstm_dims &d) { return (d.B + BM - 1) / BM; }

}   // namespace

extern "C" int cat_lstm_abi_version(void) { return CAT_LSTM_ABI_VERSION; }
extern "C" const char *cat_lstm_last_error(void) { return g_err; }

extern "C" int cat_lstm_blocks(const cat_lstm_dims *d) { return d && dims_ok(*d) ? blocks_of(*d) : CAT_LSTM_ERR_BAD_ARG; }

extern "C" size_t cat_lstm_saved_acts_bytes(const cat_lstm_dims *d)
{
    return d && dims_ok(*d) ? (size_t)d->T * d->G * blocks_of(*d) * NW * 4 * HH * LANES * 4 * 2 : 0;
}
extern "C" size_t cat_lstm_saved_cell_bytes(const cat_lstm_dims *d)
{
    return d && dims_ok(*d) ? (size_t)d->T * d->G * blocks_of(*d) * NW * 2 * HH * LANES * 4 * 2 : 0;
}

extern "C" int cat_lstm_seq_forward(const cat_lstm_fwd *a, void *stream)
{
    if (!a || !dims_ok(a->d)) return fail(CAT_LSTM_ERR_BAD_ARG, "cat_lstm_seq_forward: bad dimensions");
    if (!a->xproj || !a->w_hh || !a->h0 || !a->c0 || !a->out || !a->h_last || !a->c_last)
        return fail(CAT_LSTM_ERR_BAD_ARG, "cat_lstm_seq_forward: a required buffer is NULL");
    if ((a->saved_acts == nullptr) != (a->saved_cell == nullptr))
        return fail(CAT_LSTM_ERR_BAD_ARG, "cat_lstm_seq_forward: saved_acts and saved_cell go together");
    if (!aligned(a->w_hh, 16) || (a->sw_g % 8) || !aligned(a->xproj, 8) || (a->sx_g % 4) || (a->sx_t % 4) || (a->sx_b % 4) ||
        !aligned(a->out, 8) || (a->so_g % 4) || (a->so_t % 4) || (a->so_b % 4) || !aligned(a->h0, 8) || !aligned(a->c0, 8) ||
        !aligned(a->h_last, 8) || !aligned(a->c_last, 8) || !aligned(a->h_in, 8) || !aligned(a->saved_acts, 8) || !aligned(a->saved_cell, 8) ||
        !aligned(a->bias, 8) || (a->bias && (a->sb_g % 4)) || !aligned(a->bias2, 8) || (a->bias2 && (a->sb2_g % 4)) || (a->bias2 && !a->bias))
        return fail(CAT_LSTM_ERR_BAD_ARG, "cat_lstm_seq_forward: misaligned buffer or stride");
    // more blocks than one resident round of the device (one workgroup per CU at this register count): fold them
    const int nb = blocks_of(a->d);
    if (a->saved_acts) {
        if (!a->h_in) return fail(CAT_LSTM_ERR_BAD_ARG, "cat_lstm_seq_forward: h_in goes with the saved buffers");
        const int rounds = (nb * a->d.G + 255) / 256;
        hipLaunchKernelGGL((lstm_seq_fwd_kernel<true, NW>), dim3((nb + rounds - 1) / rounds, a->d.G), dim3(NW * LANES), 0, (hipStream_t)stream, *a);
    } else {
        const int rounds = (nb * a->d.G + 511) / 512;      // two 4-wave workgroups per CU are resident
        hipLaunchKernelGGL((lstm_seq_fwd_kernel<false, 4>), dim3((nb + rounds - 1) / rounds, a->d.G), dim3(4 * LANES), 0, (hipStream_t)stream, *a);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_LSTM_OK : fail(CAT_LSTM_ERR_HIP, hipGetErrorString(e));
}

extern "C" int cat_lstm_seq_backward(const cat_lstm_bwd *a, void *stream)
{
    if (!a || !dims_ok(a->d)) return fail(CAT_LSTM_ERR_BAD_ARG, "cat_lstm_seq_backward: bad dimensions");
    if (!a->w_hh || !a->saved_acts || !a->saved_cell || !a->d_xproj)
        return fail(CAT_LSTM_ERR_BAD_ARG, "cat_lstm_seq_backward: a required buffer is NULL");
    if (!aligned(a->w_hh, 2) || !aligned(a->d_xproj, 8) || (a->sx_g % 4) || (a->sx_t % 4) || (a->sx_b % 4) || !aligned(a->d_out, 8) ||
        (a->d_out && ((a->so_g % 4) || (a->so_t % 4) || (a->so_b % 4))) || !aligned(a->d_h_last, 8) || !aligned(a->d_c_last, 8) ||
        !aligned(a->d_h0, 8) || !aligned(a->d_c0, 8) || !aligned(a->saved_acts, 8) || !aligned(a->saved_cell, 8) || !aligned(a->part_dbias, 16))
        return fail(CAT_LSTM_ERR_BAD_ARG, "cat_lstm_seq_backward: misaligned buffer or stride");
    hipLaunchKernelGGL(lstm_seq_bwd_kernel, dim3(blocks_of(a->d), a->d.G), dim3(NW * LANES), 0, (hipStream_t)stream, *a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_LSTM_OK : fail(CAT_LSTM_ERR_HIP, hipGetErrorString(e));
}
