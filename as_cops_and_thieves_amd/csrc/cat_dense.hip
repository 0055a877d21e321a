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
// chunks) and added up through LDS
__global__ __launch_bounds__(BLOCK) void sum_chunks_kernel(const float *partial, int chunks, int n, __bf16 *dst0, int64_t sd0, __bf16 *dst1,
                                                           int64_t sd1, int accumulate)
{
    __shared__ float lds[BLOCK];
    const int g = blockIdx.y, cl = threadIdx.x & 31, lane = threadIdx.x >> 5, j = blockIdx.x * 32 + cl;
    float s = 0.0f;
    if (j < n) {
        const float *p = partial + (size_t)g * chunks * n + j;
        for (int c = lane; c < chunks; c += 8) s += p[(size_t)c * n];
    }
    lds[threadIdx.x] = s;
    __syncthreads();
    if (lane == 0 && j < n) {
#pragma unroll
        for (int k = 1; k < 8; ++k) s += lds[32 * k + cl];
        if (accumulate) {
            dst0[(size_t)g * sd0 + j] = (__bf16)(s + (float)dst0[(size_t)g * sd0 + j]);
            if (dst1) dst1[(size_t)g * sd1 + j] = (__bf16)(s + (float)dst1[(size_t)g * sd1 + j]);
        } else {
            dst0[(size_t)g * sd0 + j] = (__bf16)s;
            if (dst1) dst1[(size_t)g * sd1 + j] = (__bf16)s;
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
    if (!partial || !dst0 || G <= 0 || G > 65535 || chunks <= 0 || n <= 0)
        return fail(CAT_DENSE_ERR_BAD_ARG, "cat_dense_sum_chunks: bad dimensions or NULL buffer");
    hipLaunchKernelGGL(sum_chunks_kernel, dim3((n + 31) / 32, G), dim3(BLOCK), 0, (hipStream_t)stream, partial, chunks, n,
                       (__bf16 *)dst0, sd0_g, (__bf16 *)dst1, sd1_g, accumulate);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_DENSE_OK : fail(CAT_DENSE_ERR_HIP, hipGetErrorString(e));
}
