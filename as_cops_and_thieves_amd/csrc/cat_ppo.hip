// cat_ppo.hip -- libcat_learn.so, part 3: PPO loss + gradient and the optimiser step (include/cat_ppo.h).
//
// Both are a few flops per element over at most a few million elements: as library elementwise kernels they are
// ~110 launches of ~5 us per minibatch step inside the replayed HIP graph (each a dependent graph node), about a tenth
// of the step.  Here: one launch for the loss, its four per-agent statistics and the gradient w.r.t. logits and
// values (the derivative is written out analytically, no autograd graph for this part), and two launches for the
// gradient-norm clip + masked Adam + bf16 refresh.  Reductions are two-stage (per-block partial sums, added up by the
// consumer), never a semaphore-style single-pass reduction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "cat_ppo.h"

namespace {

constexpr int BLOCK = 256;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// sum of v over the block; valid in thread 0
template <int N>
__device__ __forceinline__ void block_sum(float (&v)[N], float *lds)
{
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_down(v[i], off, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0)
#pragma unroll
        for (int i = 0; i < N; ++i) lds[w * N + i] = v[i];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = lds[i] + lds[N + i] + lds[2 * N + i] + lds[3 * N + i];
}

__global__ __launch_bounds__(BLOCK) void ppo_loss_kernel(const cat_ppo_loss a)
{
    __shared__ float lds[4 * 4];
    const int g = blockIdx.y, M = a.M;
    const size_t base = (size_t)g * M;
    const float inv_m = 1.0f / (float)M, lo = 1.0f - a.ratio_clip, hi = 1.0f + a.ratio_clip;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < M; i += gridDim.x * BLOCK) {
        const size_t s = base + i;
        const f32x4 z = __builtin_convertvector(*(const bf16x4 *)((const __bf16 *)a.logits + 4 * s), f32x4);
        const float zmax = fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3]));
        float e[4], lp[4], p[4], sum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { e[j] = __expf(z[j] - zmax); sum += e[j]; }
        const float lse = zmax + __logf(sum), rs = 1.0f / sum;
        float ent = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { lp[j] = z[j] - lse; p[j] = e[j] * rs; ent -= p[j] * lp[j]; }
        const int act = (int)a.actions[s] & 3;
        const float logp = lp[act], d = logp - a.old_logp[s], ratio = __expf(d), adv = a.adv[s];
        const float s1 = adv * ratio, s2 = adv * fminf(fmaxf(ratio, lo), hi);
        const bool inside = ratio >= lo && ratio <= hi;
        // d min(s1, s2) / d logp: inside the clip range both branches are the same function; outside, the clipped
        // branch is constant, so the gradient flows only when the unclipped branch is the smaller one
        const float dsurr = (inside || s1 < s2) ? s1 : 0.0f;           // adv * ratio * (d ratio / d logp = ratio ... ) = s1
        acc[0] += fminf(s1, s2);
        const float v = (float)((const __bf16 *)a.values)[s], dv = v - a.ret[s];
        acc[1] += dv * dv;
        acc[2] += ent;
        acc[3] += (ratio - 1.0f) - d;
        f32x4 dz;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float onehot = (j == act) ? 1.0f : 0.0f;
            dz[j] = inv_m * (-dsurr * (onehot - p[j]) + a.entropy_scale * p[j] * (lp[j] + ent));
        }
        *(bf16x4 *)((__bf16 *)a.d_logits + 4 * s) = __builtin_convertvector(dz, bf16x4);
        ((__bf16 *)a.d_values)[s] = (__bf16)(2.0f * a.value_scale * inv_m * dv);
    }
    block_sum(acc, lds);
    if (threadIdx.x == 0) {
        float *out = a.partial + ((size_t)g * gridDim.x + blockIdx.x) * 4;
        out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2]; out[3] = acc[3];
    }
}

__global__ __launch_bounds__(BLOCK) void grad_norm_kernel(const cat_ppo_adam a)
{
    __shared__ float lds[4];
    const int g = blockIdx.y, P = a.P;
    const float *ar = a.ar + (size_t)g * (P + 1), *col = a.col_train + (size_t)g * P;
    float acc[1] = {0.f};
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < P; i += gridDim.x * BLOCK) {
        const float x = ar[i] * col[i];
        acc[0] += x * x;
    }
    block_sum(acc, lds);
    if (threadIdx.x == 0) a.norm_partial[(size_t)g * gridDim.x + blockIdx.x] = acc[0];
}

__global__ __launch_bounds__(BLOCK) void adam_kernel(const cat_ppo_adam a)
{
    __shared__ float lds[4];
    __shared__ float s_scale, s_active;
    const int g = blockIdx.y, P = a.P;
    float acc[1] = {0.f};
    for (int i = threadIdx.x; i < a.chunks; i += BLOCK) acc[0] += a.norm_partial[(size_t)g * a.chunks + i];
    block_sum(acc, lds);
    const float *ar = a.ar + (size_t)g * (P + 1);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(acc[0]), kl = ar[P];
        s_scale = fminf(a.grad_norm_clip / (norm + 1e-6f), 1.0f);
        // the gate is idempotent (a 0/1 factor), so blocks that read epoch_active after block 0 rewrote it agree
        const float active = a.epoch_active[g] * ((a.kl_threshold > 0.0f && !(kl <= a.kl_threshold)) ? 0.0f : 1.0f);
        s_active = active;
        if (blockIdx.x == 0) { a.epoch_active[g] = active; a.kl_out[g] = kl; }
    }
    __syncthreads();
    const float scale = s_scale, active = s_active, b1 = a.beta1, b2 = a.beta2;
    const size_t row = (size_t)g * P;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < P; i += gridDim.x * BLOCK) {
        const float col = a.col_train[row + i], gate = active * col;
        const float gr = ar[i] * col * scale;
        float m = a.m[row + i], v = a.v[row + i], st = a.steps[row + i], w = a.master[row + i];
        st += gate;
        m += gate * (1.0f - b1) * (gr - m);
        v += gate * (1.0f - b2) * (gr * gr - v);
        const float s = fmaxf(st, 1.0f);
        const float bc1 = 1.0f - powf(b1, s), bc2 = 1.0f - powf(b2, s);
        w -= gate * a.lr * (m / bc1) / (sqrtf(v / bc2) + a.eps);
        a.steps[row + i] = st; a.m[row + i] = m; a.v[row + i] = v; a.master[row + i] = w;
        if (a.lp) ((__bf16 *)a.lp)[row + i] = (__bf16)w;
    }
}

// one thread per (agent, env): the reverse scan of the rollout's T ticks; lanes = envs, so every access is coalesced
__global__ __launch_bounds__(BLOCK) void gae_kernel(const cat_ppo_gae a)
{
    const int n = blockIdx.x * BLOCK + threadIdx.x, g = blockIdx.y;
    if (n >= a.N) return;
    const size_t col = (size_t)g * a.T * a.N + n;
    float last = 0.0f, nxt = a.last_values[(size_t)g * a.N + n];
    const float gl = a.gamma * a.lambda;
    for (int t = a.T - 1; t >= 0; --t) {
        const size_t o = col + (size_t)t * a.N;
        const float nd = a.dones[(size_t)t * a.N + n] ? 0.0f : 1.0f, v = a.values[o];
        const float delta = a.rewards[o] + a.gamma * nxt * nd - v;
        last = delta + gl * nd * last;
        a.adv[o] = last;
        a.ret[o] = last + v;
        nxt = v;
    }
}

thread_local char g_err[256] = "";
int fail(int code, const char *msg)
{
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

}   // namespace

extern "C" int cat_ppo_abi_version(void) { return CAT_PPO_ABI_VERSION; }
extern "C" const char *cat_ppo_last_error(void) { return g_err; }

extern "C" int cat_ppo_loss_grad(const cat_ppo_loss *a, void *stream)
{
    if (!a || a->G <= 0 || a->G > 65535 || a->M <= 0 || a->chunks <= 0 || a->chunks > CAT_PPO_MAX_CHUNKS)
        return fail(CAT_PPO_ERR_BAD_ARG, "cat_ppo_loss_grad: bad dimensions");
    if (!a->logits || !a->values || !a->actions || !a->old_logp || !a->adv || !a->ret || !a->d_logits || !a->d_values || !a->partial)
        return fail(CAT_PPO_ERR_BAD_ARG, "cat_ppo_loss_grad: a required buffer is NULL");
    if (((uintptr_t)a->logits % 8) || ((uintptr_t)a->d_logits % 8))
        return fail(CAT_PPO_ERR_BAD_ARG, "cat_ppo_loss_grad: logits must be 8-byte aligned");
    hipLaunchKernelGGL(ppo_loss_kernel, dim3(a->chunks, a->G), dim3(BLOCK), 0, (hipStream_t)stream, *a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_PPO_OK : fail(CAT_PPO_ERR_HIP, hipGetErrorString(e));
}

extern "C" int cat_ppo_gae_scan(const cat_ppo_gae *a, void *stream)
{
    if (!a || a->G <= 0 || a->G > 65535 || a->T <= 0 || a->N <= 0) return fail(CAT_PPO_ERR_BAD_ARG, "cat_ppo_gae_scan: bad dimensions");
    if (!a->rewards || !a->values || !a->dones || !a->last_values || !a->adv || !a->ret)
        return fail(CAT_PPO_ERR_BAD_ARG, "cat_ppo_gae_scan: a required buffer is NULL");
    hipLaunchKernelGGL(gae_kernel, dim3((a->N + BLOCK - 1) / BLOCK, a->G), dim3(BLOCK), 0, (hipStream_t)stream, *a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_PPO_OK : fail(CAT_PPO_ERR_HIP, hipGetErrorString(e));
}

extern "C" int cat_ppo_adam_step(const cat_ppo_adam *a, void *stream)
{
    if (!a || a->G <= 0 || a->G > 65535 || a->P <= 0 || a->chunks <= 0 || a->chunks > CAT_PPO_MAX_CHUNKS)
        return fail(CAT_PPO_ERR_BAD_ARG, "cat_ppo_adam_step: bad dimensions");
    if (!a->ar || !a->col_train || !a->epoch_active || !a->m || !a->v || !a->steps || !a->master || !a->kl_out || !a->norm_partial)
        return fail(CAT_PPO_ERR_BAD_ARG, "cat_ppo_adam_step: a required buffer is NULL");
    hipLaunchKernelGGL(grad_norm_kernel, dim3(a->chunks, a->G), dim3(BLOCK), 0, (hipStream_t)stream, *a);
    hipLaunchKernelGGL(adam_kernel, dim3(a->chunks, a->G), dim3(BLOCK), 0, (hipStream_t)stream, *a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_PPO_OK : fail(CAT_PPO_ERR_HIP, hipGetErrorString(e));
}
