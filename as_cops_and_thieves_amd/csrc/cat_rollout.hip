// cat_rollout.hip -- libcat_learn.so, part 5: observation packing and action sampling of a rollout tick
// (include/cat_rollout.h).  Byte/half-word shuffling at a few hundred KB per tick: what matters is that each is ONE
// node of the replayed rollout graph.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>

#include "cat_rollout.h"

namespace {

constexpr int BLOCK = 256;

__global__ __launch_bounds__(BLOCK) void pack_kernel(const cat_rollout_pack_args a)
{
    const int g = blockIdx.y, R = a.R;
    const int ai = a.agent[g], si = a.first_agent_state ? 0 : ai;          // whose observation / whose shared state
    const int team = si < a.n_cops ? 0 : 1;
    const __half *od = (const __half *)a.obs_distance, *sd = (const __half *)a.shared_distance;
    const uint8_t *ot = (const uint8_t *)a.obs_type, *st = (const uint8_t *)a.shared_type;
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < (long)a.N * R; i += (long)gridDim.x * BLOCK) {
        const int n = (int)(i / R), r = (int)(i - (long)n * R);
        __bf16 *p = (__bf16 *)a.policy_in + (size_t)g * a.sp_g + (size_t)n * a.sp_n;
        __bf16 *v = (__bf16 *)a.value_in + (size_t)g * a.sv_g + (size_t)n * a.sv_n;
        const size_t own = ((size_t)n * a.A + ai) * R + r, st_own = ((size_t)n * a.A + si) * R + r, sh = ((size_t)n * 2 + team) * R + r;
        p[r] = (__bf16)(__half2float(od[own]) * a.distance_scale);
        p[R + r] = (__bf16)((float)ot[own] * a.type_scale);
        v[r] = (__bf16)(__half2float(sd[sh]) * a.distance_scale);
        v[R + r] = (__bf16)((float)st[sh] * a.type_scale);
        v[2 * R + r] = (__bf16)(__half2float(od[st_own]) * a.distance_scale);
        v[3 * R + r] = (__bf16)((float)ot[st_own] * a.type_scale);
    }
}

using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(BLOCK) void sample_kernel(const cat_rollout_sample_args a)
{
    const int g = blockIdx.y;
    for (int n = blockIdx.x * BLOCK + threadIdx.x; n < a.N; n += gridDim.x * BLOCK) {
        const size_t s = (size_t)g * a.N + n;
        const f32x4 z = __builtin_convertvector(*(const bf16x4 *)((const __bf16 *)a.logits + 4 * s), f32x4);
        const float zmax = fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3]));
        float e[4], sum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { e[j] = __expf(z[j] - zmax); sum += e[j]; }
        const float u = a.uniform[s] * sum;          // inverse CDF on the unnormalised masses
        const int act = (u >= e[0]) + (u >= e[0] + e[1]) + (u >= e[0] + e[1] + e[2]);
        a.act_out[(size_t)g * a.sa_g + n] = act;
        a.logp_out[(size_t)g * a.sl_g + n] = z[act] - zmax - __logf(sum);
        if (a.value_out) a.value_out[(size_t)g * a.sl_g + n] = (float)((const __bf16 *)a.values)[s];
        a.actions[(size_t)n * a.A + a.agent[g]] = act;
    }
}

__global__ __launch_bounds__(BLOCK) void post_kernel(const cat_rollout_post_args a)
{
    for (int n = blockIdx.x * BLOCK + threadIdx.x; n < a.N; n += gridDim.x * BLOCK) {
        for (int g = 0; g < a.G; ++g) a.reward_out[(size_t)g * a.sr_g + n] = a.reward[(size_t)n * a.A + a.agent[g]];
        const uint8_t d = a.terminated[n] ? 1 : 0;
        if (a.done_out) a.done_out[n] = d;
        if (a.start_out) a.start_out[n] = d;
        if (a.keep_out) a.keep_out[n] = d ? 0.0f : 1.0f;
    }
}

thread_local char g_err[256] = "";
int fail(int code, const char *msg)
{
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

}   // namespace

extern "C" int cat_rollout_abi_version(void) { return CAT_ROLLOUT_ABI_VERSION; }
extern "C" const char *cat_rollout_last_error(void) { return g_err; }

extern "C" int cat_rollout_pack(const cat_rollout_pack_args *a, void *stream)
{
    if (!a || a->N <= 0 || a->R <= 0 || a->A <= 0 || a->A > CAT_ROLLOUT_MAX_AGENTS || a->G <= 0 || a->G > CAT_ROLLOUT_MAX_AGENTS)
        return fail(CAT_ROLLOUT_ERR_BAD_ARG, "cat_rollout_pack: bad dimensions");
    for (int g = 0; g < a->G; ++g)
        if (a->agent[g] < 0 || a->agent[g] >= a->A) return fail(CAT_ROLLOUT_ERR_BAD_ARG, "cat_rollout_pack: agent index out of range");
    if (!a->obs_distance || !a->obs_type || !a->shared_distance || !a->shared_type || !a->policy_in || !a->value_in)
        return fail(CAT_ROLLOUT_ERR_BAD_ARG, "cat_rollout_pack: a required buffer is NULL");
    long work = (long)a->N * a->R;
    int blocks = (int)((work + BLOCK - 1) / BLOCK);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pack_kernel, dim3(blocks, a->G), dim3(BLOCK), 0, (hipStream_t)stream, *a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_ROLLOUT_OK : fail(CAT_ROLLOUT_ERR_HIP, hipGetErrorString(e));
}

extern "C" int cat_rollout_sample(const cat_rollout_sample_args *a, void *stream)
{
    if (!a || a->N <= 0 || a->A <= 0 || a->A > CAT_ROLLOUT_MAX_AGENTS || a->G <= 0 || a->G > CAT_ROLLOUT_MAX_AGENTS)
        return fail(CAT_ROLLOUT_ERR_BAD_ARG, "cat_rollout_sample: bad dimensions");
    for (int g = 0; g < a->G; ++g)
        if (a->agent[g] < 0 || a->agent[g] >= a->A) return fail(CAT_ROLLOUT_ERR_BAD_ARG, "cat_rollout_sample: agent index out of range");
    if (!a->logits || !a->uniform || !a->act_out || !a->logp_out || !a->actions || (a->value_out && !a->values) || ((uintptr_t)a->logits % 8))
        return fail(CAT_ROLLOUT_ERR_BAD_ARG, "cat_rollout_sample: NULL or misaligned buffer");
    int blocks = (a->N + BLOCK - 1) / BLOCK;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sample_kernel, dim3(blocks, a->G), dim3(BLOCK), 0, (hipStream_t)stream, *a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_ROLLOUT_OK : fail(CAT_ROLLOUT_ERR_HIP, hipGetErrorString(e));
}

extern "C" int cat_rollout_post(const cat_rollout_post_args *a, void *stream)
{
    if (!a || a->N <= 0 || a->A <= 0 || a->A > CAT_ROLLOUT_MAX_AGENTS || a->G <= 0 || a->G > CAT_ROLLOUT_MAX_AGENTS)
        return fail(CAT_ROLLOUT_ERR_BAD_ARG, "cat_rollout_post: bad dimensions");
    for (int g = 0; g < a->G; ++g)
        if (a->agent[g] < 0 || a->agent[g] >= a->A) return fail(CAT_ROLLOUT_ERR_BAD_ARG, "cat_rollout_post: agent index out of range");
    if (!a->reward || !a->terminated || !a->reward_out) return fail(CAT_ROLLOUT_ERR_BAD_ARG, "cat_rollout_post: a required buffer is NULL");
    int blocks = (a->N + BLOCK - 1) / BLOCK;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(post_kernel, dim3(blocks), dim3(BLOCK), 0, (hipStream_t)stream, *a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAT_ROLLOUT_OK : fail(CAT_ROLLOUT_ERR_HIP, hipGetErrorString(e));
}
