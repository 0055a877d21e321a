// Microbenchmark: VALU issue cost per wave-instruction on gfx950 for f64 / f32 / packed-f32 / int ops at 1, 2 and 4 waves per
// SIMD (what the ray fan's "f32 pre-classification instead of f64" can and cannot buy).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP> __global__ void k(float *out, int iters)
{
    double a = threadIdx.x * 1e-3 + 1.0, b = 1.0000001, c = 0.5;
    double a2 = a + 1, a3 = a + 2, a4 = a + 3;
    float fa = (float)a, fb = 1.0000001f, fc = 0.5f, fa2 = fa + 1, fa3 = fa + 2, fa4 = fa + 3;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pa = {fa, fa2}, pb = {fb, fb}, pc = {fc, fc}, pa2 = {fa3, fa4};
    int ia = threadIdx.x, ib = 3, ia2 = ia + 1, ia3 = ia + 2, ia4 = ia + 3;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (OP == 0) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
                           asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c)); }
            if (OP == 1) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa) : "v"(fb), "v"(fc)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa2) : "v"(fb), "v"(fc));
                           asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa3) : "v"(fb), "v"(fc)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa4) : "v"(fb), "v"(fc)); }
            if (OP == 2) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa) : "v"(pb), "v"(pc)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa2) : "v"(pb), "v"(pc));
                           asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa) : "v"(pb), "v"(pc)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa2) : "v"(pb), "v"(pc)); }
            if (OP == 3) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia) : "v"(ib)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia2) : "v"(ib));
                           asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia3) : "v"(ib)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia4) : "v"(ib)); }
            if (OP == 4) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a2) : "v"(b));
                           asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a3) : "v"(b)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a4) : "v"(b)); }
            if (OP == 5) { asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(a), "v"(b), "v"(ia), "v"(ib) : "vcc");
                           asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(a2), "v"(b), "v"(ia2), "v"(ib) : "vcc"); }
            if (OP == 6) { asm volatile("v_rcp_f64 %0, %1" : "=v"(a) : "v"(a2)); asm volatile("v_rcp_f64 %0, %1" : "=v"(a3) : "v"(a4));
                           asm volatile("v_rsq_f64 %0, %1" : "=v"(a) : "v"(a2)); asm volatile("v_rsq_f64 %0, %1" : "=v"(a3) : "v"(a4)); }
            if (OP == 7) { asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_min_f64 %0, %0, %1" : "+v"(a2) : "v"(b));
                           asm volatile("v_max_f64 %0, %0, %1" : "+v"(a3) : "v"(b)); asm volatile("v_min_f64 %0, %0, %1" : "+v"(a4) : "v"(b)); }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = (float)(a + a2 + a3 + a4) + fa + fa2 + fa3 + fa4 + pa.x + pa.y + pa2.x + pa2.y + (float)(ia + ia2 + ia3 + ia4);
}

template <int OP> void run(const char *name, float *d)
{
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves : {4, 8, 16}) {   // waves per CU (one workgroup per CU): 1, 2, 4 per SIMD
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 0, 0, d, iters);   // warm
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
        const double n = (double)iters * 64;                 // wave-instructions per wave
        printf("%-26s waves/SIMD %d: kernel %.1f us -> %.3f ns per wave-instruction per SIMD; oldest wave: %.2f counter ticks per instruction\n",
               name, waves / 4, ms * 1e3, ms * 1e6 / (n * (waves / 4)), cyc / n);
    }
}

int main()
{
    float *d; hipMalloc(&d, 4 * (1 + 256 * 1024));
    run<1>("v_fma_f32", d); run<0>("v_fma_f64", d); run<4>("v_add_f64 / v_mul_f64", d); run<7>("v_max_f64 / v_min_f64", d); run<5>("v_cmp_lt_f64 + v_cndmask", d);
    run<2>("v_pk_fma_f32", d); run<3>("v_add_u32", d); run<6>("v_rcp_f64 / v_rsq_f64", d);
    return 0;
}
