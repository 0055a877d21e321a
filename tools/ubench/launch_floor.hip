// Microbenchmark: duration (events attached to the dispatch) of kernels with tick_kernel's launch geometry -- 256 workgroups x 1024
// threads, ~150 KB of dynamic LDS -- that do (a) nothing, (b) one global round trip + barrier, (c) three dependent round trips +
// barrier (parameters -> descriptor -> geometry, as the tick's staging).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k_empty(const int *, int *) {}
__global__ __launch_bounds__(1024) void k_one(const int *in, int *out)
{
    extern __shared__ int sm[];
    sm[threadIdx.x] = in[blockIdx.x * 1024 + threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = sm[1023];
}
__global__ __launch_bounds__(1024) void k_three(const int *in, int *out)
{
    extern __shared__ int sm[];
    int a = in[blockIdx.x];                       // "parameters"
    int b = in[(a & 1023) + 4096];                // "descriptor"
    sm[threadIdx.x] = in[((b & 255) * 1024 + threadIdx.x) & 0xFFFFF];   // "geometry"
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = sm[1023];
}
template <class F> void run(const char *name, F f, size_t lds, const int *in, int *out)
{
    hipFuncSetAttribute((const void *)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float tot = 0; const int n = 200;
    for (int i = 0; i < n + 20; i++) {
        hipExtLaunchKernelGGL(f, dim3(256), dim3(1024), lds, 0, e0, e1, 0, in, out);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (i >= 20) tot += ms;
    }
    printf("%-44s %6.2f us (dynamic LDS %zu KB)\n", name, 1e3 * tot / n, lds / 1024);
}
int main()
{
    int *in, *out; hipMalloc(&in, 4 << 20); hipMalloc(&out, 4096); hipMemset(in, 0, 4 << 20);
    for (size_t lds : {(size_t)4096, (size_t)150 * 1024}) {
        run("empty kernel", k_empty, lds, in, out);
        run("one global round trip + barrier", k_one, lds, in, out);
        run("three dependent round trips + barrier", k_three, lds, in, out);
    }
    return 0;
}
