// Cost of a grid-wide barrier in a cooperative launch (one 1024-thread workgroup per CU) and of a
// hand-written arrive/spin barrier, per iteration, with a little dependent global traffic in between.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(1024) void k_cg(int *data, int iters) {
    cg::grid_group g = cg::this_grid();
    for (int it = 0; it < iters; ++it) {
        if (threadIdx.x == 0) atomicAdd(&data[(blockIdx.x + it) % gridDim.x], 1);
        g.sync();
    }
}
// hand-written: monotone counter, every workgroup's thread 0 arrives and spins until all have
__global__ __launch_bounds__(1024) void k_manual(int *data, unsigned int *bar, int iters) {
    for (int it = 0; it < iters; ++it) {
        if (threadIdx.x == 0) atomicAdd(&data[(blockIdx.x + it) % gridDim.x], 1);
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned int target = (unsigned int)(it + 1) * gridDim.x;
            atomicAdd(bar, 1u);
            long spins = 0;
            while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < 100000000L) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}
int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount;
    int *d; unsigned int *bar;
    CK(hipMalloc(&d, sizeof(int) * grid)); CK(hipMemset(d, 0, sizeof(int) * grid));
    CK(hipMalloc(&bar, 64)); CK(hipMemset(bar, 0, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int iters : {1, 101, 1001}) {
        void *args[] = {&d, &iters};
        CK(hipLaunchCooperativeKernel((void *)k_cg, dim3(grid), dim3(1024), args, 0, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        CK(hipLaunchCooperativeKernel((void *)k_cg, dim3(grid), dim3(1024), args, 0, 0));
        CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("cooperative groups grid.sync: grid %d, %4d iterations: %.1f us total, %.2f us per iteration\n", grid, iters, ms * 1000, ms * 1000 / iters);
        CK(hipMemset(bar, 0, 64));
        CK(hipEventRecord(a));
        k_manual<<<grid, 1024>>>(d, bar, iters);
        CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, a, b));
        printf("hand-written arrive/spin    : grid %d, %4d iterations: %.1f us total, %.2f us per iteration\n", grid, iters, ms * 1000, ms * 1000 / iters);
    }
    return 0;
}
