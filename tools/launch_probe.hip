// Launch-latency probe: dependent tiny kernels on one stream, launched one by one vs as a captured
// graph.  Prints us per kernel (wall) for both, and the GPU-side span from events.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_tiny(int *p, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += v; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    int *d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int N = 32, R = 200;
    for (int grid : {1, 512}) {
        for (int w = 0; w < 50; ++w) k_tiny<<<grid, 1024, 0, s>>>(d, 1);
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(a, s));
        for (int r = 0; r < R; ++r) for (int i = 0; i < N; ++i) k_tiny<<<grid, 1024, 0, s>>>(d, 1);
        CK(hipEventRecord(b, s));
        auto t1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(s));
        auto t2 = std::chrono::steady_clock::now();
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("grid %3d direct : host enqueue %.2f us/launch, wall %.2f us/launch, gpu span %.2f us/launch\n", grid,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / (N * R), std::chrono::duration<double, std::micro>(t2 - t0).count() / (N * R), ms * 1000 / (N * R));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) k_tiny<<<grid, 1024, 0, s>>>(d, 1);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(a, s));
        for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(b, s));
        t1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(s));
        t2 = std::chrono::steady_clock::now();
        CK(hipEventElapsedTime(&ms, a, b));
        printf("grid %3d graph  : host enqueue %.2f us/kernel, wall %.2f us/kernel, gpu span %.2f us/kernel\n", grid,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / (N * R), std::chrono::duration<double, std::micro>(t2 - t0).count() / (N * R), ms * 1000 / (N * R));
        // one graph launch, synchronous: latency of a 32-kernel replan-like chain
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 50; ++r) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
        t1 = std::chrono::steady_clock::now();
        printf("grid %3d graph  : launch + sync of one 32-kernel graph: %.1f us\n", grid, std::chrono::duration<double, std::micro>(t1 - t0).count() / 50);
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 50; ++r) { for (int i = 0; i < N; ++i) k_tiny<<<grid, 1024, 0, s>>>(d, 1); CK(hipStreamSynchronize(s)); }
        t1 = std::chrono::steady_clock::now();
        printf("grid %3d direct : 32 launches + sync: %.1f us\n", grid, std::chrono::duration<double, std::micro>(t1 - t0).count() / 50);
    }
    return 0;
}
