// Clock probe: ratio of shader clock (s_memtime) to the 100 MHz real-time counter inside a
// short latency-bound kernel launched like the relax kernels (many tiny launches).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void probe(unsigned long long* out, int iters) {
    __shared__ int x;
    if (threadIdx.x == 0) x = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int v = 0;
    for (int i = 0; i < iters; ++i) { v += atomicAdd(&x, 1); }   // dependent LDS atomic round trips
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = v; }
}
__global__ void probe_alu(unsigned long long* out, int iters, float a) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float v = a;
    for (int i = 0; i < iters; ++i) { v = v * 1.0001f + 0.5f; v = __builtin_amdgcn_sqrtf(v); }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)v; }
}
int main() {
    unsigned long long *d, h[3];
    hipMalloc(&d, 24);
    for (int rep = 0; rep < 3; ++rep) {
        for (int busy = 0; busy < 2; ++busy) {
            // idle-ish: single tiny launches with host sync in between; busy: 2000 back-to-back
            int n = busy ? 2000 : 1;
            auto w0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; ++i) probe<<<16, 64>>>(d, 1000);
            hipDeviceSynchronize();
            auto w1 = std::chrono::steady_clock::now();
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            printf("lds-atomic chain: %s: shader ticks %llu, realtime ticks(100MHz) %llu -> clock %.0f MHz, %.1f cycles per LDS atomic round trip, wall/launch %.1f us\n",
                   busy ? "back-to-back" : "single", h[0], h[1], 100.0 * h[0] / h[1], (double)h[0] / 1000.0,
                   std::chrono::duration<double, std::micro>(w1 - w0).count() / n);
            probe_alu<<<16, 64>>>(d, 1000, 3.0f);
            hipDeviceSynchronize();
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            printf("alu chain (mul,add,sqrt dependent): clock %.0f MHz, %.1f cycles per iteration\n", 100.0 * h[0] / h[1], (double)h[0] / 1000.0);
        }
    }
    return 0;
}
