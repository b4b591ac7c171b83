// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters for THIS engine's access pattern
// (one dword per lane, 128-byte row segments), as MI355X_MICROARCH.md (HBM section) prescribes:
// stream a buffer far larger than the 256 MiB Infinity Cache with dword loads / stores and compare
// the counter with the known byte count.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_read_dword(const float *in, float *out, size_t n) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
    if (acc == 12345.678f) out[0] = acc;   // keep the loads alive
}
__global__ void calib_write_dword(float *out, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = v;
}
int main() {
    const size_t n = (size_t)1 << 28;   // 1 GiB of floats
    float *a, *b;
    if (hipMalloc(&a, n * 4) != hipSuccess || hipMalloc(&b, 4096) != hipSuccess) return 1;
    calib_write_dword<<<2048, 256>>>(a, n, 1.0f);
    hipDeviceSynchronize();
    calib_read_dword<<<2048, 256>>>(a, b, n);
    hipDeviceSynchronize();
    std::printf("calib: read %zu bytes with dword loads, wrote %zu bytes with dword stores\n", n * 4, n * 4);
    return 0;
}
