// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for this repo's access pattern: 8 bytes per lane, coalesced
// (global_load_dwordx2 / global_store_dwordx2), known byte counts.  Build: hipcc --offload-arch=gfx950 -O3 -o calib_stream calib_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_read8(const double* __restrict__ a, double* out, size_t n)
{
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s == 123.456) out[0] = s;
}
__global__ void calib_write8(double* a, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (double)i;
}
int main()
{
    const size_t n = (size_t)1 << 28;   // 2 GiB of doubles: far beyond L2 and the 256 MiB Infinity Cache
    double *a, *o;
    hipMalloc(&a, n * 8); hipMalloc(&o, 8);
    hipMemset(a, 0, n * 8);
    hipLaunchKernelGGL(calib_write8, dim3(4096), dim3(256), 0, 0, a, n);
    hipLaunchKernelGGL(calib_read8, dim3(4096), dim3(256), 0, 0, a, o, n);
    hipDeviceSynchronize();
    printf("bytes %zu\n", n * 8);
    return 0;
}
