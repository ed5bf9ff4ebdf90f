// Developer micro-benchmark (not part of the library): what one SIMD of an MI355X CU does with FP64 vector instructions.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fp64_issue scripts/ubench/fp64_issue.hip && /tmp/fp64_issue
// Prints, for 1..4 wavefronts per SIMD and dependent chains of ILP 1 / 2 / 4: core cycles per wave-instruction, and the core clock
// (s_memtime against the 100 MHz s_memrealtime) while every CU is busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ void k_chain(double* out, unsigned long long* cyc, unsigned long long* wall, int iters, double a, double b)
{
    double x[ILP];
    for (int k = 0; k < ILP; ++k) x[k] = threadIdx.x * 1e-3 + k;
    __syncthreads();
    const unsigned long long w0 = wall_clock64();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int k = 0; k < ILP; ++k) x[k] = __builtin_fma(x[k], a, b);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    double s = 0;
    for (int k = 0; k < ILP; ++k) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    // the slowest wavefront of the workgroup counts (the issue arbiter favours the oldest one)
    __shared__ unsigned long long s_c, s_w;
    if (threadIdx.x == 0) { s_c = 0; s_w = 0; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { atomicMax(&s_c, t1 - t0); atomicMax(&s_w, w1 - w0); }
    __syncthreads();
    if (threadIdx.x == 0) { cyc[blockIdx.x] = s_c; wall[blockIdx.x] = s_w; }
}

template <int ILP>
void run(int waves_per_simd, int blocks)
{
    const int threads = 64 * 4 * waves_per_simd, iters = 4000;
    double* out; unsigned long long *cyc, *wall;
    hipMalloc(&out, sizeof(double) * blocks * threads); hipMalloc(&cyc, 8 * blocks); hipMalloc(&wall, 8 * blocks);
    // 100 KB of dynamic LDS: one workgroup per CU
    hipFuncSetAttribute((const void*)k_chain<ILP>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 2; ++rep) k_chain<ILP><<<blocks, threads, 100 * 1024>>>(out, cyc, wall, iters, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks), w(blocks);
    hipMemcpy(c.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost); hipMemcpy(w.data(), wall, 8 * blocks, hipMemcpyDeviceToHost);
    double cs = 0, ws = 0;
    for (int i = 0; i < blocks; ++i) { cs += c[i]; ws += w[i]; }
    const double n_inst = (double)iters * 16 * ILP;
    printf("blocks %4d waves/SIMD %d ILP %d: %.2f ticks per instruction of a wave, %.2f ticks per wave-instruction of the SIMD, counter rate %.0f MHz\n",
           blocks, waves_per_simd, ILP, cs / blocks / n_inst, cs / blocks / n_inst / waves_per_simd, cs / ws * 100.0);
    hipFree(out); hipFree(cyc); hipFree(wall);
}

int main()
{
    for (int blocks : {1, 256})
        for (int w = 1; w <= 4; ++w) { run<1>(w, blocks); run<2>(w, blocks); run<4>(w, blocks); }
    return 0;
}
