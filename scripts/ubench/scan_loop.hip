// Developer micro-benchmark (not part of the library): the common path of the resident kernel's collision broad-phase in isolation.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/scan_loop scripts/ubench/scan_loop.hip && scripts/ubench/scan_loop
// One workgroup of 768 threads per CU (135 KB of LDS, like k_robot_steps<768>); thread i < ns tests all ns candidates against its own
// position: squared distance, compare with a threshold nothing passes.  Variants: candidate poses read by every lane from LDS
// (broadcast reads, as the committed scan does: index, then x y z), with and without the indirection through the index array; and
// the arithmetic alone (candidate = a function of j in registers).  Prints cycles per candidate and wavefront.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int BLOCK = 768;

template <int MODE>      // 0: shi[j] -> ps[...] (committed scan)  1: ps[j] directly  2: no memory at all
__global__ __launch_bounds__(BLOCK, 1) void k_scan(int ns, double thresh, int* out_cnt, unsigned long long* out_cyc)
{
    extern __shared__ double lds[];
    double* ps = lds;                    // [4][BLOCK]
    int* shi = (int*)(lds + 4 * BLOCK);  // [BLOCK]
    const int tid = threadIdx.x;
    ps[tid] = 1e-3 * (tid % 10); ps[BLOCK + tid] = 1e-3 * ((tid / 10) % 10); ps[2 * BLOCK + tid] = 1e-3 * (tid / 100); ps[3 * BLOCK + tid] = 1e-3;
    shi[tid] = (tid * 7) % BLOCK;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    int cnt = 0;
    if (tid < ns) {
        const double px = ps[tid], py = ps[BLOCK + tid], pz = ps[2 * BLOCK + tid];
        for (int j0 = 0; j0 < ns; j0 += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                double qx, qy, qz;
                if (MODE == 0) { const int l = shi[j < ns ? j : ns - 1] & 1023; qx = ps[l]; qy = ps[BLOCK + l]; qz = ps[2 * BLOCK + l]; }
                else if (MODE == 1) { const int l = j < ns ? j : ns - 1; qx = ps[l]; qy = ps[BLOCK + l]; qz = ps[2 * BLOCK + l]; }
                else { qx = 1e-3 * (j % 10); qy = 1e-3 * ((j / 10) % 10); qz = 1e-3 * (j / 100); }
                if (j >= ns || j == tid) continue;
                const double dx = px - qx, dy = py - qy, dz = pz - qz;
                const double d2 = dx * dx + dy * dy + dz * dz;
                if (!(d2 < thresh)) continue;
                ++cnt;
            }
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    out_cnt[blockIdx.x * BLOCK + tid] = cnt;
    if (tid == 0) out_cyc[blockIdx.x] = t1 - t0;
}

// MODE 3: the committed data path (LDS broadcast reads through the index array), but NO control flow per candidate: the three tests
// as integer arithmetic on compare results, summed.  MODE 4: the same, and a (never taken) branch per group of eight on their OR.
template <int MODE>
__global__ __launch_bounds__(BLOCK, 1) void k_scan_flat(int ns, double thresh, int* out_cnt, unsigned long long* out_cyc)
{
    extern __shared__ double lds[];
    double* ps = lds;
    int* shi = (int*)(lds + 4 * BLOCK);
    const int tid = threadIdx.x;
    ps[tid] = 1e-3 * (tid % 10); ps[BLOCK + tid] = 1e-3 * ((tid / 10) % 10); ps[2 * BLOCK + tid] = 1e-3 * (tid / 100); ps[3 * BLOCK + tid] = 1e-3;
    shi[tid] = (tid * 7) % BLOCK;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    int cnt = 0;
    if (tid < ns) {
        const double px = ps[tid], py = ps[BLOCK + tid], pz = ps[2 * BLOCK + tid];
        for (int j0 = 0; j0 < ns; j0 += 8) {
            int any = 0, pass[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                const int l = MODE == 5 ? j : (shi[j < ns ? j : ns - 1] & 1023);      // (5: staged by ordinal; past the end: inside the array)
                const double dx = px - ps[l], dy = py - ps[BLOCK + l], dz = pz - ps[2 * BLOCK + l];
                const double d2 = dx * dx + dy * dy + dz * dz;
                pass[u] = (int)(d2 < thresh) & (int)(j != tid) & (int)(j < ns);
                any |= pass[u];
            }
            if (MODE == 3) {
#pragma unroll
                for (int u = 0; u < 8; ++u) cnt += pass[u];
            } else if (any) {                  // (rare in the real scan: the accepted-pair work would sit here)
#pragma unroll
                for (int u = 0; u < 8; ++u) cnt += pass[u];
            }
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    out_cnt[blockIdx.x * BLOCK + tid] = cnt;
    if (tid == 0) out_cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run_flat(int blocks, int ns)
{
    int* cnt; unsigned long long* cyc;
    hipMalloc(&cnt, sizeof(int) * blocks * BLOCK); hipMalloc(&cyc, 8 * blocks);
    const size_t lds = 135 * 1024;
    hipFuncSetAttribute((const void*)k_scan_flat<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) k_scan_flat<MODE><<<blocks, BLOCK, lds>>>(ns, -1.0, cnt, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : c) s += (double)v;
    printf("%-68s %4d workgroups: %8.0f cycles per run = %6.1f per candidate and wavefront\n",
           MODE == 3 ? "committed data path, tests as arithmetic, no branch" : (MODE == 4 ? "committed data path, tests as arithmetic, one branch per eight" : "poses staged by ordinal, tests as arithmetic, one branch per eight"), blocks, s / blocks, s / blocks / ns);
    hipFree(cnt); hipFree(cyc);
}

template <int MODE>
void run(int blocks, int ns)
{
    int* cnt; unsigned long long* cyc;
    hipMalloc(&cnt, sizeof(int) * blocks * BLOCK); hipMalloc(&cyc, 8 * blocks);
    const size_t lds = 135 * 1024;
    hipFuncSetAttribute((const void*)k_scan<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) k_scan<MODE><<<blocks, BLOCK, lds>>>(ns, -1.0, cnt, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : c) s += (double)v;
    static const char* what[3] = {"LDS broadcast reads through the index array (the committed scan)", "LDS broadcast reads, no index array", "arithmetic only"};
    printf("%-68s %4d workgroups: %8.0f cycles per run = %6.1f per candidate and wavefront\n", what[MODE], blocks, s / blocks, s / blocks / ns);
    hipFree(cnt); hipFree(cyc);
}

int main()
{
    const int ns = 655;
    for (int blocks : {1, 256}) { run<0>(blocks, ns); run<1>(blocks, ns); run<2>(blocks, ns); run_flat<3>(blocks, ns); run_flat<4>(blocks, ns); run_flat<5>(blocks, ns); }
    return 0;
}
