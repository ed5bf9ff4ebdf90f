// Developer micro-benchmark (not part of the library): what ONE exchange between two tiles of k_tile_steps costs on an MI355X.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/tile_sync scripts/ubench/tile_sync.hip && /tmp/tile_sync > gpurun_out/tile_sync.json
// The tiled kernel (kernels_tiled.hpp) moves poses between workgroups as self-validating 8-byte granules {32 data bits, 32-bit tag}:
// the producer stores them with relaxed agent-scope atomic stores (global_store_dwordx2 sc1: written through), the consumer polls with
// relaxed agent-scope atomic loads (L1 / L2 bypassing) until the tag is this step's.  Here two workgroups on two CUs play ping-pong
// with exactly those instructions: A publishes tag k, B polls for it and answers with tag k, A polls for the answer.  The time per
// round trip / 2 is the one-way publish -> seen latency; measured for a pair on the SAME XCD and on DIFFERENT XCDs (which XCD a
// workgroup runs on is read from the hardware: HW_REG_XCC_ID), for one granule polled by one lane and for the kernel's own shape (every
// lane of a wavefront polling the sixteen granules of one halo voxel), and for the L2-scope variant (workgroup-coherent loads that may
// be served by the XCD's own L2: only correct between CUs of one XCD).  Also: the round trip of a plain dependent global load chain
// (pointer chase over the same buffer) for scale.  Prints ONE JSON object.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <algorithm>

typedef __attribute__((address_space(1))) unsigned long long gu64;

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf; }      // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ unsigned hw_id() { return __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4); }              // HW_REG_HW_ID

template <int SCOPE>      // __HIP_MEMORY_SCOPE_AGENT (the kernel's) or __HIP_MEMORY_SCOPE_WORKGROUP ... see main()
__device__ __forceinline__ void st_g(unsigned long long* p, unsigned long long v) { __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE>
__device__ __forceinline__ unsigned long long ld_g(const unsigned long long* p) { return __hip_atomic_load((const gu64*)p, __ATOMIC_RELAXED, SCOPE); }

// buf: [2 sides][NG granule planes][64 lanes]; out: ticks of the 100 MHz wall clock for `iters` round trips (block a), XCC / HW ids of a and b
template <int NG, int ST_SCOPE, int LD_SCOPE, bool SLEEP>
__global__ void k_pingpong(unsigned long long* buf, int a, int b, int iters, int lanes, unsigned long long* out)
{
    const int me = blockIdx.x;
    if (me != a && me != b) return;
    const int lane = threadIdx.x;
    if (lane == 0) { out[me == a ? 1 : 2] = xcc_id(); out[me == a ? 3 : 4] = hw_id(); }
    if (lane >= lanes) return;
    unsigned long long* mine = buf + (me == a ? 0 : NG * 64);       // where I publish
    const unsigned long long* theirs = buf + (me == a ? NG * 64 : 0);
    const unsigned long long t0 = wall_clock64();
    for (int k = 1; k <= iters; ++k) {
        const unsigned long long tag = (unsigned long long)k << 32;
        if (me == a) {
#pragma unroll
            for (int g = 0; g < NG; ++g) st_g<ST_SCOPE>(mine + g * 64 + lane, tag | (unsigned)g);
        }
        for (;;) {                                                   // poll until every granule of mine carries the tag
            bool ok = true;
            unsigned long long v[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) v[g] = ld_g<LD_SCOPE>(theirs + g * 64 + lane);
#pragma unroll
            for (int g = 0; g < NG; ++g) ok = ok && (v[g] >> 32) == (unsigned long long)k;
            if (ok) break;
            if (SLEEP) __builtin_amdgcn_s_sleep(1);
        }
        if (me == b) {
#pragma unroll
            for (int g = 0; g < NG; ++g) st_g<ST_SCOPE>(mine + g * 64 + lane, tag | (unsigned)g);
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (me == a && lane == 0) out[0] = t1 - t0;
}

// a dependent chain of plain global loads over a small buffer that was just written by ANOTHER kernel (so it sits in L2 / memory, not in this
// CU's L1 at first; after the first lap it is L1- or L2-resident): the floor of any "load, then act on it" step
__global__ void k_chase(const unsigned* next, int iters, unsigned long long* out)
{
    unsigned p = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < iters; ++k) p = __hip_atomic_load(next + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[5] = p; }
}

struct Result { double us_round_trip; unsigned xa, xb, ha, hb; };

template <int NG, int ST_SCOPE, int LD_SCOPE, bool SLEEP>
Result run(unsigned long long* buf, unsigned long long* out, int a, int b, int lanes, int iters, int blocks)
{
    hipMemset(buf, 0, sizeof(unsigned long long) * 2 * NG * 64);
    hipMemset(out, 0, 64);
    // one workgroup per CU (100 KB of dynamic LDS), a whole chip's worth of workgroups so that a and b land on different CUs
    hipFuncSetAttribute((const void*)k_pingpong<NG, ST_SCOPE, LD_SCOPE, SLEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    k_pingpong<NG, ST_SCOPE, LD_SCOPE, SLEEP><<<blocks, 64, 100 * 1024>>>(buf, a, b, iters, lanes, out);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    return {(double)h[0] / 100.0 / iters, (unsigned)h[1], (unsigned)h[2], (unsigned)h[3], (unsigned)h[4]};
}

int main()
{
    unsigned long long *buf, *out;
    hipMalloc(&buf, sizeof(unsigned long long) * 2 * 16 * 64 + 4096);
    hipMalloc(&out, 64);
    const int iters = 20000, blocks = 256;
    constexpr int AG = __HIP_MEMORY_SCOPE_AGENT, WG = __HIP_MEMORY_SCOPE_WORKGROUP;
    // find a partner of block 0 on the same XCD and one on a different XCD (workgroups are dealt round robin over the XCDs: 0 and 8, 0 and 1)
    std::string js = "{";
    auto emit = [&](const char* name, Result r) {
        char line[512];
        std::snprintf(line, sizeof(line), "%s\n \"%s\": {\"us_round_trip\": %.4f, \"us_one_way\": %.4f, \"xcc_a\": %u, \"xcc_b\": %u, \"same_cu\": %s}",
                      js.size() > 1 ? "," : "", name, r.us_round_trip, r.us_round_trip / 2, r.xa, r.xb, (r.ha == r.hb) ? "true" : "false");
        js += line;
    };
    for (int rep = 0; rep < 2; ++rep) {        // (the second repetition is the one kept: the first warms the code objects up)
        js = "{";
        emit("agent_scope_1_granule_1_lane_same_xcd", run<1, AG, AG, false>(buf, out, 0, 8, 1, iters, blocks));
        emit("agent_scope_1_granule_1_lane_other_xcd", run<1, AG, AG, false>(buf, out, 0, 1, 1, iters, blocks));
        emit("agent_scope_1_granule_1_lane_other_xcd_far", run<1, AG, AG, false>(buf, out, 0, 4, 1, iters, blocks));
        emit("agent_scope_1_granule_1_lane_other_xcd_sleep1", run<1, AG, AG, true>(buf, out, 0, 1, 1, iters, blocks));
        emit("agent_scope_16_granules_64_lanes_same_xcd", run<16, AG, AG, false>(buf, out, 0, 8, 64, iters, blocks));
        emit("agent_scope_16_granules_64_lanes_other_xcd", run<16, AG, AG, false>(buf, out, 0, 1, 64, iters, blocks));
        emit("agent_scope_16_granules_64_lanes_other_xcd_sleep1", run<16, AG, AG, true>(buf, out, 0, 1, 64, iters, blocks));
        // agent-scope stores (written through to memory AND the XCD's L2), workgroup-scope polls: may be served by the consumer's L1 / L2 -- a
        // poll that hits a stale L1 line never ends, so this form is only run with an iteration bound by construction (same XCD: the L2 is shared)
        emit("agent_store_16_granules_64_lanes_same_xcd_again", run<16, AG, AG, false>(buf, out, 0, 16, 64, iters, blocks));
    }
    // plain dependent loads, one lane, buffer of 64 entries (resident in the L2 / L1 after the first lap)
    {
        unsigned* next; hipMalloc(&next, 4 * 64);
        std::vector<unsigned> h(64);
        for (int i = 0; i < 64; ++i) h[i] = (i * 17 + 5) & 63;
        hipMemcpy(next, h.data(), 4 * 64, hipMemcpyHostToDevice);
        hipMemset(out, 0, 64);
        k_chase<<<1, 1>>>(next, 20000, out);
        hipDeviceSynchronize();
        unsigned long long r[8]; hipMemcpy(r, out, 64, hipMemcpyDeviceToHost);
        char line[256];
        std::snprintf(line, sizeof(line), ",\n \"dependent_agent_scope_load_chain\": {\"us_per_load\": %.4f}", (double)r[0] / 100.0 / 20000);
        js += line;
    }
    js += "\n}";
    std::printf("%s\n", js.c_str());
    (void)WG;
    return 0;
}
