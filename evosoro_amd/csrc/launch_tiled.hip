// k_tile_steps: one translation unit (launch.hpp)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include "kernels.hpp"
#include "launch.hpp"

namespace vxh {

// (one table of granted dynamic-LDS sizes per kernel instance, shared by the launch and by the occupancy query: the limit only ever grows)
template <bool TABG, bool MESH, bool FLUID, bool SMALL>
static size_t (&granted_lds())[64] { static size_t table[64] = {}; return table; }

// k_tile_steps must keep NOTHING in scratch (round 6).  Its tiles exchange poses through relaxed agent-scope stores and polled loads, and
// every scratch access queues behind those in the wavefront's one in-order memory counter (a reload right after the publication waited
// for the write-through stores to be acknowledged: ~1 us of a 7.9-us step, rounds 2-5).  And it is where round 5's unexplained abort
// lived: a build of the FLUID variant with ~500 bytes of scratch per lane stored granules through address registers reloaded from scratch
// whose upper lanes had never been written (rocgdb, precise memory: lanes 48-62 of the faulting global_store's address pair held garbage
// -- a slot filled under a narrower lane mask than the one it was read back under), a memory access fault at the first step (HISTORY
// "Round 6").  Four wavefronts per workgroup leave every wavefront 512 registers, the compiler spills into the upper 256 (AGPRs), and
// this check keeps it that way: a build that needs scratch again is refused with VXH_ERR_HIP before it is launched.
// (VXH_TILE_SCRATCH_LIMIT: bytes per lane tolerated, default 0; the GPU test sets -1 to see the refusal.)
template <bool TABG, bool MESH, bool FLUID, bool SMALL>
static void refuse_scratch()
{
    static std::mutex lock;
    static long bytes[64];
    static bool known[64] = {};
    int dev = 0;
    hip_check(hipGetDevice(&dev), "hipGetDevice");
    dev = dev >= 0 && dev < 64 ? dev : 0;
    long have;
    {
        std::lock_guard<std::mutex> hold(lock);
        if (!known[dev]) {
            hipFuncAttributes attr;
            hip_check(hipFuncGetAttributes(&attr, (const void*)k_tile_steps<TABG, MESH, FLUID, SMALL>), "hipFuncGetAttributes(k_tile_steps)");
            bytes[dev] = (long)attr.localSizeBytes; known[dev] = true;
        }
        have = bytes[dev];
    }
    const char* env = std::getenv("VXH_TILE_SCRATCH_LIMIT");
    const long limit = env ? std::atol(env) : 0;
    if (have > limit)
        throw std::runtime_error("HIP: k_tile_steps<" + std::to_string((int)TABG) + ", " + std::to_string((int)MESH) + ", " + std::to_string((int)FLUID) + ", " + std::to_string((int)SMALL) + "> was compiled with " +
                                 std::to_string(have) + " bytes of scratch per lane (limit " + std::to_string(limit) + "): refused, see evosoro_amd/csrc/launch_tiled.hip");
}

template <bool TABG, bool MESH, bool FLUID, bool SMALL>
static void launch_tiles(const DBatch& B, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters, unsigned gen)
{
    refuse_scratch<TABG, MESH, FLUID, SMALL>();
    grant_dynamic_lds((const void*)k_tile_steps<TABG, MESH, FLUID, SMALL>, granted_lds<TABG, MESH, FLUID, SMALL>(), lds);
    hipLaunchKernelGGL((k_tile_steps<TABG, MESH, FLUID, SMALL>), dim3(count), dim3(VXH_TILE_THREADS), lds, s, B, B.robot, B.tiles, list, cap, iters, gen);
}

template <bool SMALL>
static void launch_tile_group_of(const DBatch& B, bool tabg, int mesh_kind, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters, unsigned gen)
{
    if (mesh_kind == 2) { if (tabg) launch_tiles<true, true, true, SMALL>(B, list, count, lds, s, cap, iters, gen); else launch_tiles<false, true, true, SMALL>(B, list, count, lds, s, cap, iters, gen); }
    else if (mesh_kind) { if (tabg) launch_tiles<true, true, false, SMALL>(B, list, count, lds, s, cap, iters, gen); else launch_tiles<false, true, false, SMALL>(B, list, count, lds, s, cap, iters, gen); }
    else if (tabg) launch_tiles<true, false, false, SMALL>(B, list, count, lds, s, cap, iters, gen);
    else launch_tiles<false, false, false, SMALL>(B, list, count, lds, s, cap, iters, gen);
}

void launch_tile_group(const DBatch& B, bool tabg, int mesh_kind, bool small, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters, unsigned gen)
{
    if (small) launch_tile_group_of<true>(B, tabg, mesh_kind, list, count, lds, s, cap, iters, gen);
    else launch_tile_group_of<false>(B, tabg, mesh_kind, list, count, lds, s, cap, iters, gen);
}

int tile_threads() { return VXH_TILE_THREADS; }

// Workgroups of a k_tile_steps instance one CU keeps resident at `lds` bytes of dynamic LDS.  All tiles of a robot spin on each other, so
// an over-estimate ends in VXH_ROBOT_SYNC_TIMEOUT replays, an under-estimate only in smaller launches: the runtime's occupancy figure
// (queried with the kernel's dynamic-LDS limit already raised to `lds`, or it answers for the 64 KB default) is clamped to ONE -- four
// wavefronts of up to 512 registers each fill the register files of a CU whatever the LDS says (round 6; rounds 2-5: five wavefronts of
// 256, "two per CU by the registers" was already one in practice) -- and a failing query is reported once instead of silently read as 1.
template <bool TABG, bool MESH, bool FLUID, bool SMALL>
static int occupancy_of(size_t lds)
{
    grant_dynamic_lds((const void*)k_tile_steps<TABG, MESH, FLUID, SMALL>, granted_lds<TABG, MESH, FLUID, SMALL>(), lds);
    int n = 0;
    const hipError_t err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_tile_steps<TABG, MESH, FLUID, SMALL>, VXH_TILE_THREADS, lds);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        static bool told = false;
        if (!told) { told = true; std::fprintf(stderr, "vxhip: hipOccupancyMaxActiveBlocksPerMultiprocessor(k_tile_steps, %zu B of LDS) failed (%s): one workgroup per CU assumed\n", lds, hipGetErrorString(err)); }
        return 1;
    }
    return n;
}

long long tile_workgroups_per_cu(int tabg, int mesh, size_t lds)
{
    // (the generic instances answer for their SMALL twins: same threads, same register budget)
    const int n = mesh == 2 ? (tabg ? occupancy_of<true, true, true, false>(lds) : occupancy_of<false, true, true, false>(lds))
                : mesh == 1 ? (tabg ? occupancy_of<true, true, false, false>(lds) : occupancy_of<false, true, false, false>(lds))
                            : (tabg ? occupancy_of<true, false, false, false>(lds) : occupancy_of<false, false, false, false>(lds));
    return std::min(1, std::max(1, n));
}

}  // namespace vxh
