// k_tile_steps: one translation unit (launch.hpp)
#include <algorithm>
#include <cstdio>
#include "kernels.hpp"
#include "launch.hpp"

namespace vxh {

// (one table of granted dynamic-LDS sizes per kernel instance, shared by the launch and by the occupancy query: the limit only ever grows)
template <bool TABG, bool MESH, bool FLUID>
static size_t (&granted_lds())[64] { static size_t table[64] = {}; return table; }

template <bool TABG, bool MESH, bool FLUID>
static void launch_tiles(const DBatch& B, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters, unsigned gen)
{
    grant_dynamic_lds((const void*)k_tile_steps<TABG, MESH, FLUID>, granted_lds<TABG, MESH, FLUID>(), lds);
    hipLaunchKernelGGL((k_tile_steps<TABG, MESH, FLUID>), dim3(count), dim3(VXH_TILE_THREADS), lds, s, B, B.robot, B.tiles, list, cap, iters, gen);
}

void launch_tile_group(const DBatch& B, bool tabg, int mesh_kind, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters, unsigned gen)
{
    if (mesh_kind == 2) { if (tabg) launch_tiles<true, true, true>(B, list, count, lds, s, cap, iters, gen); else launch_tiles<false, true, true>(B, list, count, lds, s, cap, iters, gen); }
    else if (mesh_kind) { if (tabg) launch_tiles<true, true, false>(B, list, count, lds, s, cap, iters, gen); else launch_tiles<false, true, false>(B, list, count, lds, s, cap, iters, gen); }
    else if (tabg) launch_tiles<true, false, false>(B, list, count, lds, s, cap, iters, gen);
    else launch_tiles<false, false, false>(B, list, count, lds, s, cap, iters, gen);
}

int tile_threads() { return VXH_TILE_THREADS; }

// Workgroups of a k_tile_steps instance one CU keeps resident at `lds` bytes of dynamic LDS.  All tiles of a robot spin on each other, so
// an over-estimate ends in VXH_ROBOT_SYNC_TIMEOUT replays, an under-estimate only in smaller launches: the runtime's occupancy figure
// (queried with the kernel's dynamic-LDS limit already raised to `lds`, or it answers for the 64 KB default) is clamped to ONE -- four
// wavefronts of up to 512 registers each fill the register files of a CU whatever the LDS says (round 6; rounds 2-5: five wavefronts of
// 256, "two per CU by the registers" was already one in practice) -- and a failing query is reported once instead of silently read as 1.
template <bool TABG, bool MESH, bool FLUID>
static int occupancy_of(size_t lds)
{
    grant_dynamic_lds((const void*)k_tile_steps<TABG, MESH, FLUID>, granted_lds<TABG, MESH, FLUID>(), lds);
    int n = 0;
    const hipError_t err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_tile_steps<TABG, MESH, FLUID>, VXH_TILE_THREADS, lds);
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
    const int n = mesh == 2 ? (tabg ? occupancy_of<true, true, true>(lds) : occupancy_of<false, true, true>(lds))
                : mesh == 1 ? (tabg ? occupancy_of<true, true, false>(lds) : occupancy_of<false, true, false>(lds))
                            : (tabg ? occupancy_of<true, false, false>(lds) : occupancy_of<false, false, false>(lds));
    return std::min(1, std::max(1, n));
}

}  // namespace vxh
