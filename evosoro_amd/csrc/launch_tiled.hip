// k_tile_steps: one translation unit (launch.hpp)
#include <algorithm>
#include "kernels.hpp"
#include "launch.hpp"

namespace vxh {

template <bool TABG, bool MESH, bool FLUID>
static void launch_tiles(const DBatch& B, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters, unsigned gen)
{
    static size_t granted[64] = {};
    grant_dynamic_lds((const void*)k_tile_steps<TABG, MESH, FLUID>, granted, lds);
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

long long tile_workgroups_per_cu(int tabg, int mesh, size_t lds)
{
    const void* f = mesh == 2 ? (tabg ? (const void*)k_tile_steps<true, true, true> : (const void*)k_tile_steps<false, true, true>)
                  : mesh == 1 ? (tabg ? (const void*)k_tile_steps<true, true, false> : (const void*)k_tile_steps<false, true, false>)
                              : (tabg ? (const void*)k_tile_steps<true, false, false> : (const void*)k_tile_steps<false, false, false>);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, VXH_TILE_THREADS, lds) != hipSuccess) { (void)hipGetLastError(); n = 1; }
    return std::max(1, n);
}

}  // namespace vxh
