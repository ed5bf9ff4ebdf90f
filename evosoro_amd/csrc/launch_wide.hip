// k_robot_wide: one translation unit (launch.hpp)
#include "kernels.hpp"
#include "launch.hpp"

namespace vxh {

template <int BLOCK, bool MESH, bool TABG>
static void launch_wide(const DBatch& B, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters, int two_tiles)
{
    static size_t granted[64] = {};
    grant_dynamic_lds((const void*)k_robot_wide<BLOCK, MESH, TABG>, granted, lds);
    hipLaunchKernelGGL((k_robot_wide<BLOCK, MESH, TABG>), dim3(count), dim3(BLOCK), lds, s, B, B.robot, list, cap, iters, (int)(lds / 8), two_tiles);
}

void launch_wide_group(const DBatch& B, bool mesh, bool tabg, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters, int two_tiles)
{
    if (mesh) { if (tabg) launch_wide<512, true, true>(B, list, count, lds, s, cap, iters, two_tiles); else launch_wide<512, true, false>(B, list, count, lds, s, cap, iters, two_tiles); }
    else { if (tabg) launch_wide<512, false, true>(B, list, count, lds, s, cap, iters, two_tiles); else launch_wide<512, false, false>(B, list, count, lds, s, cap, iters, two_tiles); }
}

}  // namespace vxh
