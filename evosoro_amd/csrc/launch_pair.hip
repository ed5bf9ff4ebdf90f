// k_robot_pair (developer library, -DVXH_PAIR): one translation unit (launch.hpp)
#include "kernels.hpp"
#include "launch.hpp"

namespace vxh {
#ifdef VXH_PAIR
template <bool TABG, bool SEL>
static void launch_pair(const DBatch& B, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters)
{
    static size_t granted[64] = {};
    grant_dynamic_lds((const void*)k_robot_pair<TABG, SEL>, granted, lds);
    hipLaunchKernelGGL((k_robot_pair<TABG, SEL>), dim3(count), dim3(VXH_PAIR_T), lds, s, B, B.robot, list, cap, iters, (int)(lds / 8));
}

void launch_pair_group(const DBatch& B, bool tabg, bool sel, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters)
{
    if (sel) { if (tabg) launch_pair<true, true>(B, list, count, lds, s, cap, iters); else launch_pair<false, true>(B, list, count, lds, s, cap, iters); }
    else { if (tabg) launch_pair<true, false>(B, list, count, lds, s, cap, iters); else launch_pair<false, false>(B, list, count, lds, s, cap, iters); }
}
#endif
}  // namespace vxh
