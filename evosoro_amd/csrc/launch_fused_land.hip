// k_robot_steps, the _voxcad (no surface mesh) variants: one translation unit (launch.hpp)
#include "kernels.hpp"
#include "launch.hpp"

namespace vxh {

template <int BLOCK, int NACC, bool TABG>
static void launch_variant(const DBatch& B, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters,
                           const unsigned long long* order_in, unsigned long long* order_out)
{
    static size_t granted[64] = {};
    grant_dynamic_lds((const void*)k_robot_steps<BLOCK, NACC, false, TABG>, granted, lds);
    hipLaunchKernelGGL((k_robot_steps<BLOCK, NACC, false, TABG>), dim3(count), dim3(BLOCK), lds, s, B, B.robot, list, cap, iters, (int)(lds / 8), order_in, order_out);
}

template <bool TABG>
static void launch_sized(const DBatch& B, int block, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters,
                         const unsigned long long* order_in, unsigned long long* order_out)
{
    if (block == 256) launch_variant<256, 2, TABG>(B, list, count, lds, s, cap, iters, order_in, order_out);
    else if (block == 512) launch_variant<512, 2, TABG>(B, list, count, lds, s, cap, iters, order_in, order_out);
    else if (block == 768) launch_variant<768, 2, TABG>(B, list, count, lds, s, cap, iters, order_in, order_out);
    else launch_variant<1024, 1, TABG>(B, list, count, lds, s, cap, iters, order_in, order_out);
}

void launch_fused_land(const DBatch& B, int block, bool tabg, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters,
            const unsigned long long* order_in, unsigned long long* order_out)
{
    if (tabg) launch_sized<true>(B, block, list, count, lds, s, cap, iters, order_in, order_out);
    else launch_sized<false>(B, block, list, count, lds, s, cap, iters, order_in, order_out);
}

}  // namespace vxh
