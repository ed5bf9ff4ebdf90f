// Host-side launch entry points of the stepping kernels.  Each kernel family is its own translation unit (launch_fused_*.hip,
// launch_wide.hip, launch_tiled.hip, launch_pair.hip) so that the Makefile compiles them side by side and a change to one family
// recompiles that family only; engine.hip (batch assembly, the call loop, the streaming kernels) sees these declarations.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdexcept>
#include <string>
#include "device_types.hpp"

namespace vxh {

inline void hip_check(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string("HIP: ") + what + ": " + hipGetErrorString(e));
}

// The opt-in to more than 64 KB of dynamic LDS is per function AND per device (engines on several devices and threads may live in
// one process): the largest size granted on each device is remembered per kernel, under a lock.
inline void grant_dynamic_lds(const void* kernel, size_t (&granted)[64], size_t lds)
{
    static std::mutex lock;
    int dev = 0;
    hip_check(hipGetDevice(&dev), "hipGetDevice");
    std::lock_guard<std::mutex> hold(lock);
    if (dev < 0 || dev >= 64 || lds > granted[dev]) {
        hip_check(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(dynamic LDS)");
        if (dev >= 0 && dev < 64) granted[dev] = lds;
    }
}

// k_robot_steps<block, nacc, FLUID, TABG> (kernels_fused.hpp): block in {256, 512, 768, 1024}; land robots / land_water robots
void launch_fused_land(const DBatch& B, int block, bool tabg, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters,
                       const unsigned long long* order_in, unsigned long long* order_out);
void launch_fused_mesh(const DBatch& B, int block, bool tabg, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters,
                       const unsigned long long* order_in, unsigned long long* order_out);
// k_robot_wide<512, MESH, TABG> (kernels_wide.hpp)
void launch_wide_group(const DBatch& B, bool mesh, bool tabg, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters, int two_tiles);
// k_tile_steps<TABG, MESH, FLUID> (kernels_tiled.hpp); mesh_kind: 0 = _voxcad, 1 = land_water robot on land, 2 = in a fluid
// small: every tile of the launch within VXH_TILE_S_OWN / _HALO / _BONDS (device_types.hpp): the instances with compile-time LDS strides
void launch_tile_group(const DBatch& B, bool tabg, int mesh_kind, bool small, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters, unsigned gen);
// workgroups of that k_tile_steps instance one CU keeps resident at `lds` bytes of dynamic LDS
long long tile_workgroups_per_cu(int tabg, int mesh_kind, size_t lds);
// threads of a tile's workgroup (the host sizes launches and LDS with the kernel's own constants)
int tile_threads();
#ifdef VXH_PAIR
// k_robot_pair<TABG, SEL> (kernels_pair.hpp; developer library only)
void launch_pair_group(const DBatch& B, bool tabg, bool sel, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters);
#endif

}  // namespace vxh
