// Plain structs shared by host code and HIP kernels (HBM layout, see DESIGN.md "Data layout").
#pragma once

namespace vxh {

// per-class constant tables: a handful of entries for a whole population (all robots of an evosoro run share
// one palette), read through the scalar/L1 caches instead of streaming ~170 B of constants per bond
struct DVoxClass {
    double mass, mass_inv, inertia_inv, c_lin, c_ang, E, k_floor, u_static, u_dynamic, cte, nom_size;
    int mat, pad;
};
struct DBondClass {
    double L, a1, a2, b1, b2, b3;
    double sq_a1m1, sq_a1m2, sq_a2i1, sq_a2i2, sq_b1m1, sq_b1m2, sq_b2fm1, sq_b2fm2, sq_b3i1, sq_b3i2;
    double stress_E1, stress_E2, area_sum;
    int homogeneous, pad;
};

enum RobotFlags { RF_SELF_COL = 1, RF_GRAV = 2, RF_FLOOR = 4, RF_TEMP = 8, RF_STICKY = 16, RF_FLUID = 32, RF_LW = 64,
                  RF_HORIZON_COL = 128 };

struct DRobot {               // constant per robot
    int vox_begin, nvox, surf_begin, nsurf, flags, stop_type;
    double dt, lat, bond_z_half, slow_z, col_z, grav_acc;
    double init_cm_time, stop_value, afterlife, temp_period_d;
    double min_temp_fact, growth_amplitude, col_horizon, filter_dist2, drag_coef;
    float temp_amplitude, temp_period;
};

struct DRobotState {          // mutable per robot
    double cur_time, dt_prev, max_disp;
    double ini_cm[3];
    double eol_post_y;
    unsigned long long maxvel2_bits;
    int steps, status, cm_init, active, diverged, col_overflow, ncol_links, rebuilds;
};

enum { VXH_MAXCOL = 64 };     // collision partners kept per surface voxel (overflow -> VXH_ROBOT_COL_OVERFLOW)

// all device pointers of a batch; passed to kernels by value
struct DBatch {
    int n_robots, nv;                 // nv = total padded voxel slots (multiple of 64 per robot)
    const DRobot* robot;
    DRobotState* rstate;
    const int* wave_robot;            // [nv/64] robot of each 64-voxel group
    const DVoxClass* vclass_tab;
    const DBondClass* bclass_tab;
    // voxel constants
    const unsigned short* vclass;     // [nv]
    const short* bclass;              // [3*nv] axis-major, -1 = no bond
    const int* nbr;                   // [6*nv] direction-major, global voxel slot or -1
    const float* phase;               // [nv]
    const float* amp_damp;            // [nv]
    // voxel state: pos/scale double-buffered (collision forces read other voxels' previous positions)
    double* pos[2][3];
    double* scale[2];
    double* quat[4];                  // w, x, y, z
    double* lin_mom[3];
    double* ang_mom[3];
    // bond state (axis-major slots 3*nv): history _LastPos2, _LastAngle1, _LastAngle2 and the small-angle flag
    double* hist[9];
    unsigned char* small_angle;
    // bond outputs of the current step: F1, M1, F2, M2
    double* bout[12];
    // collisions
    const int* surf;                  // global voxel slots of surface voxels, per robot contiguous
    const int* surf_ord;              // [nv] ordinal in the robot's surface list or -1
    const int* near_off;              // [nv+1] CSR of the CalcNearby exclusion lists (global voxel slots)
    const int* near_idx;
    int* col_cnt;                     // [total surface voxels]
    int* col_partner;                 // [total surface voxels * VXH_MAXCOL] global voxel slots
    // constants from Vec3D.h evaluated by the host libm (thresholds of the small-angle logic)
    double small_angle_w, smallish_angle_w, slthresh_acos2sqrt;
};

}  // namespace vxh
