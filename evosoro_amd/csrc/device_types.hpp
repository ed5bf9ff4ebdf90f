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
    double L100, a2, b1, b2, b3;  // L100 = 100 * L: the bond has diverged when its elongation exceeds it (strain > 100)
    // per unit ELONGATION (strain = elongation / L, the 1 / L is folded in): axial force stress_k * (area1 + area2) / 2 / L
    // (model.cpp make_bond_class) and the half-bond strains CurStrainV1/V2
    double kf_L, strain_a1_L, strain_a2_L;
    // AddDampForces (VXS_BondInternal.cpp:310-346): the 2 sqrt(k m) terms of VX_Bond.h:65-71, already multiplied by the
    // robot's BondDampingZ / 2 (moments: / 4) and by 1 / dt, so the kernel multiplies plain differences of the bond-frame
    // pose: forces on voxel 1 / 2 (A linear-x, B linear-yz, F angular), moments on voxel 1 / 2 (T twist, G linear, H angular)
    double dA1, dB1, dF1, dA2, dB2, dF2, dT1, dG1, dH1, dT2, dG2, dH2;
    int homogeneous, pad;
};

enum RobotFlags { RF_SELF_COL = 1, RF_GRAV = 2, RF_FLOOR = 4, RF_TEMP = 8, RF_STICKY = 16, RF_FLUID = 32, RF_LW = 64,
                  RF_HORIZON_COL = 128,
                  // _voxcad development (VXS_Voxel.cpp:236-328): any layer present / which ones
                  RF_DEV = 256, RF_DEV_SIZE = 512 /* Initial- or FinalVoxelSize */, RF_DEV_FSIZE = 1024, RF_DEV_FPHASE = 2048, RF_DEV_FTAD = 4096 };

struct DRobot {               // constant per robot
    int vox_begin, nvox, surf_begin, nsurf, flags, stop_type, excl_wpr, pad0;
    long long excl_begin;     // first word of this robot's exclusion rows in DBatch::excl
    int vert_begin, nmv;      // surface-mesh vertices of this robot (land_water robots)
    int facet_begin, nfacet;  // its facets in DBatch::facet_vox / facet_vert
    int vtab_begin, n_vclass; // this robot's rows of DBatch::vclass_tab (class ids stored per voxel are robot-local)
    int btab_begin, n_bclass; // ... and of DBatch::bclass_tab
    double dt, lat, bond_z_half, slow_z, col_z, grav_acc;
    double init_cm_time, stop_value, afterlife, temp_period_d;
    double min_temp_fact, growth_amplitude, col_horizon, filter_dist2, drag_coef;
    double midlife_freeze_time;
    float temp_amplitude, temp_period;
};

struct DRobotState {          // mutable per robot
    double cur_time, dt_prev, max_disp;
    double ini_cm[3];
    double eol_post_y;
    double act_sin, act_cos;      // streaming path: sincos of the actuation phase of the current step (actuation_sincos)
    unsigned long long maxvel2_bits;
    int steps, status, cm_init, active, diverged, col_overflow, rebuild_now, rebuilds;
};

enum { VXH_MAXCOL = 64 };     // collision partners kept per surface voxel (overflow -> VXH_ROBOT_COL_OVERFLOW)

// all device pointers of a batch; passed to kernels by value
struct DBatch {
    int n_robots, nv;                 // nv = total padded voxel slots (multiple of 64 per robot)
    int dbg, pad1;                    // developer switches (tests/dev_gpu_diag.py), 0 in production
    const DRobot* robot;
    DRobotState* rstate;
    const int* wave_robot;            // [nv/64] robot of each 64-voxel group
    const DVoxClass* vclass_tab;      // per-robot class tables, concatenated (DRobot::vtab_begin / btab_begin)
    const DBondClass* bclass_tab;
    // voxel constants
    const unsigned short* vclass;     // [nv] robot-local class id
    const short* bclass;              // [3*nv] axis-major, robot-local class id, -1 = no bond
    const int* nbr;                   // [6*nv] direction-major, global voxel slot or -1
    const int* blist;                 // [3*nv] fused path: per robot and axis the COMPACTED list of its bonds, entry t of axis a at
                                      // [a*nv + vox_begin + t] = local negative-end voxel | local positive-end voxel << 10 |
                                      // bond class << 20, -1 past the end of the list
    const double* act_sb;             // [nv] sin / cos of 2 pi' * PhaseOffset of the voxel (pi' = 3.1415926f)
    const double* act_cb;
    const float* amp_damp;            // [nv]
    const float* dev;                 // [7][nv] development robots: initialVoxelSize, finalVoxelSize, startGrowthTime, growthTime,
                                      // phaseOffset, finalPhaseOffset, finalTempAmpDamp (float members of CVXS_Voxel)
    // voxel state, 18 component planes of nv doubles each: [0..3] pos xyz + scale (buffer 0), [4..7] the same
    // (buffer 1; positions/scale are double-buffered because collision forces read OTHER voxels' previous
    // positions), [8..11] quaternion w x y z, [12..14] linear momentum, [15..17] angular momentum
    double* vs;
    // bond history, 6 planes of 3*nv (axis-major slots) + a flag byte per slot.  The reference keeps _LastPos2,
    // _LastAngle1, _LastAngle2 (9 doubles), but three of them are always exactly zero: after a small-angle step
    // _LastAngle1 = 0, after a large-angle step _LastPos2.y = _LastPos2.z = _LastAngle1.x = 0.  Planes 0..2 hold
    // _LastPos2 xyz (small layout) or _LastPos2.x, _LastAngle1.y, _LastAngle1.z (large layout), planes 3..5
    // _LastAngle2.  Flag byte: bit 0 = SmallAngle, bit 1 = history is in the large layout.
    double* hist;
    unsigned char* small_angle;
    // streaming path only: bond outputs of the current step, 12 planes of 3*nv: F1, M1, F2, M2
    double* bout;
    // collisions
    const int* surf;                  // global voxel slots of surface voxels, per robot contiguous
    const int* surf_ord;              // [nv] ordinal in the robot's surface list or -1
    const unsigned long long* excl;   // CalcNearby exclusion as bit rows: per robot nsurf rows of excl_wpr 64-bit words,
                                      // bit j of row i set when surface voxels i and j are within the hop horizon
    int* col_cnt;                     // [total surface voxels]
    int col_rows, pad2;               // total surface voxels of colliding robots
    int* col_partner;                 // [VXH_MAXCOL][col_rows] (partner-major) global voxel slots
    double* col_a1;                   // same shape: linear stiffness a1 of that collision bond
    // land_water fluid drag (LW/VX_Sim.cpp:1516-1597): deformable surface mesh of every fluid robot
    int total_mv, pad3;               // mesh vertices of all fluid robots
    const int* vert_pack;             // [3][total_mv] the <= 7 voxels sharing the vertex, by the corner code (NNN..PPP = 0..7) they
                                      // touch it with: word0 = l0 | l1 << 10 | l2 << 20, word1 = l3 | l4 << 10 | l5 << 20,
                                      // word2 = l6 | l7 << 10 | (bit c set when corner c is present) << 20; l = robot-local voxel
    const double* vert_v0;            // [3][total_mv] rest position
    int total_facet, pad4;
    const int* facet_vox;             // [total_facet] local voxel owning the facet (reference order: per voxel, per face, two triangles)
    const int* facet_vert;            // [3][total_facet] its three robot-local mesh vertices
    const int* facet_first;           // [nv] first facet of the voxel, relative to the robot's facet_begin
    const unsigned char* facet_count; // [nv]
    double* strain;                   // [6][nv] StrainPosDirsCur xyz, StrainNegDirsCur xyz (land_water robots only)
    const unsigned char* streamed;    // [n_robots] 1 = this robot is stepped by the streaming kernels in the current call
    // streaming kernels only (robots in a fluid that do not fit the resident kernel): mesh in HBM
    int n_mv, n_facet;                // real counts (total_mv / total_facet are the plane strides, >= 1)
    const int* vert_vox;              // [8][total_mv] global voxel slot touching the vertex with corner code c, or -1
    const int* vert_robot;            // [total_mv]
    const int* facet_robot;           // [total_facet]
    double* mesh_pos;                 // [3][total_mv] current vertex positions
    double* fdrag;                    // [3][total_facet] drag of every facet in the current step
    unsigned long long* prof;         // developer builds (-DVXH_PHASE_TIMING): per-wave phase cycle sums, else null
    // constants from Vec3D.h evaluated by the host libm (thresholds of the small-angle logic)
    double small_angle_w, smallish_angle_w, slthresh_acos2sqrt;
};

}  // namespace vxh
