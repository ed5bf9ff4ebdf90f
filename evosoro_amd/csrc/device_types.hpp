// Plain structs shared by host code and HIP kernels (HBM layout, see DESIGN.md "Data layout").
#pragma once

namespace vxh {

// per-class constant tables: a handful of entries for a whole population (all robots of an evosoro run share
// one palette), read through the scalar/L1 caches instead of streaming ~170 B of constants per bond
struct DVoxClass {
    double mass, mass_inv, inertia_inv, c_lin, c_ang, E, k_floor, u_static, u_dynamic, cte, nom_size;
    double prenatal_k;   // (float)nom_size / nom_size - 1: the factor of the _voxcad PreNatal term (VXS_Voxel.cpp:236-246: initialVoxelSize is a float
                         // member), a constant of the class -- the kernels used to divide for it per voxel and step
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

enum { VXH_ORDER_MAX_STEPS = 128,    // launches of at most this many steps are dispatched "robots due for a broad-phase run first" (kernels_fused.hpp)
       VXH_ORDER_HORIZON = 32 };     // ... and what a longer launch calls "due" when it leaves the flags for the launch behind it
enum { VXH_RIMG_CAP = 2048 };
// pair kernel (kernels_pair.hpp; compiled with -DVXH_PAIR only: the developer library): threads, voxel slots, wavefronts, voxel wavefronts
enum { VXH_PAIR_T = 512, VXH_PAIR_NV = 1024, VXH_PAIR_NW = 8, VXH_PAIR_NVW = 16 };     // entries of a saved contact-row image (the LDS pool holds at most 24 KB / 12 B)

struct DRobot {               // constant per robot
    int vox_begin, nvox, surf_begin, nsurf, flags, stop_type, excl_wpr;
    int sched_begin;          // resident kernel: this robot's bond schedule in DBatch::bsched ([3][workgroup size] entries, see there)
    long long excl_begin;     // first word of this robot's exclusion rows in DBatch::excl
    int vert_begin, nmv;      // surface-mesh vertices of this robot (land_water robots)
    int facet_begin, nfacet;  // its facets in DBatch::facet_vox / facet_vert
    int vtab_begin, n_vclass; // this robot's rows of DBatch::vclass_tab (class ids stored per voxel are robot-local)
    int btab_begin, n_bclass; // ... and of DBatch::bclass_tab
    double dt, lat, bond_z_half, slow_z, col_z, grav_acc;
    double init_cm_time, stop_value, afterlife, temp_period_d;
    double min_temp_fact, growth_amplitude, col_horizon, filter_dist2, drag_coef;
    double midlife_freeze_time;
    double trace_dt;              // <TimeBetweenTraces>, 0 = no centre-of-mass trace
    int trace_begin, trace_cap;   // this robot's entries of DBatch::trace
    float temp_amplitude, temp_period;
    long long col_begin;          // first entry of this robot's block of contact rows in DBatch::col_partner / col_a1 / col_code
    int col_cap;
    int img_index;                // resident kernel: this robot's slot in DBatch::rimg_* (the saved LDS image of its contact rows), -1 = none
                                  // col_cap: partners a row can hold.  Default nsurf - 1: every other surface voxel, i.e. unbounded like the
                                  // reference's lists (CreateColBond, VX_Sim.cpp:753-769); engine option col_cap sets a smaller one
    // wide kernel (kernels_wide.hpp): the robot's combined bond list in DBatch::wlist and its length, the record number that stays zero (what the
    // missing directions of a voxel point at), doubles of LDS its bond records / scratch take
    int wl_begin, wnbond, wzidx, wregion;
};

struct DRobotState {          // mutable per robot
    double cur_time, dt_prev, max_disp;
    double ini_cm[3];
    double eol_post_y;
    double act_sin, act_cos;      // streaming path: sincos of the actuation phase of the current step (actuation_sincos)
    unsigned long long maxvel2_bits;
    int steps, status, cm_init, active, diverged, col_overflow, rebuild_now, rebuilds;
    int rows_img, reb_step;       // resident kernel: DBatch::rimg_* hold the LDS image of this robot's contact rows as the last launch left it;
                                  // `steps` at the last broad-phase run (the dispatch order of short launches: fused_dispatch_slot)
    int col_tiled, ntrace;        // the contact rows were built by the tiled kernel (DBatch::col_code / tile_xh are valid for them) ; points
                                  // of the centre-of-mass trace recorded so far (SS.CMTrace, VX_Sim.cpp:1537-1547)
    double last_trace_time;
    double act_time;              // resident / wide / tiled kernels: act_sin / act_cos are those of this simulated time (stashed at the end of a launch
                                  // for the first step of the next one; -1: nothing stashed)
};

// Tiled path (kernels_tiled.hpp): a robot cut into `ntiles` tiles, one workgroup each.  A tile OWNS n_own voxels (it
// integrates them) and mirrors n_halo more (the far ends of the bonds that leave the tile, owned by neighbour tiles); its
// bond list holds every bond with at least one owned end, so a bond that crosses a tile boundary is evaluated by both tiles
// (same inputs, same code, same bits) and no force ever travels between tiles -- only poses do.
struct DTile {
    int robot;            // robot index
    int tile0, ntiles;    // first tile of the robot in the batch-wide tile numbering (flags, mv) and their number
    int n_own, n_halo;    // voxels: DBatch::tile_vox[vox_off ..) holds the n_own owned global slots, then the n_halo mirrored ones
    int nb;               // bonds: DBatch::tile_bond / tile_bcls / tile_bslot [bond_off ..)
    int vox_off, bond_off;
    int xoff, pad;        // first exchange slot of the tile's owned voxels (DBatch::xch), a multiple of 64
    // robots in a FLUID (round 5): the part of the drag mesh the tile's owned voxels carry facets on -- its vertices in DBatch::tile_mvert /
    // tile_mv0 [mv_off ..), its facets in DBatch::tile_facet [f_off ..) (per owned voxel contiguous, in the reference's order)
    int mv_off, n_mv, f_off, n_f;
    int mx_off, n_mx;     // ... and the voxels those vertices average over (DBatch::tile_mvox [mx_off ..): exchange slots), staged in LDS every step
};

// dynamic LDS of a tile's workgroup, in doubles from the base; the host sizes the launch with the same function
#if defined(__HIP__)
#define VXH_HD __host__ __device__
#else
#define VXH_HD
#endif
enum { VXH_TILE_BLOCK = 256,          // worker threads of a tile's workgroup = most voxels a tile can own
       VXH_TILE_THREADS = 256,        // threads of its workgroup: four wavefronts, one per SIMD, 512 registers each (rounds 2-5: 320, a fifth wavefront
                                      // for the per-robot barrier and the control block -- now the last worker wavefront's job; kernels_tiled.hpp)
       VXH_TILE_HASH_BITS = 9, VXH_TILE_HASH = 512,   // broad-phase: LDS hash set of the contact partners owned by other tiles
       VXH_TILE_MAX_TILES = 256,      // most tiles of one robot (its max-|v|^2 words are polled by one wavefront)
       VXH_TILE_MV_STRIDE = 512,      // granules between the max-|v|^2 words of two tiles: 4 KB, so that the wavefronts polling them
                                      // (one per tile, all at once) spread over the memory channels instead of queueing at one
       VXH_TILE_ROWPOOL = 1024,       // contact-row entries (partner code + pair stiffness) a tile keeps in LDS (rows beyond: read from memory)
       VXH_TILE_XH = 192,             // most contact partners owned by other tiles that a tile mirrors in LDS (the rest: fetched from memory)
       VXH_TILE_CH = 128,             // chunk of the whole-robot passes (latch, broad-phase) staged through LDS
       VXH_TILE_STATIC_LDS = 1024,    // upper bound of the kernel's static __shared__ variables
       // the SMALL size class (round 6): tiles of at most one wavefront of owned voxels, two of mirrored ones and one bond per lane -- the 4x4x4
       // tiles of configs[4], the tiles of a swimmer above 1024 voxels -- are stepped by kernel instances whose LDS plane strides are these
       // constants instead of the tile's own counts
       VXH_TILE_S_OWN = 64, VXH_TILE_S_HALO = 128, VXH_TILE_S_BONDS = 256 };
struct TileLayout { int np, no, nbp, nmvp, nfp, nmxp, o_ps, o_pl, o_hl, o_pht, o_sl, o_sc, o_px, o_rc, o_tab, o_int, o_mesh, total; };
VXH_HD inline TileLayout tile_layout(int n_own, int n_halo, int nb, int tab_doubles, bool mesh, int n_mv = 0, int n_f = 0, int n_mx = 0)
{
    TileLayout L;
    L.np = (n_own + n_halo + 1) & ~1; L.no = (n_own + 1) & ~1; L.nbp = (nb + 1) & ~1;
    L.o_ps = 0;
    L.o_pl = L.o_ps + 8 * L.np;
    L.o_hl = L.o_pl + 36 * L.no;
    L.o_pht = L.o_hl + 6 * L.nbp;
    L.o_sl = L.o_pht + 2 * L.no;          // [6][no] directional strains of the owned voxels (land_water robots only -- `mesh`: RobotVolume tags)
    L.o_sc = L.o_sl + (mesh ? 6 * L.no : 0);
    L.o_px = L.o_sc + 5 * VXH_TILE_CH + VXH_TILE_CH / 2;
    L.o_rc = L.o_px + 4 * VXH_TILE_XH;
    // (round 6: the integer region in front of the class tables -- whose size is the robot's -- so that every offset up to here is a
    // function of np / no / nbp alone: compile-time constants in the SMALL size class of the kernel, immediates of its LDS instructions)
    L.o_int = L.o_rc + VXH_TILE_ROWPOOL;
    L.o_tab = L.o_int + (3 * L.nbp + VXH_TILE_XH + 2 * VXH_TILE_HASH + VXH_TILE_ROWPOOL + 1) / 2;
    // a tile of a robot in a fluid: [3][nmvp] vertices of its part of the drag mesh, [3][nfp] drag of its facets, [3][no] velocities of its voxels,
    // [13][nmxp] position, quaternion and the six strains of every voxel its vertices average over
    L.nmvp = (n_mv + 1) & ~1; L.nfp = (n_f + 1) & ~1; L.nmxp = (n_mx + 1) & ~1;
    L.o_mesh = L.o_tab + ((tab_doubles + 1) & ~1);
    // ... and the constant tables of that mesh (copied once per launch): [3][nmvp] rest positions, [8][nmvp] ints voxel per corner code, [4][nfp] ints facets
    L.total = L.o_mesh + ((n_mv > 0 || n_f > 0) ? 3 * L.nmvp + 3 * L.nfp + 3 * L.no + 13 * L.nmxp + 3 * L.nmvp + 4 * L.nmvp + 2 * L.nfp : 0);
    return L;
}

// what CVX_SimGA::WriteResultFile needs of a robot's final state, reduced on the device (k_results): the host derives every tag from it
struct DResult {
    double cm[3];                 // SS.CurCM: mass-weighted sum in voxel order, like GetCM (VX_Sim.cpp:2415-2430)
    double d2max, d2min;          // max / min over the voxels of (x - IniCM.x)^2 + (y - IniCM.y)^2 (getAnteriorDist / getPosteriorDist :2584-2616)
    double ymax, ymin;            // max / min y of the voxels that are not material 5 (getAnteriorY / getPosteriorY :2620-2656)
    int touching, feet;           // voxels below the floor plane, and those of material 6 among them (GetNumTouchingFloor :2660-2712)
};

// (collision partners per surface voxel: DRobot::col_cap; a row that would need more -> VXH_ROBOT_COL_OVERFLOW)

// all device pointers of a batch; passed to kernels by value
struct DBatch {
    int n_robots, nv;                 // nv = total padded voxel slots (multiple of 64 per robot)
    int dbg, pad1;                    // developer switches (scripts/dev_gpu_diag.py), 0 in production
    const DRobot* robot;
    DRobotState* rstate;
    DRobotState* rstate_mirror;       // the same blocks in pinned host memory (or null): the resident kernels write a robot's block to both when a launch ends
    const int* wave_robot;            // [nv/64] robot of each 64-voxel group
    const DVoxClass* vclass_tab;      // per-robot class tables, concatenated (DRobot::vtab_begin / btab_begin)
    const DBondClass* bclass_tab;
    // voxel constants
    const unsigned short* vclass;     // [nv] robot-local class id
    const short* bclass;              // [3*nv] axis-major, robot-local class id, -1 = no bond
    const int* nbr;                   // [6*nv] direction-major, global voxel slot or -1
    const int* wlist;                 // wide kernel: per robot (DRobot::wl_begin) ALL its bonds in one list, axis after axis, entry = local
                                      // negative-end voxel | local positive-end voxel << 9 | axis << 18 | bond class << 20
    const int* wgather;               // [2][nv] wide kernel: the records of the voxel's six bonds (+X -X +Y -Y +Z -Z, 10 bits each, three
                                      // per word; DRobot::wzidx where the voxel has no bond in that direction)
    const int* bsched;                // resident kernel (kernels_fused.hpp): per robot [3][BLOCK] packed bond entries (negative-end voxel | positive-end
                                      // voxel << 10 | class << 20, -1 = none): what thread t evaluates in the X, Y and Z slot of a step.  X and Z: the
                                      // t-th bond of the axis (compacted lists); Y: whole 64-bond chunks dealt to the wavefronts so that X and Y TOGETHER
                                      // load the four SIMDs evenly (two accumulator tiles: X and Y are evaluated without a barrier between them)
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
    double* hist_aos;                 // resident kernel (kernels_fused.hpp): the same six doubles per bond slot as ONE 48-byte record, [3 * nv][6] --
                                      // three 16-byte loads / stores per bond instead of six 8-byte ones (its lanes walk compacted bond lists: the planes
                                      // of `hist` are not read coalesced anyway).  A robot's history lives in the array of the kernel that steps it.
    unsigned char* small_angle;
    // streaming path only: bond outputs of the current step, 12 planes of 3*nv: F1, M1, F2, M2
    double* bout;
    // collisions
    const int* surf;                  // global voxel slots of surface voxels, per robot contiguous
    const int* surf_code;             // same shape as surf: robot-local voxel index | voxel class << 10 (what the broad-phase stages per ordinal)
    const int* surf_ord;              // [nv] ordinal in the robot's surface list or -1
    const unsigned long long* excl;   // CalcNearby exclusion as bit rows: per robot nsurf rows of excl_wpr 64-bit words,
                                      // bit j of row i set when surface voxels i and j are within the hop horizon
    int* col_cnt;                     // [total surface voxels]
    // resident kernel: the workgroup's LDS copy of a robot's contact rows (rows_to_lds: pair codes, pair stiffnesses, the wavefronts'
    // segments, every thread's row descriptor), saved whenever it is built and streamed back at the start of the next launch -- one
    // round trip of coalesced loads instead of ordinal -> count -> scan -> rows, three dependent round trips and three barriers
    int* rimg_code; double* rimg_a1; int* rimg_seg; int* rimg_rowd;     // [n_img * VXH_RIMG_CAP] x 2, [n_img * 64], [nv]
    int col_rows, pad2;               // total surface voxels of colliding robots
    int* col_partner;                 // per robot a block [col_cap][nsurf] (partner-major; kernels.hpp col_at) of global voxel slots
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
    // tiled path (robots stepped by k_tile_steps; all null / zero otherwise)
    const DTile* tiles;
    const int* tile_vox;              // per tile: owned voxels (ascending global slot), then the EXCHANGE slots of its halo voxels
    const int* tile_bond;             // per tile and bond: local index of the negative end | local index of the positive end << 10
                                      // | axis << 20 (local = position in the tile's voxel list, halo entries after the owned ones)
    const int* tile_bcls;             // ... its bond class (robot-local)
    const int* tile_bslot;            // ... its canonical slot axis * nv + negative-end voxel (DBatch::hist, small_angle)
    // Exchange between the tiles of a robot, in 8-byte GRANULES {32 data bits, 32-bit tag}: a double travels as two granules
    // (low word, high word), each written by one write-through store, so a granule is never torn and validates itself -- the
    // tag names the launch and the publication (kernels_tiled.hpp tile_tag) -- and no flag, fence or drain orders anything.
    const int* xslot;                 // [nv] exchange slot of every voxel of a tiled robot: owner tile's xoff + its index there
    int nx, xplanes;                  // exchange slots in all (every tile's range padded to a multiple of 64); granule planes per buffer: 16, or 28 when
                                      // a tiled robot is in a fluid (planes 16 .. 27: low / high granules of the voxel's six directional strains)
    const int* tile_mvox;             // fluid tiles: the voxels a tile's mesh vertices average over, as exchange slots (DTile::mx_off, n_mx)
    const int* tile_mvert;            // [8][n_tmv] fluid tiles: per mesh vertex of a tile and corner code, the voxel touching it there as an index into the tile's tile_mvox list, or -1
    const double* tile_mv0;           // [3][n_tmv] its rest position
    const int* tile_facet;            // [4][n_tf] per facet of a tile: owner (index among the tile's owned voxels), its three vertices (tile-local)
    const int* tile_ffirst;           // [as tile_vox] per owned voxel: first facet (tile-local) ...
    const unsigned char* tile_fcount; // ... and their number
    int n_tmv, n_tf;
    unsigned long long* xch;          // [3][nx / 64][xplanes][64] poses (+ strains), a ring of three buffers, slot = step count mod 3, the planes blocked by 64
                                      // exchange slots (kernels_tiled.hpp xch_at): planes 2c / 2c+1 = low / high granule of component c
                                      // (pos xyz, scale, quaternion wxyz) of every voxel of a tiled robot, published by its owner
    unsigned long long* tile_mv;      // [3][total tiles][VXH_TILE_MV_STRIDE] first two words: low / high granule of the max |v|^2 of the tile's voxels in the step
                                      // that produced the published poses (negative: a bond of the tile diverged in that step)
    int n_tiles, pad5;
    const int* tile_of;               // [nv] owner tile (batch-wide numbering) and position among its owned voxels, -1 = not tiled
    const int* tile_lidx;
    int* col_code;                    // same shape as col_partner: where the tile that owns the row finds the partner's position in its
                                      // LDS (owned voxel: its index; mirrored partner: np + entry of tile_xh), -1 = fetch from memory
    int* tile_xh;                     // [total tiles][VXH_TILE_XH] contact partners owned by other tiles that the tile mirrors (global slots)
    int* tile_xhn;                    // [total tiles] their number
    double* trace;                    // centre-of-mass traces: 4 doubles (time, x, y, z) per entry, robots one after the other (DRobot::trace_begin)
    unsigned long long* prof;         // developer builds (-DVXH_PHASE_TIMING): per-wave phase cycle sums, else null
    // constants from Vec3D.h evaluated by the host libm (thresholds of the small-angle logic)
    double small_angle_w, smallish_angle_w, slthresh_acos2sqrt;
};

}  // namespace vxh
