/* libvxhip — C ABI of the MI355X-native batched Voxelyze time-stepper.
 *
 * This is the drop-in boundary for evosoro's "one `voxelyze -f file.vxa` process per robot" path.  Each
 * entry point names the reference interface it replaces (paths relative to the reference repository,
 * VX/ = evosoro/_voxcad/Voxelyze/, LW/ = evosoro/_voxcad_land_water/Voxelyze/):
 *
 *   vxh_add_vxa_file / _buffer  CVX_Sim::LoadVXAFile + ReadVXA            VX/VX_Sim.cpp:159-203
 *                               (voxelyzeMain/main.cpp:55, `-f <file>` argument :33-36)
 *   vxh_run                     Import + the `while(!StopConditionMet()) TimeStep()` loop
 *                               voxelyzeMain/main.cpp:62-111; VX/VX_Sim.cpp:488-717,1054-1156,1763-1933
 *                               for EVERY queued robot at once (replaces the N concurrent processes of
 *                               evosoro/tools/evaluation.py:89-90)
 *   vxh_get_result              the values CVX_SimGA::WriteResultFile prints  VX/VX_SimGA.cpp:33-168,
 *                               LW/VX_SimGA.cpp:33-77
 *   vxh_write_result_xml        CVX_SimGA::SaveResultFile(FitnessFileName)    VX/VX_SimGA.cpp:25-30,
 *                               voxelyzeMain/main.cpp:130 (same tags, 6 significant digits)
 *   vxh_get_state               CVXS_Voxel::GetCurPos/GetCurAngle/GetCurScale/GetCurVel (trajectory parity)
 *   vxh_step                    CVX_Sim::TimeStep called n times (parity tests on early steps)
 *
 * Conventions: plain C types only; every function returns 0 (VXH_OK) or a negative vxh_status; the caller
 * owns every buffer it passes; a handle is not thread-safe (use one per thread / per GPU process).
 * There is NO CPU fallback: vxh_create fails with VXH_ERR_NO_DEVICE when no HIP device is usable.
 */
#ifndef VXHIP_H
#define VXHIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vxh_engine vxh_engine;

enum vxh_status {
    VXH_OK = 0,
    VXH_ERR_ARG = -1,        /* bad argument / index out of range */
    VXH_ERR_NO_DEVICE = -2,  /* no usable HIP device (the engine has no CPU path) */
    VXH_ERR_PARSE = -3,      /* malformed .vxa */
    VXH_ERR_IO = -4,         /* cannot read/write a file */
    VXH_ERR_HIP = -5,        /* a HIP runtime call failed, see vxh_last_error */
    VXH_ERR_STATE = -6,      /* call order (e.g. results before vxh_run) */
    VXH_ERR_UNSUPPORTED = -7 /* .vxa uses a feature outside the supported scope */
};

enum vxh_variant { VXH_VOXCAD = 0, VXH_VOXCAD_LAND_WATER = 1 };

/* per-robot outcome; the reference has no such codes: a diverged or empty robot makes its process spin
 * forever and evosoro times it out (evaluation.py:107-119). */
enum vxh_robot_status { VXH_ROBOT_PENDING = 0, VXH_ROBOT_FINISHED = 1, VXH_ROBOT_DIVERGED = 2, VXH_ROBOT_EMPTY = 3,
                        VXH_ROBOT_COL_OVERFLOW = 4,
                        VXH_ROBOT_SYNC_TIMEOUT = 5 /* device side only, never reported: the call is made again without the tiled kernel -- the
                                                      batch is stepped from its imported state up to where the call was to end (also
                                                      a call in the middle of a run: the evaluation is deterministic) */ };

typedef struct vxh_result {
    int status;            /* vxh_robot_status */
    int steps;             /* CurStepCount */
    int nvox, nbond;
    double dt;             /* DtFrac*OptimalDt */
    double cur_time;       /* CurTime */
    double lifetime;       /* <Lifetime> */
    double ini_cm[3];      /* IniCM */
    double cur_cm[3];      /* SS.CurCM after the last step */
    /* _voxcad tags, VX/VX_SimGA.cpp:145-168 */
    double norm_final_dist, norm_regime_dist, norm_frozen_dist, final_dist, final_dist_y;
    double anterior_dist, posterior_dist, anterior_y, posterior_y, end_of_life_posterior_y, fall_adj_post_y;
    double num_non_feet_touching_floor, num_touching_floor;
    /* _voxcad_land_water tags, LW/VX_SimGA.cpp:58-62 */
    double norm_abs_disp, norm_dist_x, norm_dist_y, norm_dist_z;
    double robot_volume_start, robot_volume_end;   /* <RobotVolumeStart/End>, LW/VX_MeshUtil.cpp:908-952 */
    int col_rebuilds;      /* diagnostic: how often CalcL1Bonds ran (VX/VX_Sim.cpp:1741-1747) */
    int reserved;
    double hull_volume_start, hull_volume_end;     /* <ConvexHullVolumeStart/End>: convex hull of the surface-mesh vertices, which the
                                                      reference gets from an external qhull (LW/VX_MeshUtil.cpp:775-900); the
                                                      <ShapeComplexity*> tags: see vxh_get_angle_excess */
} vxh_result;
/* land_water shape descriptor, per vertex of the deformable surface mesh: the angle excess 2 pi - sum of the angles of the facets
 * meeting in the vertex, exactly as CVX_MeshUtil::computeShapeComplexity forms it (LW/VX_MeshUtil.cpp:956-1014) before handing the
 * vector to curvatureEntropy.py through <CurvaturesTmpFile> (:1016-1036).  That script is not in the reference repository: its call fails,
 * the reference reads ONE number back from the file it wrote itself, and <ShapeComplexityStart/End> are the first vertex's angle excess
 * as printed (six digits) -- which is what vxh_write_result_xml prints too: the tags EMULATE THE REFERENCE'S FAILURE MODE, they are not
 * an entropy (with the script installed a reference user gets the entropy instead); -1 without a <CurvaturesTmpFile> or when that file
 * cannot be opened for writing, where the reference returns -1 before computing anything (:1024-1032).  The
 * vector is what can be computed and pinned: at_end = 0 the rest state (what
 * computeInitialShapeComplexity sees, voxelyzeMain/main.cpp:65), 1 the state after the last step (computeFinalShapeComplexity, :117).
 * `count_out` receives the number of mesh vertices (0 for a _voxcad robot); at most `capacity` values are written.
 * vxh_write_result_xml also writes the final vector to the robot's <CurvaturesTmpFile>, tab-separated, six significant digits. */
int  vxh_get_angle_excess(const vxh_engine* e, int robot, int at_end, double* out, int capacity, int* count_out);
/* ... of the rest state straight from a .vxa text, host-only (no GPU needed) */
/* The deformable surface mesh of a robot that carries one -- every land_water robot; a _voxcad robot added to an engine whose option
 * "shape_descriptors" was 1 -- and what CVX_MeshUtil derives from it: what voxelyzeMain/main.cpp:65-88,113-126 computes and, under -p,
 * prints with --computeShapeDescriptors (computeAndStoreQHullStart/End, computeAndStoreRobotVolumeStart/End,
 * computeInitial/FinalShapeComplexity, printAllMeshInfo; VX_MeshUtil.cpp:733-772,775-1076).  at_end 0 = the rest state right after the
 * import, 1 = the current state.  vxh_get_mesh: vertex positions (v + DrawOffset) in the reference's vertex order, facets as vertex
 * triples in its facet order; counts are returned even when the capacities are 0.  vxh_get_shape_descriptors: enclosed volume (signed
 * tetrahedra), volume of the convex hull of the vertices (own incremental hull where the reference shells out to qhull), and the
 * shape complexity AS THE REFERENCE BINARY PRINTS IT (see vxh_get_angle_excess: the first vertex's angle excess; -1 without a
 * <CurvaturesTmpFile>).  A robot without a mesh: counts 0, descriptors -1. */
int  vxh_get_mesh(const vxh_engine*, int robot, int at_end, double* verts3, int vert_capacity, int* n_verts, int* facets3, int facet_capacity, int* n_facets);
int  vxh_get_shape_descriptors(const vxh_engine*, int robot, int at_end, double* robot_volume, double* hull_volume, double* shape_complexity);
int  vxh_inspect_angle_excess(const char* xml, size_t len, int variant, double* out, int capacity, int* count_out, char* errbuf, size_t errcap);
/* volume of the convex hull of n points (x, y, z per point): the computation behind hull_volume_*, exposed for testing */
double vxh_convex_hull_volume(const double* xyz, int n);

typedef struct vxh_counters {
    double voxel_steps;        /* sum over robots of nvox * steps taken in vxh_run/vxh_step so far */
    double bond_steps;         /* sum of nbond * steps */
    double algorithmic_bytes;  /* sum of (224*nvox + 144*nbond) * steps, SURVEY.md section 8(d) */
    double kernel_seconds;     /* HIP-event time of the stepping region on the engine's stream */
    double run_seconds;        /* host wall time of vxh_run incl. upload/download */
    long long launches;        /* step-kernel launches */
    long long max_steps;       /* largest per-robot step count */
    /* the dominant kernel of the last vxh_run/vxh_step call (the size class holding most voxels on the fused path,
     * k_bonds on the streaming path): what bench.py prices against the HBM roofline */
    int dominant_block;             /* workgroup size of k_robot_steps<BLOCK>; + 1: k_robot_wide; 1026: k_robot_pair; 1 = k_tile_steps; 0 = streaming path */
    int dominant_robots;
    long long dominant_launches;
    double dominant_seconds;        /* HIP-event time from its first to its last launch on its own stream */
    double dominant_alg_bytes;      /* algorithmic bytes its launches processed in that call */
    double dominant_voxel_steps;
} vxh_counters;

/* Host-only view of what Import() would build for a .vxa (no device needed): voxel/bond/surface counts in the
 * reference's ordering, OptimalDt (CalcMaxDt, VX/VX_Sim.cpp:1693-1727), dt = DtFrac*OptimalDt and the number of
 * TimeStep() calls the reference main loop would make (voxelyzeMain/main.cpp:89-111 with StopConditionMet). */
typedef struct vxh_model_info {
    int nvox, nbond, nsurf, n_vox_classes, n_bond_classes, reserved;
    double opt_dt, dt;
    long long planned_steps;
    double alg_bytes_per_step;   /* 224*nvox + 144*nbond */
} vxh_model_info;
int  vxh_inspect_vxa_buffer(const char* xml, size_t len, int variant, vxh_model_info* out, char* errbuf, size_t errcap);

/* Host-only view of how the engine would cut a robot into tiles for its multi-workgroup kernel (evosoro_amd/csrc/kernels_tiled.hpp;
 * the loop being tiled: CVX_Sim::Integrate, VX/VX_Sim.cpp:1763-1933): the grid, the largest tile, and the owner tile of every voxel
 * (tile_of_out, `capacity` entries, may be NULL).  k_request tiles or somewhat fewer. */
typedef struct vxh_tiling_info {
    int k, kx, ky, kz;       /* tiles = kx * ky * kz boxes */
    int max_own;             /* most voxels owned by a tile */
    int max_local;           /* ... owned + mirrored (halo) */
    int max_bonds;           /* most bonds evaluated by a tile (bonds crossing a tile boundary count on both sides) */
    int total_bonds;         /* sum over tiles */
} vxh_tiling_info;
/* Host-only: the constants the import leaves on every voxel (12 doubles each: mass, 1/mass, inertia, 1/inertia, first moment,
 * _2xSqMxExS, _2xSqIxExSxSxS, Vox_E, nominal size, uStatic, uDynamic, CTE -- CVX_Voxel::SetMaterial, VX_Voxel.cpp:94-128) and on
 * every internal bond in the reference's creation order (23 doubles each: voxel 1, voxel 2, axis 0..2, homogeneous, L, a1, a2,
 * b1y b2y b3y, b1z b2z b3z, _2xSqA1xM1/2, _2xSqA2xI1/2, _2xSqB1YxM1/2, _2xSqB2YxFM1/2, _2xSqB3YxI1/2 -- CVX_Bond::UpdateConstants,
 * VX_Bond.cpp:95-173).  For the bit-for-bit comparison with the oracle (tests/test_capi.py); the kernels read tables derived
 * from these per class. */
int  vxh_inspect_constants(const char* xml, size_t len, int variant, double* vox12n, int vox_capacity, double* bond23n, int bond_capacity,
                           char* errbuf, size_t errcap);
int  vxh_plan_tiles_buffer(const char* xml, size_t len, int variant, int k_request, vxh_tiling_info* out, int* tile_of_out, int capacity,
                           char* errbuf, size_t errcap);

int  vxh_create(vxh_engine** out, int variant, int device_id);
/* One handle over several GPUs of the node: the robots are partitioned over the devices by cost (voxels x
 * planned steps, largest first) at the first vxh_run / vxh_step after an addition, every device steps its share from its own host
 * thread, and every other call works on the global robot numbering as with one device.  (The results of one process live in its
 * host memory: no collective is involved; the RCCL gather belongs to the one-process-per-GPU route, evosoro_amd/parallel.py.)
 * The SAME device id given more than once makes several engines on that one GPU, and such a handle PIPELINES a generation handed
 * over with vxh_add_robots: the additions are checked and copied (everything that can be refused is refused there), and vxh_run
 * builds, uploads and launches chunk after chunk -- one per engine -- before it waits for the first, so the device steps chunk k
 * while the host cores build chunk k + 1.  Results do not depend on it, bit for bit.  A reader before the run (vxh_robot_dims ...)
 * builds on demand; a generation with a lattice of more than 1024 voxels is not pipelined (its kernel must own the device). */
int  vxh_create_multi(vxh_engine** out, int variant, const int* device_ids, int n_devices);
void vxh_destroy(vxh_engine* e);

int  vxh_add_vxa_file(vxh_engine* e, const char* path, int* robot_index_out);
int  vxh_add_vxa_buffer(vxh_engine* e, const char* xml, size_t len, int* robot_index_out);
/* A whole generation at once (the files evaluate_all wrote, evosoro/tools/evaluation.py:62,89): read, parsed and built on
 * all host cores, appended in the order given; on error nothing is appended and the first failing file is reported. */
int  vxh_add_vxa_files(vxh_engine* e, const char* const* paths, int n, int* first_index_out);
/* In-memory hand-off of a generation (replaces, for the robots of one generation, evaluation.py:59-62 + read_write_voxelyze.py:345-399:
 * per-voxel state arrays -> text -> file -> parser): the <Simulator>, <Environment> and <Palette> of the generation come from ONE
 * .vxa text (`template_vxa`: the file of any individual of the generation; its <Structure> data are ignored), every robot's lattice
 * and per-voxel layers from arrays in the order the writer prints them (z slowest, then y, x fastest).  Layer tags as in the file:
 * "PhaseOffset", "TempAmpDamp", "Stiffness", "FinalPhaseOffset", "FinalTempAmpDamp", "InitialVoxelSize", "FinalVoxelSize",
 * "GrowthTime", "StartGrowthTime".  round_like_text != 0: every layer value goes through the decimal text the Python-2 writer would
 * have printed ("%.12g") and back, so the robot is bit-identical to the one the file route builds.  Robots are built on all host
 * cores and appended in order; on error nothing is appended. */
typedef struct vxh_robot_arrays {
    int nx, ny, nz;
    const unsigned char* material;         /* [nz*ny*nx] palette indices 0..n, 0 = empty: the digits of <Data> */
    int n_layers;
    const char* const* layer_tags;         /* [n_layers] */
    const double* const* layers;           /* [n_layers] pointers to [nz*ny*nx] doubles */
    const char* fitness_file_name;         /* <FitnessFileName> of this robot, or NULL */
} vxh_robot_arrays;
int  vxh_add_robots(vxh_engine* e, const char* template_vxa, size_t template_len, const vxh_robot_arrays* robots, int n,
                    int round_like_text, int* first_index_out);
int  vxh_num_robots(const vxh_engine* e);
int  vxh_robot_dims(const vxh_engine* e, int robot, int* nvox, int* nbond, double* dt, long long* planned_steps);

/* Actuation parameters of one voxel as the reference holds them -- the float members CVXS_Voxel::TempAmplitude / TempPeriod /
 * phaseOffset (VXS_Voxel.h:92-111, assigned in CVX_Sim::ResetSimulation, VX_Sim.cpp:880-990) widened to double: what `voxelyze -p`
 * prints for Vox[0] every 100 steps (voxelyzeMain/main.cpp:99-101). */
int  vxh_voxel_actuation(const vxh_engine* e, int robot, int voxel, double* temp_amplitude, double* temp_period, double* phase_offset);
int  vxh_run(vxh_engine* e);                         /* every robot to its own stop condition */
int  vxh_step(vxh_engine* e, long long nsteps);      /* at most nsteps more TimeStep()s per robot */
int  vxh_reset(vxh_engine* e);                       /* back to the imported state (ResetSimulation) */
int  vxh_clear(vxh_engine* e);                       /* drop all robots */

int  vxh_get_result(const vxh_engine* e, int robot, vxh_result* out);
int  vxh_write_result_xml(const vxh_engine* e, int robot, const char* path_or_null);
int  vxh_fitness_file_name(const vxh_engine* e, int robot, char* buf, size_t cap);
/* per voxel 14 doubles: pos3, quat(w,x,y,z), scale, vel3, angvel3; capacity in voxels */
int  vxh_get_state(const vxh_engine* e, int robot, double* out14n, int capacity);
/* the centre-of-mass trace of a _voxcad robot whose .vxa sets <TimeBetweenTraces> > 0 (SS.CMTraceTime / SS.CMTrace, VX/VX_Sim.cpp:1537-1547;
 * recorded on the device at the end of the step in which it falls due): per point 4 doubles (time, x, y, z); count_out = points
 * recorded (may exceed capacity).  With <SaveTraces> the result XML carries the same points as <CMTrace> (VX/VX_SimGA.cpp:170-184). */
int  vxh_get_cm_trace(const vxh_engine* e, int robot, double* out4n, int capacity, int* count_out);
int  vxh_get_counters(const vxh_engine* e, vxh_counters* out);
/* how many internal bonds of the batch are currently in the large-angle branch of CVXS_BondInternal::CalcLinForce
 * (VX/VXS_BondInternal.cpp:72-126, the SmallAngle flag): the two branches differ 2.5x in arithmetic, so a throughput figure
 * should say which mix it was measured on */
int  vxh_count_bond_modes(const vxh_engine* e, long long* large_angle_out, long long* total_out);
/* Options (all have working defaults):
 *   "tiled"             0 = never; 1 (default) = the robots the one-workgroup-per-robot kernel cannot take (more than 1024 voxels,
 *                       oversized class tables) are stepped by the multi-workgroup kernel; 2 = every robot it supports (tests).
 *                       With the default, which kernel steps a robot depends on the robot alone, so its result is the same, bit for
 *                       bit, whatever batch it is in.
 *   "tile_small"        1 = also tile large robots (more than 512 voxels) when the population occupies at most a quarter of the CUs
 *                       (measured in round 6 on one 64-robot shard of the bench population: 14.0 us per step against 12.0 with the default
 *                       kernels -- it no longer pays there; kept for A/B), at the price that such a robot's last bits then
 *                       depend on the population it is evaluated in (the two kernels agree to 1e-12 voxel, not to the bit).  Default 0.
 *   "tiles_per_robot"   > 0 requests a tile count (tests).
 *   "wide"              1 (default) = robots of up to 512 voxels and 1023 bonds are stepped by the wide kernel (one lane per bond, all
 *                       three axes at once, forces summed in the reference's order); 0 = by the resident kernel with its three bond
 *                       rounds (cross-checks).  Like "tiled" a function of the robot alone.
 *   "wide_two_tiles"    1 (default) = the wide kernel keeps a second pose tile in LDS where the robot's layout leaves room for it and
 *                       steps with two workgroup barriers instead of three; 0 = one tile (cross-checks).  Same arithmetic, same bits.
 *   "col_cap"           partners a collision row can hold.  0 (default) = every other surface voxel of the robot, i.e. unbounded like
 *                       the reference's lists (CVX_Sim::CreateColBond has no cap); memory: 12-16 bytes x nsurf^2 per colliding robot
 *                       (5 MB for a 10x10x10 robot), touched only as far as rows really grow.  n > 0 bounds the rows at n partners
 *                       (less memory); a robot one of whose rows would need more ends with VXH_ROBOT_COL_OVERFLOW.
 *   "pair"              1 = robots of 769-1024 voxels without a surface mesh are stepped by k_robot_pair (512 threads, two voxels and up to
 *                       two bonds per axis per lane; same bits as k_robot_steps<1024>), 2 = also those of 513-768; 0 (default): measured
 *                       20-30 % slower than the kernels it replaces (DESIGN.md "Pair path").  "pair_sel": its rotation-vector form (A/B).
 *   "fused"             0 = robots the tiled kernel does not take go through the streaming kernels (cross-checks)
 *                       These (tiled .. fused) belong to the uploaded batch: set them before the first vxh_run / vxh_step or right
 *                       after vxh_reset (VXH_ERR_STATE once a step has been taken: a robot that changed kernels in the middle of a run
 *                       would read another kernel's bond history).  The tiles of a robot wait for each other on the
 *                       device: an engine that tiles must own its GPU (with another process on the same GPU set "tiled" to 0).
 *   "steps_per_launch"  time steps per launch of the resident / tiled kernels (default 1024).  A launch of a self-colliding population
 *                       carries ~0.07 ms of fixed cost (0.27 until round 3), and a call ~0.04 ms on the host, so vxh_step(e, n) with a
 *                       small n is paid for: 20 steps at a time run at ~35 us per step where 1000 at a time run at ~30.5 (512 robots
 *                       of 10x10x10).
 *   "graph_steps"       streaming kernels: step rounds per captured hipGraph (0 = plain launches)
 *   "shape_descriptors" 1 = _voxcad robots added FROM NOW ON carry the deformable surface mesh land_water robots always carry (they are stepped
 *                       by the kernel variants that record the directional strains the deformed mesh needs): vxh_get_mesh /
 *                       vxh_get_shape_descriptors then answer for them -- `voxelyze --computeShapeDescriptors`.  Default 0.
 *   "host_results"      1 = vxh_get_result evaluates every tag on the host from the downloaded voxel state instead of from the
 *                       device-side reductions (cross-checks; the numbers are the same) */
int  vxh_set_option(vxh_engine* e, const char* key, double value);

const char* vxh_strerror(int status);
const char* vxh_last_error(const vxh_engine* e);
const char* vxh_version(void);
/* HIP devices this process can use (0 without a GPU): what a one-process-per-GPU launcher takes LOCAL_RANK modulo
 * (evosoro_amd/parallel.py), independent of whether the caller's torch build sees the GPUs */
int  vxh_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
