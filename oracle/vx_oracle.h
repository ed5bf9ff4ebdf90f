/* TEST INFRASTRUCTURE ONLY (oracle/): scalar CPU restatement of the reference Voxelyze hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (evosoro_amd/, libvxhip.so) never links, imports or executes it.
 *
 * PARITY PINNED: this restatement is checked, state-for-state, against traces produced by the reference C++
 * compiled from its own sources (oracle/_ref/vxprobe, built by oracle/Makefile) and against the result XMLs
 * written by the reference binary (tests/golden/expected/, tests/test_oracle_vs_reference.py).
 */
#ifndef VX_ORACLE_H
#define VX_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

/* Already-parsed .vxa content (the XML reading is done by the caller: tests use Python's ElementTree so
 * that the oracle path shares no parsing code with the product's C++ reader). Defaults the caller must
 * apply for absent tags are those of the reference readers (VX_Sim.cpp:263-354, VX_Environment.cpp:123-234,
 * VX_Object.cpp:1064-1073,1344-1441,1733-1900). */
typedef struct vxo_model {
    int variant;                 /* 0 = _voxcad, 1 = _voxcad_land_water */
    /* VXC */
    int nx, ny, nz;
    double lattice_dim;          /* Lattice_Dim; *_Dim_Adj assumed 1, offsets 0 (what evosoro writes) */
    const unsigned char* structure; /* nx*ny*nz material indices, x fastest then y then z; 0 = empty */
    int nmat;                    /* palette size INCLUDING index 0 ("Erase") */
    const double* mat_E;         /* [nmat] Elastic_Mod */
    const double* mat_rho;       /* [nmat] Density */
    const double* mat_nu;        /* [nmat] Poissons_Ratio */
    const double* mat_cte;       /* [nmat] CTE */
    const double* mat_us;        /* [nmat] uStatic */
    const double* mat_ud;        /* [nmat] uDynamic */
    const double* phase_offset;  /* [nvox] by simulation index, or NULL (tag <PhaseOffset>) */
    const double* temp_amp_damp; /* [nvox] or NULL (<TempAmpDamp>) ; _voxcad only */
    const double* stiffness;     /* [nvox] or NULL (<Stiffness>) evolved per-voxel modulus */
    /* _voxcad development layers (VX_Object.cpp:1910-2140), each [nvox] by simulation index or NULL */
    const double* final_phase_offset;   /* <FinalPhaseOffset> */
    const double* final_temp_amp_damp;  /* <FinalTempAmpDamp> */
    const double* initial_voxel_size;   /* <InitialVoxelSize>, in units of GrowthAmplitude around the nominal size */
    const double* final_voxel_size;     /* <FinalVoxelSize> */
    const double* growth_time;          /* <GrowthTime>, fraction of the available lifetime */
    const double* start_growth_time;    /* <StartGrowthTime>, fraction of the lifetime after InitCmTime */
    /* Simulator */
    double dt_frac, bond_damping_z, col_damping_z, slow_damping_z;
    int self_col_enabled, col_system;
    double collision_horizon;
    int stop_type;               /* StopCondition enum, VX_Enums.h:52-62 */
    double stop_value, afterlife_time, midlife_freeze_time, init_cm_time;
    double min_temp_fact;
    /* Environment */
    int grav_enabled; double grav_acc; int floor_enabled;
    int temp_enabled; double temp_amplitude, temp_base, temp_period; int vary_temp_enabled;
    double growth_amplitude;
    double min_growth_time;      /* <MinGrowthTime>, VX_Environment.cpp:210 */
    int sticky_floor;
    /* land_water only */
    int fluid_env; double aggregate_drag_coef;
    /* _voxcad only: centre-of-mass trace (<TimeBetweenTraces>, VX_Environment.cpp:215; recorded in UpdateStats, VX_Sim.cpp:1537-1547) */
    double time_between_traces;
} vxo_model;

typedef struct vxo_info {
    int nvox, nbond, nsurf, ncol, steps, status; /* status 0 running, 1 finished, 2 diverged, 3 empty */
    int cm_initialized, n_small_angle, col_rebuilds, reserved;
    double opt_dt, dt, cur_time, max_vox_vel;
    double cur_cm[3], ini_cm[3];
} vxo_info;

/* every number the reference writes into its result XML (VX_SimGA.cpp:145-168; LW/VX_SimGA.cpp:58-68) */
typedef struct vxo_result {
    int status, steps, nvox, nbond;
    double dt, cur_time, lifetime;
    double ini_cm[3], cur_cm[3];
    double norm_final_dist, norm_regime_dist, norm_frozen_dist, final_dist, final_dist_y;
    double anterior_dist, posterior_dist, anterior_y, posterior_y, end_of_life_posterior_y, fall_adj_post_y;
    double num_non_feet_touching_floor, num_touching_floor;
    /* land_water */
    double norm_abs_disp, norm_dist_x, norm_dist_y, norm_dist_z;
} vxo_result;

typedef struct vxo_sim vxo_sim;

vxo_sim* vxo_create(const vxo_model* m);          /* Import + ResetSimulation + CalcMaxDt */
void     vxo_destroy(vxo_sim* s);
long     vxo_step(vxo_sim* s, long max_steps);    /* main loop; returns steps taken (stops on stop condition / divergence) */
void     vxo_get_info(const vxo_sim* s, vxo_info* out);
void     vxo_get_state(const vxo_sim* s, double* out14n); /* per voxel: pos3, quat wxyz, scale, vel3, angvel3 */
void     vxo_get_bond_table(const vxo_sim* s, int* vox1, int* vox2, int* axis);
void     vxo_get_bond_modes(const vxo_sim* s, int* small);      /* instrument: SmallAngle flag per bond */
void     vxo_get_result(const vxo_sim* s, vxo_result* out);
void     vxo_get_constants(const vxo_sim* s, double* vox12n, double* bond23n); /* per voxel / per bond constants of Import (see vx_oracle.c) */
void     vxo_jitter(vxo_sim* s, unsigned seed);   /* test instrument: every position component one ulp up or down (see vx_oracle.c) */
void     vxo_set_state(vxo_sim* s, const double* in14n); /* test instrument: overwrite the voxels' state (layout of vxo_get_state) */
int      vxo_get_cm_trace(const vxo_sim* s, double* out4n, int capacity); /* (time, x, y, z) per entry; returns the number recorded */
double   vxo_alg_bytes_per_step(const vxo_sim* s); /* 224*nvox + 144*nbond, SURVEY.md 8(d) */

#ifdef __cplusplus
}
#endif
#endif
