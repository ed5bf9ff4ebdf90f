// Batched engine: owns the robots of one population shard and their SoA state on ONE HIP device.
#pragma once
#include <algorithm>
#include <stdexcept>
#include <cstdlib>
#include <functional>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/vxhip.h"
#include "model.hpp"

namespace vxh {

struct HostState {                      // final/current state of one robot, downloaded from the device
    std::vector<double> pos, quat, scale, lin_mom, ang_mom;   // pos [3*n] xyz-interleaved, quat [4*n] wxyz
    std::vector<double> strain;                                // land_water: [6*n] StrainPosDirsCur xyz, StrainNegDirsCur xyz of the last step
    double cur_time = 0, ini_cm[3] = {0, 0, 0}, eol_post_y = 0;
    int steps = 0, status = 0, cm_init = 0, rebuilds = 0;
    // reductions of the final state done on the device (k_results); valid when `reduced`
    bool reduced = false;
    double red_cm[3] = {0, 0, 0}, d2max = 0, d2min = 0, ymax = 0, ymin = 0;
    int touching = 0, feet = 0;
    std::vector<double> cm_trace;                              // [4*k] (time, x, y, z): SS.CMTraceTime / SS.CMTrace
};

// the tiles of a robot gave up waiting for each other (VXH_ROBOT_SYNC_TIMEOUT): Engine::advance makes a call that started from the
// imported state again without the tiled kernel; elsewhere it surfaces as VXH_ERR_HIP
struct TileTimeout : std::runtime_error { using std::runtime_error::runtime_error; };

class Engine {
public:
    Engine(int variant, int device_id);       // throws std::runtime_error when no HIP device is usable
    ~Engine();
    int add_vxa(const char* data, size_t len);              // returns robot index; throws
    int add_vxa_files(const std::vector<std::string>& paths);   // parse + build on all host cores, append in order; returns first index
    int add_arrays(const char* template_vxa, size_t len, const vxh_robot_arrays* robots, int n, bool round_like_text);   // returns first index
    // the same in two steps: parse + build (throws, changes nothing), then append
    std::vector<RobotModel> build_vxa(const char* data, size_t len) const;
    std::vector<RobotModel> build_vxa_files(const std::vector<std::string>& paths) const;
    std::vector<RobotModel> build_arrays(const char* template_vxa, size_t len, const vxh_robot_arrays* robots, int n, bool round_like_text) const;
    std::vector<VxaModel> models_from_arrays(const char* template_vxa, size_t len, const vxh_robot_arrays* robots, int n, bool round_like_text) const;   // checks + copies
    std::vector<RobotModel> build_models(std::vector<VxaModel>&& models) const;                                                                 // the expensive part
    int append(std::vector<RobotModel>&& built);             // returns first index
    int num_robots() const { return (int)robots_.size(); }
    const RobotModel& robot(int i) const { return robots_[i]; }
    void run();                                // to completion
    void run_launch();                         // ... in two halves: enqueue everything / wait + accounting (EngineSet pipelines engines with them)
    void run_finish();
    void step(long long n);                    // at most n more steps per robot
    void reset();
    void clear();
    void result(int robot, vxh_result* out);
    const std::vector<double>& trace_of(int robot);            // (downloads the state if needed)
    void state14(int robot, double* out, int capacity);
    void counters(vxh_counters* out) const { *out = counters_; }
    int cm_trace(int robot, double* out4n, int capacity);      // returns the number of points recorded
    std::vector<double> angle_excess(int robot, bool at_end);  // land_water: discrete curvature of every mesh vertex, rest state / current state
    void shape(int robot, bool at_end, struct MeshShape& out); // the surface mesh and its descriptors, rest state / current state
    void bond_modes(long long* large_angle, long long* total);   // SmallAngle flags of every bond, downloaded
    void set_option(const std::string& key, double value);
    void check_option(const std::string& key, double value) const;   // throws what set_option would throw, changes nothing
    int variant() const { return variant_; }
    int device() const { return device_id_; }
    // EngineSet moves robots between the engines of a handle (host-side models only; the batch is rebuilt on the next run)
    std::vector<RobotModel> take_robots();
    // The handle's veto on the multi-workgroup kernel, apart from the user's option `tiled`: two engines of ONE handle on ONE device
    // must not both run it (the tiles of a robot wait for each other).  Takes effect at the next batch assembly; the option keeps
    // its value and acts again when the veto is lifted (EngineSet::gather / clear).
    void set_tiling_allowed(bool on);
    void add_run_seconds(double s) { counters_.run_seconds += s; }
    void give_robots(std::vector<RobotModel>&& models);
    void drop_graph();

private:
    struct Device;                             // HIP-side members (engine.hip)
    void prepare();                            // build + upload the batch
    void advance(long long max_rounds);        // launch step rounds and wait
    void advance_launch(long long max_rounds);
    void advance_finish();
    void quiesce();                            // after a launch sequence that failed halfway: every stream idle, nothing pending
    void download();
    void download_control(bool already_copied = false);
    void download_reduced();                   // k_results + the traces: what results need, without the voxel state
    int variant_, device_id_;
    std::vector<RobotModel> robots_;
    std::vector<HostState> host_;
    std::unique_ptr<Device> dev_;
    bool prepared_ = false, state_downloaded_ = false, control_downloaded_ = false, reduced_downloaded_ = false;
    bool host_results_ = false;                // option host_results: every result from the downloaded voxel state on the host (cross-check)
    long long rounds_done_ = 0;
    int graph_steps_ = 32;                     // streaming path: step rounds captured per hipGraph launch (0 = plain launches)
    int dbg_ = 0;
    bool fused_ = true;                        // one-workgroup-per-robot fused kernel when every robot has <= 1024 voxels
    // fused and tiled paths: time steps per kernel launch.  A launch of a self-colliding population carries ~0.07 ms of fixed cost (0.27 ms until round 3; it
    // ends with its slowest workgroup: DESIGN.md "The cost of a launch"); a whole evaluation of the bench population (7806 steps),
    // us per step by launch length: 256: 31.2 | 512: 30.7 | 1024: 30.6 | 2048: 30.6 | 8192 (one launch): 30.3.  (256 until late in round 2.)
    int steps_per_launch_ = 1024;
    bool tiling_allowed_ = true;               // the handle's veto (set_tiling_allowed): several engines of one handle share this device
    int tiled_ = 1;                            // several workgroups per robot (kernels_tiled.hpp): 0 never, 1 when the population is too
                                               // small to fill the CUs one robot each or a robot has more than 1024 voxels, 2 always
    int tiles_per_robot_ = 0;                  // 0 = chosen from the population size; > 0: requested for every tiled robot (tests)
    // fault injection for tests/test_gpu_tiled.py: VXH_INJECT_TILE_TIMEOUT=n, the n-th tiled call of the engine reports a timeout (1 = the first)
    int inject_tile_timeout_ = std::getenv("VXH_INJECT_TILE_TIMEOUT") ? std::max(1, std::atoi(std::getenv("VXH_INJECT_TILE_TIMEOUT"))) : 0;
    bool wide_two_tiles_ = true;               // ... with a second pose tile in LDS where it fits (two barriers per step instead of three); 0: cross-checks
    bool wide_ = true;                         // small robots (up to 512 voxels, 1023 bonds) go to the wide kernel (kernels_wide.hpp); 0: resident kernel
    // k_robot_pair (kernels_pair.hpp: 512 threads, two voxels and up to two bonds per axis per lane) instead of k_robot_steps<1024> for robots of
    // 769-1024 voxels without a surface mesh (1), also instead of <768> for those of 513-768 (2).  OFF by default: bit-identical to <1024>, but
    // measured 20-30 % slower (round 5, DESIGN.md section 4 "Pair path": two wavefronts per SIMD do not hide the latency of a dependent FP64 chain)
    int pair_ = 0;
    bool shape_descriptors_ = false;           // _voxcad robots added from now on carry the deformable surface mesh (voxelyze --computeShapeDescriptors): they are
                                               // stepped by the MESH kernel variants, which record the directional strains the final mesh needs
    bool pair_sel_ = false;                    // ... with the rotation-vector factor in select form (kernels.hpp rotvec_factor<SEL>; A/B switch)
    int col_cap_ = 0;                          // partners a contact row can hold; 0 = every other surface voxel (unbounded, like the reference)
    bool tile_small_ = false;                  // also tile large robots the resident kernel could take when the population is small (see prepare())
    unsigned tile_gen_ = 0;                    // launch generation of the tiled kernel (high half of the tiles' flag words)
    vxh_counters counters_{};
};

// One handle of the C ABI: one engine per device.  With a single device everything is forwarded; with several (vxh_create_multi)
// the robots are collected by the first engine and, at the first run / step after an addition, partitioned over the devices by
// cost (greedy longest-processing-time on voxels x planned steps, like evosoro_amd/parallel.py does across processes); every
// device then steps its share from its own host thread.  The results stay in host memory of the one process, so this path needs
// no collective at all -- the RCCL gather of SURVEY section 8(e) belongs to the one-process-per-GPU route (parallel.py).
class EngineSet {
public:
    EngineSet(int variant, const std::vector<int>& device_ids);
    int add_vxa(const char* data, size_t len);
    int add_vxa_files(const std::vector<std::string>& paths);
    int add_arrays(const char* template_vxa, size_t len, const vxh_robot_arrays* robots, int n, bool round_like_text);
    int num_robots() const;
    const RobotModel& robot(int i) const;
    void run();
    void step(long long n);
    void reset();
    void clear();
    void result(int robot, vxh_result* out);
    void state14(int robot, double* out, int capacity);
    void counters(vxh_counters* out) const;
    void set_option(const std::string& key, double value);
    int cm_trace(int robot, double* out4n, int capacity);
    const std::vector<double>& trace_of(int robot);
    std::vector<double> angle_excess(int robot, bool at_end);
    void shape(int robot, bool at_end, struct MeshShape& out);
    void bond_modes(long long* large_angle, long long* total);
    int n_devices() const { return (int)engines_.size(); }

private:
    void gather();          // every robot back to engine 0, in order (before an addition)
    void distribute();      // ... and out to the devices (before a run)
    void each(const std::function<void(Engine&)>& body);   // on every engine that holds robots, concurrently; rethrows the first failure
    void no_tiling_if_shared();   // several engines of the handle on one device are about to hold robots: no multi-workgroup kernel
    void tiling_allowed_again();  // ... the robots are back on one engine
    bool repeated_ = false;       // the device list names a device more than once
    void flush_pending();   // build what a pipelining handle has only checked so far (any reader or addition before the run)
    std::vector<std::unique_ptr<Engine>> engines_;
    std::vector<std::pair<int, int>> where_;   // robot -> (engine, index there), valid while distributed_
    bool distributed_ = false;
    // Several engines on ONE device (vxh_create_multi with a repeated device id): the handle pipelines a generation handed over as
    // arrays -- vxh_add_robots only checks and copies, vxh_run builds, uploads and launches chunk after chunk (one per engine) before
    // it waits for the first, so the device steps chunk k while the host cores build chunk k + 1
    bool pipelined_ = false;
    std::vector<VxaModel> pending_;
};

int hip_device_count();      // engine.hip: HIP devices this process can use (0 when the runtime reports none)

// results.cpp: the numbers of CVX_SimGA::WriteResultFile from a final state, and the XML text
void compute_result(const RobotModel& model, const HostState& st, vxh_result* out);
std::string result_xml(const RobotModel& model, const vxh_result& res, const std::vector<double>& cm_trace, double shape_start = -1.0, double shape_end = -1.0);
double shape_complexity_as_the_reference_prints_it(const RobotModel& model, const std::vector<double>& angle_excess);
const std::vector<double>& empty_trace();
double convex_hull_volume(const std::vector<double>& xyz);   // results.cpp: what stands in for the reference's external qhull
// results.cpp: per-vertex angle excess of the surface mesh (LW/VX_MeshUtil.cpp:956-1014); pos / quat / strain null = the rest state
void mesh_angle_excess(const RobotModel& model, const double* pos, const double* quat, const double* strain, std::vector<double>& out);
// results.cpp: everything CVX_MeshUtil knows of a robot's surface mesh in one state (voxelyzeMain/main.cpp:65-88,113-126): vertex positions,
// facets (vertex triples), enclosed volume, volume of the convex hull, per-vertex angle excesses
struct MeshShape { std::vector<double> verts; std::vector<int> facets; double robot_volume = 0, hull_volume = 0; std::vector<double> angle_excess; };
void mesh_shape(const RobotModel& model, const double* pos, const double* quat, const double* strain, MeshShape& out);

}  // namespace vxh
