// Engine implementation: batch assembly, HBM upload, step-round launches (hipGraph), download.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <exception>
#include <fstream>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <thread>

#include "engine.hpp"
#include "kernels.hpp"
#include "launch.hpp"

namespace vxh {

namespace {

#define HIP_OK(call) hip_check((call), #call)

}  // namespace

int hip_device_count()
{
    int count = 0;
    return hipGetDeviceCount(&count) == hipSuccess ? count : 0;
}

struct Engine::Device {
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // fused path: robots grouped by kernel variant (workgroup size 256/512/768/1024, exchange buffers, fluid), one
    // stream per group so that the groups fill the chip together
    struct Group {
        int block = 0, nacc = 0, fluid = 0, tabg = 0;   // template arguments of k_robot_steps
        int wide = 0;                     // 1: k_robot_wide<block, fluid, tabg> (kernels_wide.hpp)
        int two_tiles = 0;                // wide kernel: a second pose tile in LDS (two barriers per step instead of three)
        int pair = 0;                     // 1: k_robot_pair<tabg, sel> (kernels_pair.hpp; block = 512 threads, 1024 voxel slots); 2: its SEL form
        int count = 0;
        const int* list = nullptr;
        // dispatch order of short launches (kernels_fused.hpp fused_dispatch_slot): one bit per list entry, written by a launch for the next
        unsigned long long* order_bits[2] = {nullptr, nullptr};
        int order_cur = 0;
        bool order_valid = false, order_use = false;
        size_t lds = 0;                   // dynamic LDS bytes
        hipStream_t stream = nullptr;
        hipEvent_t t0 = nullptr, t1 = nullptr;
        std::vector<int> robots;
    };
    std::vector<Group> groups;
    std::vector<hipStream_t> group_streams;   // created on demand, reused across prepare() calls
    std::vector<hipEvent_t> group_events;
    const unsigned char* streamed_all = nullptr;    // masks for DBatch::streamed: every robot / the robots outside the launch groups
    const unsigned char* streamed_rest = nullptr;
    int n_rest = 0, n_all = 0;            // robots (with voxels) that only the streaming kernels can step / that they step with fused = 0
    const unsigned char* graph_mask = nullptr;      // the mask the captured graph was recorded with
    bool any_fluid = false;               // some robot is in a fluid: the streaming rounds include the drag kernels
    std::vector<void*> allocs;
    DBatch B{};
    std::vector<DRobot> h_robot;
    std::vector<int> vox_begin;           // per robot, global slot of voxel 0
    std::vector<int> trace_begin, trace_cap;   // per robot, entries of DBatch::trace
    int total_trace = 0;
    std::vector<int> surf_begin;
    int total_surf = 0;
    long long max_planned = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_rounds = 0;
    int max_nvox = 0;                     // largest robot of the batch (selects the fused block size)
    // tiled path: the robots cut into tiles, packed into launches that fit the chip (all tiles of a robot in one launch)
    struct TileLaunch {
        int tabg = 0, mesh = 0, count = 0;   // template arguments of k_tile_steps (mesh: land_water robots, whose strains are part of the state)
        int small = 0;                    // ... and SMALL: every tile of the launch within VXH_TILE_S_* (compile-time LDS strides)
        const int* list = nullptr;        // tile ids
        size_t lds = 0;
        std::vector<int> tile_ids, robots;
    };
    std::vector<TileLaunch> tile_launches;
    std::vector<unsigned char> robot_tiled;   // per robot
    std::vector<int> robot_tiles;             // per robot: number of tiles (0 = not tiled)
    hipStream_t tile_stream = nullptr;        // ONE stream for every tiled launch: two of them must never share the chip
    hipEvent_t tile_t0 = nullptr, tile_t1 = nullptr;
    int n_cu = 0;
    DResult* results = nullptr;               // [robots] output of k_results (freed with the batch)
    DRobotState* h_rstate = nullptr;          // pinned host mirror of DBatch::rstate: copied back on the call's own stream, behind its kernels
    size_t h_rstate_cap = 0;
    // a call that has been launched and not yet waited for (advance_launch / advance_finish)
    struct Pending {
        bool active = false;
        long long todo = 0, launches = 0, tile_launch_count = 0;
        bool fused = false, tiled = false, streaming = false, single = false;
        std::vector<int> steps_before;
        std::vector<long long> group_launches;
    } pending;
    int reb_blocks = 0;                   // streaming path: collision-rebuild blocks appended to k_bonds
    const int* reb_robot = nullptr;
    const int* reb_i0 = nullptr;

    // Allocations and uploads go through the engine's own stream (stream-ordered pool): hipMalloc / hipMemcpy / hipFree synchronise
    // with the whole device, and a handle that pipelines a generation prepares chunk k + 1 while the kernels of chunk k run --
    // measured: "allocations + uploads" of the second chunk 6 ms alone, 38 ms next to a running kernel with the synchronous calls.
    // The host vectors are pageable: hipMemcpyAsync returns once it has staged them; prepare() ends with a wait for this stream.
    bool async_alloc = std::getenv("VXH_SYNC_ALLOC") == nullptr;
    bool dbg_alloc = std::getenv("VXH_DBG_ALLOC") != nullptr;
    std::vector<size_t> alloc_bytes;
    std::vector<std::pair<void*, std::vector<char>>> dbg_uploads;      // (VXH_DBG_ALLOC: what every upload sent, for verify_uploads)
    void verify_uploads(const char* when)
    {
        if (!dbg_alloc) return;
        int n_bad = 0;
        for (size_t k = 0; k < dbg_uploads.size(); ++k) {
            std::vector<char> back(dbg_uploads[k].second.size());
            if (hipMemcpy(back.data(), dbg_uploads[k].first, back.size(), hipMemcpyDeviceToHost) != hipSuccess) { std::fprintf(stderr, "vxhip: VERIFY %s: read-back of upload #%zu failed\n", when, k); continue; }
            size_t first = back.size(), count = 0, zeros = 0;
            for (size_t i = 0; i < back.size(); ++i) if (back[i] != dbg_uploads[k].second[i]) { if (first == back.size()) first = i; ++count; zeros += back[i] == 0; }
            if (count) { ++n_bad; std::fprintf(stderr, "vxhip: VERIFY %s: upload #%zu at %p (%zu bytes) differs in %zu bytes from offset %zu on (%zu of them read 0)\n", when, k, dbg_uploads[k].first, back.size(), count, first, zeros); }
        }
        std::fprintf(stderr, "vxhip: VERIFY %s: %d of %zu uploads differ\n", when, n_bad, dbg_uploads.size());
    }
    void* raw_alloc(size_t bytes)
    {
        void* p = nullptr;
        if (async_alloc) HIP_OK(hipMallocAsync(&p, bytes, stream)); else HIP_OK(hipMalloc(&p, bytes));
        if (dbg_alloc) {
            for (size_t k = 0; k < allocs.size(); ++k)
                if ((char*)p < (char*)allocs[k] + alloc_bytes[k] && (char*)allocs[k] < (char*)p + bytes)
                    std::fprintf(stderr, "vxhip: ALLOCATION OVERLAP: new %p + %zu against live #%zu %p + %zu\n", p, bytes, k, allocs[k], alloc_bytes[k]);
            alloc_bytes.push_back(bytes);
        }
        allocs.push_back(p);
        return p;
    }
    template <class T>
    T* upload(const std::vector<T>& h, size_t min_count = 1)
    {
        size_t n = std::max(h.size(), min_count);
        void* p = raw_alloc(n * sizeof(T));
        if (!h.empty()) {
            if (async_alloc) HIP_OK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, stream));
            else HIP_OK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
            if (dbg_alloc) dbg_uploads.emplace_back(p, std::vector<char>((const char*)h.data(), (const char*)h.data() + h.size() * sizeof(T)));
        }
        return (T*)p;
    }
    template <class T>
    T* alloc_zero(size_t n)
    {
        if (n == 0) n = 1;
        void* p = raw_alloc(n * sizeof(T));
        if (async_alloc) HIP_OK(hipMemsetAsync(p, 0, n * sizeof(T), stream)); else HIP_OK(hipMemset(p, 0, n * sizeof(T)));
        return (T*)p;
    }
    template <class T>
    T* alloc_raw(size_t n)
    {
        if (n == 0) n = 1;
        return (T*)raw_alloc(n * sizeof(T));
    }
    void free_all()
    {
        if (h_rstate) { hipHostFree(h_rstate); h_rstate = nullptr; h_rstate_cap = 0; }
        if (graph_exec) { hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
        if (graph) { hipGraphDestroy(graph); graph = nullptr; }
        if (async_alloc && stream) { for (void* p : allocs) hipFreeAsync(p, stream); if (!allocs.empty()) hipStreamSynchronize(stream); }
        else for (void* p : allocs) hipFree(p);
        allocs.clear();
        alloc_bytes.clear();
        dbg_uploads.clear();
        results = nullptr;
        B = DBatch{};
    }
};

Engine::Engine(int variant, int device_id) : variant_(variant), device_id_(device_id), dev_(new Device)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        throw std::runtime_error("no HIP device available (libvxhip has no CPU path)");
    if (device_id < 0 || device_id >= count) throw std::runtime_error("device id out of range");
    HIP_OK(hipSetDevice(device_id));
    // The stream-ordered pool keeps what the engines free (release threshold: everything) instead of handing it back to the driver at every
    // synchronisation.  Round 6, found by looping the GPU tests: a call that failed halfway (a refused tile kernel), then reset() -- free,
    // release, acquire again, upload -- left the FIRST TWO uploads of the new batch (DRobot and DRobotState, the first 0x290 bytes of the
    // re-acquired chunk) zeroed in 3 % of fresh processes: correct right behind their copies, zero at the end of prepare() (VXH_DBG_ALLOC=1
    // reads every upload back), i.e. cleared by something that is not the engine's; never with hipMalloc (VXH_SYNC_ALLOC), never with the
    // memory kept (0 of 250 against 7-8 of 250; gpurun_out/r6/refusedloop*.txt, HISTORY "Round 6").  A population's footprint is what the
    // process needed at its peak anyway, and re-mapping it every generation costs time.  VXH_POOL_RELEASE=1: the runtime's default back.
    if (!std::getenv("VXH_POOL_RELEASE")) {
        hipMemPool_t pool = nullptr;
        if (hipDeviceGetDefaultMemPool(&pool, device_id) == hipSuccess && pool) {
            uint64_t keep = ~0ull;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
        (void)hipGetLastError();
    }
    HIP_OK(hipStreamCreateWithFlags(&dev_->stream, hipStreamNonBlocking));
    HIP_OK(hipEventCreate(&dev_->ev0));
    HIP_OK(hipEventCreate(&dev_->ev1));
    HIP_OK(hipStreamCreateWithFlags(&dev_->tile_stream, hipStreamNonBlocking));
    HIP_OK(hipEventCreate(&dev_->tile_t0));
    HIP_OK(hipEventCreate(&dev_->tile_t1));
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, device_id));
    dev_->n_cu = prop.multiProcessorCount;
    // VXH_ENGINE_OPTIONS="key=value,key=value": defaults of vxh_set_option for every engine of the process, so that a caller
    // that cannot pass options (the voxelyze command line, a test matrix) still selects e.g. tiled=0
    if (const char* env = std::getenv("VXH_ENGINE_OPTIONS")) {
        std::stringstream list(env);
        std::string item;
        while (std::getline(list, item, ',')) {
            const size_t eq = item.find('=');
            if (eq == std::string::npos || eq == 0) throw std::runtime_error("VXH_ENGINE_OPTIONS: expected key=value, got '" + item + "'");
            set_option(item.substr(0, eq), std::stod(item.substr(eq + 1)));
        }
    }
}

Engine::~Engine()
{
    if (dev_) {
        hipSetDevice(device_id_);
        dev_->free_all();
        if (dev_->ev0) hipEventDestroy(dev_->ev0);
        if (dev_->ev1) hipEventDestroy(dev_->ev1);
        if (dev_->stream) hipStreamDestroy(dev_->stream);
        if (dev_->tile_stream) hipStreamDestroy(dev_->tile_stream);
        if (dev_->tile_t0) hipEventDestroy(dev_->tile_t0);
        if (dev_->tile_t1) hipEventDestroy(dev_->tile_t1);
        for (hipStream_t st : dev_->group_streams) hipStreamDestroy(st);
        for (hipEvent_t ev : dev_->group_events) hipEventDestroy(ev);
    }
}

// The add_* calls are build_* (parse + model building: may throw, touch nothing) followed by append(): a refused call leaves the
// engine -- and, behind a handle over several devices, the distribution of the robots -- as it was.
int Engine::append(std::vector<RobotModel>&& built)
{
    const int first = (int)robots_.size();
    for (auto& m : built) robots_.push_back(std::move(m));
    prepared_ = false;
    state_downloaded_ = control_downloaded_ = reduced_downloaded_ = false;
    return first;
}

std::vector<RobotModel> Engine::build_vxa(const char* data, size_t len) const
{
    VxaModel vxa = read_vxa(data, len, variant_);
    if (shape_descriptors_) vxa.want_mesh = true;
    if (!vxa.unsupported.empty()) {
        std::string msg = "unsupported .vxa feature(s):";
        for (const auto& u : vxa.unsupported) msg += " [" + u + "]";
        throw std::invalid_argument(msg);
    }
    std::vector<RobotModel> out;
    out.push_back(build_robot(vxa));
    return out;
}
int Engine::add_vxa(const char* data, size_t len) { return append(build_vxa(data, len)); }
int Engine::add_vxa_files(const std::vector<std::string>& paths) { return append(build_vxa_files(paths)); }
int Engine::add_arrays(const char* template_vxa, size_t len, const vxh_robot_arrays* in, int n, bool round_like_text)
{ return append(build_arrays(template_vxa, len, in, n, round_like_text)); }

// A generation arrives as hundreds of .vxa files; reading, XML parsing and model building (hop-distance lists, bond
// classes, drag mesh) are independent per robot and take longer than the GPU needs to simulate them, so they are
// spread over the host cores.  Robots are appended in the order of `paths`; the first failure (in that order) is
// rethrown and nothing is appended.
std::vector<RobotModel> Engine::build_vxa_files(const std::vector<std::string>& paths) const
{
    const int n = (int)paths.size();
    std::vector<RobotModel> built(n);
    std::vector<std::exception_ptr> errors(n);
    std::atomic<int> next{0};
    auto worker = [&]() {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            try {
                std::ifstream in(paths[i], std::ios::binary);
                if (!in) throw std::runtime_error("io: cannot open " + paths[i]);
                std::stringstream ss;
                ss << in.rdbuf();
                const std::string text = ss.str();
                VxaModel vxa = read_vxa(text.data(), text.size(), variant_);
                if (shape_descriptors_) vxa.want_mesh = true;
                if (!vxa.unsupported.empty()) {
                    std::string msg = "unsupported .vxa feature(s):";
                    for (const auto& u : vxa.unsupported) msg += " [" + u + "]";
                    throw std::invalid_argument(msg + " in " + paths[i]);
                }
                built[i] = build_robot(vxa);
            } catch (...) {
                errors[i] = std::current_exception();
            }
        }
    };
    // 32 workers: on the 256-thread MI355X host 512 robots import in 15 ms with 32 threads and in 200 ms with 64 (allocator contention)
    const int nthreads = std::max(1, std::min({n, (int)std::thread::hardware_concurrency(), 32}));
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    for (int i = 0; i < n; ++i) if (errors[i]) std::rethrow_exception(errors[i]);
    return built;
}

// A generation handed over as arrays (vxh_add_robots): one parsed template + per-robot lattice and layers.  The VxaModel of a robot
// is exactly what read_vxa builds from the file the writer would have produced: material digits as they are, layer values taken by
// occupied-voxel counter in file order (VX_Object.cpp:1879-1900), optionally through the writer's decimal text.
std::vector<RobotModel> Engine::build_arrays(const char* template_vxa, size_t len, const vxh_robot_arrays* in, int n, bool round_like_text) const
{
    return build_models(models_from_arrays(template_vxa, len, in, n, round_like_text));
}

// ... in two steps: the cheap one (template parse, argument checks, the arrays copied into .vxa models: everything that can be
// REFUSED) and the expensive one (model building).  A handle that pipelines a generation does the first when the robots are
// added and the second chunk by chunk while earlier chunks already step (EngineSet::run).
std::vector<RobotModel> Engine::build_models(std::vector<VxaModel>&& models) const
{
    const int n = (int)models.size();
    std::vector<RobotModel> built(n);
    std::vector<std::exception_ptr> errors(n);
    std::atomic<int> next{0};
    auto worker = [&]() {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            try { built[i] = build_robot(models[i]); } catch (...) { errors[i] = std::current_exception(); }
        }
    };
    const int nthreads = std::max(1, std::min({n, (int)std::thread::hardware_concurrency(), 32}));
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    for (int i = 0; i < n; ++i) if (errors[i]) std::rethrow_exception(errors[i]);
    return built;
}

std::vector<VxaModel> Engine::models_from_arrays(const char* template_vxa, size_t len, const vxh_robot_arrays* in, int n, bool round_like_text) const
{
    VxaModel base = read_vxa(template_vxa, len, variant_);
    if (shape_descriptors_) base.want_mesh = true;
    if (!base.unsupported.empty()) {
        std::string msg = "unsupported .vxa feature(s):";
        for (const auto& u : base.unsupported) msg += " [" + u + "]";
        throw std::invalid_argument(msg + " in the template");
    }
    struct LayerSlot { const char* tag; bool VxaModel::*has; std::vector<double> VxaModel::*values; bool land_only; };
    static const LayerSlot slots[] = {
        {"PhaseOffset", &VxaModel::has_phase_offset, &VxaModel::phase_offset, false},
        {"TempAmpDamp", &VxaModel::has_temp_amp_damp, &VxaModel::temp_amp_damp, false},
        {"Stiffness", &VxaModel::has_stiffness, &VxaModel::stiffness, false},
        {"FinalPhaseOffset", &VxaModel::has_final_phase_offset, &VxaModel::final_phase_offset, true},
        {"FinalTempAmpDamp", &VxaModel::has_final_temp_amp_damp, &VxaModel::final_temp_amp_damp, true},
        {"InitialVoxelSize", &VxaModel::has_initial_voxel_size, &VxaModel::initial_voxel_size, true},
        {"FinalVoxelSize", &VxaModel::has_final_voxel_size, &VxaModel::final_voxel_size, true},
        {"GrowthTime", &VxaModel::has_growth_time, &VxaModel::growth_time, true},
        {"StartGrowthTime", &VxaModel::has_start_growth_time, &VxaModel::start_growth_time, true},
    };
    for (const auto& sl : slots) { base.*(sl.has) = false; (base.*(sl.values)).clear(); }
    base.structure.clear();
    std::vector<VxaModel> built(n);
    std::vector<std::exception_ptr> errors(n);
    std::atomic<int> next{0};
    auto worker = [&]() {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            try {
                const vxh_robot_arrays& A = in[i];
                if (A.nx < 1 || A.ny < 1 || A.nz < 1 || (long long)A.nx * A.ny * A.nz > (1LL << 26) || !A.material) throw std::invalid_argument("bad lattice");
                if (A.n_layers < 0 || (A.n_layers > 0 && (!A.layer_tags || !A.layers))) throw std::invalid_argument("bad layer arrays");
                for (int l = 0; l < A.n_layers; ++l) if (!A.layer_tags[l] || !A.layers[l]) throw std::invalid_argument("null layer tag or layer");
                VxaModel m = base;
                m.nx = A.nx; m.ny = A.ny; m.nz = A.nz;
                const size_t cells = (size_t)A.nx * A.ny * A.nz;
                m.structure.assign(A.material, A.material + cells);
                for (size_t k = 0; k < cells; ++k)
                    if (m.structure[k] >= m.palette.size()) throw std::invalid_argument("material index outside the palette");
                if (A.fitness_file_name) m.fitness_file_name = A.fitness_file_name;
                for (int l = 0; l < A.n_layers; ++l) {
                    const LayerSlot* slot = nullptr;
                    for (const auto& sl : slots) if (std::strcmp(sl.tag, A.layer_tags[l]) == 0) slot = &sl;
                    if (!slot) throw std::invalid_argument(std::string("unsupported per-voxel layer <") + A.layer_tags[l] + ">");
                    if (slot->land_only && variant_ != 0) throw std::invalid_argument(std::string("unsupported <") + A.layer_tags[l] + "> development layer (land_water)");
                    std::vector<double>& out = m.*(slot->values);
                    m.*(slot->has) = true;
                    out.clear();
                    for (size_t k = 0; k < cells; ++k) {
                        if (m.structure[k] == 0) continue;
                        double v = A.layers[l][k];
                        if (round_like_text) { char buf[48]; std::snprintf(buf, sizeof(buf), "%.12g", v); v = std::atof(buf); }
                        out.push_back(v);
                    }
                }
                built[i] = std::move(m);
            } catch (...) {
                errors[i] = std::current_exception();
            }
        }
    };
    const int nthreads = std::max(1, std::min({n, (int)std::thread::hardware_concurrency(), 32}));
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    for (int i = 0; i < n; ++i) if (errors[i]) std::rethrow_exception(errors[i]);
    return built;
}

std::vector<RobotModel> Engine::take_robots()
{
    HIP_OK(hipSetDevice(device_id_));
    dev_->free_all();
    std::vector<RobotModel> out = std::move(robots_);
    robots_.clear();
    host_.clear();
    prepared_ = state_downloaded_ = control_downloaded_ = reduced_downloaded_ = false;
    rounds_done_ = 0;
    return out;
}

void Engine::set_tiling_allowed(bool on)
{
    if (on == tiling_allowed_) return;
    if (prepared_ && rounds_done_ > 0) throw std::logic_error("the kernel choice of a batch that has stepped cannot change");
    tiling_allowed_ = on;
    prepared_ = false;
}

void Engine::give_robots(std::vector<RobotModel>&& models)
{
    for (auto& m : models) robots_.push_back(std::move(m));
    prepared_ = state_downloaded_ = control_downloaded_ = reduced_downloaded_ = false;
}

void Engine::clear()
{
    HIP_OK(hipSetDevice(device_id_));
#ifdef VXH_PHASE_TIMING
    if (std::getenv("VXH_COL_STATS") && dev_->B.col_rows > 0) {    // lengths of the contact rows as the last broad-phase runs left them
        std::vector<int> cnt(dev_->B.col_rows);
        HIP_OK(hipMemcpy(cnt.data(), dev_->B.col_cnt, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost));
        long long hist[9] = {0}, total = 0; int mx = 0;
        for (int c : cnt) { total += c; mx = std::max(mx, c); ++hist[c == 0 ? 0 : std::min(8, 1 + (c - 1) / 8)]; }
        fprintf(stderr, "contact rows: %zu surface voxels, %.1f partners on average, longest %d; rows by length 0 | 1-8 | 9-16 | ... | 57-64:", cnt.size(), (double)total / cnt.size(), mx);
        for (long long v : hist) fprintf(stderr, " %lld", v);
        long long per_robot_max = 0, wave_max_sum = 0, waves = 0;
        for (size_t r = 0; r < robots_.size(); ++r) {
            long long sum = 0;
            const int b = dev_->surf_begin[r], n = robots_[r].nsurf;
            for (int i = 0; i < n; ++i) sum += cnt[b + i];
            per_robot_max = std::max(per_robot_max, sum);
        }
        fprintf(stderr, "\n  most partners in one robot: %lld\n", per_robot_max);
        (void)wave_max_sum; (void)waves;
    }
    if (dev_->B.prof) {
        unsigned long long h[16 * 8 + 256 * 8];
        HIP_OK(hipMemcpy(h, dev_->B.prof, sizeof(h), hipMemcpyDeviceToHost));
        if (std::getenv("VXH_PROF_TILES")) {    // the first tiles at one step of the last launch: real-time counter (10 ns ticks) at the step's boundaries
            unsigned long long t0 = ~0ull;
            for (int t = 0; t < std::min(256, dev_->B.n_tiles); ++t) if (h[128 + t * 8]) t0 = std::min(t0, h[128 + t * 8]);
            fprintf(stderr, "tile: top  halo-done  bond-done  voxel-done  C-passed | svc: barrier-resolved | next top | mv-published   (us after the first tile's top)\n");
            for (int t = 0; t < std::min(256, dev_->B.n_tiles); ++t) {
                fprintf(stderr, "tile %3d:", t);
                for (int k : {0, 1, 2, 3, 4, 5, 6, 7}) fprintf(stderr, " %7.2f", 0.01 * (double)(long long)(h[128 + t * 8 + k] - t0));
                fprintf(stderr, "\n");
            }
        }
        if (h[2112]) fprintf(stderr, "contact rows in LDS: %llu wavefront copies, %llu did not fit (mean capacity %.0f pairs); cross-check (dbg 8): %llu lanes, %llu with different bits, %llu whose LDS row differs from memory; of the differing lanes: %llu with fewer mask bits than pairs in reach, %llu with more, %llu with as many\n",
                            h[2112], h[2113], h[2114] ? (double)h[2114] * 12.0 / (double)h[2112] : 0.0, h[2115], h[2116], h[2117], h[2118], h[2119], h[2111]);
        if (h[2108]) fprintf(stderr, "broad-phase runs (resident kernel, with the copy of the rows): %llu, %.0f cycles each on average; workgroup-launches with at least one: %llu; "
                            "most cycles one workgroup spent in them in one launch (maximum over ALL launches): %llu\n", h[2108], (double)h[2109] / (double)h[2108], h[2107], h[2110]);
        if (h[2108] && h[2100]) fprintf(stderr, "   of a broad-phase run, cycles on average (first wavefront): staging the surface list %.0f | its own scan / block pairs %.0f | wait for the "
                                      "slowest wavefront %.0f | bit matrix -> rows %.0f | the rest (copy of the rows to LDS) %.0f\n", (double)h[2100] / h[2108], (double)h[2101] / h[2108], (double)h[2102] / h[2108],
                                      (double)h[2126] / h[2108], ((double)h[2109] - (double)h[2100] - (double)h[2101] - (double)h[2102] - (double)h[2126]) / h[2108]);
        if (h[2104]) fprintf(stderr, "   broad-phase cross-check (dbg 16): %llu rows built by both scans, %llu differ (%llu longer, %llu shorter than the plain scan's; longest row %llu against %llu; partners in all %llu against %llu)\n",
                             h[2104], h[2105], h[2120], h[2121], h[2122], h[2123], h[2124], h[2125]);
        if (h[2103]) fprintf(stderr, "   prologue of the resident kernel, cycle sums of the first thread over all workgroup-launches: tables + first barrier %.3e | state load, zeroing, "
                             "first control %.3e | rows_to_lds %.3e  = (its parts, incl. the calls after broad-phase runs) ordinal -> count loads + barrier %.3e | scan + allotment %.3e | copy + barrier %.3e\n",
                             (double)h[2103], (double)h[2106], (double)h[2113], (double)h[2114], (double)h[2115], (double)h[2116]);
        if (h[2142]) fprintf(stderr, "   tiled kernel, voxel phase of the first thread of every tile, cycle sums: six-direction sums %.3e | contacts %.3e | voxel update %.3e | pose granules out %.3e\n",
                             (double)h[2140], (double)h[2141], (double)h[2142], (double)h[2143]);
        if (h[2131]) fprintf(stderr, "   drag, facet pass (first thread, cycle sums): barrier behind the vertices %.3e | facet loop %.3e | barrier %.3e | per-voxel sums %.3e | barrier %.3e\n",
                             (double)h[2130], (double)h[2131], (double)h[2132], (double)h[2133], (double)h[2134]);
        static const char* names[6] = {"ctl+barA", "aux", "bond", "barB", "voxel", "barC+pub"};
        static const char* tnames[8] = {"halo-wait", "bond", "svc:poll", "barB", "latch/rebuild", "voxel", "barC+mv", "svc:reduce+horizon"};
        const bool tiled = !dev_->tile_launches.empty();
        if (tiled && h[121]) fprintf(stderr, "robot barrier: %.2f poll rounds per barrier, %.0f cycles per barrier\n", (double)h[120] / h[121], (double)h[122] / h[121]);
        for (int w = 0; w < 15; ++w) {
            double tot = 0; for (int k = 0; k < (tiled ? 8 : 6); ++k) tot += (double)h[w * 8 + k];
            if (tot == 0) continue;
            fprintf(stderr, "wave %2d:", w);
            if (tiled) for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %5.1f%%", tnames[k], 100.0 * h[w * 8 + k] / tot);
            else for (int k = 0; k < 6; ++k) fprintf(stderr, " %s %5.1f%%", names[k], 100.0 * h[w * 8 + k] / tot);
            fprintf(stderr, "  total %.3e cycles", tot);
            if (!tiled && h[w * 8 + 6] + h[w * 8 + 7]) fprintf(stderr, "  (MESH: mesh vertices / facets inside aux; else: before the step loop / after it, cycles)  %.3e  %.3e", (double)h[w * 8 + 6], (double)h[w * 8 + 7]);
            fprintf(stderr, "\n");
        }
    }
#endif
    dev_->free_all();
    robots_.clear();
    host_.clear();
    prepared_ = state_downloaded_ = control_downloaded_ = reduced_downloaded_ = false;
    rounds_done_ = 0;
}

// A captured step graph holds the kernel arguments (the whole DBatch, by value) of the moment it was captured: every option that
// changes one of them, or which kernels a round consists of, drops it; the next graph-sized call captures afresh.
void Engine::drop_graph()
{
    HIP_OK(hipSetDevice(device_id_));
    if (dev_->graph_exec) { hipGraphExecDestroy(dev_->graph_exec); dev_->graph_exec = nullptr; }
    if (dev_->graph) { hipGraphDestroy(dev_->graph); dev_->graph = nullptr; }
}

// throws what set_option would throw for this key / value in the engine's present state, and changes nothing: a handle over several
// devices applies an option to all of its engines or to none
void Engine::check_option(const std::string& key, double value) const
{
#ifdef VXH_PHASE_TIMING
    if (key == "dbg") return;                 // physics-skipping what-if switches: developer library only
#endif
    if (key == "host_results") return;
    if (key == "shape_descriptors") { if (value != 0 && value != 1) throw std::invalid_argument("shape_descriptors: 0 or 1"); return; }
    if (key == "steps_per_launch") { if (!(value >= 1 && value <= 200000)) throw std::invalid_argument("steps_per_launch out of range"); return; }
    if (key == "graph_steps") { if (!(value >= 0 && value <= 1e6)) throw std::invalid_argument("graph_steps out of range"); return; }
    if (key == "tiled" || key == "tiles_per_robot" || key == "tile_small" || key == "wide" || key == "wide_two_tiles" || key == "col_cap" || key == "fused" || key == "pair" || key == "pair_sel") {
        // which kernel steps a robot, its tiling and the size of its contact rows are part of the uploaded batch: set before the first
        // vxh_run / vxh_step, or right after vxh_reset (no step taken yet: the batch is then assembled again at the next run)
        {   // (an unchanged value is always fine: nothing would be assembled again)
            const double now = key == "wide" ? (double)wide_ : key == "wide_two_tiles" ? (double)wide_two_tiles_ : key == "col_cap" ? (double)col_cap_ :
                               key == "tiled" ? (double)tiled_ : key == "tile_small" ? (double)tile_small_ : key == "fused" ? (double)fused_ :
                               key == "pair" ? (double)pair_ : key == "pair_sel" ? (double)pair_sel_ : (double)tiles_per_robot_;
            const bool flag = key == "wide" || key == "wide_two_tiles" || key == "tile_small" || key == "fused" || key == "pair_sel";
            if (prepared_ && rounds_done_ > 0 && now != (flag ? (double)(value != 0) : (double)(int)value))
                throw std::logic_error("option " + key + " must be set before the first vxh_run/vxh_step (or right after vxh_reset)");
        }
        // (`fused` too, since round 5: the resident kernels keep the bond history in their own array and a saved image of the contact rows, so
        // a robot that changed kernels in the middle of a run would read stale history -- ADVICE round 4)
        if ((key == "wide" || key == "wide_two_tiles" || key == "fused" || key == "pair_sel") && value != 0 && value != 1) throw std::invalid_argument(key + ": 0 or 1");
        if (key == "pair" && value != 0 && value != 1 && value != 2) throw std::invalid_argument("pair: 0, 1 or 2");
#ifndef VXH_PAIR
        if (key == "pair" && value != 0) throw std::invalid_argument("pair: k_robot_pair is compiled into the developer library only (make -C evosoro_amd/csrc prof: -DVXH_PAIR)");
#endif
        if (key == "col_cap" && !(value >= 0 && value <= 1e6)) throw std::invalid_argument("col_cap out of range");
        if (key == "tiled" && value != 0 && value != 1 && value != 2) throw std::invalid_argument("tiled: 0, 1 or 2");
        if (key == "tile_small" && value != 0 && value != 1) throw std::invalid_argument("tile_small: 0 or 1");
        if (key == "tiles_per_robot" && !(value >= 0 && value <= 4096)) throw std::invalid_argument("tiles_per_robot out of range");
        return;
    }
    throw std::invalid_argument("unknown option " + key);
}

void Engine::set_option(const std::string& key, double value)
{
    check_option(key, value);
#ifdef VXH_PHASE_TIMING
    if (key == "dbg") { dbg_ = (int)value; dev_->B.dbg = dbg_; drop_graph(); return; }
#endif
    if (key == "host_results") { host_results_ = value != 0; reduced_downloaded_ = false; }
    else if (key == "shape_descriptors") shape_descriptors_ = value != 0;      // (robots added FROM NOW ON carry the surface mesh)
    else if (key == "steps_per_launch") steps_per_launch_ = (int)value;
    else if (key == "graph_steps") { graph_steps_ = (int)value; drop_graph(); }
    else {
        const double now = key == "wide" ? (double)wide_ : key == "wide_two_tiles" ? (double)wide_two_tiles_ : key == "col_cap" ? (double)col_cap_ :
                           key == "tiled" ? (double)tiled_ : key == "tile_small" ? (double)tile_small_ : key == "fused" ? (double)fused_ :
                           key == "pair" ? (double)pair_ : key == "pair_sel" ? (double)pair_sel_ : (double)tiles_per_robot_;
        const double want = (key == "wide" || key == "wide_two_tiles" || key == "tile_small" || key == "fused" || key == "pair_sel") ? (double)(value != 0) : (double)(int)value;
        if (now == want) return;              // (unchanged: the assembled batch stays)
        prepared_ = false;                    // (the batch is assembled again at the next run)
        if (key == "fused") { fused_ = value != 0; drop_graph(); }
        else if (key == "pair") pair_ = (int)value;
        else if (key == "pair_sel") pair_sel_ = value != 0;
        else if (key == "wide") wide_ = value != 0;
        else if (key == "wide_two_tiles") wide_two_tiles_ = value != 0;
        else if (key == "col_cap") col_cap_ = (int)value;
        else if (key == "tiled") tiled_ = (int)value;
        else if (key == "tile_small") tile_small_ = value != 0;
        else tiles_per_robot_ = (int)value;
    }
}

// where the host-side time of a run goes (VXH_PROF_HOST=1: one line per prepare() / advance() on stderr)
namespace {
struct HostStages {
    const bool on = std::getenv("VXH_PROF_HOST") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    std::string line;
    void mark(const char* what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        char buf[96]; std::snprintf(buf, sizeof(buf), " %s %.3f ms |", what, 1e3 * std::chrono::duration<double>(now - t).count());
        line += buf; t = now;
    }
    void print(const char* head) { if (on) std::fprintf(stderr, "%s:%s\n", head, line.c_str()); }
};
}

void Engine::prepare()
{
    HostStages hs;
    HIP_OK(hipSetDevice(device_id_));
    Device& D = *dev_;
    D.free_all();
    hs.mark("free");
    const int nr = (int)robots_.size();
    // Wide kernel (kernels_wide.hpp; robots of up to WIDE_BLOCK voxels and 1023 bonds): its bond records share their LDS region with
    // the scratch of the whole-robot passes (12 * BLOCK doubles, then the mesh vertices of a robot in a fluid); the record that stays
    // zero lies behind both.  A function of the robot alone, like every kernel choice.
    constexpr int WIDE_BLOCK = 512;
    struct WideLayout { int zidx = 0, region = 0; };
    auto wide_layout = [&](const RobotModel& M) {
        WideLayout W;
        const bool in_fluid = variant_ == 1 && M.vxa.fluid_env;
        const int scratch = 12 * WIDE_BLOCK + (in_fluid ? 3 * ((M.nvox + 63) / 64 * 64) + 3 * M.nmv : 0);
        W.zidx = std::max(M.nbond, (scratch + VXH_WIDE_REC - 1) / VXH_WIDE_REC);
        W.region = (VXH_WIDE_REC * (W.zidx + 1) + 1) & ~1;
        return W;
    };
    auto wide_listed = [&](const RobotModel& M) {
        return wide_ && M.nvox > 0 && M.nvox <= WIDE_BLOCK && M.nbond <= 1023 && M.bond_classes.size() <= 4095 && wide_layout(M).zidx <= 1023;
    };
    std::vector<DVoxClass> vtab;
    std::vector<DBondClass> btab;
    D.h_robot.assign(nr, DRobot{});
    D.vox_begin.assign(nr, 0);
    D.surf_begin.assign(nr, 0);
    int nv = 0, ns = 0;
    D.max_planned = 0;
    D.max_nvox = 0;
    for (int r = 0; r < nr; ++r) {
        D.vox_begin[r] = nv;
        D.surf_begin[r] = ns;
        nv += (robots_[r].nvox + 63) / 64 * 64;
        ns += robots_[r].vxa.self_col_enabled ? robots_[r].nsurf : 0;
        if (robots_[r].nvox > 0) D.max_planned = std::max(D.max_planned, robots_[r].planned_steps);
        D.max_nvox = std::max(D.max_nvox, robots_[r].nvox);
    }
    if (nv == 0) nv = 64;
    // the kernels address a component plane as 32-bit `plane * nv + slot` (up to 36 planes of bond slots)
    if ((long long)nv * 36 >= (1LL << 32))
        throw std::invalid_argument("batch too large for one engine (" + std::to_string(nv) + " voxel slots; limit 119 million): split the population");
    D.total_surf = ns;
    long long col_total = 0;
    // centre-of-mass traces (<TimeBetweenTraces>): one entry at most every trace_dt of simulated time after InitCmTime
    D.trace_begin.assign(nr, 0); D.trace_cap.assign(nr, 0); D.total_trace = 0;
    for (int r = 0; r < nr; ++r) {
        const RobotModel& M = robots_[r];
        D.trace_begin[r] = D.total_trace;
        if (variant_ == 0 && M.vxa.time_between_traces > 0 && M.nvox > 0) {
            const double span = std::max(0.0, (double)M.planned_steps * M.dt - M.vxa.init_cm_time);
            const double n = span / M.vxa.time_between_traces + 8;
            if (!(n < 4e6)) throw std::invalid_argument("TimeBetweenTraces asks for more than 4 million trace points");
            D.trace_cap[r] = (int)n;
            D.total_trace += D.trace_cap[r];
        }
    }

    std::vector<int> wave_robot(nv / 64, -1), nbr((size_t)6 * nv, -1), surf(std::max(ns, 1), 0), surf_code(std::max(ns, 1), 0), surf_ord(nv, -1);
    std::vector<unsigned long long> excl;
    std::vector<unsigned short> vclass(nv, 0);
    std::vector<short> bclass((size_t)3 * nv, -1);
    // bond schedules of the resident kernel: [3][block] per robot, block = the workgroup size of the robot's size class
    auto resident_block = [](int n) { return n <= 256 ? 256 : (n <= 512 ? 512 : (n <= 768 ? 768 : 1024)); };
    // Pair kernel (kernels_pair.hpp): 512 threads, two voxels and up to two bonds per axis per lane -- the large robots without a surface
    // mesh that the wide kernel does not take.  A function of the robot alone (and of the engine's options), like every kernel choice.
    auto uses_pair = [&](const RobotModel& M) {
        return pair_ > 0 && M.nmv == 0 && M.nvox > (pair_ == 2 ? 512 : 768) && M.nvox <= 1024 && M.bond_classes.size() <= 4095 && !wide_listed(M);
    };
    std::vector<int> sched_off(nr + 1, 0);
    for (int r = 0; r < nr; ++r)
        sched_off[r + 1] = sched_off[r] + ((robots_[r].nvox > 0 && robots_[r].nvox <= 1024) ? (uses_pair(robots_[r]) ? 6 * VXH_PAIR_T : 3 * resident_block(robots_[r].nvox)) : 0);
    std::vector<int> bsched(std::max(sched_off[nr], 1), -1);
    std::vector<int> wgather((size_t)2 * nv, 0);
    std::vector<float> amp_damp(nv, 1.f);
    std::vector<double> act_sb(nv, 0.0), act_cb(nv, 1.0);
    bool any_dev = false;
    for (int r = 0; r < nr; ++r) any_dev = any_dev || robots_[r].development;
    std::vector<float> dev(any_dev ? (size_t)7 * nv : 7, 0.f);
    std::vector<double> px(nv, 0), py(nv, 0), pz(nv, 0), sc(nv, 0), qw(nv, 1.0);
    std::vector<unsigned char> small((size_t)3 * nv, 1);
    // land_water robots: deformable surface mesh (fluid drag, RobotVolume tags)
    int total_mv = 0;
    std::vector<int> mv_begin(nr, 0);
    bool any_mesh = false;
    for (int r = 0; r < nr; ++r) { mv_begin[r] = total_mv; total_mv += robots_[r].nmv; any_mesh = any_mesh || robots_[r].nmv > 0; }
    std::vector<int> vert_pack((size_t)3 * std::max(total_mv, 1), 0);
    // robots in a fluid: the same mesh with global indices for the streaming kernels (robots that do not fit the resident one)
    bool any_fluid = false;
    for (int r = 0; r < nr; ++r) any_fluid = any_fluid || (variant_ == 1 && robots_[r].vxa.fluid_env && robots_[r].nmv > 0);
    std::vector<int> vert_vox(any_fluid ? (size_t)8 * std::max(total_mv, 1) : 8, -1), vert_robot(any_fluid ? std::max(total_mv, 1) : 1, 0);
    std::vector<double> vert_v0((size_t)3 * std::max(total_mv, 1), 0.0);
    int total_facet = 0;
    std::vector<int> facet_begin(nr, 0);
    for (int r = 0; r < nr; ++r) { facet_begin[r] = total_facet; total_facet += (int)robots_[r].facet_vox.size(); }
    std::vector<int> facet_vox(std::max(total_facet, 1), 0), facet_vert((size_t)3 * std::max(total_facet, 1), 0), facet_first(any_mesh ? nv : 1, 0);
    std::vector<unsigned char> facet_count(any_mesh ? nv : 1, 0);
    std::vector<int> facet_robot(any_fluid ? std::max(total_facet, 1) : 1, 0);
    std::vector<DRobotState> rstate(nr);

    // offsets of the per-robot pieces of the shared tables, then the robots are assembled on the host cores (disjoint ranges)
    std::vector<int> vtab_off(nr + 1, 0), btab_off(nr + 1, 0), wl_off(nr + 1, 0);
    std::vector<long long> excl_off(nr + 1, 0), col_off(nr + 1, 0);
    std::vector<int> col_cap(nr, 0);
    std::vector<int> img_idx(nr, -1);          // slot of the robot's saved contact-row image (DBatch::rimg_*), colliding robots of up to 1024 voxels
    { int n_img = 0; for (int r = 0; r < nr; ++r) if (robots_[r].vxa.self_col_enabled && robots_[r].nvox > 0 && robots_[r].nvox <= 1024) img_idx[r] = n_img++; }
    // contact rows: as long as the physics makes them -- a surface voxel can list every other one (CreateColBond, VX_Sim.cpp:753-769,
    // has no cap) -- unless the option col_cap bounds them; 12-16 bytes per entry, untouched beyond the partners a row really has.
    // That is nsurf^2 entries of address space per robot (75 MB for a 20^3 lattice, 1.5 GB for 512 robots of 10^3: nothing on this
    // device), but it grows with the FOURTH power of a lattice's edge: when the uncapped rows of a batch would take more than a third of
    // the device's free memory (a lattice of 100^3 and beyond) every robot's rows are cut to an equal share of that third, never
    // below 64, and the engine says so -- a row that then overflows gives its robot VXH_ROBOT_COL_OVERFLOW, as with a col_cap of the
    // user's.  (Round 3 allocated nsurf^2 whatever the size and ran out of memory in this function for lattices it used to step.)
    long long row_limit = 0x7fffffff;
    if (col_cap_ <= 0) {
        double want = 0, rows = 0;
        for (int r = 0; r < nr; ++r) if (robots_[r].vxa.self_col_enabled) { want += 16.0 * (double)robots_[r].nsurf * (double)std::max(1, robots_[r].nsurf - 1); rows += robots_[r].nsurf; }
        // (the budget is a third of the device's TOTAL memory, not of what happens to be free: how long a row can grow -- and with it whether a
        // robot ends FINISHED or COL_OVERFLOW -- must not depend on what other processes hold on the GPU at this moment; if the rows then do
        // not fit what IS free, the allocation fails with an error instead of changing outcomes silently.  ADVICE round 4)
        size_t free_b = 0, total_b = 0;
        if (want > 0 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && want > (double)total_b / 3.0) {
            row_limit = std::max<long long>(64, (long long)((double)total_b / 3.0 / 16.0 / std::max(1.0, rows)));
            std::fprintf(stderr, "vxhip: the contact rows of this batch would take %.1f GB uncapped (device memory: %.1f GB); rows are cut to %lld partners "
                                 "(option col_cap sets the length; a robot whose row overflows ends with status COL_OVERFLOW)\n", want / 1e9, (double)total_b / 1e9, row_limit);
        }
    }
    for (int r = 0; r < nr; ++r) {
        const RobotModel& M = robots_[r];
        if (M.vox_classes.size() > 32767 || M.bond_classes.size() > 32767) throw std::runtime_error("too many distinct voxel/bond classes in one robot");
        vtab_off[r + 1] = vtab_off[r] + (int)M.vox_classes.size();
        btab_off[r + 1] = btab_off[r] + (int)M.bond_classes.size();
        excl_off[r + 1] = excl_off[r] + (M.vxa.self_col_enabled ? (long long)M.nsurf * ((M.nsurf + 63) / 64) : 0);
        col_cap[r] = M.vxa.self_col_enabled ? std::max(1, col_cap_ > 0 ? std::min(col_cap_, M.nsurf - 1) : (int)std::min<long long>(row_limit, M.nsurf - 1)) : 0;
        col_off[r + 1] = col_off[r] + (long long)col_cap[r] * (M.vxa.self_col_enabled ? M.nsurf : 0);
        wl_off[r + 1] = wl_off[r] + (wide_listed(M) ? M.nbond : 0);
    }
    std::vector<int> wlist(std::max(wl_off[nr], 1), -1);
    col_total = col_off[nr];
    vtab.resize(vtab_off[nr]);
    btab.resize(btab_off[nr]);
    excl.assign((size_t)excl_off[nr], 0ull);
    auto assemble = [&](int r) {
        const RobotModel& M = robots_[r];
        const VxaModel& X = M.vxa;
        const int base = D.vox_begin[r];
        const int vtab_begin = vtab_off[r], btab_begin = btab_off[r];
        for (size_t i = 0; i < M.vox_classes.size(); ++i) {
            const VoxClass& c = M.vox_classes[i];
            DVoxClass d;
            std::memset(&d, 0, sizeof(d));
            d.mass = c.mass; d.mass_inv = c.mass_inv; d.inertia_inv = c.inertia_inv; d.c_lin = c.c_lin; d.c_ang = c.c_ang;
            d.E = c.E; d.k_floor = c.k_floor; d.u_static = c.u_static; d.u_dynamic = c.u_dynamic; d.cte = c.cte;
            d.nom_size = c.nom_size; d.mat = c.mat;
            d.prenatal_k = ((double)(float)c.nom_size / c.nom_size) - 1;
            vtab[vtab_begin + i] = d;
        }
        for (size_t i = 0; i < M.bond_classes.size(); ++i) {
            const BondClass& c = M.bond_classes[i];
            DBondClass d;
            std::memset(&d, 0, sizeof(d));
            const double zi = (0.5 * X.bond_damping_z) * (1.0 / M.dt), zh = 0.5 * zi;   // BondDampingZ/2 (moments: /4) over dt
            d.L100 = 100.0 * c.L; d.a2 = c.a2; d.b1 = c.b1; d.b2 = c.b2; d.b3 = c.b3;
            d.kf_L = c.stress_k * c.area_sum / 2 / c.L; d.strain_a1_L = c.strain_a1 / c.L; d.strain_a2_L = c.strain_a2 / c.L;
            d.dA1 = c.sq_a1m1 * zi; d.dB1 = c.sq_b1m1 * zi; d.dF1 = c.sq_b2fm1 * zi;
            d.dA2 = c.sq_a1m2 * zi; d.dB2 = c.sq_b1m2 * zi; d.dF2 = c.sq_b2fm2 * zi;
            d.dT1 = c.sq_a2i1 * zh; d.dG1 = c.sq_b2fm1 * zh; d.dH1 = c.sq_b3i1 * zh;
            d.dT2 = c.sq_a2i2 * zh; d.dG2 = c.sq_b2fm2 * zh; d.dH2 = c.sq_b3i2 * zh;
            d.homogeneous = c.homogeneous;
            btab[btab_begin + i] = d;
        }
        for (int w = base / 64; w < (base + (M.nvox + 63) / 64 * 64) / 64; ++w) wave_robot[w] = r;
        for (int v = 0; v < M.nvox; ++v) {
            const int g = base + v;
            vclass[g] = (unsigned short)M.vox_class[v];
            { const double b = (double)(2 * 3.1415926f) * (double)M.phase_offset[v]; act_sb[g] = std::sin(b); act_cb[g] = std::cos(b); }
            amp_damp[g] = M.temp_amp_damp[v];
            if (M.development) {
                const float vals[7] = {M.initial_voxel_size[v], M.final_voxel_size[v], M.start_growth_time[v], M.growth_time[v],
                                       M.phase_offset[v], M.final_phase_offset[v], M.final_temp_amp_damp[v]};
                for (int k = 0; k < 7; ++k) dev[(size_t)k * nv + g] = vals[k];
            }
            px[g] = M.nom_pos[3 * v]; py[g] = M.nom_pos[3 * v + 1]; pz[g] = M.nom_pos[3 * v + 2];
            sc[g] = M.vox_classes[M.vox_class[v]].nom_size;
            for (int d = 0; d < 6; ++d) { int o = M.nbr[(size_t)v * 6 + d]; nbr[(size_t)d * nv + g] = o < 0 ? -1 : base + o; }
            for (int a = 0; a < 3; ++a) { int c = M.bond_class[(size_t)v * 3 + a]; bclass[(size_t)a * nv + g] = c < 0 ? (short)-1 : (short)c; }
        }
        DRobot& R = D.h_robot[r];
        R.sched_begin = sched_off[r];
        R.img_index = img_idx[r];
        if (M.nvox > 0 && M.nvox <= 1024 && M.bond_classes.size() <= 4095) {
            // The resident kernel's bond schedule.  X and Z: bond j of the axis' compacted list to thread j.  Y: the kernel variants
            // with two accumulator tiles (up to 768 threads) evaluate X and Y without a barrier between them -- an accumulator entry
            // gets at most one X and one Y contribution per tile, both added to zero with LDS atomics, and a + b == b + a, so the sums
            // are those of the barrier-separated rounds bit for bit -- and a wavefront's bond is a latency-bound chain (a lone
            // wavefront issues a dependent FP64 instruction every ~7 cycles, three that share a SIMD one every ~4): the 64-bond
            // chunks of Y are dealt to the wavefronts that X leaves idle, least-loaded SIMD first (wavefront w runs on SIMD w mod 4),
            // so that X + Y occupy every SIMD with three to four chunk executions spread over its three wavefronts instead of
            // two rounds of two.  The 1024-thread variant (one tile, two barrier-separated sub-steps per round) keeps Y in list order.
            const int block = resident_block(M.nvox), nw = block / 64;
            int* sx = &bsched[sched_off[r]], *sy = sx + block, *sz = sy + block;
            std::vector<int> list[3];
            for (int a = 0; a < 3; ++a)
                for (int v = 0; v < M.nvox; ++v) {
                    const int c = M.bond_class[(size_t)v * 3 + a];
                    if (c >= 0) list[a].push_back((int)((unsigned)v | ((unsigned)M.nbr[(size_t)v * 6 + 2 * a] << 10) | ((unsigned)c << 20)));
                }
            if (uses_pair(M)) {
                // k_robot_pair: [3 axes][2 slots][512 threads].  X and Y are evaluated in ONE barrier-free stretch (four slots per lane), Z
                // in a second one (two): the 64-bond chunks of X, then of Y, go round the eight wavefronts (wavefront w runs on SIMD w mod 4),
                // so the four SIMDs carry the same number of chunk executions to within one; Z likewise on its own.
                int* const sch = &bsched[sched_off[r]];
                int used_xy[VXH_PAIR_NW][2] = {}, next_wave = 0;      // slots taken per wavefront: [X, Y]
                for (int a = 0; a < 2; ++a) {
                    const int nch = ((int)list[a].size() + 63) / 64;
                    for (int c = 0; c < nch; ++c) {
                        int w = next_wave;
                        while (used_xy[w][a] >= 2) w = (w + 1) % VXH_PAIR_NW;          // (at most 16 chunks per axis: a free slot exists)
                        next_wave = (w + 1) % VXH_PAIR_NW;
                        int* dst = sch + (2 * a + used_xy[w][a]++) * VXH_PAIR_T + 64 * w;
                        for (int k = 0; k < 64 && (size_t)(64 * c + k) < list[a].size(); ++k) dst[k] = list[a][(size_t)64 * c + k];
                    }
                }
                const int nchz = ((int)list[2].size() + 63) / 64;
                for (int c = 0; c < nchz; ++c) {
                    int* dst = sch + (4 + c / VXH_PAIR_NW) * VXH_PAIR_T + 64 * (c % VXH_PAIR_NW);
                    for (int k = 0; k < 64 && (size_t)(64 * c + k) < list[2].size(); ++k) dst[k] = list[2][(size_t)64 * c + k];
                }
            } else {
            for (size_t j = 0; j < list[0].size(); ++j) sx[j] = list[0][j];
            for (size_t j = 0; j < list[2].size(); ++j) sz[j] = list[2][j];
            if (block == 1024) { for (size_t j = 0; j < list[1].size(); ++j) sy[j] = list[1][j]; }
            else {
                const int nchx = ((int)list[0].size() + 63) / 64, nchy = ((int)list[1].size() + 63) / 64;
                std::vector<int> simd_load(4, 0), wave_load(nw, 0), has_y(nw, 0);
                for (int w = 0; w < nchx; ++w) { ++simd_load[w % 4]; ++wave_load[w]; }
                for (int c = 0; c < nchy; ++c) {
                    int best = -1;
                    for (int w = 0; w < nw; ++w) {
                        if (has_y[w]) continue;
                        if (best < 0 || simd_load[w % 4] < simd_load[best % 4] ||
                            (simd_load[w % 4] == simd_load[best % 4] && wave_load[w] < wave_load[best])) best = w;
                    }
                    has_y[best] = 1; ++simd_load[best % 4]; ++wave_load[best];
                    for (int k = 0; k < 64 && (size_t)(64 * c + k) < list[1].size(); ++k) sy[64 * best + k] = list[1][(size_t)64 * c + k];
                }
            }
            }
        }
        R.wl_begin = wl_off[r]; R.wnbond = 0; R.wzidx = 0; R.wregion = 0;
        if (wide_listed(M)) {
            // wide kernel: all bonds in one list, axis after axis; a bond's place in it is the number of its record, and every voxel
            // learns the records of its six directions (+A: the bond it is the negative end of; -A: that of its -A neighbour)
            const WideLayout W = wide_layout(M);
            std::vector<int> slot_of((size_t)M.nvox * 3, -1);
            int t = 0;
            for (int a = 0; a < 3; ++a)
                for (int v = 0; v < M.nvox; ++v) {
                    const int c = M.bond_class[(size_t)v * 3 + a];
                    if (c < 0) continue;
                    slot_of[(size_t)v * 3 + a] = t;
                    wlist[wl_off[r] + t++] = (int)((unsigned)v | ((unsigned)M.nbr[(size_t)v * 6 + 2 * a] << 9) | ((unsigned)a << 18) | ((unsigned)c << 20));
                }
            for (int v = 0; v < M.nvox; ++v) {
                int idx[6];
                for (int a = 0; a < 3; ++a) {
                    idx[2 * a] = slot_of[(size_t)v * 3 + a];
                    const int n = M.nbr[(size_t)v * 6 + 2 * a + 1];
                    idx[2 * a + 1] = n >= 0 ? slot_of[(size_t)n * 3 + a] : -1;
                }
                for (int d = 0; d < 6; ++d) if (idx[d] < 0) idx[d] = W.zidx;
                wgather[base + v] = idx[0] | (idx[1] << 10) | (idx[2] << 20);
                wgather[(size_t)nv + base + v] = idx[3] | (idx[4] << 10) | (idx[5] << 20);
            }
            R.wnbond = M.nbond; R.wzidx = W.zidx; R.wregion = W.region;
        }
        if (X.self_col_enabled)
            for (int i = 0; i < M.nsurf; ++i) {
                surf[D.surf_begin[r] + i] = base + M.surf[i]; surf_ord[base + M.surf[i]] = i;
                surf_code[D.surf_begin[r] + i] = (int)((unsigned)M.surf[i] | ((unsigned)M.vox_class[M.surf[i]] << 10));     // (robots the resident kernels take: < 1024 voxels)
            }
        // CalcNearby exclusion lists as bit rows over surface ordinals
        const long long excl_begin = excl_off[r];
        int wpr = 0;
        if (X.self_col_enabled) {
            wpr = (M.nsurf + 63) / 64;
            std::vector<int> ord(M.nvox, -1);
            for (int i = 0; i < M.nsurf; ++i) ord[M.surf[i]] = i;
            for (int i = 0; i < M.nsurf; ++i) {
                const int vi = M.surf[i];
                for (int k = M.near_off[vi]; k < M.near_off[vi + 1]; ++k) {
                    const int j = ord[M.near_idx[k]];
                    if (j >= 0) excl[(size_t)excl_begin + (size_t)i * wpr + (j >> 6)] |= 1ull << (j & 63);
                }
            }
        }
        R.col_begin = col_off[r]; R.col_cap = col_cap[r];
        R.vox_begin = base; R.nvox = M.nvox; R.surf_begin = D.surf_begin[r]; R.nsurf = X.self_col_enabled ? M.nsurf : 0;
        R.flags = (X.self_col_enabled ? RF_SELF_COL : 0) | (X.grav_enabled ? RF_GRAV : 0) | (X.floor_enabled ? RF_FLOOR : 0) |
                  (X.temp_enabled ? RF_TEMP : 0) | ((X.sticky_floor && variant_ == 0) ? RF_STICKY : 0) |
                  ((variant_ == 1 && X.fluid_env) ? RF_FLUID : 0) | (variant_ == 1 ? RF_LW : 0) |
                  ((X.col_system == 2 || X.col_system == 3) ? RF_HORIZON_COL : 0) |
                  (M.development ? RF_DEV : 0) | ((X.has_initial_voxel_size || X.has_final_voxel_size) ? RF_DEV_SIZE : 0) |
                  (X.has_final_voxel_size ? RF_DEV_FSIZE : 0) | (X.has_final_phase_offset ? RF_DEV_FPHASE : 0) |
                  (X.has_final_temp_amp_damp ? RF_DEV_FTAD : 0);
        R.stop_type = X.stop_type; R.excl_wpr = wpr; R.excl_begin = excl_begin;
        R.vert_begin = mv_begin[r]; R.nmv = M.nmv;
        R.facet_begin = facet_begin[r]; R.nfacet = (int)M.facet_vox.size();
        for (size_t f = 0; f < M.facet_vox.size(); ++f) {
            facet_vox[facet_begin[r] + f] = M.facet_vox[f];
            if (any_fluid) facet_robot[facet_begin[r] + f] = r;
            for (int k = 0; k < 3; ++k) facet_vert[(size_t)k * std::max(total_facet, 1) + facet_begin[r] + f] = M.facet_vert[f * 3 + k];
        }
        R.vtab_begin = vtab_begin; R.n_vclass = (int)M.vox_classes.size(); R.btab_begin = btab_begin; R.n_bclass = (int)M.bond_classes.size();
        if (M.nmv > 0) {
            const size_t tm = (size_t)std::max(total_mv, 1);
            for (int i = 0; i < M.nmv; ++i) {
                unsigned w[3] = {0, 0, 0};
                for (int q = 0; q < 8; ++q) {
                    const int c = M.vert_comp[(size_t)i * 8 + q];
                    if (c < 0 || M.nvox > 1024) continue;       // (only the fused path, robots <= 1024 voxels, reads it)
                    const unsigned corner = (unsigned)c & 7u, l = (unsigned)c >> 3;
                    w[corner / 3] |= l << (10 * (corner % 3));
                    w[2] |= 1u << (20 + corner);
                }
                for (int k = 0; k < 3; ++k) vert_pack[k * tm + mv_begin[r] + i] = (int)w[k];
                if (any_fluid) {
                    vert_robot[mv_begin[r] + i] = r;
                    for (int q = 0; q < 8; ++q) {
                        const int c = M.vert_comp[(size_t)i * 8 + q];
                        if (c >= 0) vert_vox[(size_t)(c & 7) * tm + mv_begin[r] + i] = base + (c >> 3);
                    }
                }
                for (int k = 0; k < 3; ++k) vert_v0[k * tm + mv_begin[r] + i] = M.vert_v0[(size_t)i * 3 + k];
            }
            for (int v = 0; v < M.nvox; ++v) {
                facet_first[base + v] = M.facet_first[v]; facet_count[base + v] = M.facet_count[v];
            }
        }
        R.dt = M.dt; R.lat = X.lattice_dim; R.bond_z_half = 0.5 * X.bond_damping_z; R.slow_z = X.slow_damping_z; R.col_z = X.col_damping_z;
        R.grav_acc = X.grav_acc; R.init_cm_time = X.init_cm_time; R.stop_value = X.stop_value;
        R.afterlife = variant_ == 0 ? X.afterlife_time : 0.0;
        R.temp_period_d = X.temp_period; R.min_temp_fact = X.min_temp_fact; R.growth_amplitude = X.growth_amplitude;
        R.col_horizon = X.collision_horizon;
        { double fd = X.collision_horizon * 1.5 * X.lattice_dim; R.filter_dist2 = fd * fd; }
        R.drag_coef = X.aggregate_drag_coef;
        R.midlife_freeze_time = X.midlife_freeze_time;
        R.trace_dt = D.trace_cap[r] > 0 ? X.time_between_traces : 0.0; R.trace_begin = D.trace_begin[r]; R.trace_cap = D.trace_cap[r];
        R.temp_amplitude = (float)X.temp_amplitude; R.temp_period = (float)X.temp_period;
        DRobotState& S = rstate[r];
        std::memset(&S, 0, sizeof(S));
        S.max_disp = (double)FLT_MAX;      // ClearAll, VX_Sim.cpp:369: forces a collision-list build on the first step
        S.act_time = -1.0;
        S.status = M.nvox == 0 ? 3 : 0;
    };
    {
        std::atomic<int> next{0};
        auto worker = [&]() { for (int r = next.fetch_add(1); r < nr; r = next.fetch_add(1)) assemble(r); };
        const int nthreads = std::max(1, std::min({nr / 8, (int)std::thread::hardware_concurrency(), 32}));
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; ++t) pool.emplace_back(worker);
        worker();
        for (auto& t : pool) t.join();
    }
    hs.mark("host assembly");
    DBatch& B = D.B;
    B.n_robots = nr; B.nv = nv; B.dbg = dbg_;
    B.robot = D.upload(D.h_robot);
    B.rstate = D.upload(rstate);
    {   // ... and their mirror in pinned host memory, which the resident kernels write as well (a call that runs one launch group and
        // nothing else needs no copy back: ~10-15 us of a 0.7 ms call); start values for the robots no kernel touches
        if (D.h_rstate_cap < (size_t)nr) {
            if (D.h_rstate) HIP_OK(hipHostFree(D.h_rstate));
            D.h_rstate = nullptr; D.h_rstate_cap = 0;
            HIP_OK(hipHostMalloc((void**)&D.h_rstate, sizeof(DRobotState) * std::max(nr, 1), hipHostMallocDefault));
            D.h_rstate_cap = (size_t)std::max(nr, 1);
        }
        std::memcpy(D.h_rstate, rstate.data(), sizeof(DRobotState) * nr);
        void* dp = nullptr;
        B.rstate_mirror = hipHostGetDevicePointer(&dp, D.h_rstate, 0) == hipSuccess ? (DRobotState*)dp : nullptr;
    }
    B.wave_robot = D.upload(wave_robot);
    B.vclass_tab = D.upload(vtab);
    B.bclass_tab = D.upload(btab);
    B.vclass = D.upload(vclass);
    B.bclass = D.upload(bclass);
    B.nbr = D.upload(nbr);
    B.bsched = D.upload(bsched);
    B.wlist = D.upload(wlist);
    B.wgather = D.upload(wgather);
    B.act_sb = D.upload(act_sb);
    B.act_cb = D.upload(act_cb);
    B.amp_damp = D.upload(amp_damp);
    B.dev = D.upload(dev);
    {   // vs[18][nv]: nominal position + scale in both buffers, identity quaternions, zero momenta
        double* vs = D.alloc_zero<double>((size_t)18 * nv);
        const std::vector<double>* planes[5] = {&px, &py, &pz, &sc, &qw};
        for (int k = 0; k < 5; ++k)
            HIP_OK(hipMemcpyAsync(vs + (size_t)(k < 4 ? k : 8) * nv, planes[k]->data(), sizeof(double) * nv, hipMemcpyHostToDevice, D.stream));
        HIP_OK(hipMemcpyAsync(vs + (size_t)4 * nv, vs, sizeof(double) * 4 * nv, hipMemcpyDeviceToDevice, D.stream));
        B.vs = vs;
    }
    B.hist = D.alloc_zero<double>((size_t)6 * 3 * nv);
    B.hist_aos = D.alloc_zero<double>((size_t)6 * 3 * nv);
    B.small_angle = D.upload(small);
    B.bout = D.alloc_zero<double>((size_t)12 * 3 * nv);
    B.surf = D.upload(surf);
    B.surf_code = D.upload(surf_code);
    B.surf_ord = D.upload(surf_ord);
    B.excl = D.upload(excl);
    B.total_mv = std::max(total_mv, 1);
    B.vert_pack = D.upload(vert_pack);
    B.vert_v0 = D.upload(vert_v0);
    B.total_facet = std::max(total_facet, 1);
    B.facet_vox = D.upload(facet_vox);
    B.facet_vert = D.upload(facet_vert);
    B.facet_first = D.upload(facet_first);
    B.facet_count = D.upload(facet_count);
    B.strain = D.alloc_zero<double>(any_mesh ? (size_t)6 * nv : 1);
    B.n_mv = total_mv; B.n_facet = total_facet;
    B.vert_vox = D.upload(vert_vox);
    B.vert_robot = D.upload(vert_robot);
    B.facet_robot = D.upload(facet_robot);
    B.mesh_pos = D.alloc_zero<double>(any_fluid ? (size_t)3 * std::max(total_mv, 1) : 1);
    B.fdrag = D.alloc_zero<double>(any_fluid ? (size_t)3 * std::max(total_facet, 1) : 1);
    D.any_fluid = any_fluid;
    B.trace = D.alloc_zero<double>((size_t)std::max(D.total_trace, 1) * 4);
    B.col_rows = std::max(ns, 1);
    B.col_cnt = D.alloc_zero<int>(std::max(ns, 1));
    {   // saved LDS images of the contact rows (resident kernel): one slot per colliding robot of up to 1024 voxels
        int n_img = 0;
        for (int r = 0; r < nr; ++r) if (img_idx[r] >= 0) ++n_img;
        B.rimg_code = D.alloc_raw<int>((size_t)std::max(n_img, 1) * VXH_RIMG_CAP);
        B.rimg_a1 = D.alloc_raw<double>((size_t)std::max(n_img, 1) * VXH_RIMG_CAP);
        B.rimg_seg = D.alloc_zero<int>((size_t)std::max(n_img, 1) * 64);
        B.rimg_rowd = D.alloc_zero<int>((size_t)nv);
    }
    B.col_partner = D.alloc_raw<int>((size_t)std::max<long long>(col_off[nr], 1));     // (entries beyond col_cnt are never read)
    B.col_a1 = D.alloc_raw<double>((size_t)std::max<long long>(col_off[nr], 1));
    hs.mark("allocations + uploads");
    // Which kernel steps which robot.  Resident kernel (kernels_fused.hpp), one workgroup per robot: its variant is a function
    // of the robot alone (size, fluid, LDS need of its own tables), never of the batch.
    struct FusedVariant { int block = 0, nacc = 0, fluid = 0, tabg = 0, wide = 0, two_tiles = 0, pair = 0; size_t lds = 0; };
    auto fused_variant = [&](const RobotModel& M) {
        FusedVariant fv;
        const size_t lds_max = 160 * 1024 - VXH_FUSED_STATIC_LDS;
        const int n = M.nvox;
        if (n == 0 || n > 1024 || M.bond_classes.size() > 4095) return fv;   // (12 class bits in a bond entry; a robot of 1024 voxels has at most 3072 bonds)
        if (wide_listed(M)) {
            // the wide kernel when its LDS layout fits (mirrors the top of k_robot_wide): pose tile, record region, class tables,
            // the strain tile of a land_water robot, the mask + pool of the contact rows
            const WideLayout W = wide_layout(M);
            const int mesh = M.nmv > 0 ? 1 : 0;
            auto wneed = [&](bool tables_in_lds) {
                return (size_t)(8 * WIDE_BLOCK + W.region) * 8 +
                       (tables_in_lds ? M.bond_classes.size() * sizeof(DBondClass) + M.vox_classes.size() * sizeof(DVoxClass) : 0) +
                       (mesh ? (size_t)48 * WIDE_BLOCK : 0) + (M.vxa.self_col_enabled ? (size_t)8 * WIDE_BLOCK : 0);
            };
            const int wtabg = wneed(true) > lds_max ? 1 : 0;
            if (wneed(!wtabg) <= lds_max) {
                fv.block = WIDE_BLOCK; fv.nacc = 0; fv.fluid = mesh; fv.tabg = wtabg; fv.wide = 1; fv.lds = wneed(!wtabg);
                // a second pose tile (64 B per thread) where the layout leaves room for it and, for a colliding robot, for at least
                // 8 KB of contact rows: the step then has two workgroup barriers instead of three (kernels_wide.hpp)
                const size_t tile = (size_t)64 * WIDE_BLOCK, rows_min = M.vxa.self_col_enabled ? (size_t)8 * 1024 : 0;
                if (wide_two_tiles_ && fv.lds + tile + rows_min <= lds_max) fv.two_tiles = 1;
                const size_t room = lds_max - (fv.two_tiles ? tile : 0);
                if (M.vxa.self_col_enabled && room > fv.lds) fv.lds += std::min<size_t>(room - fv.lds, (size_t)24 * 1024);
                fv.lds &= ~(size_t)7;
                if (fv.two_tiles) fv.lds += tile;
                return fv;
            }
        }
        if (uses_pair(M)) {
            // pair kernel (mirrors the top of k_robot_pair): pose tile, class tables, one accumulator tile; a colliding robot gets what is
            // left of the CU's LDS (one workgroup per CU anyway): the mask, the contact-row pool, and -- the whole stretch from the
            // accumulators on -- the scratch of the broad-phase bit matrix
            const size_t tabs = (M.bond_classes.size() * sizeof(DBondClass) + M.vox_classes.size() * sizeof(DVoxClass) + 15) & ~(size_t)15;
            const size_t fixed = (size_t)(8 + 6) * VXH_PAIR_NV * 8;
            const int ptabg = fixed + tabs + (M.vxa.self_col_enabled ? (size_t)(8 + 12) * 1024 : 0) > lds_max ? 1 : 0;
            fv.block = VXH_PAIR_T; fv.nacc = 1; fv.fluid = 0; fv.tabg = ptabg; fv.pair = pair_sel_ ? 2 : 1;
            fv.lds = M.vxa.self_col_enabled ? lds_max : fixed + (ptabg ? 0 : tabs);
            fv.lds &= ~(size_t)7;
            return fv;
        }
        const int block = n <= 256 ? 256 : (n <= 512 ? 512 : (n <= 768 ? 768 : 1024));
        const int fluid = M.nmv > 0 ? 1 : 0;      // (the MESH variants: every land_water robot carries the surface mesh)
        const bool in_fluid = variant_ == 1 && M.vxa.fluid_env;
        // LDS need: pose tile, accumulator tiles, actuation phases; class tables; the mesh vertices of a robot in a
        // fluid; with two accumulator tiles the MESH variants also hold the strain tile (with one it stays in HBM)
        // (mirrors the layout at the top of k_robot_steps; SLIM = the 768-thread MESH variant, phases and strains in HBM)
        const bool slim = fluid && block == 768;
        auto need = [&](int nacc, bool tables_in_lds) {
            return (size_t)(8 + 6 * nacc + (slim ? 0 : 2)) * block * 8 +
                   (tables_in_lds ? M.bond_classes.size() * sizeof(DBondClass) + M.vox_classes.size() * sizeof(DVoxClass) : 0) +
                   (in_fluid ? (size_t)24 * M.nmv : 0) + ((fluid && nacc == 2 && !slim) ? (size_t)48 * block : 0);
        };
        // accumulator tiles: two up to 768 voxels, one for 1024; class tables in LDS unless they do not fit (per-voxel
        // evolved stiffness makes nearly every bond a class of its own), then the TABG variant reads them from HBM
        const int nacc = block == 1024 ? 1 : 2;
        const int tabg = need(nacc, true) > lds_max ? 1 : 0;
        if (need(nacc, !tabg) > lds_max) return fv;                   // (e.g. a mesh with thousands of vertices)
        fv.block = block; fv.nacc = nacc; fv.fluid = fluid; fv.tabg = tabg; fv.lds = need(nacc, !tabg);
        // colliding robots: what the workgroup can spare without lowering the number of workgroups a CU holds (two of the 256-thread
        // variant, by its registers; one of the others) takes the contact rows, 12 bytes per listed pair
        if (M.vxa.self_col_enabled) {
            const size_t room = (block == 256 ? (size_t)80 * 1024 - VXH_FUSED_STATIC_LDS : lds_max);
            if (room > fv.lds) fv.lds += std::min<size_t>(room - fv.lds, (size_t)24 * 1024);
        }
        fv.lds &= ~(size_t)7;
        return fv;
    };
    // Tiled kernel (kernels_tiled.hpp), several workgroups per robot: for robots the resident kernel cannot take, and for
    // populations too small to give every CU a robot.  The number of tiles depends on the population (the results do not:
    // the tiled kernel's arithmetic per bond and per voxel, and its summation orders, are those of the resident kernel).
    D.robot_tiled.assign(nr, 0);
    D.robot_tiles.assign(nr, 0);
    D.tile_launches.clear();
    std::vector<DTile> h_tiles;
    std::vector<int> tile_vox, tile_bond, tile_bcls, tile_bslot, xslot(nv, 0);
    std::vector<int> tmv_slot[8], tf_owner, tf_vert[3], tile_ffirst, tile_mvox;      // fluid tiles: mesh vertices (per corner code), facets, the voxels under the vertices
    std::vector<double> tmv_v0[3];
    std::vector<unsigned char> tile_fcount;
    bool any_fluid_tiles = false;
    int nx_total = 0;
    const int tiled_now = tiling_allowed_ ? tiled_ : 0;
    if (tiled_now > 0) {
        const size_t lds_cap = 160 * 1024 - VXH_TILE_STATIC_LDS;
        int n_work = 0;
        for (int r = 0; r < nr; ++r) if (robots_[r].nvox > 0) ++n_work;
        // Which robots are tiled.  By default only those the resident kernel cannot take (more than 1024 voxels, oversized tables): the
        // kernel that steps a robot is then a function of the robot ALONE, and so is its trajectory, bit for bit, whatever the batch --
        // a robot re-evaluated alone gives the bits it gave inside a generation of 512 (tests/test_gpu_parity.py
        // test_full_size_batch_properties; the tiled and the resident kernel agree to 1e-12 voxel, not to the bit).
        // Option tile_small = 1 gives that up for speed where it was measured to pay (scripts/dev_gpu_diag.py tilepolicy, one box, us per
        // population step, resident | tiled): 64 robots of 6^3 10.2 | 11.1, 7^3 11.0 | 11.3, 8^3 11.1 | 11.7, 9^3 13.7 | 14.2, 10^3 15.6 |
        // 14.7; 16 of 10^3 15.1 | 13.2; 128 of 8^3 11.3 | 14.0; 128 of 10^3 16.0 | 32.9 (their tiles no longer fit one co-resident
        // launch) -- i.e. large robots (the 768- and 1024-thread variants) in populations of at most a quarter of the CUs.  (Until late in
        // round 2 the default was "any robot, up to three quarters of the CUs": slower for every size below 10^3, 2x slower at 128 robots,
        // and population-dependent in the last bits.)
        const bool small_population = tile_small_ && n_work * 4 <= D.n_cu;
        std::vector<int> cand;
        for (int r = 0; r < nr; ++r) {
            const RobotModel& M = robots_[r];
            // land_water robots are tiled on land (round 4: the tiles keep the directional strains the RobotVolume tags need) and, since
            // round 5, in a FLUID: the tile then carries its part of the drag mesh, and the voxels a mesh vertex averages over -- up to
            // seven, across tile boundaries, diagonal neighbours included -- come through the exchange buffer with their strains
            if (M.nvox == 0) continue;
            const int block = fused_variant(M).block;
            if (tiled_now == 2 || !fused_ || block == 0 || (small_population && block >= 768)) cand.push_back(r);
        }
        // tiles per robot: one bond per lane and one WAVEFRONT of owned voxels per tile if the CUs allow it (a step then costs one bond
        // evaluation + one voxel update on one wavefront + the barrier), fewer when the candidates outnumber the CUs.  (Until late in
        // round 3 only the bonds counted: 112 tiles of 71-72 voxels for the 20^3 lattice -- a second, nearly empty wavefront in every
        // voxel phase -- where 125 tiles of 64 step it in 8.7 instead of 10.3 us; scripts/dev_gpu_diag.py cfg4tiles: 64 tiles 11.2,
        // 216 tiles 9.1, 250 tiles 8.8.)
        auto k_bonds = [&](const RobotModel& M) { return std::max(1, (int)((M.nbond * 5LL / 4 + VXH_TILE_BLOCK - 1) / VXH_TILE_BLOCK)); };
        auto k_wave = [&](const RobotModel& M) { return std::max(k_bonds(M), (M.nvox + 63) / 64); };
        long long sum_wave = 0;
        for (int r : cand) sum_wave += k_wave(robots_[r]);
        const bool by_wave = sum_wave <= D.n_cu;       // (else the round-2 rule: by bonds, scaled down to the CUs -- 64 robots of 10^3 with tile_small: 14.3 us, against 16.4 with the wavefront rule scaled down)
        auto k_latency = [&](const RobotModel& M) { return by_wave ? k_wave(M) : k_bonds(M); };
        long long sum_lat = 0;
        for (int r : cand) sum_lat += k_latency(robots_[r]);
        struct Planned { int r; TilePlan plan; int tabg; size_t lds; int mesh; int small; };
        // a robot in a fluid: per tile the mesh vertices its owned voxels' facets use and those facets (counts, for the LDS layout; the tables
        // themselves are filled when the exchange slots of the robot's voxels are known)
        auto in_fluid = [&](const RobotModel& M) { return variant_ == 1 && M.vxa.fluid_env && M.nmv > 0; };
        auto tile_mesh_counts = [&](const RobotModel& M, const TilePlan::Tile& t, int& n_mv, int& n_f, int& n_mx) {
            n_mv = n_f = n_mx = 0;
            if (!in_fluid(M)) return;
            std::vector<char> seen(M.nmv, 0), vseen(M.nvox, 0);
            for (int v : t.own)
                for (int f = M.facet_first[v]; f < M.facet_first[v] + (int)M.facet_count[v]; ++f) {
                    ++n_f;
                    for (int k = 0; k < 3; ++k) {
                        const int i = M.facet_vert[(size_t)f * 3 + k];
                        if (seen[i]) continue;
                        seen[i] = 1; ++n_mv;
                        for (int e = 0; e < 8; ++e) { const int c = M.vert_comp[(size_t)i * 8 + e]; if (c >= 0 && !vseen[c >> 3]) { vseen[c >> 3] = 1; ++n_mx; } }
                    }
                }
        };
        std::vector<Planned> planned;
        for (int r : cand) {
            const RobotModel& M = robots_[r];
            const int tab_doubles = (int)(M.bond_classes.size() * sizeof(DBondClass) / 8 + M.vox_classes.size() * sizeof(DVoxClass) / 8);
            const int tabg = (size_t)tab_doubles * 8 > 32 * 1024 ? 1 : 0;
            int k = tiles_per_robot_ > 0 ? tiles_per_robot_
                                         : (sum_lat <= D.n_cu ? k_latency(M) : std::max(1, (int)(k_latency(M) * (long long)D.n_cu / sum_lat)));
            k = std::max(1, std::min(k, M.nvox / 8));
            for (;;) {
                TilePlan P = plan_tiles(M, k);
                size_t lds = 0;
                bool small = true;                                    // every tile within the SMALL size class: its kernel instances, its (fixed) layout
                for (const auto& t : P.tiles) small = small && (int)t.own.size() <= VXH_TILE_S_OWN && (int)t.halo.size() <= VXH_TILE_S_HALO && (int)t.bond_v1.size() <= VXH_TILE_S_BONDS;
                for (const auto& t : P.tiles) {
                    int n_mv = 0, n_f = 0, n_mx = 0;
                    tile_mesh_counts(M, t, n_mv, n_f, n_mx);
                    const TileLayout TL = small ? tile_layout(VXH_TILE_S_OWN, VXH_TILE_S_HALO, VXH_TILE_S_BONDS, tabg ? 0 : tab_doubles, M.nmv > 0, n_mv, n_f, n_mx)
                                                : tile_layout((int)t.own.size(), (int)t.halo.size(), (int)t.bond_v1.size(), tabg ? 0 : tab_doubles, M.nmv > 0, n_mv, n_f, n_mx);
                    lds = std::max(lds, (size_t)TL.total * 8);
                }
                if (P.k > VXH_TILE_MAX_TILES) break;                  // (left to the other kernels)
                if (P.max_own <= VXH_TILE_BLOCK && P.max_local <= 1024 && lds <= lds_cap) { planned.push_back({r, std::move(P), tabg, lds, M.nmv > 0 ? (in_fluid(M) ? 2 : 1) : 0, small ? 1 : 0}); break; }
                if (k >= M.nvox / 8) break;                           // cannot be tiled: left to the other kernels
                k = std::min(std::max(k + 1, k * 5 / 4 + 1), std::max(1, M.nvox / 8));
            }
        }
        // launches: all tiles of a robot in one launch, a launch no larger than what the chip keeps resident (the tiles of a
        // robot wait for each other).  How many workgroups of THIS kernel a CU holds is asked of the runtime (registers, wavefront
        // slots and LDS of the compiled code: five wavefronts of 256 vector registers are one workgroup per CU, whatever the LDS says)
        for (int kind = 0; kind < 12; ++kind) {
            const int tabg = kind & 1, mesh = (kind >> 1) % 3, small = kind / 6;      // mesh: 0 _voxcad, 1 land_water on land, 2 land_water in a fluid
            Device::TileLaunch cur;
            cur.tabg = tabg; cur.mesh = mesh; cur.small = small;
            auto capacity = [&](size_t lds) { return (long long)D.n_cu * tile_workgroups_per_cu(tabg, mesh, lds); };
            for (auto& q : planned) {
                if (q.tabg != tabg || q.mesh != mesh || q.small != small) continue;
                const int k = q.plan.k;
                if (k > capacity(q.lds)) continue;                    // more tiles than the chip holds: not tiled
                const size_t lds = std::max(cur.lds, q.lds);
                if (cur.count > 0 && cur.count + k > capacity(lds)) { D.tile_launches.push_back(cur); cur = Device::TileLaunch(); cur.tabg = tabg; cur.mesh = mesh; cur.small = small; }
                cur.lds = std::max(cur.lds, q.lds);
                const int tile0 = (int)h_tiles.size(), base = D.vox_begin[q.r];
                for (int t = 0; t < k; ++t) {
                    const TilePlan::Tile& T = q.plan.tiles[t];
                    DTile d;
                    d.robot = q.r; d.tile0 = tile0; d.ntiles = k;
                    d.n_own = (int)T.own.size(); d.n_halo = (int)T.halo.size(); d.nb = (int)T.bond_v1.size();
                    d.vox_off = (int)tile_vox.size(); d.bond_off = (int)tile_bond.size();
                    d.xoff = nx_total; d.pad = 0; d.mv_off = d.n_mv = d.f_off = d.n_f = d.mx_off = d.n_mx = 0;
                    for (size_t i = 0; i < T.own.size(); ++i) { tile_vox.push_back(base + T.own[i]); xslot[base + T.own[i]] = nx_total + (int)i; }
                    nx_total += ((int)T.own.size() + 63) / 64 * 64;
                    for (int v : T.halo) tile_vox.push_back(-(base + v) - 1);      // (exchange slots once every tile of the robot has its range)
                    for (size_t b = 0; b < T.bond_v1.size(); ++b) {
                        tile_bond.push_back(T.bond_entry[b]);
                        tile_bcls.push_back(robots_[q.r].bond_class[(size_t)T.bond_v1[b] * 3 + T.bond_axis[b]]);
                        tile_bslot.push_back(T.bond_axis[b] * nv + base + T.bond_v1[b]);
                    }
                    cur.tile_ids.push_back((int)h_tiles.size());
                    h_tiles.push_back(d);
                }
                for (int t = 0; t < k; ++t) {                          // halo voxels: global slot -> exchange slot
                    const DTile& d = h_tiles[tile0 + t];
                    for (int i = 0; i < d.n_halo; ++i) { int& e = tile_vox[d.vox_off + d.n_own + i]; e = xslot[-(e + 1)]; }
                }
                tile_ffirst.resize(tile_vox.size(), 0); tile_fcount.resize(tile_vox.size(), 0);
                if (in_fluid(robots_[q.r])) {
                    // the drag mesh, tile by tile: the vertices the facets of the tile's owned voxels use (every voxel touching such a vertex named
                    // by its EXCHANGE slot: the vertex pass reads pose and strains of all of them from the exchange buffer, whoever owns them)
                    // and those facets, per owned voxel in the reference's order
                    const RobotModel& M = robots_[q.r];
                    any_fluid_tiles = true;
                    for (int t = 0; t < k; ++t) {
                        DTile& d = h_tiles[tile0 + t];
                        const TilePlan::Tile& T = q.plan.tiles[t];
                        std::vector<int> local(M.nmv, -1), verts;
                        d.f_off = (int)tf_owner.size();
                        for (size_t o = 0; o < T.own.size(); ++o) {
                            const int v = T.own[o];
                            tile_ffirst[d.vox_off + o] = (int)tf_owner.size() - d.f_off;
                            tile_fcount[d.vox_off + o] = M.facet_count[v];
                            for (int f = M.facet_first[v]; f < M.facet_first[v] + (int)M.facet_count[v]; ++f) {
                                tf_owner.push_back((int)o);
                                for (int c = 0; c < 3; ++c) {
                                    const int i = M.facet_vert[(size_t)f * 3 + c];
                                    if (local[i] < 0) { local[i] = (int)verts.size(); verts.push_back(i); }
                                    tf_vert[c].push_back(local[i]);
                                }
                            }
                        }
                        d.n_f = (int)tf_owner.size() - d.f_off;
                        d.mv_off = (int)tmv_v0[0].size(); d.n_mv = (int)verts.size();
                        d.mx_off = (int)tile_mvox.size();
                        std::vector<int> vlocal(M.nvox, -1);
                        for (int i : verts) {
                            int slot[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
                            for (int e = 0; e < 8; ++e) {
                                const int c = M.vert_comp[(size_t)i * 8 + e];
                                if (c < 0) continue;
                                const int v = c >> 3;
                                if (vlocal[v] < 0) { vlocal[v] = (int)tile_mvox.size() - d.mx_off; tile_mvox.push_back(xslot[base + v]); }
                                slot[c & 7] = vlocal[v];
                            }
                            for (int c = 0; c < 8; ++c) tmv_slot[c].push_back(slot[c]);
                            for (int c = 0; c < 3; ++c) tmv_v0[c].push_back(M.vert_v0[(size_t)i * 3 + c]);
                        }
                        d.n_mx = (int)tile_mvox.size() - d.mx_off;
                    }
                }
                cur.count += k;
                cur.robots.push_back(q.r);
                D.robot_tiled[q.r] = 1;
                D.robot_tiles[q.r] = k;
            }
            if (cur.count > 0) D.tile_launches.push_back(cur);
        }
        for (auto& L : D.tile_launches) {
            L.list = D.upload(L.tile_ids);
            // a launch with no more tiles than CUs: a CU to every tile.  Two tiles that share a CU share its SIMDs, run at half
            // speed, and the whole robot waits for them at every barrier; asking for more than half of the LDS keeps them apart.
            if (L.count <= D.n_cu) L.lds = std::max(L.lds, (size_t)(81 * 1024));
            if (std::getenv("VXH_PROF_TILES"))
                std::fprintf(stderr, "vxhip: tile launch kind tabg=%d mesh=%d: %d tiles of %zu robots, %zu B of LDS, %lld workgroups per CU\n", L.tabg, L.mesh, L.count,
                             L.robots.size(), L.lds, tile_workgroups_per_cu(L.tabg, L.mesh, L.lds));
        }
    }
    {
        std::vector<int> tile_of(h_tiles.empty() ? 1 : nv, -1), tile_lidx(h_tiles.empty() ? 1 : nv, 0);
        for (size_t t = 0; t < h_tiles.size(); ++t)
            for (int k = 0; k < h_tiles[t].n_own; ++k) { const int g = tile_vox[h_tiles[t].vox_off + k]; tile_of[g] = (int)t; tile_lidx[g] = k; }
        B.tile_of = D.upload(tile_of);
        B.tile_lidx = D.upload(tile_lidx);
        B.col_code = D.alloc_raw<int>(h_tiles.empty() ? 1 : (size_t)std::max<long long>(col_total, 1));
        B.tile_xh = D.alloc_zero<int>(std::max<size_t>(1, h_tiles.size()) * VXH_TILE_XH);
        B.tile_xhn = D.alloc_zero<int>(std::max<size_t>(1, h_tiles.size()));
    }
    B.n_tiles = (int)h_tiles.size();
    B.tiles = D.upload(h_tiles);
    B.tile_vox = D.upload(tile_vox);
    B.tile_bond = D.upload(tile_bond);
    B.tile_bcls = D.upload(tile_bcls);
    B.tile_bslot = D.upload(tile_bslot);
    B.nx = std::max(nx_total, 64);
    B.xslot = D.upload(xslot);
    B.xplanes = any_fluid_tiles ? 28 : 16;
    B.xch = D.alloc_zero<unsigned long long>(h_tiles.empty() ? 1 : (size_t)3 * B.xplanes * B.nx);
    {   // fluid tiles: their parts of the drag mesh
        const size_t nm = tmv_v0[0].size(), nf = tf_owner.size();
        std::vector<int> mvert(std::max<size_t>(1, 8 * nm), -1), facet(std::max<size_t>(1, 4 * nf), 0);
        std::vector<double> mv0(std::max<size_t>(1, 3 * nm), 0.0);
        for (int c = 0; c < 8; ++c) for (size_t i = 0; i < nm; ++i) mvert[(size_t)c * nm + i] = tmv_slot[c][i];
        for (int c = 0; c < 3; ++c) for (size_t i = 0; i < nm; ++i) mv0[(size_t)c * nm + i] = tmv_v0[c][i];
        for (size_t f = 0; f < nf; ++f) { facet[f] = tf_owner[f]; for (int c = 0; c < 3; ++c) facet[(size_t)(c + 1) * nf + f] = tf_vert[c][f]; }
        tile_ffirst.resize(std::max<size_t>(1, tile_vox.size()), 0); tile_fcount.resize(std::max<size_t>(1, tile_vox.size()), 0);
        B.n_tmv = (int)nm; B.n_tf = (int)nf;
        B.tile_mvert = D.upload(mvert); B.tile_mv0 = D.upload(mv0); B.tile_facet = D.upload(facet);
        B.tile_ffirst = D.upload(tile_ffirst); B.tile_fcount = D.upload(tile_fcount);
        if (tile_mvox.empty()) tile_mvox.push_back(0);
        B.tile_mvox = D.upload(tile_mvox);
    }
    B.tile_mv = D.alloc_zero<unsigned long long>(std::max<size_t>(1, h_tiles.size()) * 3 * VXH_TILE_MV_STRIDE);
    {   // fused path: launch groups by kernel variant; inside a group the longest-running robots first
        D.groups.clear();
        for (int r = 0; r < nr; ++r) {
            if (D.robot_tiled[r]) continue;
            const FusedVariant fv = fused_variant(robots_[r]);
            if (fv.block == 0) continue;                              // streaming kernels
            Device::Group* g = nullptr;
            for (auto& q : D.groups) if (q.block == fv.block && q.nacc == fv.nacc && q.fluid == fv.fluid && q.tabg == fv.tabg && q.wide == fv.wide && q.two_tiles == fv.two_tiles && q.pair == fv.pair) g = &q;
            if (!g) { D.groups.emplace_back(); g = &D.groups.back(); g->block = fv.block; g->nacc = fv.nacc; g->fluid = fv.fluid; g->tabg = fv.tabg; g->wide = fv.wide; g->two_tiles = fv.two_tiles; g->pair = fv.pair; }
            g->robots.push_back(r);
            g->lds = std::max(g->lds, fv.lds);
        }
        {
            // masks of the robots the streaming kernels step: everything the other kernels do not take / (option fused = 0)
            // everything but the tiled robots
            std::vector<unsigned char> all(std::max(nr, 1), 1), rest(std::max(nr, 1), 1);
            for (auto& g : D.groups) for (int r : g.robots) rest[r] = 0;
            for (int r = 0; r < nr; ++r) if (D.robot_tiled[r]) all[r] = rest[r] = 0;
            D.n_rest = D.n_all = 0;
            for (int r = 0; r < nr; ++r) { if (rest[r] && robots_[r].nvox > 0) ++D.n_rest; if (all[r] && robots_[r].nvox > 0) ++D.n_all; }
            D.streamed_all = D.upload(all);
            D.streamed_rest = D.upload(rest);
        }
        size_t gi = 0;
        for (auto& g : D.groups) {
            std::stable_sort(g.robots.begin(), g.robots.end(), [&](int a, int b) {
                return (double)robots_[a].planned_steps * robots_[a].nvox > (double)robots_[b].planned_steps * robots_[b].nvox; });
            g.count = (int)g.robots.size();
            g.list = D.upload(g.robots);
            g.order_valid = false; g.order_cur = 0; g.order_use = false;
            if (!g.wide && !g.pair) {       // (k_robot_steps; colliding robots only: the others have no broad-phase runs to spread)
                for (int r : g.robots) if (robots_[r].vxa.self_col_enabled) g.order_use = true;
                if (g.order_use) for (int k = 0; k < 2; ++k) g.order_bits[k] = D.alloc_zero<unsigned long long>((size_t)(g.count + 63) / 64 + 1);
            }
            if (D.group_streams.size() <= gi) {
                hipStream_t st; hipEvent_t e0, e1;
                // launch groups of one call are meant to run side by side (two size classes of one population: 64 + 64 workgroups on 256
                // CUs).  Streams share a few hardware queues, handed out round robin as streams are created: two group streams that
                // land on the same queue run their kernels one after the other (measured: the same 64 robots of two size classes
                // 14.8 or 24.3 us per step, depending on how many streams the process had created before).  Streams of different
                // priority never share a queue, so the group streams cycle through the priorities the device offers.
                int least = 0, greatest = 0;
                HIP_OK(hipDeviceGetStreamPriorityRange(&least, &greatest));
                const int span = least - greatest + 1;                 // (numerically: greatest priority = smallest number)
                const int prio = span > 1 ? greatest + (int)(D.group_streams.size() % (size_t)span) : 0;
                HIP_OK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio));
                HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
                D.group_streams.push_back(st); D.group_events.push_back(e0); D.group_events.push_back(e1);
            }
            g.stream = D.group_streams[gi]; g.t0 = D.group_events[2 * gi]; g.t1 = D.group_events[2 * gi + 1];
            ++gi;
        }
    }
    {   // streaming path: one rebuild block per 256 surface voxels of every colliding robot
        std::vector<int> rr, ri;
        for (int r = 0; r < nr; ++r)
            if (robots_[r].vxa.self_col_enabled)
                for (int i0 = 0; i0 < robots_[r].nsurf; i0 += 256) { rr.push_back(r); ri.push_back(i0); }
        D.reb_blocks = (int)rr.size();
        D.reb_robot = D.upload(rr);
        D.reb_i0 = D.upload(ri);
    }
#ifdef VXH_PHASE_TIMING
    B.prof = D.alloc_zero<unsigned long long>(16 * 8 + 256 * 8);
#endif
    B.small_angle_w = std::cos(VXH_SMALL_ANGLE_RAD * 0.5);                    // Vec3D.h:55-59
    B.smallish_angle_w = std::cos(VXH_HYST * VXH_SMALL_ANGLE_RAD * 0.5);
    B.slthresh_acos2sqrt = 1.0 - 0.9988 * 0.9988;
    HIP_OK(hipStreamSynchronize(D.stream));      // every upload has left the host vectors above
    D.verify_uploads("at the end of prepare()");
    hs.mark("kernel choice, tiling, launch groups");
    hs.print("prepare");
    prepared_ = true;
    state_downloaded_ = control_downloaded_ = reduced_downloaded_ = false;
    host_.clear();
    rounds_done_ = 0;
    counters_ = vxh_counters{};
}

void Engine::reset() { if (!robots_.empty()) prepare(); }

static void launch_group(const DBatch& B, int block, bool fluid, bool tabg, bool wide, const int* list, int count, size_t lds, hipStream_t s, long long cap, int iters, int two_tiles = 0, int pair = 0,
                         const unsigned long long* order_in = nullptr, unsigned long long* order_out = nullptr)
{
#ifdef VXH_PAIR
    if (pair) { launch_pair_group(B, tabg, pair == 2, list, count, lds, s, cap, iters); return; }
#else
    (void)pair;                               // (check_option refuses `pair` in a library built without -DVXH_PAIR)
#endif
    if (wide) { launch_wide_group(B, fluid, tabg, list, count, lds, s, cap, iters, two_tiles); return; }
    if (fluid) launch_fused_mesh(B, block, tabg, list, count, lds, s, cap, iters, order_in, order_out);
    else launch_fused_land(B, block, tabg, list, count, lds, s, cap, iters, order_in, order_out);
}

// A call in two halves: advance_launch enqueues every kernel of the call and returns; advance_finish waits for the device, reads
// the control blocks back and does the accounting.  advance() = one after the other; a handle that pipelines a generation over
// several engines of one device (EngineSet::run) launches them all before it waits for the first.
void Engine::advance(long long max_rounds)
{
    const long long before = rounds_done_;
    try {
        advance_launch(max_rounds);
        advance_finish();
    } catch (const TileTimeout& e) {
        // The tiles of a robot are co-resident workgroups that wait for each other: a second process on the GPU can keep some of them
        // off the chip until the bounded spins give up.  The robots' state of before the call is gone -- but the evaluation is a
        // deterministic function of the imported state, which the engine still holds: the batch is assembled again without the tiled
        // kernel (streaming kernels for what the resident one cannot take: slower, the same trajectories to 1e-12 voxel) and stepped
        // from the start up to where this call was to end -- a call in the middle of a run too (round 4; until then only a call that
        // started from the imported state was made again, and with several `voxelyze` processes on one GPU, the reference's own
        // launch pattern, this is the path that is hit).  The tiled kernel stays off for this engine.
        if (tiled_ == 0 || !tiling_allowed_) throw;
        std::fprintf(stderr, "vxhip: %s -- this batch is stepped again%s without the tiled kernel, which stays off for this engine\n", e.what(),
                     before != 0 ? " from its imported state" : "");
        dev_->pending.active = false;
        tiled_ = 0;
        prepare();
        const long long huge = 0x7fffffffffffffffLL / 4;
        advance_launch(max_rounds >= huge - before ? huge : before + max_rounds);
        advance_finish();
    } catch (...) {
        // a launch sequence that ended halfway (a refused kernel, a HIP error): nothing of it may still be queued when the caller goes on
        quiesce();
        throw;
    }
}

// every stream of this engine idle, no launched call pending, no sticky error: the state a failed call leaves behind
void Engine::quiesce()
{
    Device& D = *dev_;
    (void)hipSetDevice(device_id_);
    for (hipStream_t st : D.group_streams) (void)hipStreamSynchronize(st);
    if (D.tile_stream) (void)hipStreamSynchronize(D.tile_stream);
    if (D.stream) (void)hipStreamSynchronize(D.stream);
    (void)hipGetLastError();
    D.pending.active = false;
}

void Engine::advance_launch(long long max_rounds)
{
    HIP_OK(hipSetDevice(device_id_));
    Device& D = *dev_;
    DBatch& B = D.B;
    const long long cap_all = 0x7fffffffffffffffLL;
    const long long remaining = std::max(0LL, D.max_planned - rounds_done_);
    const long long todo = std::min(max_rounds, remaining);
    const long long cap = (max_rounds >= remaining) ? cap_all : rounds_done_ + max_rounds;
    // robots that fit the resident kernel are stepped by their launch groups, the others (more than 1024 voxels, oversized
    // mesh) by the streaming kernels, side by side on their own streams; with the option fused = 0 everything streams
    const bool fused = fused_ && !D.groups.empty();
    const bool tiled = !D.tile_launches.empty();
    const bool streaming = fused ? D.n_rest > 0 : D.n_all > 0;
    B.streamed = fused ? D.streamed_rest : D.streamed_all;
    // per-robot step counts before, to attribute the work of this call
    if (D.pending.active) throw std::logic_error("a launched call has not been finished");
    std::vector<int>& steps_before = D.pending.steps_before;
    steps_before.assign(robots_.size(), 0);
    for (size_t r = 0; r < robots_.size(); ++r) steps_before[r] = host_.size() == robots_.size() ? host_[r].steps : 0;
    HostStages hs_l;
    HIP_OK(hipEventRecord(D.ev0, D.stream));
    hs_l.mark("first event");
    long long launches = 0;
    std::vector<long long>& group_launches = D.pending.group_launches;
    group_launches.assign(D.groups.size(), 0);
    // one launch group and nothing else (the usual population: robots of one size class): everything on the engine's own stream --
    // no event hand-offs between streams, which cost the command processor ~10-20 us each; a 20-step call is ~0.65 ms of kernel
    const bool single = fused && D.groups.size() == 1 && !tiled && !streaming;
    // (Tried in round 3 and taken out again: for short calls, the robots closest to their next broad-phase run dispatched first --
    // by displacement since the last one, from the previous call's control blocks -- so that the CUs that get them take fewer robots
    // afterwards.  The timed 20-step launch got 3 % shorter, 0.656 -> 0.634-0.650 ms; the host side of the call got 0.04-0.06 ms
    // longer in every variant tried -- list re-sorted and copied, partitioned and copied from pinned memory, read by the kernel from
    // mapped host memory -- which is more than the launch gained.  Polling the last event instead of sleeping on the stream: no gain.)
    if (fused) {
        const int iters = std::max(1, steps_per_launch_);
        for (auto& g : D.groups) {
            if (single) break;               // (its span is the call's: ev0 .. ev1; every event record is a packet the command processor works through)
            HIP_OK(hipStreamWaitEvent(g.stream, D.ev0, 0));
            HIP_OK(hipEventRecord(g.t0, g.stream));
        }
        for (long long done = 0; done < todo || done == 0; done += iters) {
            for (size_t k = 0; k < D.groups.size(); ++k) {
                auto& g = D.groups[k];
                // a short launch (what is left of the call, or the launch length itself): flagged robots first, flags for the next one
                const long long len = std::min<long long>(iters, std::max<long long>(1, todo - done));
                const bool dyn = g.order_use && g.order_valid && len <= VXH_ORDER_MAX_STEPS;
                launch_group(B, g.block, g.fluid != 0, g.tabg != 0, g.wide != 0, g.list, g.count, g.lds, single ? D.stream : g.stream, cap, iters, g.two_tiles, g.pair,
                             dyn ? g.order_bits[g.order_cur] : nullptr, g.order_use ? g.order_bits[g.order_cur ^ 1] : nullptr);
                if (g.order_use) { g.order_cur ^= 1; g.order_valid = true; }      // (every launch leaves the flags for the next)
                ++launches; ++group_launches[k];
            }
        }
        if (!single) for (auto& g : D.groups) HIP_OK(hipEventRecord(g.t1, g.stream));
    }
    long long tile_launch_count = 0;
    if (tiled) {
        // every tiled launch on the one tile stream: the tiles of a robot wait for each other, so two such launches must never
        // compete for the CUs
        // ... nor with anything else: the launches are sized for an empty chip (capacity() in prepare()), and the tiles of a robot that
        // are already placed spin -- holding their CUs -- until the missing ones arrive.  In a mixed batch the tiled launches therefore
        // run AFTER the launch groups of the resident robots, and the streaming kernels after them.
        const int iters = std::max(1, steps_per_launch_);
        HIP_OK(hipStreamWaitEvent(D.tile_stream, D.ev0, 0));
        if (fused) for (auto& g : D.groups) HIP_OK(hipStreamWaitEvent(D.tile_stream, g.t1, 0));
        HIP_OK(hipEventRecord(D.tile_t0, D.tile_stream));
        for (long long done = 0; done < todo || done == 0; done += iters)
            for (const auto& L : D.tile_launches) {
                ++tile_gen_;
                launch_tile_group(B, L.tabg != 0, L.mesh, L.small != 0, L.list, L.count, L.lds, D.tile_stream, cap, iters, tile_gen_);
                ++launches; ++tile_launch_count;
            }
        HIP_OK(hipEventRecord(D.tile_t1, D.tile_stream));
    }
    if (streaming) {
        if (tiled) HIP_OK(hipStreamWaitEvent(D.stream, D.tile_t1, 0));
        const int nb_b = (3 * B.nv + 255) / 256, nb_v = (B.nv + 255) / 256;
        auto round = [&](long long c) {
            hipLaunchKernelGGL(k_step_begin, dim3(B.n_robots), dim3(256), 0, D.stream, B, c, 1);
            if (D.any_fluid) {
                hipLaunchKernelGGL(k_mesh_vertices, dim3((B.n_mv + 255) / 256), dim3(256), 0, D.stream, B);
                hipLaunchKernelGGL(k_facets, dim3((B.n_facet + 255) / 256), dim3(256), 0, D.stream, B);
            }
            hipLaunchKernelGGL(k_bonds, dim3(nb_b + D.reb_blocks), dim3(256), 0, D.stream, B, nb_b, D.reb_robot, D.reb_i0);
            hipLaunchKernelGGL(k_voxels, dim3(nb_v), dim3(256), 0, D.stream, B);
        };
        long long launched = 0;
        if (graph_steps_ > 1 && cap == cap_all && todo >= graph_steps_) {
            if (!D.graph_exec || D.graph_rounds != graph_steps_ || D.graph_mask != B.streamed) {
                if (D.graph_exec) { hipGraphExecDestroy(D.graph_exec); D.graph_exec = nullptr; }
                if (D.graph) { hipGraphDestroy(D.graph); D.graph = nullptr; }
                HIP_OK(hipStreamBeginCapture(D.stream, hipStreamCaptureModeThreadLocal));
                for (int k = 0; k < graph_steps_; ++k) round(cap_all);
                HIP_OK(hipStreamEndCapture(D.stream, &D.graph));
                HIP_OK(hipGraphInstantiate(&D.graph_exec, D.graph, nullptr, nullptr, 0));
                D.graph_rounds = graph_steps_;
                D.graph_mask = B.streamed;
            }
            while (todo - launched >= graph_steps_) { HIP_OK(hipGraphLaunch(D.graph_exec, D.stream)); launched += graph_steps_; }
        }
        for (; launched < todo; ++launched) round(cap);
        hipLaunchKernelGGL(k_step_begin, dim3(B.n_robots), dim3(256), 0, D.stream, B, cap, 0);   // finish the last step
        launches += (D.any_fluid ? 5 : 3) * todo + 1;
    }
    if (fused && !single) for (auto& g : D.groups) HIP_OK(hipStreamWaitEvent(D.stream, g.t1, 0));
    if (tiled) HIP_OK(hipStreamWaitEvent(D.stream, D.tile_t1, 0));
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventRecord(D.ev1, D.stream));
    {   // the control blocks come back on the same stream, behind the kernels: one wait for both
        const size_t nr = robots_.size();
        if (D.h_rstate_cap < nr) throw std::logic_error("control-block mirror smaller than the batch");      // (sized by prepare())
        if (!(single && B.rstate_mirror))      // (one launch group and nothing else: its kernel has written the mirror itself)
            HIP_OK(hipMemcpyAsync(D.h_rstate, B.rstate, sizeof(DRobotState) * nr, hipMemcpyDeviceToHost, D.stream));
    }
    hs_l.mark("kernels + last event queued");
    hs_l.print("advance_launch");
    D.pending.active = true; D.pending.todo = todo; D.pending.launches = launches; D.pending.tile_launch_count = tile_launch_count;
    D.pending.fused = fused; D.pending.tiled = tiled; D.pending.streaming = streaming; D.pending.single = single;
}

void Engine::advance_finish()
{
    HIP_OK(hipSetDevice(device_id_));
    Device& D = *dev_;
    if (!D.pending.active) return;
    D.pending.active = false;
    const long long todo = D.pending.todo, launches = D.pending.launches, tile_launch_count = D.pending.tile_launch_count;
    const bool fused = D.pending.fused, tiled = D.pending.tiled, streaming = D.pending.streaming;
    const std::vector<int>& steps_before = D.pending.steps_before;
    const std::vector<long long>& group_launches = D.pending.group_launches;
    HostStages hs;
    HIP_OK(hipStreamSynchronize(D.stream));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, D.ev0, D.ev1));
    counters_.kernel_seconds += ms * 1e-3;
    counters_.launches += launches;
    rounds_done_ += todo;
    state_downloaded_ = reduced_downloaded_ = false;
    hs.mark("wait for the GPU");
    download_control(true);
    hs.mark("control blocks back");

    // dominant kernel of this call: the launch group that processed most voxel-steps (fused) / the whole call (streaming)
    std::vector<double> grp_vs(D.groups.size(), 0.0), grp_ab(D.groups.size(), 0.0);
    double all_vs = 0, all_ab = 0;
    auto work = [&](int r, double& vs, double& ab) {
        const double ds = host_[r].steps - steps_before[r];
        vs += ds * robots_[r].nvox; ab += ds * (224.0 * robots_[r].nvox + 144.0 * robots_[r].nbond);
    };
    for (size_t r = 0; r < robots_.size(); ++r) if (robots_[r].nvox) work((int)r, all_vs, all_ab);
    double tile_vs = 0, tile_ab = 0;                        // what the tiled kernel did
    int n_tiled = 0;
    for (size_t r = 0; r < robots_.size(); ++r) if (D.robot_tiled[r]) { work((int)r, tile_vs, tile_ab); ++n_tiled; }
    double grp_best = 0;
    if (fused) for (size_t k = 0; k < D.groups.size(); ++k) { double v = 0, a = 0; for (int r : D.groups[k].robots) work(r, v, a); grp_best = std::max(grp_best, v); }
    if (tiled && tile_vs >= grp_best && tile_vs >= all_vs - tile_vs - grp_best) {
        float cms = 0;
        HIP_OK(hipEventElapsedTime(&cms, D.tile_t0, D.tile_t1));
        counters_.dominant_block = 1;                      // (1 = k_tile_steps; the resident kernel reports its workgroup size)
        counters_.dominant_robots = n_tiled;
        counters_.dominant_launches = tile_launch_count; counters_.dominant_seconds = cms * 1e-3;
        counters_.dominant_alg_bytes = tile_ab; counters_.dominant_voxel_steps = tile_vs;
    } else if (fused && !D.groups.empty()) {
        size_t best = 0;
        for (size_t k = 0; k < D.groups.size(); ++k) {
            for (int r : D.groups[k].robots) work(r, grp_vs[k], grp_ab[k]);
            if (grp_vs[k] > grp_vs[best]) best = k;
        }
        double rest_vs = all_vs - tile_vs, rest_ab = all_ab - tile_ab;   // what the streaming kernels did next to the groups
        for (size_t k = 0; k < D.groups.size(); ++k) { rest_vs -= grp_vs[k]; rest_ab -= grp_ab[k]; }
        if (streaming && rest_vs > grp_vs[best]) {
            counters_.dominant_block = 0; counters_.dominant_robots = D.n_rest;
            counters_.dominant_launches = todo; counters_.dominant_seconds = ms * 1e-3;
            counters_.dominant_alg_bytes = rest_ab; counters_.dominant_voxel_steps = rest_vs;
        } else {
            float cms = 0;
            if (D.pending.single) cms = ms; else HIP_OK(hipEventElapsedTime(&cms, D.groups[best].t0, D.groups[best].t1));
            // (the resident kernel's workgroup size; + 1: the wide kernel; 1026: the pair kernel -- 512 threads, 1024 voxel slots)
            counters_.dominant_block = D.groups[best].pair ? VXH_PAIR_NV + 2 : D.groups[best].block + (D.groups[best].wide ? 1 : 0); counters_.dominant_robots = D.groups[best].count;
            counters_.dominant_launches = group_launches[best]; counters_.dominant_seconds = cms * 1e-3;
            counters_.dominant_alg_bytes = grp_ab[best]; counters_.dominant_voxel_steps = grp_vs[best];
        }
    } else {
        counters_.dominant_block = 0; counters_.dominant_robots = (int)robots_.size() - n_tiled;
        counters_.dominant_launches = todo; counters_.dominant_seconds = ms * 1e-3;
        counters_.dominant_alg_bytes = all_ab - tile_ab; counters_.dominant_voxel_steps = all_vs - tile_vs;
    }
    hs.mark("accounting");
    hs.print("advance_finish");
}

void Engine::run()
{
    auto t0 = std::chrono::steady_clock::now();
    if (robots_.empty()) return;
    if (!prepared_) prepare();
    advance(0x7fffffffffffffffLL / 4);
    counters_.run_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void Engine::run_launch()
{
    if (robots_.empty()) return;
    if (!prepared_) prepare();
    advance_launch(0x7fffffffffffffffLL / 4);
}
void Engine::run_finish() { if (!robots_.empty()) advance_finish(); }

void Engine::step(long long n)
{
    auto t0 = std::chrono::steady_clock::now();
    if (robots_.empty() || n <= 0) return;
    if (!prepared_) prepare();
    advance(n);
    counters_.run_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// per-robot control blocks (a few hundred bytes each): status, step counts, IniCM -> counters
void Engine::download_control(bool already_copied)
{
    Device& D = *dev_;
    const int nr = (int)robots_.size();
    std::vector<DRobotState> own;
    const DRobotState* rstate = D.h_rstate;          // (already_copied: advance_launch queued the copy into the pinned mirror, advance_finish waited for it)
    if (!already_copied || !D.h_rstate) {
        own.resize(nr);
        HIP_OK(hipMemcpy(own.data(), D.B.rstate, sizeof(DRobotState) * nr, hipMemcpyDeviceToHost));
        rstate = own.data();
    }
    if ((int)host_.size() != nr) host_.assign(nr, HostState());
    double vs = 0, bs = 0, ab = 0; long long mx = 0;
    for (int r = 0; r < nr; ++r) {
        const RobotModel& M = robots_[r];
        HostState& H = host_[r];
        const DRobotState& S = rstate[r];
        H.cur_time = S.cur_time; H.steps = S.steps; H.status = S.status; H.cm_init = S.cm_init; H.rebuilds = S.rebuilds;
        H.eol_post_y = S.eol_post_y;
        H.cm_trace.assign((size_t)4 * std::min(S.ntrace, D.trace_cap[r]), 0.0);      // (filled by download())
        for (int k = 0; k < 3; ++k) H.ini_cm[k] = S.ini_cm[k];
        if (S.col_overflow) H.status = VXH_ROBOT_COL_OVERFLOW;
        if (S.status == 5 || (r == 0 && inject_tile_timeout_ > 0 && !D.tile_launches.empty() && --inject_tile_timeout_ == 0)) {
            throw TileTimeout("HIP: the tiles of robot " + std::to_string(r) + " timed out waiting for each other (is another process "
                              "using this GPU? set the engine option tiled = 0 then)");
        }
        vs += (double)M.nvox * S.steps; bs += (double)M.nbond * S.steps; ab += (224.0 * M.nvox + 144.0 * M.nbond) * S.steps;
        mx = std::max(mx, (long long)S.steps);
    }
    counters_.voxel_steps = vs; counters_.bond_steps = bs; counters_.algorithmic_bytes = ab; counters_.max_steps = mx;
    control_downloaded_ = true;
}

// full voxel state (lazy: only when results or trajectories are requested)
void Engine::download()
{
    HIP_OK(hipSetDevice(device_id_));
    Device& D = *dev_;
    const DBatch& B = D.B;
    const int nr = (int)robots_.size(), nv = B.nv;
    if (!control_downloaded_) download_control();
    std::vector<double> planes((size_t)18 * nv);
    HIP_OK(hipMemcpy(planes.data(), B.vs, sizeof(double) * planes.size(), hipMemcpyDeviceToHost));
    auto plane = [&](int comp) { return planes.data() + (size_t)comp * nv; };
    for (int r = 0; r < nr; ++r) {
        const RobotModel& M = robots_[r];
        HostState& H = host_[r];
        const int n = M.nvox, base = D.vox_begin[r], b = H.steps & 1;
        H.pos.resize((size_t)3 * n); H.quat.resize((size_t)4 * n); H.scale.resize(n); H.lin_mom.resize((size_t)3 * n); H.ang_mom.resize((size_t)3 * n);
        for (int v = 0; v < n; ++v) {
            for (int k = 0; k < 3; ++k) { H.pos[3 * v + k] = plane(4 * b + k)[base + v]; H.lin_mom[3 * v + k] = plane(12 + k)[base + v]; H.ang_mom[3 * v + k] = plane(15 + k)[base + v]; }
            for (int k = 0; k < 4; ++k) H.quat[4 * v + k] = plane(8 + k)[base + v];
            H.scale[v] = plane(4 * b + 3)[base + v];
        }
    }
    // land_water robots: directional strains of the last step (RobotVolumeEnd)
    bool any_mesh = false;
    for (int r = 0; r < nr; ++r) any_mesh = any_mesh || robots_[r].nmv > 0;
    if (any_mesh) {
        std::vector<double> st((size_t)6 * nv);
        HIP_OK(hipMemcpy(st.data(), B.strain, sizeof(double) * st.size(), hipMemcpyDeviceToHost));
        for (int r = 0; r < nr; ++r) {
            const RobotModel& M = robots_[r];
            if (M.nmv == 0) continue;
            HostState& H = host_[r];
            H.strain.resize((size_t)6 * M.nvox);
            for (int v = 0; v < M.nvox; ++v)
                for (int k = 0; k < 6; ++k) H.strain[(size_t)6 * v + k] = st[(size_t)k * nv + D.vox_begin[r] + v];
        }
    } else {
        for (int r = 0; r < nr; ++r) host_[r].strain.clear();
    }
    state_downloaded_ = true;
}

const std::vector<double>& Engine::trace_of(int robot)
{
    if (!prepared_) throw std::logic_error("trace requested before vxh_run/vxh_step");
    if (!reduced_downloaded_) download_reduced();
    return host_[robot].cm_trace;
}

int Engine::cm_trace(int robot, double* out4n, int capacity)
{
    const std::vector<double>& t = trace_of(robot);
    const int n = (int)(t.size() / 4);
    for (int k = 0; k < std::min(n, capacity) * 4; ++k) out4n[k] = t[k];
    return n;
}

std::vector<double> Engine::angle_excess(int robot, bool at_end)
{
    const RobotModel& M = robots_[robot];
    std::vector<double> out;
    if (!at_end) { mesh_angle_excess(M, nullptr, nullptr, nullptr, out); return out; }
    if (!prepared_) throw std::logic_error("angle excesses of the final state requested before vxh_run/vxh_step");
    if (!state_downloaded_) download();
    const HostState& H = host_[robot];
    if (H.steps == 0 || (int)H.strain.size() != 6 * M.nvox) mesh_angle_excess(M, nullptr, nullptr, nullptr, out);
    else mesh_angle_excess(M, H.pos.data(), H.quat.data(), H.strain.data(), out);
    return out;
}

void Engine::shape(int robot, bool at_end, MeshShape& out)
{
    const RobotModel& M = robots_[robot];
    if (!at_end) { mesh_shape(M, nullptr, nullptr, nullptr, out); return; }
    if (!prepared_) throw std::logic_error("the final mesh requested before vxh_run/vxh_step");
    if (!state_downloaded_) download();
    const HostState& H = host_[robot];
    if (H.steps == 0 || (int)H.strain.size() != 6 * M.nvox) mesh_shape(M, nullptr, nullptr, nullptr, out);
    else mesh_shape(M, H.pos.data(), H.quat.data(), H.strain.data(), out);
}

void Engine::bond_modes(long long* large_angle, long long* total)
{
    if (!prepared_) throw std::logic_error("bond modes requested before vxh_run/vxh_step");
    HIP_OK(hipSetDevice(device_id_));
    const int nv = dev_->B.nv;
    std::vector<unsigned char> flags((size_t)3 * nv);
    HIP_OK(hipMemcpy(flags.data(), dev_->B.small_angle, flags.size(), hipMemcpyDeviceToHost));
    long long l = 0, t = 0;
    for (size_t r = 0; r < robots_.size(); ++r) {
        const RobotModel& M = robots_[r];
        for (int v = 0; v < M.nvox; ++v)
            for (int a = 0; a < 3; ++a)
                if (M.bond_class[(size_t)v * 3 + a] >= 0) { ++t; if (!(flags[(size_t)a * nv + dev_->vox_begin[r] + v] & 1)) ++l; }
    }
    *large_angle = l; *total = t;
}

// What the results need of the final state, reduced on the device (k_results: 72 bytes per robot instead of 144 per voxel), and the
// centre-of-mass traces
void Engine::download_reduced()
{
    HIP_OK(hipSetDevice(device_id_));
    Device& D = *dev_;
    const int nr = (int)robots_.size();
    if (!control_downloaded_) download_control();
    if (nr > 0) {
        if (!D.results) D.results = D.alloc_zero<DResult>(nr);
        hipLaunchKernelGGL(k_results, dim3(nr), dim3(256), 0, D.stream, D.B, D.results);
        HIP_OK(hipGetLastError());
        std::vector<DResult> h(nr);
        HIP_OK(hipMemcpyAsync(h.data(), D.results, sizeof(DResult) * nr, hipMemcpyDeviceToHost, D.stream));
        HIP_OK(hipStreamSynchronize(D.stream));
        for (int r = 0; r < nr; ++r) {
            HostState& H = host_[r];
            H.reduced = true;
            for (int k = 0; k < 3; ++k) H.red_cm[k] = h[r].cm[k];
            H.d2max = h[r].d2max; H.d2min = h[r].d2min; H.ymax = h[r].ymax; H.ymin = h[r].ymin; H.touching = h[r].touching; H.feet = h[r].feet;
        }
    }
    if (D.total_trace > 0) {
        std::vector<double> tr((size_t)D.total_trace * 4);
        HIP_OK(hipMemcpy(tr.data(), D.B.trace, sizeof(double) * tr.size(), hipMemcpyDeviceToHost));
        for (int r = 0; r < nr; ++r)
            for (size_t k = 0; k < host_[r].cm_trace.size(); ++k) host_[r].cm_trace[k] = tr[(size_t)D.trace_begin[r] * 4 + k];
    }
    reduced_downloaded_ = true;
}

void Engine::result(int robot, vxh_result* out)
{
    if (!prepared_) throw std::logic_error("results requested before vxh_run/vxh_step");
    // _voxcad: every tag from the device-side reductions; land_water: the RobotVolume tags need the surface mesh of the final state
    // (poses + strains of every voxel), evaluated on the host
    const bool need_state = variant_ == 1 || host_results_;
    if (need_state && !state_downloaded_) download();
    if (!need_state && !reduced_downloaded_) download_reduced();
    HostState& H = host_[robot];
    if (need_state) { const bool keep = H.reduced; H.reduced = false; compute_result(robots_[robot], H, out); H.reduced = keep; }
    else compute_result(robots_[robot], H, out);
}

void Engine::state14(int robot, double* out, int capacity)
{
    const RobotModel& M = robots_[robot];
    if (capacity < M.nvox) throw std::invalid_argument("state buffer too small");
    if (!prepared_) prepare();
    if (!state_downloaded_) download();
    const HostState& H = host_[robot];
    for (int v = 0; v < M.nvox; ++v) {
        const VoxClass& C = M.vox_classes[M.vox_class[v]];
        double* o = out + (size_t)14 * v;
        for (int k = 0; k < 3; ++k) o[k] = H.pos[3 * v + k];
        for (int k = 0; k < 4; ++k) o[3 + k] = H.quat[4 * v + k];
        o[7] = H.scale[v];
        for (int k = 0; k < 3; ++k) { o[8 + k] = H.lin_mom[3 * v + k] * C.mass_inv; o[11 + k] = H.ang_mom[3 * v + k] * C.inertia_inv; }
    }
}

}  // namespace vxh
