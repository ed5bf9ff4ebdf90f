// EngineSet: the engines behind one C-ABI handle (engine.hpp).  Host-only code.
#include <algorithm>
#include <chrono>
#include <exception>
#include <iterator>
#include <stdexcept>
#include <thread>

#include "engine.hpp"

namespace vxh {

EngineSet::EngineSet(int variant, const std::vector<int>& device_ids)
{
    if (device_ids.empty()) throw std::invalid_argument("no device");
    for (int d : device_ids) engines_.emplace_back(new Engine(variant, d));
    // two engines on ONE device would have their multi-workgroup launches compete for the CUs (kernels_tiled.hpp: the tiles of a robot
    // wait for each other): when such a handle spreads robots over its engines they step with the other kernels (no_tiling_if_shared)
    std::vector<int> sorted = device_ids;
    std::sort(sorted.begin(), sorted.end());
    repeated_ = std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end();
    pipelined_ = device_ids.size() > 1 && sorted.front() == sorted.back();
}

// The veto is the handle's, not the user's option: it lasts while several engines of a repeated device hold robots and is lifted when
// the robots are back on one engine (gather, clear), so a later generation with a lattice above 1024 voxels -- stepped by ONE engine
// -- gets the tiled kernel again, and a `tiled` / `tile_small` value the user set is never overwritten.
void EngineSet::no_tiling_if_shared()
{
    if (repeated_ && engines_.size() > 1)
        for (auto& e : engines_) e->set_tiling_allowed(false);
}
void EngineSet::tiling_allowed_again() { for (auto& e : engines_) e->set_tiling_allowed(true); }

void EngineSet::flush_pending()
{
    if (pending_.empty()) return;
    std::vector<RobotModel> m = engines_[0]->build_models(std::move(pending_));
    pending_.clear();
    gather();
    engines_[0]->append(std::move(m));
}

void EngineSet::gather()
{
    if (!distributed_) return;
    std::vector<std::vector<RobotModel>> held(engines_.size());
    for (size_t k = 0; k < engines_.size(); ++k) held[k] = engines_[k]->take_robots();
    std::vector<RobotModel> all(where_.size());
    for (size_t i = 0; i < where_.size(); ++i) all[i] = std::move(held[where_[i].first][where_[i].second]);
    engines_[0]->give_robots(std::move(all));
    where_.clear();
    distributed_ = false;
    tiling_allowed_again();
}

void EngineSet::distribute()
{
    if (distributed_) return;
    const int n = engines_[0]->num_robots(), nd = (int)engines_.size();
    where_.assign(n, {0, 0});
    if (nd == 1) {
        for (int i = 0; i < n; ++i) where_[i] = {0, i};
        distributed_ = true;
        return;
    }
    std::vector<RobotModel> all = engines_[0]->take_robots();
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    auto cost = [&](int i) { return (double)all[i].nvox * (double)std::max<long long>(1, all[i].planned_steps); };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost(a) > cost(b); });
    std::vector<double> load(nd, 0.0);
    std::vector<std::vector<int>> share(nd);
    for (int i : order) {
        const int k = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        share[k].push_back(i);
        load[k] += cost(i);
    }
    // several engines of ONE device about to step side by side: no multi-workgroup kernel (its tiles must own the device); a batch
    // that ends up on a single engine -- one large lattice -- keeps it
    int busy = 0;
    for (int k = 0; k < nd; ++k) busy += share[k].empty() ? 0 : 1;
    if (busy > 1) no_tiling_if_shared(); else tiling_allowed_again();
    for (int k = 0; k < nd; ++k) {
        std::sort(share[k].begin(), share[k].end());
        std::vector<RobotModel> mine;
        for (size_t j = 0; j < share[k].size(); ++j) { where_[share[k][j]] = {k, (int)j}; mine.push_back(std::move(all[share[k][j]])); }
        engines_[k]->give_robots(std::move(mine));
    }
    distributed_ = true;
}

void EngineSet::each(const std::function<void(Engine&)>& body)
{
    std::vector<std::exception_ptr> errors(engines_.size());
    std::vector<std::thread> pool;
    auto work = [&](size_t k) { try { if (engines_[k]->num_robots() > 0) body(*engines_[k]); } catch (...) { errors[k] = std::current_exception(); } };
    for (size_t k = 1; k < engines_.size(); ++k) pool.emplace_back(work, k);
    work(0);
    for (auto& t : pool) t.join();
    for (auto& e : errors) if (e) std::rethrow_exception(e);
}

// additions: parse and build first -- a refused call (bad file, unsupported feature) leaves the robots where they are, with the
// results of earlier runs readable -- then bring every robot back to engine 0 and append
int EngineSet::add_vxa(const char* data, size_t len)
{ std::vector<RobotModel> m = engines_[0]->build_vxa(data, len); flush_pending(); gather(); return engines_[0]->append(std::move(m)); }
int EngineSet::add_vxa_files(const std::vector<std::string>& paths)
{ std::vector<RobotModel> m = engines_[0]->build_vxa_files(paths); flush_pending(); gather(); return engines_[0]->append(std::move(m)); }
int EngineSet::add_arrays(const char* t, size_t len, const vxh_robot_arrays* robots, int n, bool round_like_text)
{
    if (pipelined_ && !distributed_ && engines_[0]->num_robots() == 0) {     // a fresh generation on a pipelining handle: check + copy now, build at the run
        std::vector<VxaModel> m = engines_[0]->models_from_arrays(t, len, robots, n, round_like_text);
        const int first = (int)pending_.size();
        for (auto& x : m) pending_.push_back(std::move(x));
        return first;
    }
    std::vector<RobotModel> m = engines_[0]->build_arrays(t, len, robots, n, round_like_text);
    flush_pending(); gather();
    return engines_[0]->append(std::move(m));
}

int EngineSet::num_robots() const { return (distributed_ ? (int)where_.size() : engines_[0]->num_robots()) + (int)pending_.size(); }
const RobotModel& EngineSet::robot(int i) const
{
    const_cast<EngineSet*>(this)->flush_pending();                          // (a reader before the run: the models are needed now)
    return distributed_ ? engines_[where_[i].first]->robot(where_[i].second) : engines_[0]->robot(i);
}

void EngineSet::run()
{
    if (!pending_.empty() && !distributed_ && engines_[0]->num_robots() == 0) {
        // the pipeline: chunk k = robots [k n / K, (k + 1) n / K) goes to engine k: built on the host cores, uploaded, its kernels
        // enqueued -- and the host moves on to chunk k + 1 while the device steps; then the engines are waited for in order, so the
        // control blocks of chunk k come back while chunk k + 1 is still stepping.  Which kernel steps a robot, and therefore its
        // result, does not depend on the chunking.
        const int n = (int)pending_.size();
        where_.assign(n, {0, 0});
        std::vector<VxaModel> all = std::move(pending_);
        pending_.clear();
        // a lattice of more than 1024 voxels needs the tiled kernel, which must own the device: such a generation is not pipelined
        bool large = false;
        for (const auto& m : all) { size_t occ = 0; for (unsigned char c : m.structure) occ += c != 0; large = large || occ > 1024; }
        const int K = large ? 1 : (int)engines_.size();
        const auto t0 = std::chrono::steady_clock::now();
        int launched = 0;
        try {
            if (K == 1) {
                // one engine owns the device: its own run(), i.e. with the tiled kernel and with the fall-back of a tile timeout
                // (Engine::advance: the batch is stepped again without that kernel), and its own accounting of the run's wall time
                tiling_allowed_again();
                engines_[0]->append(engines_[0]->build_models(std::move(all)));
                for (int i = 0; i < n; ++i) where_[i] = {0, i};
                distributed_ = true;
                engines_[0]->run();
                return;
            }
            no_tiling_if_shared();            // (no multi-workgroup kernel from here on: nothing below can time out on a tile)
            for (int k = 0; k < K; ++k) {
                const int lo = (int)((long long)n * k / K), hi = (int)((long long)n * (k + 1) / K);
                if (hi <= lo) continue;
                std::vector<VxaModel> chunk(std::make_move_iterator(all.begin() + lo), std::make_move_iterator(all.begin() + hi));
                engines_[k]->append(engines_[k]->build_models(std::move(chunk)));
                for (int i = lo; i < hi; ++i) where_[i] = {k, i - lo};
                engines_[k]->run_launch();
                launched = k + 1;
            }
            distributed_ = true;
            for (int k = 0; k < K; ++k) engines_[k]->run_finish();
        } catch (...) {
            // A chunk could not be built, uploaded or stepped.  The models of the chunks behind it were never built and those in front
            // hold a run that is not the generation's: the handle is left EMPTY (vxh_num_robots 0, every reader refuses the index)
            // rather than with indices that silently point at the wrong robot; the caller sees the error and hands the generation over again.
            for (int k = 0; k < launched; ++k) { try { engines_[k]->run_finish(); } catch (...) {} }
            clear();
            throw;
        }
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (int k = 0; k < K; ++k) engines_[k]->add_run_seconds(wall);      // (counters(): the slowest engine's = the run's)
        return;
    }
    flush_pending();
    distribute();
    each([](Engine& e) { e.run(); });
}
void EngineSet::step(long long n) { flush_pending(); distribute(); each([n](Engine& e) { e.step(n); }); }
void EngineSet::reset() { flush_pending(); distribute(); each([](Engine& e) { e.reset(); }); }      // (distribute first: not the whole population on device 0)
void EngineSet::clear() { for (auto& e : engines_) e->clear(); where_.clear(); pending_.clear(); distributed_ = false; tiling_allowed_again(); }

// (readers distribute first, like run(): after an addition, or a reset, the engine that holds the robot says what is missing)
void EngineSet::result(int robot, vxh_result* out)
{
    flush_pending(); distribute();
    engines_[where_[robot].first]->result(where_[robot].second, out);
}
void EngineSet::state14(int robot, double* out, int capacity)
{
    flush_pending(); distribute();
    engines_[where_[robot].first]->state14(where_[robot].second, out, capacity);
}
int EngineSet::cm_trace(int robot, double* out4n, int capacity)
{
    flush_pending(); distribute();
    return engines_[where_[robot].first]->cm_trace(where_[robot].second, out4n, capacity);
}
const std::vector<double>& EngineSet::trace_of(int robot)
{
    flush_pending(); distribute();
    return engines_[where_[robot].first]->trace_of(where_[robot].second);
}
std::vector<double> EngineSet::angle_excess(int robot, bool at_end)
{
    flush_pending(); distribute();
    return engines_[where_[robot].first]->angle_excess(where_[robot].second, at_end);
}
void EngineSet::shape(int robot, bool at_end, MeshShape& out)
{
    flush_pending(); distribute();
    engines_[where_[robot].first]->shape(where_[robot].second, at_end, out);
}
void EngineSet::bond_modes(long long* large_angle, long long* total)
{
    flush_pending(); distribute();
    *large_angle = *total = 0;
    for (auto& e : engines_) { if (e->num_robots() == 0) continue; long long l = 0, t = 0; e->bond_modes(&l, &t); *large_angle += l; *total += t; }
}

// sums over the devices; the times are those of the slowest device (they ran side by side), the dominant kernel that of the device whose
// dominant kernel did most of the last call's work
void EngineSet::counters(vxh_counters* out) const
{
    *out = vxh_counters{};
    double best = -1;
    for (const auto& e : engines_) {
        vxh_counters c;
        e->counters(&c);
        out->voxel_steps += c.voxel_steps; out->bond_steps += c.bond_steps; out->algorithmic_bytes += c.algorithmic_bytes;
        out->kernel_seconds = std::max(out->kernel_seconds, c.kernel_seconds); out->run_seconds = std::max(out->run_seconds, c.run_seconds);
        out->launches += c.launches; out->max_steps = std::max(out->max_steps, c.max_steps);
        if (e->num_robots() == 0 && best >= 0) continue;     // (an engine the last call left idle still shows the call before)
        const double work = e->num_robots() == 0 ? 0.0 : c.dominant_voxel_steps;
        if (work > best) {
            best = work;
            out->dominant_block = c.dominant_block; out->dominant_robots = c.dominant_robots; out->dominant_launches = c.dominant_launches;
            out->dominant_seconds = c.dominant_seconds; out->dominant_alg_bytes = c.dominant_alg_bytes; out->dominant_voxel_steps = c.dominant_voxel_steps;
        }
    }
}

// to every engine or to none: an engine without robots would accept what one that has stepped refuses
void EngineSet::set_option(const std::string& key, double value)
{
    for (auto& e : engines_) e->check_option(key, value);
    for (auto& e : engines_) e->set_option(key, value);
}

}  // namespace vxh
