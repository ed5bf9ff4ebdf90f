// C ABI of libvxhip (include/vxhip.h): exception -> status code translation around vxh::Engine.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "engine.hpp"

struct vxh_engine {
    vxh::EngineSet* impl = nullptr;
    std::string last_error;
};

namespace {

template <class F>
int guarded(vxh_engine* e, F&& body)
{
    if (!e || !e->impl) return VXH_ERR_ARG;
    try {
        body();
        return VXH_OK;
    } catch (const std::invalid_argument& ex) {
        e->last_error = ex.what();
        return std::strstr(ex.what(), "unsupported") ? VXH_ERR_UNSUPPORTED : VXH_ERR_ARG;
    } catch (const std::logic_error& ex) {
        e->last_error = ex.what();
        return VXH_ERR_STATE;
    } catch (const std::runtime_error& ex) {
        e->last_error = ex.what();
        if (std::strncmp(ex.what(), "HIP:", 4) == 0) return VXH_ERR_HIP;
        if (std::strncmp(ex.what(), "xml:", 4) == 0 || std::strncmp(ex.what(), "vxa:", 4) == 0) return VXH_ERR_PARSE;
        return VXH_ERR_PARSE;
    } catch (const std::exception& ex) {
        e->last_error = ex.what();
        return VXH_ERR_ARG;
    }
}

bool bad_robot(const vxh_engine* e, int robot) { return !e || !e->impl || robot < 0 || robot >= e->impl->num_robots(); }

}  // namespace

extern "C" {

int vxh_inspect_vxa_buffer(const char* xml, size_t len, int variant, vxh_model_info* out, char* errbuf, size_t errcap)
{
    if (!xml || !out || (variant != VXH_VOXCAD && variant != VXH_VOXCAD_LAND_WATER)) return VXH_ERR_ARG;
    auto fail = [&](int code, const char* what) {
        if (errbuf && errcap) { std::strncpy(errbuf, what, errcap - 1); errbuf[errcap - 1] = 0; }
        return code;
    };
    try {
        vxh::VxaModel vxa = vxh::read_vxa(xml, len, variant);
        if (!vxa.unsupported.empty()) return fail(VXH_ERR_UNSUPPORTED, vxa.unsupported.front().c_str());
        vxh::RobotModel m = vxh::build_robot(vxa);
        std::memset(out, 0, sizeof(*out));
        out->nvox = m.nvox; out->nbond = m.nbond; out->nsurf = m.nsurf;
        out->n_vox_classes = (int)m.vox_classes.size(); out->n_bond_classes = (int)m.bond_classes.size();
        out->opt_dt = m.opt_dt; out->dt = m.dt; out->planned_steps = m.planned_steps;
        out->alg_bytes_per_step = 224.0 * m.nvox + 144.0 * m.nbond;
        return VXH_OK;
    } catch (const std::exception& ex) {
        return fail(VXH_ERR_PARSE, ex.what());
    }
}

int vxh_plan_tiles_buffer(const char* xml, size_t len, int variant, int k_request, vxh_tiling_info* out, int* tile_of_out, int capacity,
                          char* errbuf, size_t errcap)
{
    if (!xml || !out || k_request < 1 || (variant != VXH_VOXCAD && variant != VXH_VOXCAD_LAND_WATER)) return VXH_ERR_ARG;
    auto fail = [&](int code, const char* what) {
        if (errbuf && errcap) { std::strncpy(errbuf, what, errcap - 1); errbuf[errcap - 1] = 0; }
        return code;
    };
    try {
        vxh::VxaModel vxa = vxh::read_vxa(xml, len, variant);
        if (!vxa.unsupported.empty()) return fail(VXH_ERR_UNSUPPORTED, vxa.unsupported.front().c_str());
        const vxh::RobotModel m = vxh::build_robot(vxa);
        const vxh::TilePlan plan = vxh::plan_tiles(m, k_request);
        std::memset(out, 0, sizeof(*out));
        out->k = plan.k; out->kx = plan.kx; out->ky = plan.ky; out->kz = plan.kz;
        out->max_own = plan.max_own; out->max_local = plan.max_local; out->max_bonds = plan.max_bonds;
        for (const auto& t : plan.tiles) out->total_bonds += (int)t.bond_v1.size();
        if (tile_of_out) {
            if (capacity < m.nvox) return fail(VXH_ERR_ARG, "tile_of buffer too small");
            for (int v = 0; v < m.nvox; ++v) tile_of_out[v] = plan.tile_of[v];
        }
        return VXH_OK;
    } catch (const std::exception& ex) {
        return fail(VXH_ERR_PARSE, ex.what());
    }
}

int vxh_inspect_constants(const char* xml, size_t len, int variant, double* vox12n, int vox_capacity, double* bond23n, int bond_capacity,
                          char* errbuf, size_t errcap)
{
    if (!xml || !vox12n || !bond23n || (variant != VXH_VOXCAD && variant != VXH_VOXCAD_LAND_WATER)) return VXH_ERR_ARG;
    auto fail = [&](int code, const char* what) {
        if (errbuf && errcap) { std::strncpy(errbuf, what, errcap - 1); errbuf[errcap - 1] = 0; }
        return code;
    };
    try {
        vxh::VxaModel vxa = vxh::read_vxa(xml, len, variant);
        if (!vxa.unsupported.empty()) return fail(VXH_ERR_UNSUPPORTED, vxa.unsupported.front().c_str());
        const vxh::RobotModel m = vxh::build_robot(vxa);
        if (vox_capacity < m.nvox || bond_capacity < m.nbond) return fail(VXH_ERR_ARG, "constant buffers too small");
        for (int v = 0; v < m.nvox; ++v) {
            const vxh::VoxClass& c = m.vox_classes[m.vox_class[v]];
            double* o = vox12n + (size_t)12 * v;
            o[0] = c.mass; o[1] = c.mass_inv; o[2] = c.inertia; o[3] = c.inertia_inv; o[4] = c.first_moment; o[5] = c.c_lin; o[6] = c.c_ang;
            o[7] = c.E; o[8] = c.nom_size; o[9] = c.u_static; o[10] = c.u_dynamic; o[11] = c.cte;
        }
        int k = 0;                               // bonds in the reference's creation order: per voxel, +X +Y +Z (VX_Sim.cpp:620-645)
        for (int v = 0; v < m.nvox; ++v)
            for (int a = 0; a < 3; ++a) {
                const int cls = m.bond_class[(size_t)v * 3 + a];
                if (cls < 0) continue;
                const vxh::BondClass& b = m.bond_classes[cls];
                double* o = bond23n + (size_t)23 * k++;
                o[0] = v; o[1] = m.nbr[(size_t)v * 6 + 2 * a]; o[2] = a; o[3] = b.homogeneous; o[4] = b.L; o[5] = b.a1; o[6] = b.a2;
                o[7] = b.b1; o[8] = b.b2; o[9] = b.b3; o[10] = b.b1; o[11] = b.b2; o[12] = b.b3;      // (cubic voxels: the y and z sets coincide)
                o[13] = b.sq_a1m1; o[14] = b.sq_a1m2; o[15] = b.sq_a2i1; o[16] = b.sq_a2i2;
                o[17] = b.sq_b1m1; o[18] = b.sq_b1m2; o[19] = b.sq_b2fm1; o[20] = b.sq_b2fm2; o[21] = b.sq_b3i1; o[22] = b.sq_b3i2;
            }
        return VXH_OK;
    } catch (const std::exception& ex) {
        return fail(VXH_ERR_PARSE, ex.what());
    }
}

double vxh_convex_hull_volume(const double* xyz, int n)
{
    if (!xyz || n < 0) return -1.0;
    return vxh::convex_hull_volume(std::vector<double>(xyz, xyz + (size_t)3 * n));
}

int vxh_create(vxh_engine** out, int variant, int device_id)
{
    if (!out || (variant != VXH_VOXCAD && variant != VXH_VOXCAD_LAND_WATER)) return VXH_ERR_ARG;
    *out = nullptr;
    return vxh_create_multi(out, variant, &device_id, 1);
}

int vxh_create_multi(vxh_engine** out, int variant, const int* device_ids, int n_devices)
{
    if (!out || !device_ids || n_devices < 1 || (variant != VXH_VOXCAD && variant != VXH_VOXCAD_LAND_WATER)) return VXH_ERR_ARG;
    *out = nullptr;
    vxh_engine* e = new vxh_engine;
    try {
        e->impl = new vxh::EngineSet(variant, std::vector<int>(device_ids, device_ids + n_devices));
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "libvxhip: %s\n", ex.what());
        delete e;
        return std::strncmp(ex.what(), "HIP:", 4) == 0 ? VXH_ERR_HIP : VXH_ERR_NO_DEVICE;
    }
    *out = e;
    return VXH_OK;
}

void vxh_destroy(vxh_engine* e)
{
    if (!e) return;
    delete e->impl;
    delete e;
}

int vxh_add_vxa_buffer(vxh_engine* e, const char* xml, size_t len, int* robot_index_out)
{
    if (!xml) return VXH_ERR_ARG;
    return guarded(e, [&] { int idx = e->impl->add_vxa(xml, len); if (robot_index_out) *robot_index_out = idx; });
}

int vxh_add_vxa_file(vxh_engine* e, const char* path, int* robot_index_out)
{
    if (!e || !e->impl || !path) return VXH_ERR_ARG;
    std::ifstream in(path, std::ios::binary);
    if (!in) { e->last_error = std::string("cannot open ") + path; return VXH_ERR_IO; }
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string text = ss.str();
    return vxh_add_vxa_buffer(e, text.data(), text.size(), robot_index_out);
}

int vxh_add_vxa_files(vxh_engine* e, const char* const* paths, int n, int* first_index_out)
{
    if (!e || !e->impl || n < 0 || (n > 0 && !paths)) return VXH_ERR_ARG;
    std::vector<std::string> list;
    for (int i = 0; i < n; ++i) { if (!paths[i]) return VXH_ERR_ARG; list.emplace_back(paths[i]); }
    int rc = guarded(e, [&] { int idx = e->impl->add_vxa_files(list); if (first_index_out) *first_index_out = idx; });
    if (rc == VXH_ERR_PARSE && e->last_error.compare(0, 3, "io:") == 0) rc = VXH_ERR_IO;
    return rc;
}

int vxh_add_robots(vxh_engine* e, const char* template_vxa, size_t template_len, const vxh_robot_arrays* robots, int n,
                   int round_like_text, int* first_index_out)
{
    if (!e || !e->impl || !template_vxa || n < 0 || (n > 0 && !robots)) return VXH_ERR_ARG;
    for (int i = 0; i < n; ++i)
        if (robots[i].n_layers < 0 || (robots[i].n_layers > 0 && (!robots[i].layer_tags || !robots[i].layers))) return VXH_ERR_ARG;
    return guarded(e, [&] { int idx = e->impl->add_arrays(template_vxa, template_len, robots, n, round_like_text != 0); if (first_index_out) *first_index_out = idx; });
}

int vxh_num_robots(const vxh_engine* e) { return (e && e->impl) ? e->impl->num_robots() : VXH_ERR_ARG; }

int vxh_robot_dims(const vxh_engine* e, int robot, int* nvox, int* nbond, double* dt, long long* planned_steps)
{
    if (bad_robot(e, robot)) return VXH_ERR_ARG;
    const vxh::RobotModel& m = e->impl->robot(robot);
    if (nvox) *nvox = m.nvox;
    if (nbond) *nbond = m.nbond;
    if (dt) *dt = m.dt;
    if (planned_steps) *planned_steps = m.planned_steps;
    return VXH_OK;
}

int vxh_voxel_actuation(const vxh_engine* e, int robot, int voxel, double* temp_amplitude, double* temp_period, double* phase_offset)
{
    if (bad_robot(e, robot)) return VXH_ERR_ARG;
    const vxh::RobotModel& m = e->impl->robot(robot);
    if (voxel < 0 || voxel >= m.nvox) return VXH_ERR_ARG;
    if (temp_amplitude) *temp_amplitude = (double)(float)m.vxa.temp_amplitude;
    if (temp_period) *temp_period = (double)(float)m.vxa.temp_period;
    if (phase_offset) *phase_offset = (double)m.phase_offset[(size_t)voxel];
    return VXH_OK;
}

int vxh_run(vxh_engine* e) { return guarded(e, [&] { e->impl->run(); }); }
int vxh_step(vxh_engine* e, long long nsteps) { return guarded(e, [&] { e->impl->step(nsteps); }); }
int vxh_reset(vxh_engine* e) { return guarded(e, [&] { e->impl->reset(); }); }
int vxh_clear(vxh_engine* e) { return guarded(e, [&] { e->impl->clear(); }); }

int vxh_get_result(const vxh_engine* ce, int robot, vxh_result* out)
{
    vxh_engine* e = const_cast<vxh_engine*>(ce);
    if (bad_robot(e, robot) || !out) return VXH_ERR_ARG;
    return guarded(e, [&] { e->impl->result(robot, out); });
}

int vxh_fitness_file_name(const vxh_engine* e, int robot, char* buf, size_t cap)
{
    if (bad_robot(e, robot) || !buf || cap == 0) return VXH_ERR_ARG;
    const std::string& name = e->impl->robot(robot).vxa.fitness_file_name;
    if (name.size() + 1 > cap) return VXH_ERR_ARG;
    std::memcpy(buf, name.c_str(), name.size() + 1);
    return VXH_OK;
}

int vxh_write_result_xml(const vxh_engine* ce, int robot, const char* path_or_null)
{
    vxh_engine* e = const_cast<vxh_engine*>(ce);
    if (bad_robot(e, robot)) return VXH_ERR_ARG;
    vxh_result res;
    int rc = vxh_get_result(e, robot, &res);
    if (rc != VXH_OK) return rc;
    const vxh::RobotModel& m = e->impl->robot(robot);
    std::string path = path_or_null ? path_or_null : m.vxa.fitness_file_name;
    if (path.empty()) { e->last_error = "no FitnessFileName in the .vxa and no path given"; return VXH_ERR_IO; }
    std::string text;
    std::vector<double> ex_end;       // land_water: angle excesses of the final mesh (the curvatures file below; its first value is <ShapeComplexityEnd>)
    std::FILE* curv_file = nullptr;   // <CurvaturesTmpFile>, opened before anything is computed (unwritable: both tags stay -1, like the reference)
    rc = guarded(e, [&] {
        double shape_start = -1.0, shape_end = -1.0;
        // (only when <CurvaturesTmpFile> is named AND can be opened for writing: the reference returns -1 before it computes anything
        // otherwise, LW/VX_MeshUtil.cpp:1022-1032 -- advisor, round 5)
        if (m.vxa.variant == 1 && m.nmv > 0 && !m.vxa.curvatures_tmp_file.empty()) {
            curv_file = std::fopen(m.vxa.curvatures_tmp_file.c_str(), "wb");
            if (curv_file) {
                std::vector<double> ex_start;
                vxh::mesh_angle_excess(m, nullptr, nullptr, nullptr, ex_start);
                ex_end = e->impl->angle_excess(robot, true);
                shape_start = vxh::shape_complexity_as_the_reference_prints_it(m, ex_start);
                shape_end = vxh::shape_complexity_as_the_reference_prints_it(m, ex_end);
            }
        }
        text = vxh::result_xml(m, res, e->impl->trace_of(robot), shape_start, shape_end);
    });
    // land_water: the final mesh's per-vertex angle excesses into <CurvaturesTmpFile>, tab-separated with the stream's six
    // significant digits, as CVX_MeshUtil::computeShapeComplexity leaves them for curvatureEntropy.py (LW/VX_MeshUtil.cpp:1016-1031;
    // the reference then runs that script -- absent from its repository -- and removes the file; here the file stays)
    if (curv_file) {
        for (double v : ex_end) std::fprintf(curv_file, "%g\t", v);
        std::fclose(curv_file);
    }
    if (rc != VXH_OK) return rc;
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { e->last_error = "cannot write " + path; return VXH_ERR_IO; }
    const bool ok = std::fwrite(text.data(), 1, text.size(), f) == text.size();
    std::fclose(f);
    if (!ok) { e->last_error = "short write to " + path; return VXH_ERR_IO; }
    return VXH_OK;
}

int vxh_get_angle_excess(const vxh_engine* ce, int robot, int at_end, double* out, int capacity, int* count_out)
{
    vxh_engine* e = const_cast<vxh_engine*>(ce);
    if (bad_robot(e, robot) || !count_out || (capacity > 0 && !out)) return VXH_ERR_ARG;
    return guarded(e, [&] {
        const std::vector<double> ex = e->impl->angle_excess(robot, at_end != 0);
        *count_out = (int)ex.size();
        for (int k = 0; k < std::min((int)ex.size(), capacity); ++k) out[k] = ex[k];
    });
}

int vxh_get_mesh(const vxh_engine* ce, int robot, int at_end, double* verts3, int vert_capacity, int* n_verts, int* facets3, int facet_capacity, int* n_facets)
{
    vxh_engine* e = const_cast<vxh_engine*>(ce);
    if (bad_robot(e, robot) || !n_verts || !n_facets || (vert_capacity > 0 && !verts3) || (facet_capacity > 0 && !facets3)) return VXH_ERR_ARG;
    return guarded(e, [&] {
        vxh::MeshShape sh;
        e->impl->shape(robot, at_end != 0, sh);
        *n_verts = (int)(sh.verts.size() / 3); *n_facets = (int)(sh.facets.size() / 3);
        for (int k = 0; k < 3 * std::min(*n_verts, vert_capacity); ++k) verts3[k] = sh.verts[k];
        for (int k = 0; k < 3 * std::min(*n_facets, facet_capacity); ++k) facets3[k] = sh.facets[k];
    });
}

int vxh_get_shape_descriptors(const vxh_engine* ce, int robot, int at_end, double* robot_volume, double* hull_volume, double* shape_complexity)
{
    vxh_engine* e = const_cast<vxh_engine*>(ce);
    if (bad_robot(e, robot)) return VXH_ERR_ARG;
    return guarded(e, [&] {
        vxh::MeshShape sh;
        e->impl->shape(robot, at_end != 0, sh);
        const bool none = sh.verts.empty();
        if (robot_volume) *robot_volume = none ? -1.0 : sh.robot_volume;
        if (hull_volume) *hull_volume = none ? -1.0 : sh.hull_volume;
        if (shape_complexity) *shape_complexity = none ? -1.0 : vxh::shape_complexity_as_the_reference_prints_it(e->impl->robot(robot), sh.angle_excess);
    });
}

int vxh_inspect_angle_excess(const char* xml, size_t len, int variant, double* out, int capacity, int* count_out, char* errbuf, size_t errcap)
{
    if (!xml || !count_out || (capacity > 0 && !out) || (variant != VXH_VOXCAD && variant != VXH_VOXCAD_LAND_WATER)) return VXH_ERR_ARG;
    auto fail = [&](int code, const char* what) {
        if (errbuf && errcap) { std::strncpy(errbuf, what, errcap - 1); errbuf[errcap - 1] = 0; }
        return code;
    };
    try {
        vxh::VxaModel vxa = vxh::read_vxa(xml, len, variant);
        if (!vxa.unsupported.empty()) return fail(VXH_ERR_UNSUPPORTED, vxa.unsupported.front().c_str());
        const vxh::RobotModel m = vxh::build_robot(vxa);
        std::vector<double> ex;
        vxh::mesh_angle_excess(m, nullptr, nullptr, nullptr, ex);
        *count_out = (int)ex.size();
        for (int k = 0; k < std::min((int)ex.size(), capacity); ++k) out[k] = ex[k];
        return VXH_OK;
    } catch (const std::exception& ex) {
        return fail(VXH_ERR_PARSE, ex.what());
    }
}

int vxh_get_state(const vxh_engine* ce, int robot, double* out14n, int capacity)
{
    vxh_engine* e = const_cast<vxh_engine*>(ce);
    if (bad_robot(e, robot) || !out14n) return VXH_ERR_ARG;
    return guarded(e, [&] { e->impl->state14(robot, out14n, capacity); });
}

int vxh_get_counters(const vxh_engine* e, vxh_counters* out)
{
    if (!e || !e->impl || !out) return VXH_ERR_ARG;
    e->impl->counters(out);
    return VXH_OK;
}

int vxh_get_cm_trace(const vxh_engine* ce, int robot, double* out4n, int capacity, int* count_out)
{
    vxh_engine* e = const_cast<vxh_engine*>(ce);
    if (bad_robot(e, robot) || capacity < 0 || (capacity > 0 && !out4n)) return VXH_ERR_ARG;
    return guarded(e, [&] { const int n = e->impl->cm_trace(robot, out4n, capacity); if (count_out) *count_out = n; });
}

int vxh_count_bond_modes(const vxh_engine* ce, long long* large_angle_out, long long* total_out)
{
    vxh_engine* e = const_cast<vxh_engine*>(ce);
    if (!e || !e->impl) return VXH_ERR_ARG;
    return guarded(e, [&] { long long l = 0, t = 0; e->impl->bond_modes(&l, &t); if (large_angle_out) *large_angle_out = l; if (total_out) *total_out = t; });
}

int vxh_set_option(vxh_engine* e, const char* key, double value)
{
    if (!key) return VXH_ERR_ARG;
    return guarded(e, [&] { e->impl->set_option(key, value); });
}

const char* vxh_strerror(int status)
{
    switch (status) {
    case VXH_OK: return "ok";
    case VXH_ERR_ARG: return "bad argument";
    case VXH_ERR_NO_DEVICE: return "no usable HIP device (libvxhip has no CPU path)";
    case VXH_ERR_PARSE: return "malformed .vxa";
    case VXH_ERR_IO: return "file I/O error";
    case VXH_ERR_HIP: return "HIP runtime error";
    case VXH_ERR_STATE: return "call order error";
    case VXH_ERR_UNSUPPORTED: return "unsupported .vxa feature";
    default: return "unknown status";
    }
}

const char* vxh_last_error(const vxh_engine* e) { return e ? e->last_error.c_str() : ""; }
const char* vxh_version(void) { return "vxhip 0.4.0 (gfx950)"; }

int vxh_device_count(void) { return vxh::hip_device_count(); }

}  // extern "C"
