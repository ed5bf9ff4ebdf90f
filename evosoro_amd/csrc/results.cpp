// Fitness metrics and result XML.  Reference: CVX_SimGA::WriteResultFile (evosoro/_voxcad/Voxelyze/VX_SimGA.cpp:33-203;
// land_water: evosoro/_voxcad_land_water/Voxelyze/VX_SimGA.cpp:33-77), the SS.* values of CVX_Sim::UpdateStats
// (VX_Sim.cpp:1518-1535) with their helpers GetCM / getAnteriorDist / getPosteriorY / GetNumTouchingFloor
// (VX_Sim.cpp:2415-2441,2584-2712).  Numbers are printed like `ostream << double` (6 significant digits,
// Utils/XML_Rip.h:57) in TinyXML's layout (4 blanks per level).
#include <cmath>
#include <vector>
#include <cstdio>
#include <cstring>

#include "engine.hpp"

namespace vxh {

namespace {

// CVX_MeshUtil::computeCurrentRobotVolume (LW/VX_MeshUtil.cpp:908-952): signed tetrahedra over the facets of the
// deformable surface mesh; vertices as updateDeformableMesh / GetCurVLoc (:368-428) places them: mean over the voxels
// touching the lattice corner of Pos + R(Angle) * corner, corner = +-(1 + strain) * L / 2.  `pos` etc. may be null:
// the rest state (RobotVolumeStart is taken right after Import, voxelyzeMain/main.cpp:65).
double robot_volume(const RobotModel& M, const double* pos, const double* quat, const double* strain)
{
    if (M.nmv == 0) return 0.0;
    const double nom = M.vxa.lattice_dim;
    std::vector<double> vert((size_t)M.nmv * 3);
    for (int i = 0; i < M.nmv; ++i) {
        double ax = 0, ay = 0, az = 0, tw = 0;
        for (int q = 0; q < 8; ++q) {
            const int comp = M.vert_comp[(size_t)i * 8 + q];
            if (comp < 0) break;
            const int u = comp >> 3, corner = comp & 7;
            double st[6] = {0, 0, 0, 0, 0, 0};
            if (strain) for (int k = 0; k < 6; ++k) st[k] = strain[(size_t)6 * u + k];
            const double ox = (corner & 4) ? (1 + st[0]) * nom * 0.5 : -(1 + st[3]) * nom * 0.5;
            const double oy = (corner & 2) ? (1 + st[1]) * nom * 0.5 : -(1 + st[4]) * nom * 0.5;
            const double oz = (corner & 1) ? (1 + st[2]) * nom * 0.5 : -(1 + st[5]) * nom * 0.5;
            double px = M.nom_pos[3 * u], py = M.nom_pos[3 * u + 1], pz = M.nom_pos[3 * u + 2];
            double qw = 1, qx = 0, qy = 0, qz = 0;
            if (pos) { px = pos[3 * u]; py = pos[3 * u + 1]; pz = pos[3 * u + 2]; qw = quat[4 * u]; qx = quat[4 * u + 1]; qy = quat[4 * u + 2]; qz = quat[4 * u + 3]; }
            // CQuat::RotateVec3D, Vec3D.h:293-299
            const double tw_ = ox * qx + oy * qy + oz * qz, tx = ox * qw - oy * qz + oz * qy, ty = ox * qz + oy * qw - oz * qx, tz = -ox * qy + oy * qx + oz * qw;
            ax += px + (qw * tx + qx * tw_ + qy * tz - qz * ty);
            ay += py + (qw * ty - qx * tz + qy * tw_ + qz * tx);
            az += pz + (qw * tz + qx * ty - qy * tx + qz * tw_);
            tw += 1.0;
        }
        const double inv = 1.0 / tw;
        const double v0x = M.vert_v0[(size_t)3 * i], v0y = M.vert_v0[(size_t)3 * i + 1], v0z = M.vert_v0[(size_t)3 * i + 2];
        vert[(size_t)3 * i] = v0x + (ax * inv - v0x); vert[(size_t)3 * i + 1] = v0y + (ay * inv - v0y); vert[(size_t)3 * i + 2] = v0z + (az * inv - v0z);
    }
    // corner codes (NNN..PPP) of the two triangles of faces +X,-X,+Y,-Y,+Z,-Z (LW/VX_MeshUtil.cpp:165-189)
    static const unsigned tri[6][2] = {{0x467u, 0x475u}, {0x032u, 0x013u}, {0x237u, 0x276u}, {0x051u, 0x045u}, {0x157u, 0x173u}, {0x064u, 0x026u}};
    double volume = 0.0;
    for (int v = 0; v < M.nvox; ++v)
        for (int d = 0; d < 6; ++d) {
            if (!(M.open_face[v] & (1u << d))) continue;
            for (int t = 0; t < 2; ++t) {
                const unsigned code = tri[d][t];
                const double* a = &vert[(size_t)3 * M.corner_vert[(size_t)v * 8 + ((code >> 8) & 7u)]];
                const double* b = &vert[(size_t)3 * M.corner_vert[(size_t)v * 8 + ((code >> 4) & 7u)]];
                const double* c = &vert[(size_t)3 * M.corner_vert[(size_t)v * 8 + (code & 7u)]];
                const double cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2], cz = a[0] * b[1] - a[1] * b[0];
                volume += (1.0 / 6.0) * (cx * c[0] + cy * c[1] + cz * c[2]);
            }
        }
    return volume;
}

}  // namespace

void compute_result(const RobotModel& M, const HostState& S, vxh_result* r)
{
    std::memset(r, 0, sizeof(*r));
    const VxaModel& X = M.vxa;
    r->status = S.status; r->steps = S.steps; r->nvox = M.nvox; r->nbond = M.nbond;
    r->dt = M.dt; r->cur_time = S.cur_time; r->col_rebuilds = S.rebuilds;
    const double lat = X.lattice_dim;
    // SS.CurCM after the last UpdateStats: sequential mass-weighted sum in voxel order (GetCM); zero before any step
    double cm[3] = {0, 0, 0};
    if (S.reduced) {                         // (k_results: the same sums, formed on the device in the same order)
        if (S.steps > 0 && M.nvox > 0) for (int k = 0; k < 3; ++k) cm[k] = S.red_cm[k];
    } else if (S.steps > 0 && M.nvox > 0) {
        double sx = 0, sy = 0, sz = 0, tm = 0;
        for (int v = 0; v < M.nvox; ++v) {
            const double m = M.vox_classes[M.vox_class[v]].mass;
            sx += S.pos[3 * v] * m; sy += S.pos[3 * v + 1] * m; sz += S.pos[3 * v + 2] * m; tm += m;
        }
        const double inv = 1.0 / tm;
        cm[0] = inv * sx; cm[1] = inv * sy; cm[2] = inv * sz;
    }
    for (int k = 0; k < 3; ++k) { r->cur_cm[k] = cm[k]; r->ini_cm[k] = S.ini_cm[k]; }
    if (X.variant == 0) {
        r->lifetime = S.cur_time - X.afterlife_time;
        const double fd = std::pow(std::pow(cm[0] - S.ini_cm[0], 2) + std::pow(cm[1] - S.ini_cm[1], 2), 0.5) / lat;
        double ant = 0.0, post = 100000.0, anty = 0.0, posty = 100000.0;
        int touching = 0, feet = 0;
        if (S.reduced) {
            // the extrema were taken on the device over the arguments; pow(., 0.5), the division by the lattice constant and the
            // comparisons with the start values (0 and 100000) are monotone, so they commute with max / min
            if (M.nvox > 0) {
                const double a = std::pow(S.d2max, 0.5) / lat, p = std::pow(S.d2min, 0.5) / lat;
                if (a > ant) ant = a;
                if (p < post) post = p;
                if (S.ymax > -1.0e299) { const double y1 = S.ymax / lat, y0 = S.ymin / lat; if (y1 > anty) anty = y1; if (y0 < posty) posty = y0; }
            }
            touching = S.touching; feet = S.feet;
        }
        for (int v = 0; v < (S.reduced ? 0 : M.nvox); ++v) {
            const VoxClass& C = M.vox_classes[M.vox_class[v]];
            const double x = S.pos[3 * v], y = S.pos[3 * v + 1], z = S.pos[3 * v + 2];
            const double d = std::pow(std::pow(x - S.ini_cm[0], 2) + std::pow(y - S.ini_cm[1], 2), 0.5) / lat;
            if (d > ant) ant = d;
            if (d < post) post = d;
            if (C.mat != 5) { const double yy = y / lat; if (yy > anty) anty = yy; if (yy < posty) posty = yy; }
            const double pen = 0.5 * S.scale[v] - z;
            if (pen > 0) { ++touching; if (C.mat == 6) ++feet; }
        }
        if (S.steps == 0) { ant = post = anty = posty = 0; touching = feet = 0; }   // SimState::Clear()
        r->final_dist = fd; r->norm_final_dist = fd; r->norm_frozen_dist = 0;
        r->norm_regime_dist = post - S.eol_post_y;
        r->final_dist_y = (cm[1] - S.ini_cm[1]) / lat;
        r->anterior_dist = ant; r->posterior_dist = post; r->anterior_y = anty; r->posterior_y = posty;
        r->end_of_life_posterior_y = S.eol_post_y; r->fall_adj_post_y = S.eol_post_y;
        r->num_non_feet_touching_floor = feet; r->num_touching_floor = touching;
    } else {
        r->lifetime = S.cur_time;
        const double inv = 1.0 / lat;
        const double dx = inv * (cm[0] - S.ini_cm[0]), dy = inv * (cm[1] - S.ini_cm[1]), dz = inv * (cm[2] - S.ini_cm[2]);
        r->norm_dist_x = (float)dx; r->norm_dist_y = (float)dy; r->norm_dist_z = (float)dz;   // float-typed in the reference
        r->norm_abs_disp = (float)std::sqrt(dx * dx + dy * dy + dz * dz);
        r->robot_volume_start = robot_volume(M, nullptr, nullptr, nullptr);
        if (S.steps == 0) r->robot_volume_end = r->robot_volume_start;
        else r->robot_volume_end = ((int)S.strain.size() == 6 * M.nvox) ? robot_volume(M, S.pos.data(), S.quat.data(), S.strain.data())
                                                                          : -1.0;
    }
}

namespace {
void tag(std::string& out, const char* name, double value)
{
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%g", value);
    out += "        <"; out += name; out += ">"; out += buf; out += "</"; out += name; out += ">\n";
}
}  // namespace

const std::vector<double>& empty_trace() { static const std::vector<double> none; return none; }

std::string result_xml(const RobotModel& M, const vxh_result& r, const std::vector<double>& cm_trace)
{
    std::string out = "<?xml version=\"1.0\" ?>\n<Voxelyze_Sim_Result Version=\"1.0\">\n    <Fitness>\n";
    if (M.vxa.variant == 0) {
        tag(out, "NormFinalDist", r.norm_final_dist - r.norm_frozen_dist);
        tag(out, "NormRegimeDist", r.norm_regime_dist);
        tag(out, "NormFrozenDist", r.norm_frozen_dist);
        tag(out, "FinalDist", r.final_dist);
        tag(out, "finalDistY", r.final_dist_y);
        tag(out, "AnteriorDist", r.anterior_dist);
        tag(out, "PosteriorDist", r.posterior_dist);
        tag(out, "AnteriorY", r.anterior_y);
        tag(out, "PosteriorY", r.posterior_y);
        tag(out, "EndOfLifePosteriorY", r.end_of_life_posterior_y);
        tag(out, "FallAdjPostY", r.fall_adj_post_y);
        tag(out, "NumNonFeetTouchingFloor", r.num_non_feet_touching_floor);
        tag(out, "NumTouchingFloor", r.num_touching_floor);
        tag(out, "Lifetime", r.lifetime);
        tag(out, "FoundNeedleInHaystack", 0);
        tag(out, "PushDist", 0);
    } else {
        tag(out, "VoxelNumber", r.nvox);
        tag(out, "normAbsoluteDisplacement", r.norm_abs_disp);
        tag(out, "normDistX", r.norm_dist_x);
        tag(out, "normDistY", r.norm_dist_y);
        tag(out, "normDistZ", r.norm_dist_z);
        // the hull volumes need qhull (--computeShapeDescriptors) and the shape complexity an external Python 2 script:
        // the reference prints -1 for the hull without the flag; -1 here for all four
        tag(out, "RobotVolumeStart", r.robot_volume_start);
        tag(out, "ConvexHullVolumeStart", -1);
        tag(out, "RobotVolumeEnd", r.robot_volume_end);
        tag(out, "ConvexHullVolumeEnd", -1);
        tag(out, "ShapeComplexityStart", -1);
        tag(out, "ShapeComplexityEnd", -1);
    }
    out += "    </Fitness>\n";
    if (M.vxa.variant == 0 && M.vxa.time_between_traces > 0 && M.vxa.save_traces) {     // VX_SimGA.cpp:170-184
        out += "    <CMTrace>\n";
        static const char* names[4] = {"Time", "TraceX", "TraceY", "TraceZ"};
        for (size_t i = 0; i + 3 < cm_trace.size(); i += 4) {
            out += "        <TraceStep>\n";
            for (int k = 0; k < 4; ++k) {
                char buf[64];
                std::snprintf(buf, sizeof(buf), "%g", cm_trace[i + k]);
                out += "            <"; out += names[k]; out += ">"; out += buf; out += "</"; out += names[k]; out += ">\n";
            }
            out += "        </TraceStep>\n";
        }
        out += "    </CMTrace>\n";
    }
    out += "</Voxelyze_Sim_Result>\n";
    return out;
}

}  // namespace vxh
