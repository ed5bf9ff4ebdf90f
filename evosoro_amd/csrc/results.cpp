// Fitness metrics and result XML.  Reference: CVX_SimGA::WriteResultFile (evosoro/_voxcad/Voxelyze/VX_SimGA.cpp:33-203;
// land_water: evosoro/_voxcad_land_water/Voxelyze/VX_SimGA.cpp:33-77), the SS.* values of CVX_Sim::UpdateStats
// (VX_Sim.cpp:1518-1535) with their helpers GetCM / getAnteriorDist / getPosteriorY / GetNumTouchingFloor
// (VX_Sim.cpp:2415-2441,2584-2712).  Numbers are printed like `ostream << double` (6 significant digits,
// Utils/XML_Rip.h:57) in TinyXML's layout (4 blanks per level).
#include <cmath>
#include <cstdio>
#include <cstring>

#include "engine.hpp"

namespace vxh {

void compute_result(const RobotModel& M, const HostState& S, vxh_result* r)
{
    std::memset(r, 0, sizeof(*r));
    const VxaModel& X = M.vxa;
    r->status = S.status; r->steps = S.steps; r->nvox = M.nvox; r->nbond = M.nbond;
    r->dt = M.dt; r->cur_time = S.cur_time; r->col_rebuilds = S.rebuilds;
    const double lat = X.lattice_dim;
    // SS.CurCM after the last UpdateStats: sequential mass-weighted sum in voxel order (GetCM); zero before any step
    double cm[3] = {0, 0, 0};
    if (S.steps > 0 && M.nvox > 0) {
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
        for (int v = 0; v < M.nvox; ++v) {
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

std::string result_xml(const RobotModel& M, const vxh_result& r)
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
        // shape descriptors (mesh volume / qhull) are outside the hot path: the reference prints -1 for the hull
        // values when qhull is unavailable; volumes are not computed here either
        tag(out, "RobotVolumeStart", -1);
        tag(out, "ConvexHullVolumeStart", -1);
        tag(out, "RobotVolumeEnd", -1);
        tag(out, "ConvexHullVolumeEnd", -1);
        tag(out, "ShapeComplexityStart", -1);
        tag(out, "ShapeComplexityEnd", -1);
    }
    out += "    </Fitness>\n</Voxelyze_Sim_Result>\n";
    return out;
}

}  // namespace vxh
