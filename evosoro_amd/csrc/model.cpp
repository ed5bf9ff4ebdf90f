#include "model.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <utility>

namespace vxh {

namespace {

bool same_bits(double a, double b) { return std::memcmp(&a, &b, sizeof(double)) == 0; }

// CVX_Voxel::SetMaterial (VX_Voxel.cpp:94-128); Vox_E may be an evolved per-voxel value (App. A.8 of SURVEY.md)
VoxClass make_vox_class(const VxaModel& m, int mat, double size, bool evolved, double evolved_E)
{
    const Material& pm = m.palette[mat];
    VoxClass c;
    std::memset(&c, 0, sizeof(c));   // classes are interned bytewise: keep padding deterministic
    c.mat = mat;
    c.nom_size = size;
    double volume = size * size * size;
    c.mass = volume * pm.rho;
    c.inertia = c.mass * (size * size) / 6;
    c.first_moment = c.mass * size / 2;
    c.mat_E = pm.E;
    c.E = pm.E;
    c.u_static = pm.u_static;
    c.u_dynamic = pm.u_dynamic;
    c.cte = pm.cte;
    c.mass_inv = 1 / c.mass;
    c.inertia_inv = 1 / c.inertia;
    c.c_lin = 2 * std::sqrt(c.mass * c.E * size);
    c.c_ang = 2 * std::sqrt(c.inertia * c.E * size * size * size);
    if (evolved) {
        c.E = evolved_E;
        if (m.variant == 1) {  // LW SetEMod refreshes the damping terms (LW/VX_Voxel.h:61); _voxcad does not (VX_Voxel.h:61)
            c.c_lin = 2 * std::sqrt(c.mass * c.E * size);
            c.c_ang = 2 * std::sqrt(c.inertia * c.E * size * size * size);
        }
    }
    if (m.variant == 1) {  // LW/VX_Object.cpp:1474: float true_Elastic_Mod = (voxelEmod > 0) ? voxelEmod : Elastic_Mod
        float e = (float)c.E;
        float t = (e > 0) ? e : (float)c.mat_E;
        c.stress_E = (double)t;
    } else {
        c.stress_E = c.mat_E;
    }
    c.k_floor = 2 * c.E * size;
    return c;
}

// CVX_Bond::LinkVoxels + UpdateConstants (VX_Bond.cpp:65-173).  E1/E2 are the moduli the voxels carry at
// bond-creation time: the material's in _voxcad (evolved stiffness is applied later), the evolved one in LW.
BondClass make_bond_class(const VxaModel& m, const VoxClass& c1, double E1, const VoxClass& c2, double E2)
{
    BondClass b;
    std::memset(&b, 0, sizeof(b));
    b.homogeneous = (c1.mat == c2.mat);
    if (m.variant == 1) b.homogeneous = b.homogeneous && (E1 == E2);
    double u1 = m.palette[c1.mat].nu, u2 = m.palette[c2.mat].nu;
    double E = (E1 * E2 / (E1 + E2)) * 2;
    double u = (u1 == 0 && u2 == 0) ? 0 : (u1 * u2 / (u1 + u2)) * 2;
    double size = (c1.nom_size + c2.nom_size) * 0.5;
    double Lx = size, Ly = size, Lz = size;
    b.L = Lx;
    double G = E / (2 * (1 + u));
    double A = Ly * Lz;
    double Iy = Lz * Ly * Ly * Ly / 12;
    double J = Ly * Lz * (Ly * Ly + Lz * Lz) / 12;
    b.a1 = E * A / Lx;
    b.a2 = G * J / Lx;
    b.b1 = 12 * E * Iy / (Lx * Lx * Lx);   // b1y == b1z, b2y == b2z, b3y == b3z for cubic voxels
    b.b2 = 6 * E * Iy / (Lx * Lx);
    b.b3 = 2 * E * Iy / Lx;
    b.sq_a1m1 = 2.0 * std::sqrt(b.a1 * c1.mass);          b.sq_a1m2 = 2.0 * std::sqrt(b.a1 * c2.mass);
    b.sq_a2i1 = 2.0 * std::sqrt(b.a2 * c1.inertia);       b.sq_a2i2 = 2.0 * std::sqrt(b.a2 * c2.inertia);
    b.sq_b1m1 = 2.0 * std::sqrt(b.b1 * c1.mass);          b.sq_b1m2 = 2.0 * std::sqrt(b.b1 * c2.mass);
    b.sq_b2fm1 = 2.0 * std::sqrt(b.b2 * c1.first_moment); b.sq_b2fm2 = 2.0 * std::sqrt(b.b2 * c2.first_moment);
    b.sq_b3i1 = 2.0 * std::sqrt(b.b3 * c1.inertia);       b.sq_b3i2 = 2.0 * std::sqrt(b.b3 * c2.inertia);
    b.stress_E1 = c1.stress_E;
    b.stress_E2 = c2.stress_E;
    // Axial stress (UpdateBondStrain, VXS_BondInternal.cpp:189-307, linear materials).  Same material: stress = E * strain.
    // Otherwise the reference splits the strain between the two half-bonds with up to three fixed-point iterations
    // (tolerance 5e-4).  Every iterate is proportional to the strain and the loop condition compares two quantities that
    // are both proportional to |strain|, so the whole procedure is a multiplication by three constants of the bond class;
    // they are obtained by running the reference's loop once, here, on a unit strain.
    b.stress_k = b.stress_E1; b.strain_a1 = b.strain_a2 = 1.0;
    if (!b.homogeneous) {
        double e1 = 1.0, e2 = 1.0, t1 = b.stress_E1 * e1, t2 = b.stress_E2 * e2;
        double diff = std::fabs(t1 - t2), sum = std::fabs(t1 + t2);
        for (int it = 0; it < 3 && diff > sum * .0005; ++it) {
            e1 = 2 * t2 / (t1 + t2) * e1;
            e2 = 2 * t1 / (t1 + t2) * e2;
            t1 = b.stress_E1 * e1; t2 = b.stress_E2 * e2;
            diff = std::fabs(t1 - t2); sum = std::fabs(t1 + t2);
        }
        b.stress_k = (t1 + t2) / 2; b.strain_a1 = e1; b.strain_a2 = e2;
    }
    b.area_sum = Ly * Lz + Ly * Lz;   // CSArea1 + CSArea2 (VXS_Bond.cpp:75, VXS_Voxel.cpp:625-631)
    return b;
}

template <class T>
int intern(std::vector<T>& table, const T& value)
{
    for (size_t i = 0; i < table.size(); ++i)
        if (std::memcmp(&table[i], &value, sizeof(T)) == 0) return (int)i;
    table.push_back(value);
    return (int)table.size() - 1;
}

bool stop_met(const VxaModel& m, double t, long long steps)
{
    if (m.variant == 0 && t <= m.init_cm_time) return false;           // VX_Sim.cpp:1402
    switch (m.stop_type) {
    case 1: return steps > (long long)(int)(m.stop_value + 0.5);
    case 2: return m.variant == 0 ? t > (m.stop_value + m.afterlife_time) : t > m.stop_value;
    case 3: return t > m.temp_period * m.stop_value;
    default: return false;
    }
}

}  // namespace

long long plan_steps(const VxaModel& m, double dt)
{
    if (m.stop_type == 0) throw std::runtime_error("StopConditionType 0 (none) never terminates");
    if (!(dt > 0)) throw std::runtime_error("non-positive time step");
    double t = 0;
    long long steps = 0;
    const long long cap = 2000000000LL;
    while (!stop_met(m, t, steps)) {
        t += dt;
        if (++steps > cap) throw std::runtime_error("stop condition needs more than 2e9 steps");
    }
    return steps;
}

RobotModel build_robot(const VxaModel& vxa)
{
    RobotModel r;
    r.vxa = vxa;
    const VxaModel& m = r.vxa;
    const int nx = m.nx, ny = m.ny, nz = m.nz;
    const size_t ncell = (size_t)nx * ny * nz;
    std::vector<int> x2s(ncell, -1);
    for (size_t i = 0; i < ncell; ++i)
        if (m.structure[i]) { x2s[i] = r.nvox++; r.struct_index.push_back((int)i); }
    if (r.nvox == 0) return r;
    if (m.has_phase_offset && (int)m.phase_offset.size() < r.nvox) throw std::runtime_error("<PhaseOffset> has fewer values than voxels");
    if (m.has_stiffness && (int)m.stiffness.size() < r.nvox) throw std::runtime_error("<Stiffness> has fewer values than voxels");
    if (m.has_temp_amp_damp && (int)m.temp_amp_damp.size() < r.nvox) throw std::runtime_error("<TempAmpDamp> has fewer values than voxels");
    for (const auto& layer : {std::make_pair(m.has_final_phase_offset, &m.final_phase_offset), std::make_pair(m.has_final_temp_amp_damp, &m.final_temp_amp_damp),
                              std::make_pair(m.has_initial_voxel_size, &m.initial_voxel_size), std::make_pair(m.has_final_voxel_size, &m.final_voxel_size),
                              std::make_pair(m.has_growth_time, &m.growth_time), std::make_pair(m.has_start_growth_time, &m.start_growth_time)})
        if (layer.first && (int)layer.second->size() < r.nvox) throw std::runtime_error("a development layer has fewer values than voxels");

    const double size = m.lattice_dim * 1.0;   // GetLatDimEnv().x with X_Dim_Adj == 1 (VX_Object.h:377, VX_Sim.cpp:528)
    r.nbr.assign((size_t)r.nvox * 6, -1);
    r.vox_class.resize(r.nvox);
    r.nom_pos.resize((size_t)r.nvox * 3);
    r.phase_offset.assign(r.nvox, 0.0f);
    r.temp_amp_damp.assign(r.nvox, 1.0f);
    r.development = m.variant == 0 && (m.has_final_phase_offset || m.has_final_temp_amp_damp || m.has_initial_voxel_size || m.has_final_voxel_size ||
                                        m.has_growth_time || m.has_start_growth_time || m.midlife_freeze_time > 0);
    for (auto* arr : {&r.final_phase_offset, &r.final_temp_amp_damp, &r.initial_voxel_size, &r.final_voxel_size, &r.growth_time, &r.start_growth_time})
        arr->assign(r.nvox, 0.0f);
    r.bond_class.assign((size_t)r.nvox * 3, -1);
    std::vector<double> link_E(r.nvox);     // Vox_E at LinkVoxels time
    for (int v = 0; v < r.nvox; ++v) {
        int i = r.struct_index[v];
        int iz = i / (nx * ny), iy = (i - iz * nx * ny) / nx, ix = i - iz * nx * ny - iy * nx;
        int mat = m.structure[i];
        const Material& pm = m.palette[mat];
        if (!(pm.rho > 0) || !(pm.E > 0)) throw std::runtime_error("material with non-positive density or modulus");
        bool evolved = m.has_stiffness;
        VoxClass c = make_vox_class(m, mat, size, evolved, evolved ? m.stiffness[v] : 0.0);
        r.vox_class[v] = intern(r.vox_classes, c);
        link_E[v] = (m.variant == 1 && evolved) ? m.stiffness[v] : pm.E;
        r.nom_pos[3 * v + 0] = size * (ix + 0.5);
        r.nom_pos[3 * v + 1] = size * (iy + 0.5);
        r.nom_pos[3 * v + 2] = size * (iz + 0.5);
        if (m.has_phase_offset) r.phase_offset[v] = (float)m.phase_offset[v];
        if (m.has_temp_amp_damp) r.temp_amp_damp[v] = (float)m.temp_amp_damp[v];
        if (m.variant == 0) {
            // development parameters, VX_Sim.cpp:885-975: every member is a float, every right-hand side a double
            const float bound = (float)m.stop_value;      // onsetBound = terminationBound (OnsetRelative/TerminationRelative are refused)
            r.final_phase_offset[v] = m.has_final_phase_offset ? (float)m.final_phase_offset[v] : 0.0f;
            r.final_temp_amp_damp[v] = m.has_final_temp_amp_damp ? (float)m.final_temp_amp_damp[v] : 1.0f;
            auto size_of = [&](double from_vxa) {
                double tf = 1 + (m.growth_amplitude * from_vxa);
                double eff = (tf < m.min_temp_fact) ? m.min_temp_fact : tf;
                return (float)(eff * size);
            };
            r.initial_voxel_size[v] = m.has_initial_voxel_size ? size_of(m.initial_voxel_size[v]) : (float)size;
            r.final_voxel_size[v] = m.has_final_voxel_size ? size_of(m.final_voxel_size[v]) : r.initial_voxel_size[v];
            if (m.has_start_growth_time) {
                double t0 = m.start_growth_time[v] * (bound - m.init_cm_time) + m.init_cm_time;
                r.start_growth_time[v] = (t0 >= bound - m.min_growth_time) ? (float)(bound - m.min_growth_time) : (float)t0;
            } else if (m.has_final_voxel_size || m.has_growth_time) r.start_growth_time[v] = (float)m.init_cm_time;
            else r.start_growth_time[v] = (float)(m.stop_value - m.midlife_freeze_time);
            const float span = bound - r.start_growth_time[v];            // float - float
            if (m.has_growth_time) {
                double g = m.growth_time[v] * (span - m.midlife_freeze_time);
                r.growth_time[v] = (float)((g <= m.min_growth_time) ? m.min_growth_time : g);
            } else if (m.has_final_voxel_size) r.growth_time[v] = (float)(span - m.midlife_freeze_time);
            else r.growth_time[v] = (float)m.min_growth_time;
        }
        const int step[3] = {1, nx, nx * ny};
        const int coord[3] = {ix, iy, iz}, lim[3] = {nx, ny, nz};
        for (int a = 0; a < 3; ++a) {
            if (coord[a] + 1 < lim[a] && m.structure[i + step[a]]) r.nbr[(size_t)v * 6 + 2 * a] = x2s[i + step[a]];
            if (coord[a] - 1 >= 0 && m.structure[i - step[a]]) r.nbr[(size_t)v * 6 + 2 * a + 1] = x2s[i - step[a]];
        }
    }
    // permanent bonds in reference order: for each voxel +X, +Y, +Z (VX_Sim.cpp:623-641); CalcMaxDt alongside
    double max_freq2 = 0;
    for (int v = 0; v < r.nvox; ++v)
        for (int a = 0; a < 3; ++a) {
            int o = r.nbr[(size_t)v * 6 + 2 * a];
            if (o < 0) continue;
            const VoxClass& c1 = r.vox_classes[r.vox_class[v]];
            const VoxClass& c2 = r.vox_classes[r.vox_class[o]];
            BondClass b = make_bond_class(m, c1, link_E[v], c2, link_E[o]);
            r.bond_class[(size_t)v * 3 + a] = intern(r.bond_classes, b);
            r.nbond++;
            if (b.a1 / c1.mass > max_freq2) max_freq2 = b.a1 / c1.mass;
            if (b.a1 / c2.mass > max_freq2) max_freq2 = b.a1 / c2.mass;
        }
    if (r.nbond == 0)
        for (int v = 0; v < r.nvox; ++v) {
            const VoxClass& c = r.vox_classes[r.vox_class[v]];
            if (c.E / c.mass > max_freq2) max_freq2 = c.E / c.mass;
        }
    r.opt_dt = 1.0 / (std::sqrt(max_freq2) * 2 * (double)3.1415926);   // VX_Sim.cpp:1724-1725
    r.dt = m.dt_frac * r.opt_dt;
    r.planned_steps = plan_steps(m, r.dt);

    // surface voxels + CalcNearby exclusion lists (VX_Sim.cpp:649-659)
    const int hops = (int)(m.collision_horizon * 1.5);
    std::vector<int> scratch;
    std::vector<unsigned char> mark(r.nvox, 0);
    r.near_off.assign(r.nvox + 1, 0);
    for (int v = 0; v < r.nvox; ++v) {
        bool surface = false;
        for (int d = 0; d < 6; ++d) if (r.nbr[(size_t)v * 6 + d] < 0) surface = true;
        if (surface) r.surf.push_back(v);
        if (m.self_col_enabled && surface) {   // only surface voxels are ever tested (VX_Sim.cpp:2369-2387)
            scratch.clear();
            scratch.push_back(v); mark[v] = 1;
            size_t start = 0, stop = 1;
            for (int h = 0; h < hops; ++h) {
                for (size_t j = start; j < stop; ++j)
                    for (int d = 0; d < 6; ++d) {
                        int o = r.nbr[(size_t)scratch[j] * 6 + d];
                        if (o >= 0 && !mark[o]) { mark[o] = 1; scratch.push_back(o); }
                    }
                start = stop; stop = scratch.size();
            }
            for (int s : scratch) mark[s] = 0;
            std::sort(scratch.begin(), scratch.end());
            r.near_idx.insert(r.near_idx.end(), scratch.begin(), scratch.end());
        }
        r.near_off[v + 1] = (int)r.near_idx.size();
    }
    r.nsurf = (int)r.surf.size();

    if (m.want_mesh) {      // deformable surface mesh (fluid drag; RobotVolume tags of every land_water robot; _voxcad: --computeShapeDescriptors)
        const int tx = nx + 1, ty = ny + 1, tz = nz + 1;
        std::vector<std::vector<int>> comps((size_t)tx * ty * tz);
        auto d3 = [&](int X, int Y, int Z) { return (size_t)Z * tx * ty + (size_t)Y * tx + X; };
        static const int cdx[8] = {0, 0, 0, 0, 1, 1, 1, 1}, cdy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, cdz[8] = {0, 1, 0, 1, 0, 1, 0, 1};
        r.open_face.assign(r.nvox, 0);
        for (int v = 0; v < r.nvox; ++v) {
            int i = r.struct_index[v];
            int iz = i / (nx * ny), iy = (i - iz * nx * ny) / nx, ix = i - iz * nx * ny - iy * nx;
            for (int c = 0; c < 8; ++c) comps[d3(ix + cdx[c], iy + cdy[c], iz + cdz[c])].push_back(v * 8 + c);
            for (int d = 0; d < 6; ++d) if (r.nbr[(size_t)v * 6 + d] < 0) r.open_face[v] |= (unsigned char)(1u << d);
        }
        std::vector<int> map(comps.size(), -1);
        const double lat = m.lattice_dim, half = (1.0 / 2) * lat, eps = 0.000001;
        for (int k = 0; k < tz; ++k) for (int j = 0; j < ty; ++j) for (int i = 0; i < tx; ++i) {
            const auto& c = comps[d3(i, j, k)];
            if (c.empty() || c.size() == 8) continue;
            map[d3(i, j, k)] = r.nmv++;
            for (int q = 0; q < 8; ++q) r.vert_comp.push_back(q < (int)c.size() ? c[q] : -1);
            r.vert_v0.push_back(lat * (1.0 * (0.5 + i) + eps) - half);   // LW/VX_Object.cpp:521-540, LW/VX_MeshUtil.cpp:231-240
            r.vert_v0.push_back(lat * (1.0 * (0.5 + j) + eps) - half);
            r.vert_v0.push_back(lat * 1.0 * (0.5 + k) - half);
        }
        r.corner_vert.assign((size_t)r.nvox * 8, -1);
        for (int v = 0; v < r.nvox; ++v) {
            int i = r.struct_index[v];
            int iz = i / (nx * ny), iy = (i - iz * nx * ny) / nx, ix = i - iz * nx * ny - iy * nx;
            for (int c = 0; c < 8; ++c) r.corner_vert[(size_t)v * 8 + c] = map[d3(ix + cdx[c], iy + cdy[c], iz + cdz[c])];
        }
        // corner codes (NNN..PPP = 0..7) of the two triangles of faces +X,-X,+Y,-Y,+Z,-Z (LW/VX_MeshUtil.cpp:165-189)
        static const unsigned tri[6][2] = {{0x467u, 0x475u}, {0x032u, 0x013u}, {0x237u, 0x276u}, {0x051u, 0x045u}, {0x157u, 0x173u}, {0x064u, 0x026u}};
        r.facet_first.assign(r.nvox, 0);
        r.facet_count.assign(r.nvox, 0);
        for (int v = 0; v < r.nvox; ++v) {
            r.facet_first[v] = (int)r.facet_vox.size();
            for (int d = 0; d < 6; ++d) {
                if (!(r.open_face[v] & (1u << d))) continue;
                for (int t = 0; t < 2; ++t) {
                    const unsigned code = tri[d][t];
                    r.facet_vox.push_back(v);
                    for (int k = 0; k < 3; ++k) r.facet_vert.push_back(r.corner_vert[(size_t)v * 8 + ((code >> (8 - 4 * k)) & 7u)]);
                }
            }
            r.facet_count[v] = (unsigned char)((int)r.facet_vox.size() - r.facet_first[v]);
        }
    }
    (void)same_bits;
    return r;
}

// ------------------------------------------------------------------------------------------------ tiling (kernels_tiled.hpp)
namespace {

// cut `vox` (voxel indices) into `parts` groups of equal size (+-1) along lattice axis `axis` (0 = x); ties keep voxel order
std::vector<std::vector<int>> cut_along(const std::vector<int>& vox, const std::vector<int>& coord3, int axis, int parts)
{
    std::vector<int> sorted = vox;
    std::stable_sort(sorted.begin(), sorted.end(), [&](int a, int b) { return coord3[3 * a + axis] < coord3[3 * b + axis]; });
    std::vector<std::vector<int>> out(parts);
    const size_t n = sorted.size();
    for (int p = 0; p < parts; ++p) {
        const size_t lo = n * (size_t)p / parts, hi = n * (size_t)(p + 1) / parts;
        out[p].assign(sorted.begin() + lo, sorted.begin() + hi);
    }
    return out;
}

}  // namespace

TilePlan plan_tiles(const RobotModel& M, int k_request)
{
    TilePlan P;
    const int n = M.nvox;
    if (n == 0) return P;
    const int nx = M.vxa.nx, ny = M.vxa.ny;
    std::vector<int> coord3((size_t)n * 3);
    int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-1, -1, -1};
    for (int v = 0; v < n; ++v) {
        const int i = M.struct_index[v];
        const int iz = i / (nx * ny), iy = (i - iz * nx * ny) / nx, ix = i - iz * nx * ny - iy * nx;
        const int c[3] = {ix, iy, iz};
        for (int a = 0; a < 3; ++a) { coord3[3 * (size_t)v + a] = c[a]; lo[a] = std::min(lo[a], c[a]); hi[a] = std::max(hi[a], c[a]); }
    }
    const double ext[3] = {double(hi[0] - lo[0] + 1), double(hi[1] - lo[1] + 1), double(hi[2] - lo[2] + 1)};
    const double fill = n / (ext[0] * ext[1] * ext[2]);
    // the grid: among k' in [0.8 k, k] and the factorisations k' = a b c (a <= ext x, ...), the one whose typical tile lists
    // the fewest bonds (3 per voxel inside + 2 per boundary face voxel)
    int kmax = std::max(1, std::min(k_request, n));
    double best = 1e300;
    for (int k = kmax; k >= std::max(1, (kmax * 4 + 4) / 5); --k)
        for (int a = 1; a <= k; ++a) {
            if (k % a || a > ext[0]) continue;
            for (int b = 1; b <= k / a; ++b) {
                if ((k / a) % b || b > ext[1]) continue;
                const int c = k / a / b;
                if (c > ext[2]) continue;
                const double sx = ext[0] / a, sy = ext[1] / b, sz = ext[2] / c;
                const double faces = (a > 1 ? 2 : 0) * sy * sz + (b > 1 ? 2 : 0) * sx * sz + (c > 1 ? 2 : 0) * sx * sy;
                const double bonds = 3.0 * n / k + faces * fill;
                if (bonds < best - 1e-9) { best = bonds; P.k = k; P.kx = a; P.ky = b; P.kz = c; }
            }
        }
    if (P.k == 0) { P.k = P.kx = P.ky = P.kz = 1; }
    P.tile_of.assign(n, 0);
    std::vector<int> all(n);
    for (int v = 0; v < n; ++v) all[v] = v;
    int t = 0;
    for (auto& gx : cut_along(all, coord3, 0, P.kx))
        for (auto& gy : cut_along(gx, coord3, 1, P.ky))
            for (auto& gz : cut_along(gy, coord3, 2, P.kz)) {
                for (int v : gz) P.tile_of[v] = t;
                ++t;
            }
    P.tiles.assign(P.k, TilePlan::Tile());
    for (int v = 0; v < n; ++v) P.tiles[P.tile_of[v]].own.push_back(v);       // ascending
    std::vector<int> local(n, -1);
    for (int ti = 0; ti < P.k; ++ti) {
        TilePlan::Tile& T = P.tiles[ti];
        for (int v : T.own)
            for (int d = 0; d < 6; ++d) {
                const int o = M.nbr[(size_t)v * 6 + d];
                if (o >= 0 && P.tile_of[o] != ti) T.halo.push_back(o);
            }
        std::sort(T.halo.begin(), T.halo.end());
        T.halo.erase(std::unique(T.halo.begin(), T.halo.end()), T.halo.end());
        for (size_t i = 0; i < T.own.size(); ++i) local[T.own[i]] = (int)i;
        for (size_t i = 0; i < T.halo.size(); ++i) local[T.halo[i]] = (int)(T.own.size() + i);
        for (int a = 0; a < 3; ++a) {
            // bonds of axis a with an owned end, by negative-end voxel: the owned voxels' +a bonds and the halo voxels' +a bonds
            // that end in an owned voxel
            std::vector<int> v1s;
            for (int v : T.own) if (M.nbr[(size_t)v * 6 + 2 * a] >= 0) v1s.push_back(v);
            for (int v : T.halo) { const int o = M.nbr[(size_t)v * 6 + 2 * a]; if (o >= 0 && P.tile_of[o] == ti) v1s.push_back(v); }
            std::sort(v1s.begin(), v1s.end());
            for (int v1 : v1s) {
                const int v2 = M.nbr[(size_t)v1 * 6 + 2 * a];
                T.bond_v1.push_back(v1); T.bond_axis.push_back(a);
                T.bond_entry.push_back((int)((unsigned)local[v1] | ((unsigned)local[v2] << 10) | ((unsigned)a << 20)));
            }
        }
        for (int v : T.own) local[v] = -1;
        for (int v : T.halo) local[v] = -1;
        P.max_own = std::max(P.max_own, (int)T.own.size());
        P.max_local = std::max(P.max_local, (int)(T.own.size() + T.halo.size()));
        P.max_bonds = std::max(P.max_bonds, (int)T.bond_v1.size());
    }
    return P;
}

}  // namespace vxh
