// Host-side model builder: VxaModel -> voxel/bond tables in the reference's ordering.
// Reference: CVX_Sim::Import / CreatePermBond / ResetSimulation / CalcMaxDt
// (evosoro/_voxcad/Voxelyze/VX_Sim.cpp:488-751,839-1007,1693-1727), CVX_Voxel::SetMaterial / CalcNearby
// (VX_Voxel.cpp:94-128,171-209), CVX_Bond::LinkVoxels / UpdateConstants (VX_Bond.cpp:65-173).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "vxa_reader.hpp"

namespace vxh {

// constants shared by every voxel of one (material, nominal size[, evolved stiffness]) combination
struct VoxClass {
    double mass, mass_inv, inertia, inertia_inv, first_moment;
    double c_lin;      // _2xSqMxExS
    double c_ang;      // _2xSqIxExSxSxS
    double E;          // Vox_E (floor stiffness, collision bonds)
    double mat_E;      // Elastic_Mod of the material model (axial stress)
    double stress_E;   // modulus actually multiplied with strain (land: mat_E; land_water: (float)Vox_E)
    double k_floor;    // 2*Vox_E*NominalSize
    double u_static, u_dynamic, cte, nom_size;
    int mat;
};

// constants of one internal bond, shared by every bond between the same two voxel classes
struct BondClass {
    double L, a1, a2, b1, b2, b3;
    double sq_a1m1, sq_a1m2, sq_a2i1, sq_a2i2, sq_b1m1, sq_b1m2, sq_b2fm1, sq_b2fm2, sq_b3i1, sq_b3i2;
    double stress_E1, stress_E2, area_sum;
    // UpdateBondStrain's series-spring split is linear in the strain: CurStress = stress_k * strain,
    // CurStrainV1/V2 = strain_a1/a2 * strain (see make_bond_class)
    double stress_k, strain_a1, strain_a2;
    int homogeneous;
};

struct RobotModel {
    VxaModel vxa;
    int nvox = 0, nbond = 0, nsurf = 0;
    // per voxel (simulation index order = ascending structure index of occupied cells)
    std::vector<int> struct_index;          // StoXIndexMap
    std::vector<int> nbr;                   // [nvox*6] voxel index in direction PX,NX,PY,NY,PZ,NZ or -1
    std::vector<int> vox_class;             // index into `vox_classes`
    std::vector<double> nom_pos;            // [nvox*3]
    std::vector<float> phase_offset, temp_amp_damp;
    // development (float members of CVXS_Voxel, VX_Sim.cpp:885-975); `development` = any of them differs from a plain robot
    bool development = false;
    std::vector<float> final_phase_offset, final_temp_amp_damp, initial_voxel_size, final_voxel_size, growth_time, start_growth_time;
    // per bond slot (voxel v, axis a) -> slot 3*v+a ; -1 class = no bond
    std::vector<int> bond_class;            // [nvox*3]
    // collisions
    std::vector<int> surf;                  // surface voxels, ascending
    std::vector<int> near_off, near_idx;    // CSR over ALL voxels: sorted voxel indices within N hops (self included)
    // land_water surface mesh (LW/VX_MeshUtil.cpp LinkSimVoxels :110-276; fluid drag and the RobotVolume tags): vertices = lattice
    // corners touched by 1..7 voxels; every exposed voxel face carries two triangles owned by that voxel
    int nmv = 0;
    std::vector<int> vert_comp;             // [nmv*8] voxel*8 + corner code (NNN..PPP = 0..7) or -1, in voxel order
    std::vector<double> vert_v0;            // [nmv*3] rest position (incl. the 1e-6 offset hack of GetXYZ)
    std::vector<int> corner_vert;           // [nvox*8] mesh vertex at each corner of the voxel or -1
    std::vector<unsigned char> open_face;   // [nvox] bit d set when face PX,NX,PY,NY,PZ,NZ is exposed
    // the facets in the reference's order (per voxel: faces +X,-X,+Y,-Y,+Z,-Z, two triangles each; LW/VX_MeshUtil.cpp:165-193)
    std::vector<int> facet_vox, facet_vert; // [nfacet] owning voxel ; [nfacet*3] mesh vertices
    std::vector<int> facet_first;           // [nvox] first facet of the voxel (its facets are contiguous)
    std::vector<unsigned char> facet_count; // [nvox] 0..12
    // local class tables (merged batch-wide by the engine)
    std::vector<VoxClass> vox_classes;
    std::vector<BondClass> bond_classes;
    double opt_dt = 0, dt = 0;
    long long planned_steps = 0;            // TimeStep() calls the reference main loop would make
};

RobotModel build_robot(const VxaModel& vxa);   // throws std::runtime_error

// A robot cut into tiles for the multi-workgroup kernel (kernels_tiled.hpp): a kx x ky x kz grid of boxes over the lattice,
// cut positions chosen so that the tiles hold equal numbers of voxels.  Every voxel is OWNED by exactly one tile; a tile's
// halo = the far ends of bonds that leave it; its bond list = every bond with at least one owned end, so a bond that crosses
// a boundary is listed by both tiles.  Local index = position in `own` followed by `halo` (both ascending voxel index).
struct TilePlan {
    int k = 0, kx = 1, ky = 1, kz = 1;
    std::vector<int> tile_of;               // [nvox]
    struct Tile {
        std::vector<int> own, halo;         // voxel indices
        std::vector<int> bond_v1;           // negative-end voxel of every listed bond (sorted by axis, then voxel)
        std::vector<int> bond_axis;
        std::vector<int> bond_entry;        // local negative end | local positive end << 10 | axis << 20
    };
    std::vector<Tile> tiles;
    int max_own = 0, max_local = 0, max_bonds = 0;
};
// k_request tiles or somewhat fewer (a k without a good factorisation into a grid is replaced by a smaller one with one)
TilePlan plan_tiles(const RobotModel& model, int k_request);

// number of TimeStep calls until StopConditionMet, replaying CurTime += dt in double precision
long long plan_steps(const VxaModel& vxa, double dt);

}  // namespace vxh
