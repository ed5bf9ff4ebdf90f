// .vxa reader of the product path: a small DOM XML parser + the tag -> parameter mapping of the reference
// readers (defaults included).  Reference: evosoro/_voxcad/Voxelyze/VX_Sim.cpp:177-354 (ReadVXA, ReadXML),
// VX_SimGA.cpp:216-230, VX_Environment.cpp:123-234, VX_Object.cpp:435-460,1064-1073,1344-1441,1733-1900,
// Utils/XML_Rip.h:72-79 (numbers go through atof/atoi, booleans through atoi != 0).
#pragma once
#include <memory>
#include <string>
#include <vector>

namespace vxh {

struct XmlNode {
    std::string name;
    std::string text;                                  // concatenated character data + CDATA of this element
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<std::unique_ptr<XmlNode>> children;
    const XmlNode* child(const char* tag) const;      // first direct child with that name or nullptr
    std::vector<const XmlNode*> children_named(const char* tag) const;
    const std::string* attr(const char* key) const;
};

// throws std::runtime_error on malformed input
std::unique_ptr<XmlNode> parse_xml(const char* data, size_t len);

struct Material {
    double E = 0, rho = 0, nu = 0, cte = 0, u_static = 0, u_dynamic = 0;
    int mat_model = 0;
};

// everything of a .vxa that reaches the time-stepper
struct VxaModel {
    int variant = 0;                    // 0 _voxcad, 1 _voxcad_land_water
    bool want_mesh = false;             // build the deformable surface mesh (every land_water robot: fluid drag, RobotVolume tags; a _voxcad robot of an engine
                                        // with option shape_descriptors: what voxelyzeMain/main.cpp:65-88,113-126 computes under --computeShapeDescriptors)
    // Simulator
    double dt_frac = 0.9, bond_damping_z = 0.1, col_damping_z = 1.0, slow_damping_z = 0.001;
    bool self_col_enabled = false;
    int col_system = 3;
    double collision_horizon = 3.0;
    int stop_type = 0;
    double stop_value = 0, afterlife_time = 0, midlife_freeze_time = 0, init_cm_time = 0;
    double min_temp_fact = 0.1;
    std::string fitness_file_name;
    std::string curvatures_tmp_file;       // <CurvaturesTmpFile> (land_water GA section): where the per-vertex angle excesses go
    // Environment
    bool grav_enabled = false, floor_enabled = false, temp_enabled = false, vary_temp_enabled = false;
    double grav_acc = -9.81, temp_amplitude = 0, temp_base = 25, temp_period = 0.1;
    double growth_amplitude = 0, min_growth_time = 0;
    bool sticky_floor = false;
    double time_between_traces = 0;       // <TimeBetweenTraces> (VX_Environment.cpp:215), _voxcad: a point of the CoM trace at most this often
    bool save_traces = false;             // <SaveTraces> (:214): the trace is printed into the result file
    bool fluid_env = false;
    double aggregate_drag_coef = 0;
    // VXC
    double lattice_dim = 0.001;
    std::vector<Material> palette;      // index 0 = the implicit "Erase" material
    int nx = 1, ny = 1, nz = 1;
    std::vector<unsigned char> structure;      // nx*ny*nz, x fastest
    bool has_phase_offset = false, has_temp_amp_damp = false, has_stiffness = false;
    std::vector<double> phase_offset, temp_amp_damp, stiffness;   // by occupied-voxel counter
    // _voxcad development layers (VX_Object.cpp:1910-2140), empty when the tag is absent
    bool has_final_phase_offset = false, has_final_temp_amp_damp = false, has_initial_voxel_size = false, has_final_voxel_size = false,
         has_growth_time = false, has_start_growth_time = false;
    std::vector<double> final_phase_offset, final_temp_amp_damp, initial_voxel_size, final_voxel_size, growth_time, start_growth_time;
    std::vector<std::string> unsupported;      // features present in the file that the engine does not model
};

// throws std::runtime_error
VxaModel read_vxa(const char* data, size_t len, int variant);

}  // namespace vxh
