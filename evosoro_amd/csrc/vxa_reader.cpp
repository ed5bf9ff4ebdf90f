#include "vxa_reader.hpp"

#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace vxh {

// ------------------------------------------------------------------------------------------- XML DOM
const XmlNode* XmlNode::child(const char* tag) const
{
    for (const auto& c : children)
        if (c->name == tag) return c.get();
    return nullptr;
}
std::vector<const XmlNode*> XmlNode::children_named(const char* tag) const
{
    std::vector<const XmlNode*> out;
    for (const auto& c : children)
        if (c->name == tag) out.push_back(c.get());
    return out;
}
const std::string* XmlNode::attr(const char* key) const
{
    for (const auto& a : attrs)
        if (a.first == key) return &a.second;
    return nullptr;
}

namespace {

struct Cursor {
    const char* p;
    const char* end;
    bool starts(const char* s) const
    {
        size_t n = std::strlen(s);
        return size_t(end - p) >= n && std::memcmp(p, s, n) == 0;
    }
    void skip_ws()
    {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) ++p;
    }
    void skip_until(const char* s)
    {
        size_t n = std::strlen(s);
        while (p < end && !starts(s)) ++p;
        if (p >= end) throw std::runtime_error(std::string("xml: unterminated construct, expected ") + s);
        p += n;
    }
};

void append_unescaped(std::string& out, const char* b, const char* e)
{
    while (b < e) {
        if (*b == '&') {
            struct { const char* ent; char ch; } tab[] = {{"&lt;", '<'}, {"&gt;", '>'}, {"&amp;", '&'}, {"&quot;", '"'}, {"&apos;", '\''}};
            bool hit = false;
            for (auto& t : tab) {
                size_t n = std::strlen(t.ent);
                if (size_t(e - b) >= n && std::memcmp(b, t.ent, n) == 0) { out.push_back(t.ch); b += n; hit = true; break; }
            }
            if (hit) continue;
        }
        out.push_back(*b++);
    }
}

bool name_char(char c) { return !(c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '>' || c == '/' || c == '='); }

std::unique_ptr<XmlNode> parse_element(Cursor& c, int depth)
{
    if (depth > 64) throw std::runtime_error("xml: nesting too deep");
    // c.p points at '<' of a start tag
    ++c.p;
    auto node = std::unique_ptr<XmlNode>(new XmlNode);
    const char* b = c.p;
    while (c.p < c.end && name_char(*c.p)) ++c.p;
    node->name.assign(b, c.p);
    if (node->name.empty()) throw std::runtime_error("xml: empty tag name");
    for (;;) {  // attributes
        c.skip_ws();
        if (c.p >= c.end) throw std::runtime_error("xml: unterminated start tag");
        if (*c.p == '/') {
            if (c.p + 1 < c.end && c.p[1] == '>') { c.p += 2; return node; }
            throw std::runtime_error("xml: stray '/'");
        }
        if (*c.p == '>') { ++c.p; break; }
        const char* kb = c.p;
        while (c.p < c.end && name_char(*c.p)) ++c.p;
        std::string key(kb, c.p);
        c.skip_ws();
        if (c.p >= c.end || *c.p != '=') throw std::runtime_error("xml: attribute without value");
        ++c.p;
        c.skip_ws();
        if (c.p >= c.end || (*c.p != '"' && *c.p != '\'')) throw std::runtime_error("xml: unquoted attribute");
        char q = *c.p++;
        const char* vb = c.p;
        while (c.p < c.end && *c.p != q) ++c.p;
        if (c.p >= c.end) throw std::runtime_error("xml: unterminated attribute");
        std::string val;
        append_unescaped(val, vb, c.p);
        ++c.p;
        node->attrs.emplace_back(std::move(key), std::move(val));
    }
    for (;;) {  // content
        const char* tb = c.p;
        while (c.p < c.end && *c.p != '<') ++c.p;
        append_unescaped(node->text, tb, c.p);
        if (c.p >= c.end) throw std::runtime_error("xml: missing end tag for <" + node->name + ">");
        if (c.starts("<!--")) { c.skip_until("-->"); continue; }
        if (c.starts("<![CDATA[")) {
            c.p += 9;
            const char* cb = c.p;
            while (c.p < c.end && !c.starts("]]>")) ++c.p;
            if (c.p >= c.end) throw std::runtime_error("xml: unterminated CDATA");
            node->text.append(cb, c.p);
            c.p += 3;
            continue;
        }
        if (c.starts("<?")) { c.skip_until("?>"); continue; }
        if (c.starts("</")) {
            c.p += 2;
            const char* eb = c.p;
            while (c.p < c.end && name_char(*c.p)) ++c.p;
            if (std::string(eb, c.p) != node->name) throw std::runtime_error("xml: mismatched end tag for <" + node->name + ">");
            c.skip_ws();
            if (c.p >= c.end || *c.p != '>') throw std::runtime_error("xml: malformed end tag");
            ++c.p;
            return node;
        }
        node->children.push_back(parse_element(c, depth + 1));
    }
}

// atof / atoi semantics of XML_Rip.h:72-79
double num(const XmlNode* parent, const char* tag, double dflt)
{
    const XmlNode* n = parent ? parent->child(tag) : nullptr;
    if (!n || n->text.empty()) return dflt;
    return std::atof(n->text.c_str());
}
int inum(const XmlNode* parent, const char* tag, int dflt)
{
    const XmlNode* n = parent ? parent->child(tag) : nullptr;
    if (!n || n->text.empty()) return dflt;
    return std::atoi(n->text.c_str());
}
bool flag(const XmlNode* parent, const char* tag, bool dflt)
{
    const XmlNode* n = parent ? parent->child(tag) : nullptr;
    if (!n || n->text.empty()) return dflt;
    return std::atoi(n->text.c_str()) != 0;
}
bool has(const XmlNode* parent, const char* tag) { return parent && parent->child(tag) != nullptr; }

// per-voxel float layers: one <Layer> per z, ','-separated, consumed by occupied-voxel counter
// (VX_Object.cpp:1879-1900)
void read_voxel_layers(const XmlNode* block, const VxaModel& m, std::vector<double>& out)
{
    out.clear();
    auto layers = block->children_named("Layer");
    const int per_layer = m.nx * m.ny;
    for (int z = 0; z < m.nz && z < (int)layers.size(); ++z) {
        const std::string& raw = layers[z]->text;
        size_t pos = 0;
        for (int k = 0; k < per_layer; ++k) {
            size_t comma = raw.find(',', pos);
            std::string item = raw.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
            pos = comma == std::string::npos ? raw.size() : comma + 1;
            if (m.structure[(size_t)z * per_layer + k] > 0) out.push_back(std::atof(item.c_str()));
        }
    }
}

}  // namespace

std::unique_ptr<XmlNode> parse_xml(const char* data, size_t len)
{
    Cursor c{data, data + len};
    for (;;) {
        c.skip_ws();
        if (c.p >= c.end) throw std::runtime_error("xml: no root element");
        if (c.starts("<?")) { c.skip_until("?>"); continue; }
        if (c.starts("<!--")) { c.skip_until("-->"); continue; }
        if (c.starts("<!")) { c.skip_until(">"); continue; }
        if (*c.p != '<') throw std::runtime_error("xml: text before root element");
        return parse_element(c, 0);
    }
}

VxaModel read_vxa(const char* data, size_t len, int variant)
{
    std::unique_ptr<XmlNode> root = parse_xml(data, len);
    if (root->name != "VXA") throw std::runtime_error("vxa: root element is <" + root->name + ">, expected <VXA>");
    VxaModel m;
    m.variant = variant;
    const XmlNode* sim = root->child("Simulator");
    const XmlNode* env = root->child("Environment");
    const XmlNode* vxc = root->child("VXC");
    if (!vxc) vxc = root->child("DMF");
    if (!vxc) throw std::runtime_error("vxa: no <VXC> element");

    // ---- Simulator (VX_Sim.cpp:263-354): present block + absent tag -> reader fallback; absent block ->
    //      constructor value (VX_Sim.cpp:46-95)
    if (const XmlNode* integ = sim ? sim->child("Integration") : nullptr) m.dt_frac = num(integ, "DtFrac", 0.9);
    if (const XmlNode* damp = sim ? sim->child("Damping") : nullptr) {
        m.bond_damping_z = num(damp, "BondDampingZ", 0.1);
        m.col_damping_z = num(damp, "ColDampingZ", 1.0);
        m.slow_damping_z = num(damp, "SlowDampingZ", 1.0);
    }
    if (const XmlNode* col = sim ? sim->child("Collisions") : nullptr) {
        m.self_col_enabled = flag(col, "SelfColEnabled", false);
        m.col_system = inum(col, "ColSystem", 3);
        m.collision_horizon = num(col, "CollisionHorizon", 2.0);
    }
    if (const XmlNode* feat = sim ? sim->child("Features") : nullptr) {
        if (flag(feat, "MaxVelLimitEnabled", false)) m.unsupported.push_back("MaxVelLimitEnabled");
        if (flag(feat, "BlendingEnabled", false)) m.unsupported.push_back("BlendingEnabled");
        if (flag(feat, "VolumeEffectsEnabled", false)) m.unsupported.push_back("VolumeEffectsEnabled");
    }
    if (const XmlNode* stop = sim ? sim->child("StopCondition") : nullptr) {
        m.stop_type = inum(stop, "StopConditionType", 0);
        m.stop_value = num(stop, "StopConditionValue", 0.0);
        m.afterlife_time = num(stop, "AfterlifeTime", 0.0);
        m.midlife_freeze_time = num(stop, "MidLifeFreezeTime", 0.0);
        m.init_cm_time = num(stop, "InitCmTime", 0.0);
    }
    if (const XmlNode* eq = sim ? sim->child("EquilibriumMode") : nullptr)
        if (flag(eq, "EquilibriumModeEnabled", false)) m.unsupported.push_back("EquilibriumModeEnabled");
    m.min_temp_fact = num(sim, "MinTempFact", 0.1);
    if (const XmlNode* ga = sim ? sim->child("GA") : nullptr) {
        if (const XmlNode* f = ga->child("FitnessFileName")) m.fitness_file_name = f->text;
        if (const XmlNode* f = ga->child("CurvaturesTmpFile")) m.curvatures_tmp_file = f->text;
    }
    m.want_mesh = m.variant == 1;
    if (!(m.stop_type >= 0 && m.stop_type <= 3)) m.unsupported.push_back("StopConditionType>3");

    // ---- Environment (VX_Environment.cpp:123-234; LW/VX_Environment.cpp:190-191)
    auto count_regions = [&](const char* block, const char* counter) {
        const XmlNode* b = env ? env->child(block) : nullptr;
        return b ? inum(b, counter, 0) : 0;
    };
    if (count_regions("Boundary_Conditions", "NumBCs") + count_regions("Fixed_Regions", "NumFixed") +
            count_regions("Forced_Regions", "NumForced") > 0)
        m.unsupported.push_back("boundary-condition regions");
    if (const XmlNode* grav = env ? env->child("Gravity") : nullptr) {
        m.grav_enabled = flag(grav, "GravEnabled", false);
        m.grav_acc = num(grav, "GravAcc", -9.81);
        m.floor_enabled = flag(grav, "FloorEnabled", false);
    }
    if (const XmlNode* th = env ? env->child("Thermal") : nullptr) {
        m.temp_enabled = flag(th, "TempEnabled", false);
        m.temp_base = num(th, "TempBase", 25);
        if (has(th, "TempAmplitude")) m.temp_amplitude = num(th, "TempAmplitude", 0);
        else if (has(th, "TempAmp")) m.temp_amplitude = num(th, "TempAmp", 0) - m.temp_base;   // legacy tag
        else m.temp_amplitude = 0;
        m.vary_temp_enabled = flag(th, "VaryTempEnabled", false);
        m.temp_period = num(th, "TempPeriod", 0.1);
    }
    m.growth_amplitude = num(env, "GrowthAmplitude", 0);
    m.min_growth_time = num(env, "MinGrowthTime", 0);
    m.sticky_floor = flag(env, "StickyFloor", false);
    if (variant == 0) {                   // (the land_water simulator has no traces: the tags are unknown to it)
        m.time_between_traces = num(env, "TimeBetweenTraces", 0);
        m.save_traces = flag(env, "SaveTraces", false);
    }
    m.fluid_env = flag(env, "FluidEnvironment", false);
    m.aggregate_drag_coef = num(env, "AggregateDragCoefficient", 0);
    if (variant == 0) {
        if (flag(env, "NormDistByVol", false)) m.unsupported.push_back("NormDistByVol");
        if (inum(env, "NumTimeStepsInWindow", 0) > 0) m.unsupported.push_back("NumTimeStepsInWindow");
        if (flag(env, "FallingProhibited", false)) m.unsupported.push_back("FallingProhibited");
        if (flag(env, "NeedleInHaystack", false)) m.unsupported.push_back("NeedleInHaystack");
        if (has(env, "FloorRadius")) m.unsupported.push_back("FloorRadius");
        if (has(env, "Sources")) m.unsupported.push_back("Sources");
        if (flag(env, "OnsetRelative", false) || flag(env, "TerminationRelative", false)) m.unsupported.push_back("OnsetRelative/TerminationRelative");
    }

    // ---- VXC (VX_Object.cpp)
    if (const XmlNode* lat = vxc->child("Lattice")) {
        m.lattice_dim = num(lat, "Lattice_Dim", 0.001);
        if (num(lat, "X_Dim_Adj", 1) != 1 || num(lat, "Y_Dim_Adj", 1) != 1 || num(lat, "Z_Dim_Adj", 1) != 1 ||
            num(lat, "X_Line_Offset", 0) != 0 || num(lat, "Y_Line_Offset", 0) != 0 ||
            num(lat, "X_Layer_Offset", 0) != 0 || num(lat, "Y_Layer_Offset", 0) != 0)
            m.unsupported.push_back("non-cubic lattice adjustments/offsets");
    }
    m.palette.clear();
    m.palette.push_back(Material());  // "Erase", VX_Object.cpp:138-148
    if (const XmlNode* pal = vxc->child("Palette")) {
        for (const XmlNode* mat : pal->children_named("Material")) {
            Material o;
            if (has(mat, "MatType") && inum(mat, "MatType", 0) != 0) m.unsupported.push_back("non-SINGLE material type");
            if (const XmlNode* mech = mat->child("Mechanical")) {
                o.mat_model = inum(mech, "MatModel", 0);
                o.E = num(mech, "Elastic_Mod", 0);
                o.rho = num(mech, "Density", 0);
                o.nu = num(mech, "Poissons_Ratio", 0);
                o.cte = num(mech, "CTE", 0);
                o.u_static = num(mech, "uStatic", 0);
                o.u_dynamic = num(mech, "uDynamic", 0);
                if (o.mat_model != 0) m.unsupported.push_back("non-linear material model");
            }
            m.palette.push_back(o);
        }
    }
    const XmlNode* st = vxc->child("Structure");
    if (!st) throw std::runtime_error("vxa: no <Structure>");
    const std::string* comp = st->attr("Compression");
    if (!comp || *comp != "ASCII_READABLE")
        throw std::runtime_error("vxa: only Compression=\"ASCII_READABLE\" structures are supported");
    m.nx = inum(st, "X_Voxels", 1); m.ny = inum(st, "Y_Voxels", 1); m.nz = inum(st, "Z_Voxels", 1);
    if (m.nx < 1 || m.ny < 1 || m.nz < 1 || (long long)m.nx * m.ny * m.nz > (1LL << 26)) throw std::runtime_error("vxa: bad lattice size");
    const XmlNode* dat = st->child("Data");
    if (!dat) throw std::runtime_error("vxa: no <Data>");
    auto layers = dat->children_named("Layer");
    if ((int)layers.size() < m.nz) throw std::runtime_error("vxa: fewer <Layer> elements than Z_Voxels");
    m.structure.assign((size_t)m.nx * m.ny * m.nz, 0);
    for (int z = 0; z < m.nz; ++z) {
        const std::string& raw = layers[z]->text;
        if ((int)raw.size() != m.nx * m.ny) throw std::runtime_error("vxa: voxel layer data does not match X_Voxels*Y_Voxels");
        for (int k = 0; k < m.nx * m.ny; ++k) {
            int v = (unsigned char)raw[k] - 48;
            if (v < 0 || v >= (int)m.palette.size()) throw std::runtime_error("vxa: material index outside the palette");
            m.structure[(size_t)z * m.nx * m.ny + k] = (unsigned char)v;
        }
    }
    if (const XmlNode* b = st->child("PhaseOffset")) { m.has_phase_offset = true; read_voxel_layers(b, m, m.phase_offset); }
    if (const XmlNode* b = st->child("TempAmpDamp")) { m.has_temp_amp_damp = true; read_voxel_layers(b, m, m.temp_amp_damp); }
    if (const XmlNode* b = st->child("Stiffness")) { m.has_stiffness = true; read_voxel_layers(b, m, m.stiffness); }
    if (variant == 0) {   // development layers exist in _voxcad only (VX_Object.cpp:1910-2140)
        if (const XmlNode* b = st->child("FinalPhaseOffset")) { m.has_final_phase_offset = true; read_voxel_layers(b, m, m.final_phase_offset); }
        if (const XmlNode* b = st->child("FinalTempAmpDamp")) { m.has_final_temp_amp_damp = true; read_voxel_layers(b, m, m.final_temp_amp_damp); }
        if (const XmlNode* b = st->child("InitialVoxelSize")) { m.has_initial_voxel_size = true; read_voxel_layers(b, m, m.initial_voxel_size); }
        if (const XmlNode* b = st->child("FinalVoxelSize")) { m.has_final_voxel_size = true; read_voxel_layers(b, m, m.final_voxel_size); }
        if (const XmlNode* b = st->child("GrowthTime")) { m.has_growth_time = true; read_voxel_layers(b, m, m.growth_time); }
        if (const XmlNode* b = st->child("StartGrowthTime")) { m.has_start_growth_time = true; read_voxel_layers(b, m, m.start_growth_time); }
    } else {
        for (const char* tag : {"FinalPhaseOffset", "FinalTempAmpDamp", "InitialVoxelSize", "FinalVoxelSize", "GrowthTime", "StartGrowthTime"})
            if (st->child(tag)) m.unsupported.push_back(std::string("<") + tag + "> development layer (land_water)");
    }
    for (const char* tag : {"StiffnessPlasticityRate", "VestigialLimbs"})
        if (st->child(tag)) m.unsupported.push_back(std::string("<") + tag + "> development layer");
    if (variant == 0 && m.temp_amp_damp.size() && !m.has_temp_amp_damp) m.temp_amp_damp.clear();
    return m;
}

}  // namespace vxh
