// `voxelyze` command line, drop-in for the reference headless simulator
// (evosoro/_voxcad/voxelyzeMain/main.cpp:9-133; land_water: evosoro/_voxcad_land_water/voxelyzeMain/main.cpp).
//   voxelyze -f <file.vxa> [-f <more.vxa> ...] [--list <file with one .vxa path per line>] [-p]
//            [--land-water] [--device N | --devices N,M,...] [--computeShapeDescriptors (accepted, ignored)]
// Writes each robot's result XML to the <FitnessFileName> of its .vxa.  Exit code follows the reference's
// inverted convention: 1 = completed, 0 = failed (main.cpp:28,57,132).  Several -f / --list entries are
// stepped together as one batch on the GPU.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/vxhip.h"

int main(int argc, char* argv[])
{
    std::vector<std::string> files;
    bool print_scrn = false;
    int variant = VXH_VOXCAD;
    std::vector<int> devices;
    if (const char* exe = std::strrchr(argv[0], '/')) { if (std::strstr(exe, "land_water") || std::strstr(exe, "_lw")) variant = VXH_VOXCAD_LAND_WATER; }
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "-f") && i + 1 < argc) files.push_back(argv[++i]);
        else if (!std::strcmp(argv[i], "--list") && i + 1 < argc) {
            std::ifstream in(argv[++i]);
            std::string line;
            while (std::getline(in, line)) if (!line.empty()) files.push_back(line);
        }
        else if (!std::strcmp(argv[i], "-p")) print_scrn = true;
        else if (!std::strcmp(argv[i], "--land-water")) variant = VXH_VOXCAD_LAND_WATER;
        else if ((!std::strcmp(argv[i], "--device") || !std::strcmp(argv[i], "--devices")) && i + 1 < argc) {
            for (const char* p = argv[++i]; *p;) { devices.push_back(std::atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
        }
        else if (!std::strcmp(argv[i], "--computeShapeDescriptors")) {}
    }
    if (files.empty()) { std::printf("\nInput file required. Quitting.\n"); return 0; }
    vxh_engine* e = nullptr;
    if (devices.empty()) devices.push_back(0);
    int rc = vxh_create_multi(&e, variant, devices.data(), (int)devices.size());     // several devices: the batch is partitioned by cost
    if (rc != VXH_OK) { std::fprintf(stderr, "voxelyze: %s\n", vxh_strerror(rc)); return 0; }
    for (const std::string& f : files) {
        rc = vxh_add_vxa_file(e, f.c_str(), nullptr);
        if (rc != VXH_OK) {
            if (print_scrn) std::printf("\nProblem importing VXA file. Quitting\n");
            std::fprintf(stderr, "voxelyze: %s: %s (%s)\n", f.c_str(), vxh_strerror(rc), vxh_last_error(e));
            vxh_destroy(e);
            return 0;
        }
    }
    rc = vxh_run(e);
    if (rc != VXH_OK) { std::fprintf(stderr, "voxelyze: %s (%s)\n", vxh_strerror(rc), vxh_last_error(e)); vxh_destroy(e); return 0; }
    int ok = 1;
    for (int r = 0; r < vxh_num_robots(e); r++) {
        vxh_result res;
        vxh_get_result(e, r, &res);
        if (print_scrn) std::printf("%s: status %d, %d voxels, %d steps, ended at: %g\n", files[r].c_str(), res.status, res.nvox, res.steps, res.cur_time);
        if (res.status != VXH_ROBOT_FINISHED) { std::fprintf(stderr, "voxelyze: %s did not finish (status %d)\n", files[r].c_str(), res.status); ok = 0; continue; }
        rc = vxh_write_result_xml(e, r, nullptr);
        if (rc != VXH_OK) { std::fprintf(stderr, "voxelyze: %s: %s (%s)\n", files[r].c_str(), vxh_strerror(rc), vxh_last_error(e)); ok = 0; }
    }
    vxh_destroy(e);
    return ok;
}
