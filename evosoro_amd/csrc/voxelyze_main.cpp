// `voxelyze` command line, drop-in for the reference headless simulator
// (evosoro/_voxcad/voxelyzeMain/main.cpp:9-133; land_water: evosoro/_voxcad_land_water/voxelyzeMain/main.cpp).
//   voxelyze -f <file.vxa> [-f <more.vxa> ...] [--list <file with one .vxa path per line>] [-p]
//            [--land-water] [--device N | --devices N,M,...] [--computeShapeDescriptors (accepted, ignored)]
// -p with ONE robot prints what the reference prints (main.cpp:92-104, 128): every 100 steps the time, |centre of mass| and
// voxel 0's scale, TempAmplitude, TempPeriod and phaseOffset, through std::cout like the reference (same number formatting), then
// "Ended at:"; the robot is then stepped in calls of 100 steps.  -p with several robots: one summary line per robot.
// Writes each robot's result XML to the <FitnessFileName> of its .vxa.  Exit code follows the reference's
// inverted convention: 1 = completed, 0 = failed (main.cpp:28,57,132).  Several -f / --list entries are
// stepped together as one batch on the GPU.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
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
    if (print_scrn && files.size() == 1) {
        // the reference's loop: report, then step; Step % 100 == 0 (main.cpp:92).  The reference's `Time` is its own sum of dt, which
        // is the simulator's CurTime; GetCM() is the mass-weighted centre of all voxels (VX_Sim.cpp:2415-2430) -- before the first
        // step that is the rest pose, which vxh_get_state hands out straight after the import.
        vxh_result res;
        std::vector<double> st;
        long long planned = 0;
        int nvox = 0;
        vxh_robot_dims(e, 0, &nvox, nullptr, nullptr, &planned);
        st.resize((size_t)14 * (size_t)(nvox > 0 ? nvox : 1));
        double amp = 0, per = 0, ph = 0;
        if (nvox > 0) vxh_voxel_actuation(e, 0, 0, &amp, &per, &ph);
        {   // the import's return message (main.cpp:60-63; VX_Sim.cpp:622,633-637,708: the "failed" line is printed whenever a bond
            // exists, because the reference tests bond INDEX 0 as a bool -- SURVEY.md App. A.10)
            int nbond = 0;
            vxh_robot_dims(e, 0, nullptr, &nbond, nullptr, nullptr);
            std::cout << "\nImporting Environment into simulator...\n" << "Simulation import return message:\n";
            if (nbond > 0) std::cout << "At least one bond creation failed during import.\n";
            std::cout << "Completed Simulation Import: " << nvox << " Voxels, " << nbond << "Bonds.\n" << "\n";
        }
        for (long long done = 0;; done += 100) {
            if (done > 0) { rc = vxh_step(e, 100); if (rc != VXH_OK) break; }
            if (nvox == 0) { rc = vxh_run(e); break; }
            rc = vxh_get_state(e, 0, st.data(), nvox);       // (before the first step: uploads the batch and hands back the rest pose)
            if (rc == VXH_OK) rc = vxh_get_result(e, 0, &res);
            if (rc != VXH_OK || res.status != VXH_ROBOT_PENDING) break;
            std::cout << "Time: " << res.cur_time << std::endl;
            const double* cm = res.steps > 0 ? res.cur_cm : nullptr;
            double cx = 0, cy = 0, cz = 0;
            if (cm) { cx = cm[0]; cy = cm[1]; cz = cm[2]; }
            else {      // no step taken yet: every voxel of an evosoro robot has the same density, the plain mean of the rest pose is GetCM()
                for (int v = 0; v < nvox; ++v) { cx += st[(size_t)14 * v]; cy += st[(size_t)14 * v + 1]; cz += st[(size_t)14 * v + 2]; }
                cx /= nvox; cy /= nvox; cz /= nvox;
            }
            std::cout << "CM: " << std::sqrt(cx * cx + cy * cy + cz * cz) << std::endl << std::endl;
            std::cout << "Vox[0]  Scale: " << st[7] << std::endl;
            std::cout << "Vox[0]  TempAmp: " << (float)amp << std::endl;
            std::cout << "Vox[0]  TempPer: " << (float)per << std::endl;
            std::cout << "Vox[0]  phaseOffset: " << (float)ph << std::endl;
        }
        if (rc == VXH_OK) { vxh_get_result(e, 0, &res); std::cout << "Ended at: " << res.cur_time << std::endl; }
    } else
        rc = vxh_run(e);
    if (rc != VXH_OK) { std::fprintf(stderr, "voxelyze: %s (%s)\n", vxh_strerror(rc), vxh_last_error(e)); vxh_destroy(e); return 0; }
    int ok = 1;
    for (int r = 0; r < vxh_num_robots(e); r++) {
        vxh_result res;
        vxh_get_result(e, r, &res);
        if (print_scrn && files.size() > 1) std::printf("%s: status %d, %d voxels, %d steps, ended at: %g\n", files[r].c_str(), res.status, res.nvox, res.steps, res.cur_time);
        if (res.status != VXH_ROBOT_FINISHED) { std::fprintf(stderr, "voxelyze: %s did not finish (status %d)\n", files[r].c_str(), res.status); ok = 0; continue; }
        rc = vxh_write_result_xml(e, r, nullptr);
        if (rc != VXH_OK) { std::fprintf(stderr, "voxelyze: %s: %s (%s)\n", files[r].c_str(), vxh_strerror(rc), vxh_last_error(e)); ok = 0; }
    }
    vxh_destroy(e);
    return ok;
}
