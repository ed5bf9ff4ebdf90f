// `voxelyze` command line, drop-in for the reference headless simulator
// (evosoro/_voxcad/voxelyzeMain/main.cpp:9-133; land_water: evosoro/_voxcad_land_water/voxelyzeMain/main.cpp).
//   voxelyze -f <file.vxa> [-f <more.vxa> ...] [--list <file with one .vxa path per line>] [-p]
//            [--land-water] [--device N | --devices N,M,...] [--computeShapeDescriptors (with -p and one file: the reference's console report of the surface mesh)]
//            [--direct | --broker | --broker-stat | --broker-quit]
// Writes each robot's result XML to the <FitnessFileName> of its .vxa.  Exit code follows the reference's
// inverted convention: 1 = completed, 0 = failed (main.cpp:28,57,132).  Several -f / --list entries are
// stepped together as one batch on the GPU.
//
// -p with ONE robot prints what the reference prints (main.cpp:60-63, 92-104, 128): the import message, every 100 steps the time,
// |centre of mass| and voxel 0's scale, TempAmplitude, TempPeriod and phaseOffset, through std::cout like the reference (same number
// formatting), then "Ended at:"; the robot is then stepped in calls of 100 steps.  -p with several robots: one summary line per robot.
//
// THE BROKER.  evosoro starts one `voxelyze -f x.vxa` process PER ROBOT, all of a generation at once, and collects result files as
// they appear (evosoro/tools/evaluation.py:59-90, 101-211).  Taken literally that is pop_size processes on one GPU, each paying for
// a HIP context and stepping a batch of one.  So a plain one-file invocation does not step anything itself: it hands its file to a
// per-user broker process over a UNIX socket and waits for the verdict.  The broker (this same binary, `voxelyze --broker`, started
// by the first client that finds none, detached) keeps ONE engine per simulator variant resident, collects the requests that arrive
// within a few milliseconds of each other -- a generation -- into ONE vxh_run, writes every result XML relative to ITS client's
// working directory (the <FitnessFileName> of a .vxa is a relative path), answers the clients, and exits after VXH_BROKER_IDLE_S
// seconds without work.  Unmodified evosoro thereby gets the batched rate.  Requests that arrive while a batch is stepping wait in
// the socket's queue and form the next batch.
//   --direct / VXH_BROKER=0    this process steps its files itself (what every multi-file, -p or --devices invocation does anyway)
//   VXH_BROKER_SOCKET          socket path (default /tmp/vxhip-broker-<uid>.sock)
//   VXH_BROKER_GAP_MS  (15)    a batch is closed when no request has arrived for this long ...
//   VXH_BROKER_MAX_WAIT_MS (500)  ... or when its first request has waited this long
//   VXH_BROKER_IDLE_S  (30)    the broker exits after this long without a request
//   VXH_BROKER_DEVICES (0)     device list of the broker's engines, e.g. 0,1,2,3 (vxh_create_multi: the batch is partitioned by cost)
//   VXH_BROKER_LOG             file the broker's stderr is appended to (default: /dev/null)
// If no broker can be reached or started the client falls back to stepping its file itself and says so on stderr.
#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include <fcntl.h>
#include <poll.h>
#include <signal.h>
#include <sys/file.h>
#include <sys/socket.h>
#include <sys/types.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../../include/vxhip.h"

namespace {

double env_num(const char* name, double dflt) { const char* v = std::getenv(name); return (v && *v) ? std::atof(v) : dflt; }

// Where a broker listens.  The directory is the user's own -- $XDG_RUNTIME_DIR, else /tmp/vxhip-<uid> created 0700 and checked (owner, mode, not a
// symlink): nobody else can pre-bind the name or read who connects.  The NAME carries everything a broker's answers depend on, so a
// client is only ever served by a broker that would compute what the client itself would: the executable (path, size, mtime: a rebuilt
// binary gets a fresh broker while an old one drains), the device selection (HIP_ / ROCR_ / CUDA_VISIBLE_DEVICES, VXH_BROKER_DEVICES: two
// experiments pinned to different GPUs get one broker each) and VXH_ENGINE_OPTIONS.  (Round 4 keyed it by uid alone: ADVICE.)
std::string socket_path()
{
    if (const char* p = std::getenv("VXH_BROKER_SOCKET")) if (*p) return p;
    std::string dir;
    if (const char* x = std::getenv("XDG_RUNTIME_DIR")) { struct stat st; if (*x && ::stat(x, &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == getuid() && (st.st_mode & 077) == 0 && ::access(x, W_OK) == 0) dir = x; }   // (0700 like the fallback: advisor, round 5)
    if (dir.empty()) {
        dir = "/tmp/vxhip-" + std::to_string((long)getuid());
        ::mkdir(dir.c_str(), 0700);
        struct stat st;
        if (::lstat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & 077) != 0) return std::string();   // (no safe place: no broker)
    }
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](const std::string& t) { for (unsigned char c : t) { h ^= c; h *= 1099511628211ull; } h ^= 0xff; h *= 1099511628211ull; };
    char self[4096];
    const ssize_t n = ::readlink("/proc/self/exe", self, sizeof(self) - 1);
    if (n > 0) {
        self[n] = 0;
        mix(self);
        struct stat st;
        if (::stat(self, &st) == 0) mix(std::to_string((long long)st.st_size) + ":" + std::to_string((long long)st.st_mtime) + ":" + std::to_string((long long)st.st_mtim.tv_nsec));
    }
    for (const char* key : {"HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL", "VXH_BROKER_DEVICES", "VXH_ENGINE_OPTIONS"}) {
        const char* v = std::getenv(key);
        mix(std::string(key) + "=" + (v ? v : "\x01unset"));
    }
    char name[64];
    std::snprintf(name, sizeof(name), "/broker-%016llx.sock", h);
    return dir + name;
}

// the listener must be this user's process (SO_PEERCRED): an answer from anybody else's is not a result
bool peer_is_me(int fd)
{
    struct ucred cr;
    socklen_t len = sizeof(cr);
    return ::getsockopt(fd, SOL_SOCKET, SO_PEERCRED, &cr, &len) == 0 && cr.uid == getuid();
}

std::vector<int> parse_devices(const char* p)
{
    std::vector<int> d;
    while (p && *p) { d.push_back(std::atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
    return d;
}

// ---- framing: a message is a sequence of fields, each a 32-bit length followed by that many bytes
bool write_all(int fd, const void* buf, size_t n)
{
    const char* p = (const char*)buf;
    while (n > 0) { const ssize_t w = ::send(fd, p, n, MSG_NOSIGNAL); if (w <= 0) { if (w < 0 && errno == EINTR) continue; return false; } p += w; n -= (size_t)w; }
    return true;
}
bool read_all(int fd, void* buf, size_t n)
{
    char* p = (char*)buf;
    while (n > 0) { const ssize_t r = ::recv(fd, p, n, 0); if (r <= 0) { if (r < 0 && errno == EINTR) continue; return false; } p += r; n -= (size_t)r; }
    return true;
}
bool send_field(int fd, const std::string& s) { const unsigned n = (unsigned)s.size(); return write_all(fd, &n, 4) && write_all(fd, s.data(), s.size()); }
bool recv_field(int fd, std::string& s)
{
    unsigned n = 0;
    if (!read_all(fd, &n, 4) || n > (1u << 20)) return false;
    s.resize(n);
    return n == 0 || read_all(fd, &s[0], n);
}

int connect_to(const std::string& path)
{
    const int fd = ::socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    if (fd < 0) return -1;
    sockaddr_un a;
    std::memset(&a, 0, sizeof(a));
    a.sun_family = AF_UNIX;
    if (path.size() >= sizeof(a.sun_path)) { ::close(fd); return -1; }
    std::strcpy(a.sun_path, path.c_str());
    if (::connect(fd, (sockaddr*)&a, sizeof(a)) != 0) { ::close(fd); return -1; }
    if (!peer_is_me(fd)) { ::close(fd); return -1; }
    return fd;
}

// ---- one robot (or several) stepped by this process: the round-1..3 command line
// What the reference prints of its deformable surface mesh under -p --computeShapeDescriptors (voxelyzeMain/main.cpp:72-88,113-126:
// mesh size, robot volume, convex-hull volume, shape complexity, then CVX_MeshUtil::printAllMeshInfo, VX_MeshUtil.cpp:733-772: vertices,
// facets as vertex triples, facet normals = the normalised cross product of two edges, Utils/Mesh.cpp:585-591), in the stream's default
// six significant digits.  The hull volume is computed here; the reference needs the external `qhull` and prints -1 without it.
void print_shape(vxh_engine* e, bool at_end)
{
    int nv = 0, nf = 0;
    if (vxh_get_mesh(e, 0, at_end ? 1 : 0, nullptr, 0, &nv, nullptr, 0, &nf) != VXH_OK) return;
    std::vector<double> vert((size_t)3 * (nv > 0 ? nv : 1));
    std::vector<int> fac((size_t)3 * (nf > 0 ? nf : 1));
    if (vxh_get_mesh(e, 0, at_end ? 1 : 0, vert.data(), nv, &nv, fac.data(), nf, &nf) != VXH_OK) return;
    double vol = -1, hull = -1, cplx = -1;
    vxh_get_shape_descriptors(e, 0, at_end ? 1 : 0, &vol, &hull, &cplx);
    const char* when = at_end ? "Final" : "Init";
    if (!at_end) std::cout << "Robot mesh has " << nv << " vertices and " << nf << " facets" << std::endl;
    std::cout << when << " robot volume: " << vol << std::endl
              << when << " convex hull volume: " << hull << std::endl << std::endl
              << when << " shape complexity: " << cplx << std::endl;
    const char* bar = " -----------------------------------------------------------";
    std::cout << bar << std::endl << "| \t\tPRINTING DEFORMABLE MESH VERTICES \t\t\t\t  |" << std::endl << bar << std::endl;
    for (int i = 0; i < nv; ++i) std::cout << vert[(size_t)3 * i] << " " << vert[(size_t)3 * i + 1] << " " << vert[(size_t)3 * i + 2] << std::endl;
    std::cout << bar << std::endl;
    std::cout << " ---------------------------------------------------------------------" << std::endl
              << "| \tPRINTING DEFORMABLE MESH FACETS (TERNS OF VERTICES INDICES) \t|" << std::endl
              << " ---------------------------------------------------------------------" << std::endl;
    for (int f = 0; f < nf; ++f) std::cout << fac[(size_t)3 * f] << " " << fac[(size_t)3 * f + 1] << " " << fac[(size_t)3 * f + 2] << std::endl;
    std::cout << bar << std::endl;
    std::cout << bar << std::endl << "| \t\tPRINTING DEFORMABLE MESH NORMALS \t\t\t\t  |" << std::endl << bar << std::endl;
    for (int f = 0; f < nf; ++f) {
        const double* a = &vert[(size_t)3 * fac[(size_t)3 * f]]; const double* b = &vert[(size_t)3 * fac[(size_t)3 * f + 1]]; const double* c = &vert[(size_t)3 * fac[(size_t)3 * f + 2]];
        const double ux = b[0] - a[0], uy = b[1] - a[1], uz = b[2] - a[2], wx = c[0] - a[0], wy = c[1] - a[1], wz = c[2] - a[2];
        double nx = uy * wz - uz * wy, ny = uz * wx - ux * wz, nz = ux * wy - uy * wx;
        const double l = std::sqrt(nx * nx + ny * ny + nz * nz);
        if (l > 0) { nx /= l; ny /= l; nz /= l; }      // (Vec3D::Normalized: a null vector stays null)
        std::cout << nx << " " << ny << " " << nz << std::endl;
    }
    std::cout << bar << std::endl;
}

int run_direct(const std::vector<std::string>& files, int variant, std::vector<int> devices, bool print_scrn, bool shape = false)
{
    vxh_engine* e = nullptr;
    if (devices.empty()) devices.push_back(0);
    int rc = vxh_create_multi(&e, variant, devices.data(), (int)devices.size());     // several devices: the batch is partitioned by cost
    if (rc != VXH_OK) { std::fprintf(stderr, "voxelyze: %s\n", vxh_strerror(rc)); return 0; }
    shape = shape && print_scrn && files.size() == 1;      // (_voxcad: the descriptors only ever reach the console, main.cpp:72-88 -- the result file has no such tags)
    if (shape) vxh_set_option(e, "shape_descriptors", 1);
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
        int nvox = 0, nbond = 0;
        vxh_robot_dims(e, 0, &nvox, &nbond, nullptr, nullptr);
        st.resize((size_t)14 * (size_t)(nvox > 0 ? nvox : 1));
        double amp = 0, per = 0, ph = 0;
        if (nvox > 0) vxh_voxel_actuation(e, 0, 0, &amp, &per, &ph);
        // the import's return message (main.cpp:60-63; VX_Sim.cpp:622,633-637,708: the "failed" line is printed whenever a bond
        // exists, because the reference tests bond INDEX 0 as a bool -- SURVEY.md App. A.10)
        std::cout << "\nImporting Environment into simulator...\n" << "Simulation import return message:\n";
        if (nbond > 0) std::cout << "At least one bond creation failed during import.\n";
        std::cout << "Completed Simulation Import: " << nvox << " Voxels, " << nbond << "Bonds.\n" << "\n";
        if (shape) print_shape(e, false);
        for (long long done = 0;; done += 100) {
            if (done > 0) { rc = vxh_step(e, 100); if (rc != VXH_OK) break; }
            if (nvox == 0) { rc = vxh_run(e); break; }
            rc = vxh_get_state(e, 0, st.data(), nvox);       // (before the first step: uploads the batch and hands back the rest pose)
            if (rc == VXH_OK) rc = vxh_get_result(e, 0, &res);
            if (rc != VXH_OK || res.status != VXH_ROBOT_PENDING) break;
            std::cout << "Time: " << res.cur_time << std::endl;
            double cx = 0, cy = 0, cz = 0;
            if (res.steps > 0) { cx = res.cur_cm[0]; cy = res.cur_cm[1]; cz = res.cur_cm[2]; }
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
        if (rc == VXH_OK && shape) print_shape(e, true);
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

// ---- the broker
struct Request { int fd; int variant; std::string cwd, vxa; };

void answer(Request& q, int code, const std::string& msg)
{
    if (q.fd < 0) return;
    send_field(q.fd, std::to_string(code));
    send_field(q.fd, msg);
    ::close(q.fd);
    q.fd = -1;
}

int broker_main(const std::string& path)
{
    ::signal(SIGPIPE, SIG_IGN);
    const int lfd = ::socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    if (lfd < 0) { std::perror("voxelyze --broker: socket"); return 0; }
    sockaddr_un a;
    std::memset(&a, 0, sizeof(a));
    a.sun_family = AF_UNIX;
    if (path.size() >= sizeof(a.sun_path)) { std::fprintf(stderr, "voxelyze --broker: socket path too long\n"); return 0; }
    std::strcpy(a.sun_path, path.c_str());
    const mode_t old = ::umask(0077);
    const int brc = ::bind(lfd, (sockaddr*)&a, sizeof(a));
    ::umask(old);
    if (brc != 0) { std::fprintf(stderr, "voxelyze --broker: cannot bind %s: %s\n", path.c_str(), std::strerror(errno)); return 0; }
    if (::listen(lfd, 4096) != 0) { std::perror("voxelyze --broker: listen"); ::unlink(path.c_str()); return 0; }
    // (clients may connect from here on: the HIP runtime is only started with the first batch, their requests wait in the queue)
    const double gap_ms = env_num("VXH_BROKER_GAP_MS", 15), max_wait_ms = env_num("VXH_BROKER_MAX_WAIT_MS", 500), idle_s = env_num("VXH_BROKER_IDLE_S", 30);
    std::vector<int> devices = parse_devices(std::getenv("VXH_BROKER_DEVICES"));
    if (devices.empty()) devices.push_back(0);
    vxh_engine* engines[2] = {nullptr, nullptr};
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
    long long batches = 0, robots = 0, largest = 0;
    bool quit = false;
    while (!quit) {
        // ---- collect a batch
        std::vector<Request> batch;
        std::chrono::steady_clock::time_point first = now(), last = now();
        for (;;) {
            int wait_ms;
            if (batch.empty()) wait_ms = (int)(idle_s * 1000);
            else {
                const double left = std::min(gap_ms - ms_since(last), max_wait_ms - ms_since(first));
                if (left <= 0) break;
                wait_ms = (int)std::ceil(left);
            }
            pollfd p = {lfd, POLLIN, 0};
            const int pr = ::poll(&p, 1, wait_ms);
            if (pr < 0) { if (errno == EINTR) continue; quit = true; break; }
            if (pr == 0) { if (batch.empty()) quit = true; break; }
            const int cfd = ::accept4(lfd, nullptr, nullptr, SOCK_CLOEXEC);
            if (cfd < 0) continue;
            timeval tv = {5, 0};
            ::setsockopt(cfd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));      // (a client that connects and says nothing does not stall the generation)
            Request q; q.fd = cfd; q.variant = 0;
            std::string verb, variant;
            if (!recv_field(cfd, verb) || !recv_field(cfd, variant) || !recv_field(cfd, q.cwd) || !recv_field(cfd, q.vxa)) { ::close(cfd); continue; }
            if (verb == "PING") { answer(q, 1, "pong"); continue; }
            if (verb == "QUIT") { answer(q, 1, "bye"); quit = true; break; }
            if (verb == "STAT") { answer(q, 1, "batches " + std::to_string(batches) + " robots " + std::to_string(robots) + " largest " + std::to_string(largest)); continue; }
            q.variant = std::atoi(variant.c_str()) == 1 ? 1 : 0;
            if (batch.empty()) first = now();
            last = now();
            batch.push_back(q);
        }
        if (batch.empty()) continue;
        // ---- step it: one engine per variant, resident across batches
        ++batches;
        largest = std::max<long long>(largest, (long long)batch.size());
        for (int variant = 0; variant < 2; ++variant) {
            std::vector<int> idx;
            for (size_t i = 0; i < batch.size(); ++i) if (batch[i].variant == variant) idx.push_back((int)i);
            if (idx.empty()) continue;
            vxh_engine*& e = engines[variant];
            int rc = VXH_OK;
            if (!e) rc = vxh_create_multi(&e, variant, devices.data(), (int)devices.size());
            if (rc != VXH_OK) { for (int i : idx) answer(batch[i], 0, std::string("broker: ") + vxh_strerror(rc)); e = nullptr; continue; }
            vxh_clear(e);
            std::vector<int> held;             // batch index of every robot the engine took, in engine order
            for (int i : idx) {
                rc = vxh_add_vxa_file(e, batch[i].vxa.c_str(), nullptr);
                if (rc != VXH_OK) answer(batch[i], 0, batch[i].vxa + ": " + vxh_strerror(rc) + " (" + vxh_last_error(e) + ")");
                else held.push_back(i);
            }
            if (held.empty()) continue;
            rc = vxh_run(e);
            if (rc != VXH_OK) {
                const std::string msg = std::string(vxh_strerror(rc)) + " (" + vxh_last_error(e) + ")";
                for (int i : held) answer(batch[i], 0, msg);
                vxh_destroy(e); e = nullptr;   // (a fresh engine for the next batch)
                continue;
            }
            robots += (long long)held.size();
            for (size_t r = 0; r < held.size(); ++r) {
                Request& q = batch[held[r]];
                vxh_result res;
                vxh_get_result(e, (int)r, &res);
                if (res.status != VXH_ROBOT_FINISHED) { answer(q, 0, q.vxa + " did not finish (status " + std::to_string(res.status) + ")"); continue; }
                // the result file names of a .vxa are relative to the directory its voxelyze process was started in
                if (::chdir(q.cwd.c_str()) != 0) { answer(q, 0, "broker: cannot enter " + q.cwd); continue; }
                rc = vxh_write_result_xml(e, (int)r, nullptr);
                if (rc != VXH_OK) answer(q, 0, q.vxa + ": " + vxh_strerror(rc) + " (" + vxh_last_error(e) + ")");
                else answer(q, 1, "ok: batch " + std::to_string(batches) + ", " + std::to_string(held.size()) + " robots");
            }
            if (::chdir("/") != 0) {}
        }
        for (auto& q : batch) answer(q, 0, "broker: request not served");
    }
    ::unlink(path.c_str());
    for (auto*& e : engines) if (e) { vxh_destroy(e); e = nullptr; }
    ::close(lfd);
    return 1;
}

// start a broker (detached) unless somebody else just did; true when one should be reachable
bool spawn_broker(const std::string& path, const char* self)
{
    const std::string lock = path + ".lock";
    const int lk = ::open(lock.c_str(), O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0600);
    if (lk < 0) return false;
    bool ok = false;
    if (::flock(lk, LOCK_EX) == 0) {
        int fd = connect_to(path);             // (somebody else may have started one while this process waited for the lock)
        if (fd >= 0) { ::close(fd); ok = true; }
        else {
            ::unlink(path.c_str());            // a dead broker's socket
            const pid_t pid = ::fork();
            if (pid == 0) {
                if (::fork() != 0) ::_exit(0);                 // (double fork: the broker is nobody's child)
                ::setsid();
                const int devnull = ::open("/dev/null", O_RDWR);
                const char* logp = std::getenv("VXH_BROKER_LOG");
                const int log = (logp && *logp) ? ::open(logp, O_CREAT | O_WRONLY | O_APPEND, 0600) : -1;
                if (devnull >= 0) { ::dup2(devnull, 0); ::dup2(devnull, 1); ::dup2(log >= 0 ? log : devnull, 2); }
                ::close(lk);
                ::setenv("VXH_BROKER_SOCKET", path.c_str(), 1);   // (the broker listens where THIS client will look, whatever it would compute itself: a binary replaced in between -- advisor, round 5)
                ::execl(self, self, "--broker", (char*)nullptr);
                ::_exit(127);
            }
            if (pid > 0) {
                int st = 0;
                ::waitpid(pid, &st, 0);
                for (int k = 0; k < 400 && !ok; ++k) {           // the broker listens before it touches the GPU: a few milliseconds
                    fd = connect_to(path);
                    if (fd >= 0) { ::close(fd); ok = true; } else ::usleep(5000);
                }
            }
        }
        ::flock(lk, LOCK_UN);
    }
    ::close(lk);
    return ok;
}

// 1 / 0 = the reference's exit codes; -1 = no broker: do it yourself
int run_through_broker(const std::string& file, int variant, const char* self)
{
    const std::string path = socket_path();
    if (path.empty()) return -1;
    int fd = connect_to(path);
    if (fd < 0) {
        if (!spawn_broker(path, self)) return -1;
        fd = connect_to(path);
        if (fd < 0) return -1;
    }
    char cwd[4096], real[4096];
    if (!::getcwd(cwd, sizeof(cwd))) { ::close(fd); return -1; }
    const char* abs = ::realpath(file.c_str(), real);
    if (!abs) { ::close(fd); std::fprintf(stderr, "voxelyze: %s: cannot read the file\n", file.c_str()); return 0; }   // (like a failed LoadVXAFile)
    if (!send_field(fd, "RUN") || !send_field(fd, std::to_string(variant)) || !send_field(fd, cwd) || !send_field(fd, abs)) { ::close(fd); return -1; }
    std::string code, msg;
    if (!recv_field(fd, code) || !recv_field(fd, msg)) { ::close(fd); std::fprintf(stderr, "voxelyze: the broker went away\n"); return -1; }
    ::close(fd);
    if (code != "1") { std::fprintf(stderr, "voxelyze: %s\n", msg.c_str()); return 0; }
    return 1;
}

}  // namespace

int main(int argc, char* argv[])
{
    std::vector<std::string> files;
    bool print_scrn = false, direct = false, broker = false, shape = false;
    int variant = VXH_VOXCAD;
    std::vector<int> devices;
    std::string control;
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
        else if ((!std::strcmp(argv[i], "--device") || !std::strcmp(argv[i], "--devices")) && i + 1 < argc) devices = parse_devices(argv[++i]);
        else if (!std::strcmp(argv[i], "--computeShapeDescriptors")) shape = true;
        else if (!std::strcmp(argv[i], "--direct")) direct = true;
        else if (!std::strcmp(argv[i], "--broker")) broker = true;
        else if (!std::strcmp(argv[i], "--broker-quit")) control = "QUIT";
        else if (!std::strcmp(argv[i], "--broker-stat")) control = "STAT";
    }
    if (broker) { const std::string where = socket_path(); if (where.empty()) { std::fprintf(stderr, "voxelyze --broker: no directory of this user's own for the socket\n"); return 0; } return broker_main(where); }
    if (!control.empty()) {
        const int fd = connect_to(socket_path());
        if (fd < 0) { std::printf("no broker at %s\n", socket_path().c_str()); return 0; }
        std::string code, msg;
        if (send_field(fd, control) && send_field(fd, "0") && send_field(fd, "") && send_field(fd, "") && recv_field(fd, code) && recv_field(fd, msg)) std::printf("%s\n", msg.c_str());
        ::close(fd);
        return 1;
    }
    if (files.empty()) { std::printf("\nInput file required. Quitting.\n"); return 0; }
    // one plain file, nothing asked of THIS process (no -p, no device list): the broker's case
    const char* off = std::getenv("VXH_BROKER");
    if (files.size() == 1 && !print_scrn && devices.empty() && !direct && !(off && !std::strcmp(off, "0"))) {
        char self[4096];
        const ssize_t n = ::readlink("/proc/self/exe", self, sizeof(self) - 1);
        if (n > 0) {
            self[n] = 0;
            const int rc = run_through_broker(files[0], variant, self);
            if (rc >= 0) return rc;
            std::fprintf(stderr, "voxelyze: no broker reachable at %s; stepping in this process\n", socket_path().c_str());
        }
    }
    return run_direct(files, variant, devices, print_scrn, shape);
}
