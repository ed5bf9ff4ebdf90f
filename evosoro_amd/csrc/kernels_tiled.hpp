// Tiled path of the batched Voxelyze stepper: k_tile_steps<TABG, MESH>, SEVERAL workgroups per robot, all of them resident for a
// whole launch of many time steps (included at the end of kernels.hpp).  By default it steps the robots the one-workgroup-per-
// robot kernel (kernels_fused.hpp) cannot take: lattices of more than 1024 voxels (BASELINE configs[4], one 20x20x20 robot).
// With the option tile_small also small populations of large robots (measured to pay only there: engine.hip).  Loop being tiled:
// CVX_Sim::Integrate, VX_Sim.cpp:1763-1933.
//
// A robot is cut into tiles (model.cpp plan_tiles: a grid of boxes with equal voxel counts).  A tile's workgroup
//   * OWNS n_own voxels: momenta in registers (thread t = owned voxel t), poses in its LDS pose tile;
//   * MIRRORS n_halo voxels: the far ends of the bonds that leave the tile, refreshed every step from the owners;
//   * evaluates every bond with at least one owned end, all three axes in ONE round (flat bond list, axis per lane at run
//     time: bond_compute_rt), history and mode bits resident in LDS.  A bond that crosses a tile boundary is evaluated by
//     both tiles from the same poses with the same code -- the same bits -- so forces never travel between tiles;
//   * sums each owned voxel's six bond forces from per-direction LDS planes (one writer per entry, no atomics) in the same
//     order as the fused kernel of the robot's size class, so a robot's trajectory does not depend on how it was tiled.
// What leaves the CU per step: the owned poses (8 doubles per voxel) and one max-|v|^2 word per tile, as self-validating
// 8-byte granules {32 data bits, 32-bit tag} (DBatch::xch, tile_mv), each one write-through store (global_store_dwordx2 sc1 =
// a relaxed agent-scope atomic store), read with L1-bypassing loads of the same kind.  A granule is never torn and its tag
// (launch generation, redo epoch, publication count) says which step it belongs to, so nothing needs ordering: no flag, no
// fence, no vmcnt drain (cdna_hip_programming.md Guideline 16, form R2).  A step n of a tile (five wavefronts: four of
// workers, one of SERVICE):
//   1. every worker wave polls the granules of ITS halo voxels (and of the contact partners the tile mirrors) until they carry
//      this step's tag -- the neighbours' previous voxel phase -- and writes the poses into LDS;      -> workgroup barrier
//   2. bond phase on the workers;                                                                     -> workgroup barrier
//   3. voxel phase on the workers: forces from the planes, contact partners from LDS, integration; the new pose goes out at
//      once (granules).  It runs SPECULATIVELY: the per-robot barrier of the step -- every tile has published its max |v|^2 of
//      the previous step, i.e. finished it -- is awaited meanwhile by the service wavefront, which reduces the words (the
//      replicated control block needs MaxVoxVel for the collision horizon, VX_Sim.cpp:1729-1755), learns whether a bond
//      diverged in the previous step, and decides whether the contact lists were due for a rebuild BEFORE this voxel phase
//      (UpdateCollisions precedes Integrate, VX_Sim.cpp:1100-1110);                                     -> workgroup barrier
//   4. the usual case -- no rebuild due -- commits: new poses into the pose tile, the tile's max |v|^2 goes out.  If a rebuild
//      was due (a few times per thousand steps), the voxel phase is REDONE from the kept momenta after the broad-phase
//      (the pose tile still holds the old poses); the redo epoch in the tags makes the neighbours ignore the poses of the
//      discarded attempt.  Steps with a whole-robot pass that is known in advance (IniCM latch, the end of a launch), and
//      robots whose lists are rebuilt every step (ColSystem without horizon), wait for the barrier before the voxel phase.
// The exchange buffers form a ring of three: step n writes P(n+1) over P(n-2), and every tile is past reading P(n-2) -- it
// published its max |v|^2 of step n-1, which this tile saw at the barrier of step n-1, after its own reads.
// A diverging bond (strain > 100, VX_Sim.cpp:1775) stops the robot BEFORE the voxel loop of that step; the other tiles learn
// of it at the next per-robot barrier: the voxel phase that ran meanwhile is rolled back (momenta kept a step, poses still in
// the ring); the bond history of a stopped robot is dead.
// Robots in a FLUID (template argument FLUID, round 5; fluid drag LW/VX_Sim.cpp:1516-1597).  A tile carries the part of the deformable
// surface mesh its owned voxels have facets on.  A mesh vertex is the mean of the corners of the up to seven voxels meeting in it --
// owned ones, halo ones, and diagonal neighbours that are neither -- and a corner needs its voxel's pose AND its directional strains of
// the previous step's bonds; so the owners publish the six strains with every pose (planes 16 .. 27 of the exchange buffer, same tags), and
// at the top of a step every lane fetches one of the voxels under the tile's vertices (DBatch::tile_mvox) into LDS: one protocol and one
// memory round trip whoever owns the voxel.  Then: vertices (corner-code order, as fused_drag and k_mesh_vertices), a workgroup barrier,
// the facets' drag next to the bond phase, and behind barrier (B) every voxel adds its facets in the reference's order.  Same arithmetic
// and orders as the resident / wide / streaming kernels: different tilings give the same bits (tests/test_gpu_tiled.py).
// The control block (time, stop rule, collision horizon) is replicated: every tile runs the same serial control code on the
// same inputs.  Residency: all tiles of a robot must be on the chip at the same time; the host sizes every launch to what the
// CUs admit and issues the tiled launches of an engine on ONE stream.  Spins are bounded (VXH_ROBOT_SYNC_TIMEOUT).
#pragma once

namespace vxh {

enum { VXH_TILE_SPIN_LIMIT = 1 << 22 };      // polls of one wait before the robot is given up (each costs a memory round trip: seconds)

typedef __attribute__((address_space(1))) unsigned long long gu64;
#define VXH_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// tag of the granules of publication `pub` (1 = the poses a launch starts from, i + 2 = those its i-th step produced) in redo
// epoch `ep` of launch generation `gen`; never 0, the value of never-written memory (pub >= 1)
__device__ __forceinline__ unsigned tile_tag(unsigned gen, unsigned ep, int pub) { return ((gen & 0xffu) << 24) | ((ep & 0x3fu) << 18) | ((unsigned)pub & 0x3ffffu); }

// a double as two granules {data, tag}: planes lo / hi are `stride` granules apart
__device__ __forceinline__ void st_gran2(unsigned long long* lo, size_t stride, double x, unsigned tag)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x), t = (unsigned long long)tag << 32;
    __hip_atomic_store((gu64*)lo, t | (b & 0xffffffffull), VXH_RLX_AGENT);
    __hip_atomic_store((gu64*)(lo + stride), t | (b >> 32), VXH_RLX_AGENT);
}
__device__ __forceinline__ unsigned long long ld_gran(const unsigned long long* p) { return __hip_atomic_load((const gu64*)p, VXH_RLX_AGENT); }
__device__ __forceinline__ double gran2_value(unsigned long long lo, unsigned long long hi)
{ return __longlong_as_double((long long)((lo & 0xffffffffull) | (hi << 32))); }


// Where the granules of exchange slot `xs` start in a buffer of the ring (round 6): the planes are BLOCKED by 64 slots -- [slot / 64][plane]
// [slot % 64] -- so plane k of a slot is k * 64 granules behind its plane 0: an immediate displacement of the load / store instead of one
// of sixteen 64-bit plane strides k * nx out of spilled scalar pairs.  A tile's owned voxels are one 64-aligned range of slots, so its
// publication still fills whole lines plane by plane, and a wavefront polling consecutive slots still reads them coalesced.
__device__ __forceinline__ const unsigned long long* xch_at(const unsigned long long* xq, unsigned planes, int xs)
{ return xq + (size_t)((unsigned)xs >> 6) * (planes * 64u) + ((unsigned)xs & 63u); }
__device__ __forceinline__ unsigned long long* xch_at(unsigned long long* xq, unsigned planes, int xs)
{ return xq + (size_t)((unsigned)xs >> 6) * (planes * 64u) + ((unsigned)xs & 63u); }

// position + scale of any voxel of the robot as its owner published it for the current step (broad-phase, latch, contact
// partners the tile does not mirror).  Called behind the step's per-robot barrier, when every tile has stored these granules;
// stored is not yet visible (nothing orders one wavefront's max-|v|^2 word behind another's pose granules), so this read, like
// every read of the exchange buffer, goes by the tags and repeats until they are this step's (bounded: `abort_flag`).
struct PoseFromXch {
    const unsigned long long* xq; unsigned planes;      // xq = exchange buffer of the step (xch_at), granule planes per slot
    const int* xslot;                                   // exchange slot of every voxel (DBatch::xslot)
    unsigned tag; int* abort_flag;
    __device__ __forceinline__ void at(int xs, double& x, double& y, double& z, double& s) const
    {
        unsigned long long g[8];
        for (int spins = 0;; ++spins) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = ld_gran(xch_at(xq, planes, xs) + k * 64);
#pragma unroll
            for (int k = 0; k < 8; ++k) ok = ok && (unsigned)(g[k] >> 32) == tag;
            if (ok) break;
            if (spins > VXH_TILE_SPIN_LIMIT) { *abort_flag = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        x = gran2_value(g[0], g[1]); y = gran2_value(g[2], g[3]); z = gran2_value(g[4], g[5]); s = gran2_value(g[6], g[7]);
    }
    __device__ __forceinline__ void operator()(int voxel_slot, double& x, double& y, double& z, double& s) const { at(xslot[voxel_slot], x, y, z, s); }
};

// contact partner of an owned voxel: from the tile's LDS when the broad-phase found it there (`code`: an owned voxel's index in
// the pose tile, or np + entry of the mirrored-partner planes px), else (-1) from the exchange buffer; -2 = the voxel itself
// (lanes without a partner in this round of the contact loop)
struct FetchTile {
    static constexpr bool USES_CODE = true;
    const double* ps; const double* px; int np, self;
    bool live;                              // the codes are valid (rows built by this kernel); else every partner comes from memory
    PoseFromXch remote;
    __device__ __forceinline__ bool in_memory(int code) const { return code == -1 || !live; }
    __device__ __forceinline__ void memory(int slot, double& x, double& y, double& z, double& s) const { remote(slot, x, y, z, s); }
    __device__ __forceinline__ void local(int code, double& x, double& y, double& z, double& s) const
    {
        const int l = code == -2 ? self : code;
        if (l < np) { x = ps[l]; y = ps[np + l]; z = ps[2 * np + l]; s = ps[3 * np + l]; }
        else { const int e = l - np; x = px[e]; y = px[VXH_TILE_XH + e]; z = px[2 * VXH_TILE_XH + e]; s = px[3 * VXH_TILE_XH + e]; }
    }
};

// The contact forces of an owned voxel (the head of voxel_update, same arithmetic: contact_force_add), from the tile's copy of
// its contact row in LDS (codes + stiffnesses at rc_code / rc_a1 [roff ..), every partner in LDS too), or, for a row that does
// not fit or has a partner the tile does not mirror, from the rows in memory like the other kernels (the memory requests of a
// round issued together, and only by wavefronts that have any).
__device__ __forceinline__ d3 tile_contacts(const DBatch& B, const DRobot& R, const FetchTile& fetch, d3 F, const VoxState& S, int v, int row, int ccnt, int roff,
                                            const int* rc_code, const double* rc_a1)
{
    if (ccnt <= 0) return F;
    if (roff >= 0) {
        for (int k0 = 0; k0 < ccnt; k0 += 4) {
            int code[4]; double a1[4], qx[4], qy[4], qz[4], qs[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const bool on = k0 + j < ccnt; code[j] = on ? rc_code[roff + k0 + j] : -2; a1[j] = on ? rc_a1[roff + k0 + j] : 0.0; }
#pragma unroll
            for (int j = 0; j < 4; ++j) fetch.local(code[j], qx[j], qy[j], qz[j], qs[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) if (code[j] != -2) F = contact_force_add(F, S.pos, S.scale, qx[j], qy[j], qz[j], qs[j], a1[j]);
        }
        return F;
    }
    for (int k0 = 0; k0 < ccnt; k0 += 4) {
        int o[4], code[4]; double a1[4], qx[4], qy[4], qz[4], qs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool on = k0 + j < ccnt;
            const size_t at = col_at(R, k0 + j, row);
            o[j] = on ? B.col_partner[at] : -1;
            a1[j] = on ? B.col_a1[at] : 0.0;
            code[j] = on ? B.col_code[at] : -2;
        }
        bool from_memory = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) from_memory = from_memory || (o[j] >= 0 && fetch.in_memory(code[j]));
        if (__any(from_memory)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) fetch.memory(o[j] < 0 ? v : o[j], qx[j], qy[j], qz[j], qs[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (!(o[j] >= 0 && fetch.in_memory(code[j]))) fetch.local(o[j] < 0 ? -2 : code[j], qx[j], qy[j], qz[j], qs[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (o[j] >= 0) F = contact_force_add(F, S.pos, S.scale, qx[j], qy[j], qz[j], qs[j], a1[j]);
    }
    return F;
}

// CalcL1Bonds (VX_Sim.cpp:2357-2413) for the surface voxels a tile owns (worker thread t: surface ordinal `i` of its voxel, -1 =
// none; every thread of the workgroup calls).  Like rebuild_rows (kernels.hpp) every row tests all surface voxels of the robot,
// staged through LDS in chunks of CH from the exchange buffer, and keeps the accepted ones in ascending order.  Differences:
//  * a chunk whose bounding box lies farther than the distance filter from the bounding box of the tile's own surface voxels is
//    skipped by the whole workgroup (every pair in it would fail the filter: exact);
//  * every accepted partner gets a CODE, the place where the voxel phase will find its position in LDS: the pose tile entry
//    of a partner the tile owns, or a mirrored-partner entry (DBatch::tile_xh) -- the partners owned by other tiles are
//    collected in an LDS hash set, numbered once the rows are complete, and the codes written in a second pass; beyond
//    VXH_TILE_XH distinct ones (or a full set): -1, fetched from memory.
// `sh`: 5 * CH doubles + CH ints; `hkey` / `hval`: VXH_TILE_HASH ints each.  Returns the thread's partner count.
__device__ __forceinline__ int tile_rebuild(const DBatch& B, const DRobot& R, DRobotState& rs, int ti, const PoseFromXch& pose,
                                            const double* ps, int np, int i, int* xh, int* s_xhn, double* s_box, double* sh, int* hkey, int* hval)
{
    constexpr int CH = VXH_TILE_CH, NT = VXH_TILE_THREADS, NW = VXH_TILE_THREADS / 64;
    const int tid = threadIdx.x;
    int* shv = (int*)(sh + 4 * CH);
    const bool mine = i >= 0;
    int vi = 0, cnt = 0;
    d3 pi = mk3(0, 0, 0); double si = 0;
    const int row = R.surf_begin + (mine ? i : 0);
    if (mine) { vi = B.surf[R.surf_begin + i]; pi = mk3(ps[tid], ps[np + tid], ps[2 * np + tid]); si = ps[3 * np + tid]; }
    for (int h = tid; h < VXH_TILE_HASH; h += NT) hkey[h] = 0;
    // bounding box of the tile's own surface voxels (s_box[0..5] = min xyz, max xyz), through two rounds of wave reductions
    {
        double lo[3] = {mine ? pi.x : 1e300, mine ? pi.y : 1e300, mine ? pi.z : 1e300}, hi[3] = {mine ? pi.x : -1e300, mine ? pi.y : -1e300, mine ? pi.z : -1e300};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const double l = __shfl_xor(lo[a], off), h = __shfl_xor(hi[a], off); lo[a] = l < lo[a] ? l : lo[a]; hi[a] = h > hi[a] ? h : hi[a]; }
        if ((tid & 63) == 0) { for (int a = 0; a < 3; ++a) { s_box[12 + (tid >> 6) * 6 + a] = lo[a]; s_box[12 + (tid >> 6) * 6 + 3 + a] = hi[a]; } }
        if (tid == 0) *s_xhn = 0;
        __syncthreads();
        if (tid < 3) {
            double l = s_box[12 + tid], h = s_box[12 + 3 + tid];
            for (int w = 1; w < NW; ++w) { l = fmin(l, s_box[12 + w * 6 + tid]); h = fmax(h, s_box[12 + w * 6 + 3 + tid]); }
            s_box[tid] = l; s_box[3 + tid] = h;
        }
        __syncthreads();
    }
    const double fd = vsqrt(R.filter_dist2) * (1.0 + 1e-12);      // (a hair more than the filter distance: the cull must never reject a pair the filter would pass)
    const double tlo[3] = {s_box[0] - fd, s_box[1] - fd, s_box[2] - fd}, thi[3] = {s_box[3] + fd, s_box[4] + fd, s_box[5] + fd};
    const double H = R.col_horizon;
    const unsigned long long* xrow = B.excl + R.excl_begin + (long long)(mine ? i : 0) * R.excl_wpr;
    for (int c0 = 0; c0 < R.nsurf; c0 += CH) {
        const int n = min(CH, R.nsurf - c0);
        double cx = 0, cy = 0, cz = 0;
        if (tid < n) {
            const int vj = B.surf[R.surf_begin + c0 + tid];
            double cs;
            pose(vj, cx, cy, cz, cs);
            sh[tid] = cx; sh[CH + tid] = cy; sh[2 * CH + tid] = cz; sh[3 * CH + tid] = cs; shv[tid] = vj;
        }
        if (tid < CH) {                       // the chunk's bounding box (CH = two wavefronts)
            double lo[3] = {tid < n ? cx : 1e300, tid < n ? cy : 1e300, tid < n ? cz : 1e300}, hi[3] = {tid < n ? cx : -1e300, tid < n ? cy : -1e300, tid < n ? cz : -1e300};
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { const double l = __shfl_xor(lo[a], off), h = __shfl_xor(hi[a], off); lo[a] = l < lo[a] ? l : lo[a]; hi[a] = h > hi[a] ? h : hi[a]; }
            if ((tid & 63) == 0) { for (int a = 0; a < 3; ++a) { s_box[12 + (tid >> 6) * 6 + a] = lo[a]; s_box[12 + (tid >> 6) * 6 + 3 + a] = hi[a]; } }
        }
        __syncthreads();
        bool far = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double l = fmin(s_box[12 + a], s_box[18 + a]), h = fmax(s_box[12 + 3 + a], s_box[18 + 3 + a]);
            far = far || l > thi[a] || h < tlo[a];
        }
        if (mine && !far) {
            const unsigned long long w0 = xrow[c0 >> 6], w1 = (c0 + 64 < R.nsurf) ? xrow[(c0 >> 6) + 1] : 0ull;   // chunk starts are multiples of 128
            for (int k0 = 0; k0 < n; k0 += 4) {
                double qx[4], qy[4], qz[4], qs[4], d2[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int k = min(k0 + u, n - 1); qx[u] = sh[k]; qy[u] = sh[CH + k]; qz[u] = sh[2 * CH + k]; qs[u] = sh[3 * CH + k]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) d2[u] = len2(pi - mk3(qx[u], qy[u], qz[u]));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + u, j = c0 + k;
                    if (k >= n || j == i) continue;
                    if (!(d2[u] < R.filter_dist2)) continue;
                    if (((k < 64 ? w0 : w1) >> (k & 63)) & 1ull) continue;          // !pV1->IsNearbyVox(SIndex2)
                    const double s1 = (j > i) ? si : qs[u];                          // scale of Vox1 = the earlier one, used twice (:2382)
                    const double act = H * (s1 + s1) * 0.5;
                    if (d2[u] < act * act) {
                        if (cnt < R.col_cap) {
                            const int vj = shv[k];
                            const DVoxClass& Ci = B.vclass_tab[R.vtab_begin + B.vclass[vi]];
                            const DVoxClass& Cj = B.vclass_tab[R.vtab_begin + B.vclass[vj]];
                            const size_t at = col_at(R, cnt, row);
                            B.col_partner[at] = vj;
                            B.col_a1[at] = (j > i) ? contact_a1(Ci, Cj) : contact_a1(Cj, Ci);
                            if (B.tile_of[vj] != ti) {                               // into the set of partners to mirror (linear probing)
                                unsigned h = ((unsigned)vj * 2654435761u) >> (32 - VXH_TILE_HASH_BITS);
                                for (int probe = 0; probe < 16; ++probe, h = (h + 1) & (VXH_TILE_HASH - 1)) {
                                    const int old = atomicCAS(&hkey[h], 0, vj + 1);
                                    if (old == 0 || old == vj + 1) break;
                                }
                            }
                        }
                        ++cnt;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (mine) {
        if (cnt > R.col_cap) { cnt = R.col_cap; atomicOr(&rs.col_overflow, 1); }
        B.col_cnt[row] = cnt;
    }
    for (int h = tid; h < VXH_TILE_HASH; h += NT) {               // number the distinct partners to mirror
        if (hkey[h] == 0) continue;
        const int e = atomicAdd(s_xhn, 1);
        if (e < VXH_TILE_XH) { xh[e] = B.xslot[hkey[h] - 1]; hval[h] = np + e; } else hval[h] = -1;
    }
    __syncthreads();
    if (mine) {                                                    // second pass: the codes of my row
        for (int k = 0; k < cnt; ++k) {
            const size_t at = col_at(R, k, row);
            const int vj = B.col_partner[at];
            int code = -1;
            if (B.tile_of[vj] == ti) code = B.tile_lidx[vj];
            else {
                unsigned h = ((unsigned)vj * 2654435761u) >> (32 - VXH_TILE_HASH_BITS);
                for (int probe = 0; probe < 16; ++probe, h = (h + 1) & (VXH_TILE_HASH - 1)) {
                    if (hkey[h] == vj + 1) { code = hval[h]; break; }
                    if (hkey[h] == 0) break;
                }
            }
            B.col_code[at] = code;
        }
    }
    if (tid == 0) { if (*s_xhn > VXH_TILE_XH) *s_xhn = VXH_TILE_XH; rs.col_tiled = 1; }
    __syncthreads();
    for (int e = tid; e < *s_xhn; e += NT) B.tile_xh[(size_t)ti * VXH_TILE_XH + e] = xh[e];   // for the launches to come
    if (tid == 0) B.tile_xhn[ti] = *s_xhn;
    return cnt;
}

// developer instrumentation (library built with -DVXH_PHASE_TIMING): per-wave cycle sums of the phases of a step
#ifdef VXH_PHASE_TIMING
#define VXH_TT_DECL unsigned long long tt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tt_last = __builtin_readcyclecounter();
#define VXH_TT_MARK(k) { const unsigned long long tt_now = __builtin_readcyclecounter(); tt_acc[k] += tt_now - tt_last; tt_last = tt_now; }
#define VXH_TT_FLUSH if (B.prof && (threadIdx.x & 63) == 0) { for (int k = 0; k < 8; ++k) atomicAdd(&B.prof[(threadIdx.x >> 6) * 8 + k], tt_acc[k]); }
// ... and, at step VXH_TS_STEP of a launch, the real-time counter (100 MHz, common to all CUs) at the boundaries of the step, per tile
#define VXH_TS_STEP 100
// ... and cycle sums of the parts of the voxel phase, first thread of every tile (slots 2140 ..)
#define VXH_TV_BEGIN unsigned long long t_tv = __builtin_readcyclecounter();
#define VXH_TV(slot) { const unsigned long long t_now = __builtin_readcyclecounter(); if (B.prof && tid == 0) atomicAdd(&B.prof[2140 + (slot)], t_now - t_tv); t_tv = t_now; }
#define VXH_TS(k, who) if (B.prof && it == VXH_TS_STEP && (who) && ti < 256) B.prof[128 + ti * 8 + (k)] = __builtin_amdgcn_s_memrealtime();
#else
#define VXH_TT_DECL
#define VXH_TT_MARK(k)
#define VXH_TT_FLUSH
#define VXH_TS(k, who)
#define VXH_TV_BEGIN
#define VXH_TV(slot)
#endif


template <bool TABG, bool MESH, bool FLUID = false, bool SMALL = false>      // FLUID (implies MESH): the robots of the launch are in a fluid (drag mesh per tile, strains exchanged)
__global__ __launch_bounds__(VXH_TILE_THREADS) void k_tile_steps(DBatch B, const DRobot* __restrict__ robots, const DTile* __restrict__ tiles,
                                                                 const int* __restrict__ tile_list, long long step_cap, int iters, unsigned gen)
{
    constexpr int BLOCK = VXH_TILE_BLOCK, NT = VXH_TILE_THREADS;   // worker threads (one per owned voxel), all threads
    extern __shared__ __align__(16) double lds[];
    __shared__ DRobotState rs, rs_bak, rs_bak2;
    __shared__ FusedCtl s_ctl[2];
    __shared__ unsigned long long s_mvbits;
    __shared__ double s_box[12 + 6 * (NT / 64)];
    __shared__ double s_dtp[2];               // rs.dt_prev as the control of step `it` left it (slot it & 1): what the horizon update of that step needs, while the
                                              // next step's control, on another wavefront, already overwrites the field
    __shared__ int s_div, s_divprev, s_abort, s_xhn, s_pool, s_done, s_snap;
    static_assert(3 * sizeof(DRobotState) + 2 * sizeof(FusedCtl) + sizeof(double) + sizeof(double) * (12 + 6 * (NT / 64)) + 2 * sizeof(double) + 7 * sizeof(int) + 32 <= VXH_TILE_STATIC_LDS, "static LDS bound");
    static_assert(sizeof(DRobotState) % 4 == 0 && sizeof(DRobotState) / 4 <= 64, "a wavefront copies the control block one dword per lane");

    const int tid = threadIdx.x;
    // Two wavefronts also SERVE the robot (round 6; rounds 2-5: a fifth wavefront): the last one (it holds voxels only in tiles of more than
    // 192) resolves the per-robot barrier and takes the collision-horizon decision; the one before it -- when it holds no voxel either,
    // else the last one again -- runs the serial control of the NEXT step; both beside the voxel phase of the wavefronts that hold voxels.
    constexpr int SVC0 = BLOCK - 64;
    const bool svc = tid >= SVC0;
    const int ti = __builtin_amdgcn_readfirstlane(tile_list[blockIdx.x]);
    const DTile& T = tiles[ti];
    const int r = T.robot;
    const DRobot& R = robots[r];
    const unsigned nv = B.nv;                 // (the SoA macros)
    (void)nv;
    const int n_own = T.n_own, n_halo = T.n_halo, nb = T.nb, k_tiles = T.ntiles;
    const int nbd = TABG ? 0 : R.n_bclass * (int)(sizeof(DBondClass) / 8), nvd = TABG ? 0 : R.n_vclass * (int)(sizeof(DVoxClass) / 8);
    // a robot in a FLUID (round 5): the tile carries the part of the drag mesh its owned voxels have facets on (DTile::n_mv / n_f)
    static_assert(MESH || !FLUID, "a robot in a fluid is a land_water robot");
    const bool fluid = FLUID && T.n_f > 0;
    const int n_mv = FLUID ? T.n_mv : 0, n_f = FLUID ? T.n_f : 0, n_mx = FLUID ? T.n_mx : 0;
    // SMALL (every tile of the launch within VXH_TILE_S_*): the layout of the LARGEST such tile for all of them -- np, no, nbp and every
    // offset in front of the class tables are compile-time constants, i.e. immediates of the LDS instructions instead of scalar registers
    // (the kernel spilled ~250 of those into VGPR lanes and read them back ~250 times per step and wavefront)
    const TileLayout L = SMALL ? tile_layout(VXH_TILE_S_OWN, VXH_TILE_S_HALO, VXH_TILE_S_BONDS, nbd + nvd, MESH, n_mv, n_f, n_mx)
                               : tile_layout(n_own, n_halo, nb, nbd + nvd, MESH, n_mv, n_f, n_mx);
    const int np = L.np, no = L.no, nbp = L.nbp;
    double* const ps = lds + L.o_ps;          // [8][np] pose tile: owned voxels, then halo voxels
    double* const pl = lds + L.o_pl;          // [6 directions][6][no] bond force / minus bond moment on every owned voxel
    double* const hl = lds + L.o_hl;          // [6][nbp] bond history
    double* const pht = lds + L.o_pht;        // [2][no] sin / cos of the actuation phase offsets
    double* const sl = lds + L.o_sl;          // [6][no] land_water robots: the directional strains of my voxels (CurStrainV1/V2 of their bonds, SetStrainDir)
    constexpr bool mesh = MESH;               // land_water robots: the strains are part of their state (FLUID: and travel with the poses)
    double* const sc = lds + L.o_sc;          // scratch of latch / broad-phase
    double* const px = lds + L.o_px;          // [4][VXH_TILE_XH] position + scale of the mirrored contact partners
    double* const rc_a1 = lds + L.o_rc;       // [VXH_TILE_ROWPOOL] the tile's contact rows: pair stiffnesses ...
    double* const tabs = lds + L.o_tab;
    int* const bent = (int*)(lds + L.o_int);  // [nbp] packed bond entries
    int* const bcls = bent + nbp;             // [nbp] bond classes
    int* const hf = bcls + nbp;               // [nbp] mode bits (bit 0 SmallAngle, bit 1 history layout)
    int* const xh = hf + nbp;                 // [VXH_TILE_XH] contact partners owned by other tiles, mirrored in px (their exchange slots)
    int* const hkey = xh + VXH_TILE_XH;       // [VXH_TILE_HASH] broad-phase: set of the partners to mirror
    int* const hval = hkey + VXH_TILE_HASH;
    int* const rc_code = hval + VXH_TILE_HASH;   // [VXH_TILE_ROWPOOL] ... and partner codes
    double* const msh = lds + L.o_mesh;       // fluid: [3][nmvp] my mesh vertices of this step
    double* const fdr = msh + 3 * L.nmvp;     // fluid: [3][nfp] drag of my facets
    double* const spd = fdr + 3 * L.nfp;      // fluid: [3][no] velocities of my voxels at the start of the step
    double* const mst = spd + 3 * L.no;       // fluid: [13][nmxp] position, quaternion, six strains of every voxel my vertices average over
    double* const mv0 = mst + 13 * L.nmxp;    // fluid: [3][nmvp] rest positions of my vertices (constant tables: copied once per launch)
    int* const mvt = (int*)(mv0 + 3 * L.nmvp);   // fluid: [8][nmvp] per vertex and corner code, the voxel touching it there (index into mst) or -1
    int* const fct = mvt + 8 * L.nmvp;        // fluid: [4][nfp] per facet: owner voxel, its three vertices
    const int nmvp = L.nmvp, nfp = L.nfp, nmxp = L.nmxp;
    const unsigned nx = B.nx;                 // exchange slots (every tile's owned voxels contiguous: its pose stores fill whole lines)
    const unsigned xpl = (unsigned)B.xplanes;
    const size_t xbuf = (size_t)B.xplanes * nx;   // granules per exchange buffer (a ring of three): 16 planes of poses (+ 12 of strains when a tiled robot is in a fluid)
    constexpr bool xstrain = FLUID;           // my voxels' strains travel with their poses
    const int xs_own = T.xoff + tid;          // exchange slot of my owned voxel
    const size_t mvbuf = (size_t)VXH_TILE_MV_STRIDE * B.n_tiles;

    if (tid == 0) { rs = B.rstate[r]; s_div = 0; s_divprev = 0; s_abort = 0; s_pool = 0; s_done = 0; s_snap = 0; s_mvbits = 0; s_xhn = B.rstate[r].col_tiled ? B.tile_xhn[ti] : 0; }
    bool codes_live = B.rstate[r].col_tiled != 0;      // contact rows built by this kernel (else: every partner from memory until the next broad-phase)
    for (int e = tid; e < VXH_TILE_XH; e += NT) xh[e] = B.tile_xh[(size_t)ti * VXH_TILE_XH + e];
    const DBondClass* bct;
    const DVoxClass* vct;
    if constexpr (TABG) {
        bct = B.bclass_tab + R.btab_begin;
        vct = B.vclass_tab + R.vtab_begin;
    } else {
        for (int k = tid; k < nbd; k += NT) tabs[k] = ((const double*)(B.bclass_tab + R.btab_begin))[k];
        for (int k = tid; k < nvd; k += NT) tabs[nbd + k] = ((const double*)(B.vclass_tab + R.vtab_begin))[k];
        bct = (const DBondClass*)tabs;
        vct = (const DVoxClass*)(tabs + nbd);
    }
    for (int b = tid; b < nb; b += NT) {
        bent[b] = B.tile_bond[T.bond_off + b];
        bcls[b] = B.tile_bcls[T.bond_off + b];
        const int slot = B.tile_bslot[T.bond_off + b];
        hf[b] = B.small_angle[slot] & 3;
#pragma unroll
        for (int k = 0; k < 6; ++k) hl[k * nbp + b] = HIST(k, slot);
    }
    for (int k = tid; k < 36 * no; k += NT) pl[k] = 0.0;         // directions without a bond stay zero for the whole launch

    // ---- this thread's owned voxel: momenta -> registers, pose -> LDS and -> the exchange buffer of the current step
    const int n0 = B.rstate[r].steps;                              // steps taken before this launch (every tile reads the same)
    unsigned ring = (unsigned)n0 % 3u;                             // ring slot of the poses of the current step
    unsigned ep = 0;                                               // redo epoch (voxel phases redone in this launch so far)
    const bool valid = tid < n_own;
    const int gv = valid ? B.tile_vox[T.vox_off + tid] : R.vox_begin;
    const DVoxClass& C = vct[valid ? B.vclass[gv] : 0];
    int row = -1;
    int my_ord = -1;                          // my ordinal in the robot's surface list (broad-phase)
    float amp_damp = 1.f;
    d3 lm = mk3(0, 0, 0), am = mk3(0, 0, 0);
    if (valid) {
        const int b0 = n0 & 1;
        if (R.flags & RF_SELF_COL) { my_ord = B.surf_ord[gv]; if (my_ord >= 0) row = R.surf_begin + my_ord; }
        amp_damp = B.amp_damp[gv];
        pht[tid] = B.act_sb[gv]; pht[no + tid] = B.act_cb[gv];
        if constexpr (mesh) {
#pragma unroll
            for (int k = 0; k < 6; ++k) sl[k * no + tid] = B.strain[(unsigned)k * nv + (unsigned)gv];
        }
        lm = mk3(LINMOM(0, gv), LINMOM(1, gv), LINMOM(2, gv));
        am = mk3(ANGMOM(0, gv), ANGMOM(1, gv), ANGMOM(2, gv));
        const double q8[8] = {POS(b0, 0, gv), POS(b0, 1, gv), POS(b0, 2, gv), SCALE(b0, gv), QUAT(0, gv), QUAT(1, gv), QUAT(2, gv), QUAT(3, gv)};
        unsigned long long* const xq0 = xch_at(B.xch + (size_t)ring * xbuf, xpl, xs_own);
        const unsigned tag1 = tile_tag(gen, 0, 1);
#pragma unroll
        for (int k = 0; k < 8; ++k) { ps[k * np + tid] = q8[k]; st_gran2(xq0 + (2 * k) * 64, 64, q8[k], tag1); }
        if constexpr (xstrain) {
#pragma unroll
            for (int k = 0; k < 6; ++k) st_gran2(xq0 + (16 + 2 * k) * 64, 64, sl[k * no + tid], tag1);
        }
    }
    if constexpr (FLUID) {
        for (int i = tid; i < n_mv; i += NT) {
            const size_t at = (size_t)T.mv_off + i;
#pragma unroll
            for (int c = 0; c < 3; ++c) mv0[c * nmvp + i] = B.tile_mv0[(size_t)c * B.n_tmv + at];
#pragma unroll
            for (int c = 0; c < 8; ++c) mvt[c * nmvp + i] = B.tile_mvert[(size_t)c * B.n_tmv + at];
        }
        for (int f = tid; f < n_f; f += NT) {
#pragma unroll
            for (int c = 0; c < 4; ++c) fct[c * nfp + f] = B.tile_facet[(size_t)c * B.n_tf + T.f_off + f];
        }
    }
    int my_ff = 0, my_fc = 0;                 // fluid: my voxel's facets among the tile's
    if (valid && fluid) { my_ff = B.tile_ffirst[T.vox_off + tid]; my_fc = (int)B.tile_fcount[T.vox_off + tid]; }
    // The voxel update on TWO wavefronts (round 6) when one wavefront holds all the tile's voxels (the 4x4x4 tiles of configs[4], the tiles
    // of a swimmer above 1024 voxels): translation -- force sum, contacts, floor, linear integration: voxel_update_lin -- on thread i,
    // rotation + actuation -- moment sum, quaternion update, new scale: voxel_update_ang -- on thread 64 + i, whose wavefront would idle;
    // the two halves never read each other's results within a step (voxel_update is one behind the other: same operations, same bits) and
    // each keeps its momentum in registers and publishes its part of the pose.  (Round 3 measured this split slower -- with 236-368 bytes of
    // scratch per lane, whose reloads queued behind the publication's write-through stores; the kernel has no scratch any more.)
    const bool split = n_own <= 64;
    const int av = split ? tid - 64 : tid;                         // the voxel whose rotation half this thread runs
    const bool do_ang = split ? (tid >= 64 && av < n_own) : valid;
    const int gva = do_ang ? (split ? B.tile_vox[T.vox_off + av] : gv) : R.vox_begin;
    const DVoxClass& Ca = vct[do_ang ? B.vclass[gva] : 0];
    if (split) {
        am = mk3(0, 0, 0);
        if (do_ang) { am = mk3(ANGMOM(0, gva), ANGMOM(1, gva), ANGMOM(2, gva)); amp_damp = B.amp_damp[gva]; }
    }
    const int xs_ang = T.xoff + av;           // exchange slot of that voxel
    d3 lm_bak = lm, am_bak = am;              // momenta before the last committed voxel phase (a diverged step is undone)
    // the halo voxels are spread over the worker threads from the last one down: wave 0, which owns the first voxels, gets them last
    const int hslot0 = BLOCK - 1 - tid;
    const int hv0 = hslot0 < n_halo ? B.tile_vox[T.vox_off + n_own + hslot0] : 0;      // my (first) halo voxel: its exchange slot
    __syncthreads();
    const int nvw = (n_own + 63) >> 6;        // wavefronts that hold voxels
    const int ctl0 = n_own <= SVC0 - 64 ? SVC0 - 64 : SVC0;
    const bool ctl_wave = (tid & ~63) == ctl0, ctl_thread = tid == ctl0;      // the wavefront of the next step's control, its first lane
    const bool hz_thread = tid == SVC0;       // the lane that takes the horizon decision: first of the wavefront that resolves the per-robot barrier
    // snapshots of the control block by the whole control wavefront, a dword per lane (one LDS round trip instead of ten on one lane)
    auto snapshot = [&](bool two) {
        const int l = tid - ctl0;
        if (l < (int)(sizeof(DRobotState) / 4)) {
            const int a = ((const int*)&rs_bak)[l], b = ((const int*)&rs)[l];
            if (two) ((int*)&rs_bak2)[l] = a;
            ((int*)&rs_bak)[l] = b;
        }
    };
    if (tid == 0)   // the pending max |v|^2 of the last step before this launch travels like a step's: every tile publishes the robot-wide value
        st_gran2(B.tile_mv + (size_t)ring * mvbuf + (size_t)ti * VXH_TILE_MV_STRIDE, 1, __longlong_as_double((long long)rs.maxvel2_bits), tile_tag(gen, 0, 1));
    if (ctl_thread) { fused_control_begin(R, rs, step_cap, iters > 0, s_ctl[0]); s_dtp[0] = rs.dt_prev; }
    // my contact row: its length, and a copy in LDS when the rows were built by this kernel and fit
    int ccnt = 0, roff = -1;
    auto rows_to_lds = [&]() {                // (every thread calls: workgroup barriers inside)
        if (tid == 0) s_pool = 0;
        __syncthreads();
        roff = -1;
        if (valid && codes_live && ccnt > 0) {
            bool all_local = true;
            for (int k = 0; k < ccnt; ++k) all_local = all_local && B.col_code[col_at(R, k, row)] != -1;
            if (all_local) {
                const int off = atomicAdd(&s_pool, ccnt);
                if (off + ccnt <= VXH_TILE_ROWPOOL) {
                    roff = off;
                    for (int k = 0; k < ccnt; ++k) { const size_t at = col_at(R, k, row); rc_code[off + k] = B.col_code[at]; rc_a1[off + k] = B.col_a1[at]; }
                }
            }
        }
        __syncthreads();
    };
    if (valid) ccnt = row >= 0 ? B.col_cnt[row] : 0;
    rows_to_lds();
    // robots whose lists are rebuilt every step (no horizon logic) cannot speculate on "no rebuild due"
    const bool can_speculate = !(R.flags & RF_SELF_COL) || (R.flags & RF_HORIZON_COL) != 0;

    // The serving wavefront's part of a step: wait until every tile of the robot has published its max |v|^2 for this step (= has
    // finished the previous one), reduce, note a divergence, else take the collision-horizon decision of the step -- UpdateCollisions,
    // VX_Sim.cpp:1729-1755; the arithmetic of step_control_horizon (kernels.hpp), operation for operation.  What the decision reads of the
    // control block is fetched BEFORE the wait (nobody else writes those fields between two workgroup barriers) and what it leaves is stored
    // without being read back: behind the last poll stand a wave reduction, a square root, a division and stores -- the workgroup waits at
    // barrier (C) for exactly this lane (round 6; before: a dozen dependent LDS round trips, ~1 us).  `spec`: the step's snapshot of the
    // control block (taken by the control wavefront right behind barrier (B)) gets the same values.
    constexpr int MVC = VXH_TILE_MAX_TILES / 64;
    auto robot_barrier = [&](const unsigned long long* mvq, unsigned tag, bool go, FusedCtl& K, int it, bool spec) {
        const int lane = tid - SVC0;
        // TWO requests of the words in flight, half a round trip apart (round 6): a word that arrives just behind one request is seen by the
        // other a quarter of a microsecond later instead of a whole round trip later.  Every lane always asks (a lane beyond the robot's tiles
        // for the last tile's word again): without a branch around the loads the compiler can wait for the OLDER request alone.
        unsigned long long mg[2 * MVC], mh[2 * MVC];
        auto request = [&](unsigned long long (&m)[2 * MVC]) {
#pragma unroll
            for (int c = 0; c < MVC; ++c) {
                const unsigned long long* q = mvq + (size_t)min(lane + 64 * c, k_tiles - 1) * VXH_TILE_MV_STRIDE;
                m[2 * c] = ld_gran(q); m[2 * c + 1] = ld_gran(q + 1);
            }
        };
        auto all_in = [&](const unsigned long long (&m)[2 * MVC]) {
            bool ok = true;
#pragma unroll
            for (int c = 0; c < MVC; ++c) ok = ok && (unsigned)(m[2 * c] >> 32) == tag && (unsigned)(m[2 * c + 1] >> 32) == tag;
            return __all(ok) != 0;
        };
        request(mg);
        // (while the words travel) the decision's inputs
        const double pre_disp = rs.max_disp, pre_dtp = s_dtp[it & 1], lat = R.lat, half_reach = (R.col_horizon - 1.0) / 2;
        const int pre_reb = rs.rebuilds, pre_ct = rs.col_tiled, kflags = K.flags;
        const bool self_col = (R.flags & RF_SELF_COL) != 0, by_horizon = (R.flags & RF_HORIZON_COL) != 0;
        bool aborted = __hip_atomic_load(&s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
        if (spec && hz_thread)     // (the snapshot this lane will patch is in place: the control wavefront took it right behind barrier (B); the flag makes it certain)
            for (int w = 0; __hip_atomic_load(&s_snap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != it + 1 && w < 4096; ++w) __builtin_amdgcn_s_sleep(1);
        int spins = 0;
#ifdef VXH_PHASE_TIMING
        int polls = 0;
        const unsigned long long tp0 = __builtin_readcyclecounter();
#endif
        for (;;) {
#ifdef VXH_PHASE_TIMING
            polls += 2;
#endif
            request(mh);                       // the second request goes out while the first is on its way
            if (all_in(mg)) break;
            request(mg);
            if (all_in(mh)) {
#pragma unroll
                for (int c = 0; c < 2 * MVC; ++c) mg[c] = mh[c];
                break;
            }
            if (++spins > VXH_TILE_SPIN_LIMIT) { s_abort = 1; aborted = true; break; }
        }
#ifdef VXH_PHASE_TIMING
        if (B.prof && hz_thread) { atomicAdd(&B.prof[120], (unsigned long long)polls); atomicAdd(&B.prof[121], 1ull); atomicAdd(&B.prof[122], __builtin_readcyclecounter() - tp0); }
#endif
        double mvmax = 0.0;                   // max |v|^2 over the tiles; a negative word marks a tile in which a bond diverged
        bool neg = false;
#pragma unroll
        for (int c = 0; c < MVC; ++c) { const double m = gran2_value(mg[2 * c], mg[2 * c + 1]); neg = neg || m < 0.0; mvmax = m > mvmax ? m : mvmax; }      // (clamped lanes repeat the last tile's word: harmless to a maximum)
        neg = __any(neg) != 0;
        mvmax = wave_max_nonneg(mvmax);
        if (hz_thread && !aborted) {
            s_divprev = neg ? 1 : 0;
            if (!neg) {
                unsigned long long mvb = (unsigned long long)__double_as_longlong(mvmax);
                double disp = pre_disp;
                int reb = pre_reb, ct = pre_ct;
                if (go) {
                    int rebuild = 0;
                    if (self_col) {                  // (fused_control_horizon -> step_control_horizon with c.go = 1)
                        const double mv = vsqrt_nn(mvmax);
                        disp += fabs(vdiv(mv * pre_dtp, lat));
                        mvb = 0ull;
                        if (!by_horizon || disp > half_reach) { rebuild = 1; disp = 0.0; reb += 1; ct = 0; }
                    }
                    rs.rebuild_now = rebuild; K.rebuild = rebuild;
                    if (rebuild) K.flags = kflags | 8;
                    if (spec) rs_bak.rebuild_now = rebuild;
                }
                rs.maxvel2_bits = mvb; rs.max_disp = disp; rs.rebuilds = reb; rs.col_tiled = ct;
                if (spec) { rs_bak.maxvel2_bits = mvb; rs_bak.max_disp = disp; rs_bak.rebuilds = reb; rs_bak.col_tiled = ct; }   // the snapshot of this step carries the horizon update as well
            } else if (spec) { const DRobotState two = rs_bak2; rs_bak = two; }      // the robot stopped a step ago: back two snapshots instead of one
        }
        return neg;
    };

    VXH_TT_DECL
    for (int it = 0;; ++it) {
        FusedCtl& K = s_ctl[it & 1];
        FusedCtl& Knext = s_ctl[(it + 1) & 1];
        VXH_TT_MARK(6)
        VXH_TS(0, tid == 0)
#ifdef VXH_PHASE_TIMING
        if (B.prof && it == VXH_TS_STEP + 1 && tid == 0 && ti < 256) B.prof[128 + ti * 8 + 6] = __builtin_amdgcn_s_memrealtime();   // (the next step's top, in the place of "control done")
#endif
        const unsigned ringn = ring == 2 ? 0 : ring + 1, ringp = ring == 0 ? 2 : ring - 1;
        const unsigned long long* const xq = B.xch + (size_t)ring * xbuf;      // poses at the start of this step
        unsigned long long* const xqn = B.xch + (size_t)ringn * xbuf;          // ... of the next one
        const unsigned tag = tile_tag(gen, ep, it + 1);
        const bool go = K.go != 0;
        // does this step's voxel phase run ahead of the per-robot barrier?  Not when a whole-robot pass is known to come (latch),
        // not at the end of the launch (the control block written back needs the reduced max |v|^2)
        const bool speculate = can_speculate && go && !K.latch && !K.eol && !K.trace;
        int spins = 0;
        const unsigned long long* const mvq = B.tile_mv + (size_t)ring * mvbuf + (size_t)T.tile0 * VXH_TILE_MV_STRIDE;
        // ---- 1. halo poses of this step: every worker wave waits for the granules of its own halo voxels
        if (go) {
            for (int h = hslot0; h < n_halo; h += BLOCK) {
                const int hv = h == hslot0 ? hv0 : B.tile_vox[T.vox_off + n_own + h];
                unsigned long long g[16];
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 16; ++k) g[k] = ld_gran(xch_at(xq, xpl, hv) + k * 64);
#pragma unroll
                    for (int k = 0; k < 16; ++k) ok = ok && (unsigned)(g[k] >> 32) == tag;
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > VXH_TILE_SPIN_LIMIT) { s_abort = 1; break; }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) ps[k * np + n_own + h] = gran2_value(g[2 * k], g[2 * k + 1]);
            }
            // ... and the contact partners the tile mirrors (position + scale: 8 granules), same protocol
            for (int e = hslot0; e < s_xhn; e += BLOCK) {
                const int hv = xh[e];
                unsigned long long g[8];
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 8; ++k) g[k] = ld_gran(xch_at(xq, xpl, hv) + k * 64);
#pragma unroll
                    for (int k = 0; k < 8; ++k) ok = ok && (unsigned)(g[k] >> 32) == tag;
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > VXH_TILE_SPIN_LIMIT) { s_abort = 1; break; }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) px[k * VXH_TILE_XH + e] = gran2_value(g[2 * k], g[2 * k + 1]);
            }
        }
        if constexpr (FLUID) {
            if (go && fluid) {
                // ---- 1b. fluid drag, staging: pose and strains (of the PREVIOUS step's bonds) of every voxel my part of the drag mesh averages
                // over -- mine, a neighbour tile's, or a diagonal neighbour's that is no halo voxel: ONE protocol, the exchange buffer of this
                // step -- one voxel per lane, one memory round trip for all of them
                if (valid) { const d3 sp = lm * C.mass_inv; spd[tid] = sp.x; spd[no + tid] = sp.y; spd[2 * no + tid] = sp.z; }
                for (int j = tid; j < n_mx; j += BLOCK) {
                    const int xs = B.tile_mvox[T.mx_off + j];
                    unsigned long long g[26];
                    for (;;) {
                        bool ok = true;
#pragma unroll
                        for (int k = 0; k < 6; ++k) g[k] = ld_gran(xch_at(xq, xpl, xs) + k * 64);                       // position
#pragma unroll
                        for (int k = 0; k < 20; ++k) g[6 + k] = ld_gran(xch_at(xq, xpl, xs) + (8 + k) * 64);            // quaternion, strains
#pragma unroll
                        for (int k = 0; k < 26; ++k) ok = ok && (unsigned)(g[k] >> 32) == tag;
                        if (ok) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > VXH_TILE_SPIN_LIMIT) { s_abort = 1; break; }
                    }
#pragma unroll
                    for (int k = 0; k < 13; ++k) mst[k * nmxp + j] = gran2_value(g[2 * k], g[2 * k + 1]);
                }
            }
        }
        if (tid == 0) s_div = 0;
        __syncthreads();
        VXH_TT_MARK(0)
        VXH_TS(1, tid == 0)
        const bool damp_on = K.damp_on != 0;
        if constexpr (FLUID) {
            if (go && fluid) {                 // (uniform over the workgroup: its barrier)
                // ---- 1c. the surface mesh (LW/VX_MeshUtil.cpp:388-428, as k_mesh_vertices / fused_drag): every vertex my facets use = mean over
                // the voxels touching that lattice corner of Pos + R(Angle) * corner offset, summed in corner-code order
                {
                    const double nom = R.lat;
                    for (int i = tid; i < n_mv; i += BLOCK) {
                        d3 part = mk3(0, 0, 0);
                        int count = 0;
#pragma unroll
                        for (int corner = 0; corner < 8; ++corner) {
                            const int j = mvt[corner * nmvp + i];
                            if (j < 0) continue;
                            const double* q = mst + j;
                            const double hx = (1 + q[(7 + ((corner & 4) ? 0 : 3)) * nmxp]) * nom * 0.5;      // CornerPosCur / CornerNegCur
                            const double hy = (1 + q[(7 + ((corner & 2) ? 1 : 4)) * nmxp]) * nom * 0.5;
                            const double hz = (1 + q[(7 + ((corner & 1) ? 2 : 5)) * nmxp]) * nom * 0.5;
                            const RotFwd M(mkq(q[3 * nmxp], q[4 * nmxp], q[5 * nmxp], q[6 * nmxp]));
                            part = part + (mk3(q[0], q[nmxp], q[2 * nmxp]) + M(mk3((corner & 4) ? hx : -hx, (corner & 2) ? hy : -hy, (corner & 1) ? hz : -hz)));
                            ++count;
                        }
                        const double inv = vrcp((double)count);
                        const d3 v0 = mk3(mv0[i], mv0[nmvp + i], mv0[2 * nmvp + i]);
                        const d3 npos = part * inv;
                        const d3 now = v0 + (npos - v0);                 // v + DrawOffset, as the reference stores it
                        msh[i] = now.x; msh[nmvp + i] = now.y; msh[2 * nmvp + i] = now.z;
                    }
                }
                __syncthreads();
            }
        }

        // ---- 2. bond phase, every bond with an owned end, all axes in one round
        bool div = false;
        if (go && !s_abort) {
            if constexpr (FLUID) {
                if (fluid) {
                    // ---- 2a. fluid drag of my facets (LW/VX_Sim.cpp:1516-1597; facet_drag_force), each on its voxel's velocity at the start of the step
                    for (int f = tid; f < n_f; f += BLOCK) {
                        const int u = fct[f], ia = fct[nfp + f], ib = fct[2 * nfp + f], ic = fct[3 * nfp + f];
                        const d3 speed = mk3(spd[u], spd[no + u], spd[2 * no + u]);
                        const d3 contrib = facet_drag_force(speed, normalized3(speed), mk3(msh[ia], msh[nmvp + ia], msh[2 * nmvp + ia]),
                                                            mk3(msh[ib], msh[nmvp + ib], msh[2 * nmvp + ib]), mk3(msh[ic], msh[nmvp + ic], msh[2 * nmvp + ic]), R.drag_coef);
                        fdr[f] = contrib.x; fdr[nfp + f] = contrib.y; fdr[2 * nfp + f] = contrib.z;
                    }
                }
            }
            for (int b = tid; b < nb; b += BLOCK) {
                const int e = bent[b];
                const int l1 = e & 1023, l2 = (e >> 10) & 1023, axis = (e >> 20) & 3;
                BondHist H;
                H.p0 = hl[b]; H.p1 = hl[nbp + b]; H.p2 = hl[2 * nbp + b]; H.g0 = hl[3 * nbp + b]; H.g1 = hl[4 * nbp + b]; H.g2 = hl[5 * nbp + b];
                H.flags = (unsigned)hf[b];
                H.store_hist = false;
                const d3 p1 = mk3(ps[l1], ps[np + l1], ps[2 * np + l1]);
                const double s1 = ps[3 * np + l1];
                const dq q1 = mkq(ps[4 * np + l1], ps[5 * np + l1], ps[6 * np + l1], ps[7 * np + l1]);
                const d3 p2 = mk3(ps[l2], ps[np + l2], ps[2 * np + l2]);
                const double s2 = ps[3 * np + l2];
                const dq q2 = mkq(ps[4 * np + l2], ps[5 * np + l2], ps[6 * np + l2], ps[7 * np + l2]);
                const BondOut o = bond_compute_rt(axis, B, bct[bcls[b]], H, p1, q1, s1, p2, q2, s2, damp_on);
                if (H.store_hist) { hl[b] = H.p0; hl[nbp + b] = H.p1; hl[2 * nbp + b] = H.p2; hl[3 * nbp + b] = H.g0; hl[4 * nbp + b] = H.g1; hl[5 * nbp + b] = H.g2; }
                hf[b] = (int)H.flags;
                div = div || o.diverged;
                if constexpr (mesh) {        // SetStrainDir (VXS_BondInternal.cpp:300-304): +axis side of voxel 1, -axis side of voxel 2; a bond that crosses a
                                   // tile boundary is evaluated by both tiles, each keeps the strain of its own end
                    if (l1 < n_own) sl[axis * no + l1] = o.strain1;
                    if (l2 < n_own) sl[(3 + axis) * no + l2] = o.strain2;
                }
                if (l1 < n_own) {
                    double* e1 = pl + (2 * axis) * 6 * no + l1;
                    e1[0] = o.f1.x; e1[no] = o.f1.y; e1[2 * no] = o.f1.z; e1[3 * no] = -o.m1.x; e1[4 * no] = -o.m1.y; e1[5 * no] = -o.m1.z;
                }
                if (l2 < n_own) {
                    double* e2 = pl + (2 * axis + 1) * 6 * no + l2;
                    e2[0] = o.f2.x; e2[no] = o.f2.y; e2[2 * no] = o.f2.z; e2[3 * no] = -o.m2.x; e2[4 * no] = -o.m2.y; e2[5 * no] = -o.m2.z;
                }
            }
            VXH_TT_MARK(1)
        }
        if (div) s_div = 1;
        // on steps that do not speculate (rare: latch, end of launch, trace point; robots without the horizon rule) the serving wavefront
        // resolves the per-robot barrier here, behind its bonds
        if (svc && !speculate) { robot_barrier(mvq, tag, go, K, it, false); VXH_TT_MARK(2) }
        __syncthreads();                       // (B)
        VXH_TT_MARK(3)
        VXH_TS(2, tid == 0)
        const PoseFromXch pose{xq, xpl, B.xslot, tag, &s_abort};
        if (!speculate) {
            if (s_abort) break;
            if (s_divprev) break;              // (undone below)
            // ---- whole-robot passes (rare): IniCM latch / a point of the CoM trace by the robot's first tile (the others only note that
            // it happened), broad-phase
            if (K.latch || K.eol || K.trace) {
                if (ti == T.tile0) latch_cm(B, R, rs, pose, K.latch != 0, K.eol != 0, sc, VXH_TILE_CH, K.trace != 0, K.trace_index);
                else if (tid == 0) { if (K.latch) rs.cm_init = 1; if (K.eol) rs.eol_post_y = 1.0; }
            }
            if (!go) break;
            if (K.rebuild) {
                ccnt = tile_rebuild(B, R, rs, ti, pose, ps, np, my_ord, xh, &s_xhn, s_box, sc, hkey, hval);
                codes_live = true;
                // the partners mirrored from now on: their poses of THIS step (every tile is past publishing them)
                for (int e = tid; e < s_xhn; e += NT) { double x, y, z, s1; pose.at(xh[e], x, y, z, s1); px[e] = x; px[VXH_TILE_XH + e] = y; px[2 * VXH_TILE_XH + e] = z; px[3 * VXH_TILE_XH + e] = s1; }
                rows_to_lds();
            }
        }
        VXH_TT_MARK(4)

        // (Tried in round 3 and taken out: the two halves of the voxel update -- voxel_update_lin / voxel_update_ang, independent within a
        // step -- on different wavefronts, as the wide kernel does: a 64-voxel tile has one wavefront of voxels and three without.  Each
        // half then took as long as the two together had (per-wave phase shares 28.8 % and 33.4 % against 31.7 %), scratch went from
        // 236 to 368 bytes per lane, and the 20^3 lattice from 10.4 to 11.1 us per step.)
        // ---- 3. voxel phase; on a speculating step the service wavefront resolves the per-robot barrier meanwhile, and the phase
        // is redone (attempt 1) after the broad-phase if the lists turn out to have been due for a rebuild
        const bool diverged_here = s_div != 0;
        d3 drag = mk3(0, 0, 0);
        if constexpr (FLUID) {
            if (fluid && valid)                // my facets' drag in facet order (barrier (B) stands between the facet pass and here)
                for (int k = my_ff; k < my_ff + my_fc; ++k) drag = drag + mk3(fdr[k], fdr[nfp + k], fdr[2 * nfp + k]);
        }
        double p8[8];
        d3 lm_new = lm, am_new = am;
        bool stop = false;
        for (int attempt = 0;; ++attempt) {
            double vel2 = 0;
            const unsigned tagn = tile_tag(gen, ep, it + 2);
            VXH_TV_BEGIN
            // the six bond forces / moments in the order of the fused kernel of the robot's size class: up to 768 voxels the bonds in which the
            // voxel is the negative end first (+X +Y +Z), then those in which it is the positive end; above, +X -X +Y -Y +Z -Z.  (All 18 values
            // requested before the first sum, ONE branch on the order: with the order chosen per component the compiler emitted six
            // load-wait-branch-add rounds, 1.2 k cycles of a 5.4 k-cycle phase -- round 6)
            auto six_sums = [&](int c0, int vox) {
                double pq[6][3];
#pragma unroll
                for (int d = 0; d < 6; ++d)
#pragma unroll
                    for (int c = 0; c < 3; ++c) pq[d][c] = pl[(6 * d + c0 + c) * no + vox];
                d3 out;
                if (R.nvox <= 768) {
                    out = mk3(((pq[0][0] + pq[2][0]) + pq[4][0]) + ((pq[1][0] + pq[3][0]) + pq[5][0]), ((pq[0][1] + pq[2][1]) + pq[4][1]) + ((pq[1][1] + pq[3][1]) + pq[5][1]),
                              ((pq[0][2] + pq[2][2]) + pq[4][2]) + ((pq[1][2] + pq[3][2]) + pq[5][2]));
                } else {
                    out = mk3(((((pq[0][0] + pq[1][0]) + pq[2][0]) + pq[3][0]) + pq[4][0]) + pq[5][0], ((((pq[0][1] + pq[1][1]) + pq[2][1]) + pq[3][1]) + pq[4][1]) + pq[5][1],
                              ((((pq[0][2] + pq[1][2]) + pq[2][2]) + pq[3][2]) + pq[4][2]) + pq[5][2]);
                }
                return out;
            };
            if (valid) {
                // ---- translation half: voxel_update_lin (kernels.hpp; CVXS_Voxel::EulerStep / CalcTotalForce, VXS_Voxel.cpp:169-427)
                d3 F = six_sums(0, tid);
                VXH_TV(0)
                d3 pos = mk3(ps[tid], ps[np + tid], ps[2 * np + tid]);
                const double scale = ps[3 * np + tid];
                d3 lmv = lm;
                if (!diverged_here) {
                    const d3 vel = lmv * C.mass_inv;
                    F = F + (vel * (-R.slow_z)) * C.c_lin;
                    const FetchTile fetch{ps, px, np, tid, codes_live, pose};
                    VoxState S; S.pos = pos; S.scale = scale;
                    F = tile_contacts(B, R, fetch, F, S, gv, row, ccnt, roff, rc_code, rc_a1);
                    VXH_TV(1)
                    vel2 = voxel_update_lin(B, R, C, gv, pose, F, vel, pos, lmv, scale, row, 0, FLUID, drag);
                }
                VXH_TV(2)
                lm_new = lmv;
                // out at once: the neighbours' next bond phase waits for exactly these granules (the pose tile keeps the old pose until
                // every voxel of the tile has read its contact partners, and until the step is known to stand)
                p8[0] = pos.x; p8[1] = pos.y; p8[2] = pos.z;
#pragma unroll
                for (int k = 0; k < 3; ++k) st_gran2(xch_at(xqn, xpl, xs_own) + (2 * k) * 64, 64, p8[k], tagn);
                if constexpr (xstrain) {       // ... and the strains this step's bonds left: the next step's mesh is built from them
#pragma unroll
                    for (int k = 0; k < 6; ++k) st_gran2(xch_at(xqn, xpl, xs_own) + (16 + 2 * k) * 64, 64, sl[k * no + tid], tagn);
                }
                VXH_TV(3)
            }
            if (do_ang) {
                // ---- rotation half: voxel_update_ang (angular integration, quaternion update, actuation -> new scale)
                const d3 M = six_sums(3, av);
                double scale = ps[3 * np + av];
                dq ang = mkq(ps[4 * np + av], ps[5 * np + av], ps[6 * np + av], ps[7 * np + av]);
                d3 amv = am;
                if (!diverged_here)
                    voxel_update_ang(B, R, Ca, gva, K.time, K.act_sin, K.act_cos, K.prenatal_c, M, amv, ang, scale, pht[av], pht[no + av], amp_damp);
                am_new = amv;
                p8[3] = scale; p8[4] = ang.w; p8[5] = ang.x; p8[6] = ang.y; p8[7] = ang.z;
#pragma unroll
                for (int k = 3; k < 8; ++k) st_gran2(xch_at(xqn, xpl, xs_ang) + (2 * k) * 64, 64, p8[k], tagn);
            }
            if (tid < 64 * nvw) {
                // the tile's max |v|^2 goes out with the last wavefront to finish its voxels, not behind the workgroup barrier (only the
                // wavefronts that hold voxels take part: one, in a tile of 64 voxels -- no LDS round trips at all then)
                vel2 = wave_max_nonneg(vel2);
                if ((tid & 63) == 0) {
                    unsigned long long bits = (unsigned long long)__double_as_longlong(vel2);
                    bool last = true;
                    if (nvw > 1) {
                        atomicMax(&s_mvbits, bits);
                        last = atomicAdd(&s_done, 1) == nvw - 1;
                        if (last) { bits = s_mvbits; s_mvbits = 0; s_done = 0; }
                    }
                    if (last) {
                        double mv = (R.flags & RF_SELF_COL) ? __longlong_as_double((long long)bits) : 0.0;   // SS.MaxVoxVel for the collision horizon (VX_Sim.cpp:1625-1649)
                        if (diverged_here) mv = -1.0;
                        st_gran2(B.tile_mv + (size_t)ringn * mvbuf + (size_t)ti * VXH_TILE_MV_STRIDE, 1, mv, tagn);
                        VXH_TS(7, true)
                    }
                }
            }
            if (attempt == 0) {
                // Speculating step: the NEXT step's control -- one lane's serial work (the stop rule, a sincos; the two snapshots of the control
                // block by the whole wavefront) beside the voxel phase; it shares no field with the horizon update but dt_prev (s_dtp), and
                // nothing anyone reads before the next step: K stays, Knext is written.  (Rounds 3-5: on a fifth wavefront beside the bond
                // phase; a workgroup of four leaves every wavefront 512 registers, i.e. nothing in scratch -- round 6.)  If the barrier then
                // reports that the robot stopped a step ago, the control block goes back two snapshots instead of one.
                // A step that does not speculate: the same control, one snapshot.
                if (ctl_wave && !s_abort) {
                    snapshot(speculate);
                    if (ctl_thread) {
                        __hip_atomic_store(&s_snap, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (LDS operations of one wavefront: in order)
                        fused_control_begin(R, rs, step_cap, it + 1 < iters, Knext);
                        s_dtp[(it + 1) & 1] = rs.dt_prev;
                    }
                    VXH_TT_MARK(7)
                }
                if (svc && speculate) {
                    // ... and the per-robot barrier of this step
                    robot_barrier(mvq, tag, go, K, it, true);
                    VXH_TT_MARK(2)
                    VXH_TS(5, hz_thread)
                }
            }
            VXH_TT_MARK(5)
            VXH_TS(3, tid == 0)
            __syncthreads();                   // (C) the voxel phase is complete and, on a speculating step, the per-robot barrier resolved
            if (s_abort || s_divprev) { stop = true; break; }
            if (speculate && attempt == 0 && K.rebuild) {
                // mis-speculated: UpdateCollisions would have rebuilt the lists before this voxel phase.  Broad-phase on the poses
                // of the step (all published: the barrier has been passed), then the phase again; its poses go out under a new epoch
                ++ep;
                ccnt = tile_rebuild(B, R, rs, ti, pose, ps, np, my_ord, xh, &s_xhn, s_box, sc, hkey, hval);
                codes_live = true;
                for (int e = tid; e < s_xhn; e += NT) { double x, y, z, s1; pose.at(xh[e], x, y, z, s1); px[e] = x; px[VXH_TILE_XH + e] = y; px[2 * VXH_TILE_XH + e] = z; px[3 * VXH_TILE_XH + e] = s1; }
                rows_to_lds();
                continue;
            }
            break;
        }
        if (stop) break;
        // ---- 4. commit
        VXH_TS(4, tid == 0)
        lm_bak = lm; am_bak = am;
        lm = lm_new; am = am_new;
        if (valid) {
#pragma unroll
            for (int k = 0; k < 3; ++k) ps[k * np + tid] = p8[k];
        }
        if (do_ang) {
#pragma unroll
            for (int k = 3; k < 8; ++k) ps[k * np + av] = p8[k];
        }
        ring = ringn;
        (void)ringp;
    }
    VXH_TT_FLUSH
    __syncthreads();
    if (s_abort) {
        if (tid == 0) atomicExch(&B.rstate[r].status, 5);                       // VXH_ROBOT_SYNC_TIMEOUT
        return;
    }
    if (s_divprev) {
        // a bond diverged in the last step whose poses are in the pose tile: Integrate() returned before its voxel loop
        // (VX_Sim.cpp:1777), so that step's voxel phase is undone -- momenta from the copy kept a step, poses from the ring
        lm = lm_bak; am = am_bak;
        const unsigned long long* const xqp = B.xch + (size_t)(ring == 0 ? 2 : ring - 1) * xbuf;
        if (valid) {
            unsigned long long g[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) g[k] = ld_gran(xch_at(xqp, xpl, xs_own) + k * 64);
#pragma unroll
            for (int k = 0; k < 8; ++k) ps[k * np + tid] = gran2_value(g[2 * k], g[2 * k + 1]);
        }
        if (ctl_thread) { rs = rs_bak; rs.diverged = 1; FusedCtl dummy; fused_control_begin(R, rs, step_cap, 0, dummy); }   // -> status diverged
        __syncthreads();
    }

    // ---- back to HBM: state into the buffer the step count selects, bond history, control block (the robot's first tile)
    if (valid) {
        const int b1 = rs.steps & 1;
        POS(b1, 0, gv) = ps[tid]; POS(b1, 1, gv) = ps[np + tid]; POS(b1, 2, gv) = ps[2 * np + tid];
        SCALE(b1, gv) = ps[3 * np + tid];
        QUAT(0, gv) = ps[4 * np + tid]; QUAT(1, gv) = ps[5 * np + tid]; QUAT(2, gv) = ps[6 * np + tid]; QUAT(3, gv) = ps[7 * np + tid];
        LINMOM(0, gv) = lm.x; LINMOM(1, gv) = lm.y; LINMOM(2, gv) = lm.z;
        if (!split) { ANGMOM(0, gv) = am.x; ANGMOM(1, gv) = am.y; ANGMOM(2, gv) = am.z; }
        if constexpr (mesh) {
#pragma unroll
            for (int k = 0; k < 6; ++k) B.strain[(unsigned)k * nv + (unsigned)gv] = sl[k * no + tid];
        }
    }
    if (split && do_ang) { ANGMOM(0, gva) = am.x; ANGMOM(1, gva) = am.y; ANGMOM(2, gva) = am.z; }
    for (int b = tid; b < nb; b += NT) {
        if ((bent[b] & 1023) >= n_own) continue;                  // a bond is written back by the tile that owns its negative end
        const int slot = B.tile_bslot[T.bond_off + b];
#pragma unroll
        for (int k = 0; k < 6; ++k) HIST(k, slot) = hl[k * nbp + b];
        B.small_angle[slot] = (unsigned char)(hf[b] & 3);
    }
    if (tid == 0) {
        if (rs.col_overflow) atomicOr(&B.rstate[r].col_overflow, 1);
        if (ti == T.tile0) {
            const DRobotState out = rs;
            DRobotState& g = B.rstate[r];
            g.cur_time = out.cur_time; g.dt_prev = out.dt_prev; g.max_disp = out.max_disp;
            g.ini_cm[0] = out.ini_cm[0]; g.ini_cm[1] = out.ini_cm[1]; g.ini_cm[2] = out.ini_cm[2];
            g.eol_post_y = out.eol_post_y; g.act_sin = out.act_sin; g.act_cos = out.act_cos; g.act_time = out.act_time; g.maxvel2_bits = out.maxvel2_bits;
            g.steps = out.steps; g.status = out.status; g.cm_init = out.cm_init; g.active = out.active; g.diverged = out.diverged;
            g.rebuild_now = out.rebuild_now; g.rebuilds = out.rebuilds; g.col_tiled = out.col_tiled; g.ntrace = out.ntrace; g.last_trace_time = out.last_trace_time;
        }
    }
}

}  // namespace vxh
