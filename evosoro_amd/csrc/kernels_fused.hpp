// Fused path of the batched Voxelyze stepper: k_robot_steps<BLOCK, NACC, MESH, TABG>, one workgroup per robot, the robot
// resident in the CU for a whole launch of many time steps (included at the end of kernels.hpp).  MESH = land_water
// robots, which carry the deformable surface mesh (fluid drag, RobotVolume tags); TABG = the robot's class tables are read
// from HBM instead of being copied into LDS (robots with per-voxel evolved stiffness, whose tables outgrow it).
//
// A step has two kinds of work items mapped onto the same threads:
//   voxels  thread t owns voxel t: its momenta stay in registers for the whole launch, its pose is published in LDS;
//   bonds   three slots per thread, one per axis (the axis is a compile-time constant per slot): the robot's per-axis COMPACTED bond
//           lists dealt to the threads by DBatch::bsched, so a sparse robot keeps ceil(bonds_A / 64) wavefronts busy per axis instead
//           of one lane per voxel whether it has that bond or not.  Up to 768 threads the X and Y slots are evaluated back to back
//           WITHOUT a barrier between them (their sums commute: below), the Y chunks dealt to the wavefronts X leaves idle; Z follows
//           behind a barrier.  The history of a bond is one 48-byte record in DBatch::hist_aos.
//   contact pairs  (self-colliding robots) the reach test of every listed pair -- pass 1 of fused_contact_forces -- is run for the whole
//           workgroup by the wavefronts that hold no bond of the Z slot, while the others evaluate theirs (fused_contact_reach_all): it needs
//           the poses only, and at the head of the voxel phase it was a latency chain on every wavefront at once.
// Dynamic LDS layout (doubles):
//   ps   [8][BLOCK]        pose tile: pos x y z, scale, quaternion w x y z of every voxel (read by bonds, contact forces,
//                          the broad-phase, the drag mesh)
//   acc  [NACC][6][BLOCK]  force / minus-moment accumulators of every voxel.  NACC = 2: tile 0 collects the bonds in
//                          which the voxel is the negative end (Force1/Moment1), tile 1 those in which it is the
//                          positive end; an entry gets at most one contribution per axis, added with an LDS atomic: X and Y
//                          in either order (a + b == b + a from the zero the voxel phase left), Z behind a barrier.  NACC = 1 (BLOCK 1024, where two tiles do not fit next to the pose tile): one
//                          tile, the two ends of a round are added in two barrier-separated sub-steps, which gives
//                          the reference's summation order +X -X +Y -Y +Z -Z.
//                          Also scratch of latch / broad-phase between steps (re-zeroed by the voxel phase).
//   pht  [2][BLOCK]        sin / cos of 2 pi' * PhaseOffset of every voxel (constants of the launch; kept out of registers
//                          and out of the step's load queue; read from HBM by the 768-thread MESH variant)
//   tabs                   this robot's DBondClass and DVoxClass rows (not TABG)
//   st   [6][BLOCK]        MESH, BLOCK <= 512: directional strains of the previous step (DBatch::strain otherwise)
//   mesh [3][nmv]          MESH only: vertices of the drag mesh (robots in a fluid)
#pragma once

namespace vxh {

enum { VXH_FUSED_STATIC_LDS = 480 };      // upper bound of the kernel's static __shared__ variables

// developer instrumentation (scripts/dev_gpu_diag.py phases; library built with -DVXH_PHASE_TIMING): per-wave cycle sums
// of the phases of a step
// what-if switches (contact forces off, MaxVoxVel reduction off): developer library only, compiled out of libvxhip.so
#ifdef VXH_PHASE_TIMING
#define VXH_DBG(bit) (B.dbg & (bit))
#else
#define VXH_DBG(bit) false
#endif
#ifdef VXH_PHASE_TIMING
#define VXH_T_DECL unsigned long long t_acc[6] = {0, 0, 0, 0, 0, 0}; unsigned long long t_last = __builtin_readcyclecounter();
#define VXH_T_MARK(k) { const unsigned long long t_now = __builtin_readcyclecounter(); t_acc[k] += t_now - t_last; t_last = t_now; }
#define VXH_T_FLUSH if (B.prof && (threadIdx.x & 63) == 0) { for (int k = 0; k < 6; ++k) atomicAdd(&B.prof[(threadIdx.x >> 6) * 8 + k], t_acc[k]); }
#define VXH_T_SUB_BEGIN unsigned long long t_sub = __builtin_readcyclecounter();
#define VXH_T_SUB(k) { const unsigned long long t_now = __builtin_readcyclecounter(); if (B.prof && (threadIdx.x & 63) == 0) atomicAdd(&B.prof[(threadIdx.x >> 6) * 8 + (k)], t_now - t_sub); t_sub = t_now; }
#define VXH_T_DRAG_BEGIN unsigned long long t_dr = __builtin_readcyclecounter();
#define VXH_T_DRAG(slot) { const unsigned long long t_now = __builtin_readcyclecounter(); if (B.prof && threadIdx.x == 0) atomicAdd(&B.prof[2130 + (slot)], t_now - t_dr); t_dr = t_now; }
#else
#define VXH_T_DRAG_BEGIN
#define VXH_T_DRAG(slot)
#define VXH_T_DECL
#define VXH_T_MARK(k)
#define VXH_T_FLUSH
#define VXH_T_SUB_BEGIN
#define VXH_T_SUB(k)
#endif

// The thread index as a value the compiler cannot see through.  The rare whole-robot passes of a step (CoM latch, broad-phase, the
// copy of the contact rows) index everything by the thread: taken from threadIdx.x their address arithmetic and lane predicates
// are invariants of the step loop, get computed once before it and are then held for the whole launch -- in scratch, at 128
// registers per lane, and reloaded inside the hot phases.  Zero instructions.  Effect: 5-10 registers fewer in the 256-, 512- and
// 768-thread variants; 512 random 10^3 robots (768 threads), warm, against the code without it on the same box, interleaved
// (scripts/ab_lib.py): 33.6-33.8 -> 33.2-33.4 us per population step with self-collision, 23.4-23.7 -> 23.2-23.3 without (about 1 %;
// numbers from different boxes differ by as much and must not be compared).  Not applied to the 1024-thread variant, whose scratch
// it barely changes (196 -> 192 B: that is peak pressure of the bond / voxel arithmetic at 128 registers per lane).
template <int BLOCK>
__device__ __forceinline__ int opaque_tid() { int t = (int)threadIdx.x; if constexpr (BLOCK != 1024) asm volatile("" : "+v"(t)); return t; }

// plane `plane` (of nv doubles) of a SoA array, element at byte offset voff: uniform 64-bit base + 32-bit lane offset
__device__ __forceinline__ double ld_plane(const double* base, unsigned plane, unsigned nv, unsigned voff)
{
    return *(const double*)((const char*)(base + (size_t)plane * nv) + voff);
}
__device__ __forceinline__ void st_plane(double* base, unsigned plane, unsigned nv, unsigned voff, double x)
{
    *(double*)((char*)(base + (size_t)plane * nv) + voff) = x;
}

// IniCM latch + EndOfLifetimePosteriorY from the pose tile (see latch_cm in kernels.hpp for the reference lines)
template <int BLOCK>
__device__ __forceinline__ void fused_latch_cm(const DRobot& R, DRobotState& rs, const double* ps, double* sh, bool valid,
                                               const DVoxClass& C, bool latch, bool eol, bool trace, double* trace_entry)
{
    const int tid = opaque_tid<BLOCK>();
    if (valid) sh[tid] = (C.mat == 5) ? -C.mass : C.mass;      // sign marks the material excluded from PosteriorY
    __syncthreads();
    if (tid == 0) {
#pragma clang fp contract(off)      // product and sum rounded separately, like the reference's GetCM
        double sx = 0, sy = 0, sz = 0, sm = 0, miny = 100000.0;
        for (int k = 0; k < R.nvox; ++k) {
            const double ms = sh[k], m = fabs(ms), y = ps[BLOCK + k];
            const double mx = ps[k] * m, my = y * m, mz = ps[2 * BLOCK + k] * m;
            sx = sx + mx; sy = sy + my; sz = sz + mz; sm += m;
            if (!(ms < 0)) { const double yl = y / R.lat; if (yl < miny) miny = yl; }
        }
        if (latch) { const double inv = 1.0 / sm; rs.ini_cm[0] = inv * sx; rs.ini_cm[1] = inv * sy; rs.ini_cm[2] = inv * sz; rs.cm_init = 1; }
        if (eol) rs.eol_post_y = miny;
        if (trace) { const double inv = 1.0 / sm; trace_entry[0] = rs.cur_time; trace_entry[1] = inv * sx; trace_entry[2] = inv * sy; trace_entry[3] = inv * sz; }   // SS.CMTrace (VX_Sim.cpp:1537-1547)
    }
    __syncthreads();
}

// CalcL1Bonds (VX_Sim.cpp:2357-2413) from the pose tile; thread i owns surface voxel i (see rebuild_rows in kernels.hpp).  This scan
// (every lane reads every candidate's index and pose from LDS) is what the 256-, 512- and 768-thread variants run; the 1024-thread
// variant runs fused_rebuild_staged below.  The developer build runs both (what-if switch 16) and compares the rows.
#ifdef VXH_PHASE_TIMING
#define VXH_RB_MARK(slot) { const unsigned long long t_now = __builtin_readcyclecounter(); if (B.prof && tid == 0) atomicAdd(&B.prof[slot], t_now - t_rb); t_rb = t_now; }
#else
#define VXH_RB_MARK(slot)
#endif
template <int BLOCK>
__device__ __forceinline__ void fused_rebuild_plain(const DBatch& B, const DRobot& R, DRobotState& rs, const double* ps, int* shi,
                                              const DVoxClass* vct)
{
    const int tid = opaque_tid<BLOCK>(), ns = R.nsurf;
#ifdef VXH_PHASE_TIMING
    unsigned long long t_rb = __builtin_readcyclecounter();
#endif
    for (int k = tid; k < ns; k += BLOCK) shi[k] = B.surf_code[R.surf_begin + k];      // local voxel index | class of every surface voxel
    __syncthreads();
    VXH_RB_MARK(2100)
    if (tid < ns) {
        const int i = tid, mine = shi[i], li = mine & 1023;
        const DVoxClass& Ci = vct[mine >> 10];
        const d3 pi = mk3(ps[li], ps[BLOCK + li], ps[2 * BLOCK + li]);
        const double si = ps[3 * BLOCK + li];
        const unsigned long long* row = B.excl + R.excl_begin + (long long)i * R.excl_wpr;
        const double H = R.col_horizon;
        int cnt = 0;
        unsigned long long word = 0;
        for (int j0 = 0; j0 < ns; j0 += 8) {       // eight candidates at a time: their LDS reads are issued together
            if ((j0 & 63) == 0) word = row[j0 >> 6];
            int other[8]; double qx[8], qy[8], qz[8], qs[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) other[u] = shi[min(j0 + u, ns - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int lj = other[u] & 1023; qx[u] = ps[lj]; qy[u] = ps[BLOCK + lj]; qz[u] = ps[2 * BLOCK + lj]; qs[u] = ps[3 * BLOCK + lj]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                if (j >= ns || j == i) continue;
                const d3 d = pi - mk3(qx[u], qy[u], qz[u]);
                const double d2 = len2(d);
                if (!(d2 < R.filter_dist2)) continue;
                if ((word >> (j & 63)) & 1ull) continue;          // !pV1->IsNearbyVox(SIndex2)
                const double s1 = (j > i) ? si : qs[u];           // scale of Vox1 = the earlier one, used twice (:2382)
                const double act = H * (s1 + s1) * 0.5;
                if (d2 < act * act) {
                    if (cnt < R.col_cap) {
                        const DVoxClass& Cj = vct[other[u] >> 10];
                        const size_t at = col_at(R, cnt, R.surf_begin + i);
                        B.col_partner[at] = R.vox_begin + (other[u] & 1023);
                        B.col_a1[at] = (j > i) ? contact_a1(Ci, Cj) : contact_a1(Cj, Ci);
                    }
                    ++cnt;
                }
            }
        }
        VXH_RB_MARK(2101)      // (wave 0's own scan: lanes of one wavefront finish together)
        if (cnt > R.col_cap) { cnt = R.col_cap; atomicOr(&rs.col_overflow, 1); }
        B.col_cnt[R.surf_begin + i] = cnt;
    }
    __syncthreads();
    VXH_RB_MARK(2102)      // count write + waiting for the slowest wavefront's scan
}
// CalcL1Bonds (VX_Sim.cpp:2357-2413) from the pose tile; thread i owns surface voxel i (see rebuild_rows in kernels.hpp) and tests
// all others in ascending order (= creation order of the reference's collision bonds).  A run is 4 % of an average step but what a
// LAUNCH waits for (it ends with its slowest workgroup, and in any 20-step launch a tenth of the robots run one), so its common
// path -- fetch a candidate, squared distance, compare; nearly every pair fails -- is kept to the arithmetic:
//   * the surface voxels' positions are staged by surface ordinal (`stg`, four planes behind the index array in the borrowed
//     accumulator tile), so a candidate's address is the loop counter: uniform, no index read, no per-lane address arithmetic;
//   * the three tests of a candidate are integer arithmetic on compare results, eight candidates per branch ("some lane has one
//     inside the filter": rare); the accepted-pair path recomputes its distance with the same expression.
// Where the time of the scan it replaces went was established with timers, what-if switches and a micro-benchmark of the loop
// (scripts/ubench/scan_loop.hip: 306 -> 196 cycles per candidate and wavefront for this change); DESIGN.md "The cost of a launch"
// lists what did NOT help.  Same arithmetic per pair, hence the same rows (cross-checked in the developer build).
// Used by the 1024-thread variant only.  Measured against fused_rebuild_plain on one box, interleaved: a run 355 k -> 299 k cycles;
// dense 10^3 robots (1024 threads) 53.8-54.4 -> 52.5-52.9 us per step with self-collision, 51.5-51.9 -> 50.5-50.6 without.  In the
// 768-thread variant the fixed cost of a launch fell from 265 to 230 us, but the STEP got slower, 30.5 -> 31.3 us (a population
// without collisions, which never runs this code, 22.9 -> 23.5-23.8): with the larger function inlined the kernel spills 54 scalar
// registers instead of 36, reloaded inside the hot phases.  (Hiding the function's uniform arguments from the compiler, the way
// opaque_tid hides the thread index, made that worse: 65.)
// `shi`: BLOCK ints, then (from shi + 2 * BLOCK ints = one plane of doubles on) 4 planes of BLOCK doubles; the caller re-zeroes them.
template <int BLOCK>
__device__ __forceinline__ void fused_rebuild_staged(const DBatch& B, const DRobot& R, DRobotState& rs, const double* ps, int* shi,
                                                     const DVoxClass* vct)
{
    const int tid = opaque_tid<BLOCK>(), ns = R.nsurf;
    double* const stg = (double*)shi + BLOCK;     // x y z scale by surface ordinal
    for (int k = tid; k < ns; k += BLOCK) {       // local voxel index | class of every surface voxel; its pose by ordinal
        const int g = B.surf[R.surf_begin + k], l = g - R.vox_begin;
        shi[k] = l | ((int)B.vclass[g] << 10);
        stg[k] = ps[l]; stg[BLOCK + k] = ps[BLOCK + l]; stg[2 * BLOCK + k] = ps[2 * BLOCK + l]; stg[3 * BLOCK + k] = ps[3 * BLOCK + l];
    }
    __syncthreads();
    if (tid < ns) {
        const int i = tid, mine = shi[i];
        const DVoxClass& Ci = vct[mine >> 10];
        const d3 pi = mk3(stg[i], stg[BLOCK + i], stg[2 * BLOCK + i]);
        const double si = stg[3 * BLOCK + i];
        const unsigned long long* row = B.excl + R.excl_begin + (long long)i * R.excl_wpr;
        const double H = R.col_horizon, filter2 = R.filter_dist2;
        int cnt = 0;
        unsigned long long word = 0;
        for (int j0 = 0; j0 < ns; j0 += 8) {       // (a group may reach up to 7 entries past ns: inside the planes, masked below)
            if ((j0 & 63) == 0) word = row[j0 >> 6];
            int any = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                const d3 d = pi - mk3(stg[j], stg[BLOCK + j], stg[2 * BLOCK + j]);
                any |= (int)(len2(d) < filter2) & (int)(j != i) & (int)(j < ns);
            }
            if (!any) continue;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                if (j >= ns || j == i) continue;
                const d3 d = pi - mk3(stg[j], stg[BLOCK + j], stg[2 * BLOCK + j]);
                const double d2 = len2(d);
                if (!(d2 < filter2)) continue;
                if ((word >> (j & 63)) & 1ull) continue;          // !pV1->IsNearbyVox(SIndex2)
                const double s1 = (j > i) ? si : stg[3 * BLOCK + j];   // scale of Vox1 = the earlier one, used twice (:2382)
                const double act = H * (s1 + s1) * 0.5;
                if (d2 < act * act) {
                    if (cnt < R.col_cap) {
                        const int other = shi[j];
                        const DVoxClass& Cj = vct[other >> 10];
                        const size_t at = col_at(R, cnt, R.surf_begin + i);
                        B.col_partner[at] = R.vox_begin + (other & 1023);
                        B.col_a1[at] = (j > i) ? contact_a1(Ci, Cj) : contact_a1(Cj, Ci);
                    }
                    ++cnt;
                }
            }
        }
        if (cnt > R.col_cap) { cnt = R.col_cap; atomicOr(&rs.col_overflow, 1); }
        B.col_cnt[R.surf_begin + i] = cnt;
    }
    __syncthreads();
}

// lane `lane` (a compile-time constant after unrolling) of `old` := the uniform value `sval`.  Written as compare + select on purpose:
// v_writelane_b32 through inline assembly does it in one instruction, and with it in this rarely executed function the STEP of a
// colliding population ran twice as slow (61.9 against 30.4 us per population step, same box, same source otherwise: round 3,
// scripts/ab_lib.py lc2) while a population without collisions was unaffected -- the compiler's handling of the hot loop changes
// with an asm statement anywhere in the kernel; not investigated further.
__device__ __forceinline__ unsigned writelane_u32(unsigned old, unsigned sval, int lane)
{
    return ((int)(threadIdx.x & 63) == lane) ? sval : old;
}

// CalcL1Bonds (VX_Sim.cpp:2357-2413) as a BIT MATRIX built from 64 x 64 blocks of surface-voxel pairs, every unordered pair tested
// once.  The plain scan above lets thread i walk all ns candidates (ns^2 tests, wavefront w busy for as long as its rows need, the
// SIMD with three wavefronts three times as long as the one with two); here
//   1. positions are staged by surface ordinal (`stg`: x y z planes of NS = 64 * nb doubles; slots past ns hold far-away values that
//      fail every test), so a candidate's address is uniform -- a broadcast LDS read, no index indirection; the wavefront that
//      stages block b also reduces its bounding box and the largest acceptance radius of its rows;
//   2. the nb (nb + 1) / 2 block pairs (I <= J) are handed out through a counter in LDS (whoever is free takes the next one).  A pair
//      whose bounding boxes are farther apart than any row of I accepts is answered with zeros (exact: a pair's distance is at
//      least the gap between the boxes; a margin of 1e-12 covers the roundings of the two expressions).  Else lane l holds row
//      i = 64 I + l and walks the 64 candidates j = 64 J + u: d2 < thr_i with thr_i = min(FilterDist^2, ActDist_i^2) -- the
//      reference's two tests `Dist2 < FilterDist2` and `Dist2 < ActDist * ActDist` (:2378-2383) are both `d2 <`, so one compare
//      against the smaller bound decides both, exactly; ActDist uses the scale of the EARLIER voxel (:2382), which is the row's in
//      every pair with I < J and in the upper triangle of a diagonal block.  The compare's lane mask IS row j's word for block I
//      (the test is symmetric in everything but that scale, and the earlier voxel is the same seen from either side): lane u keeps
//      it; the lane's own bit goes into row i's word for block J;
//   3. both words lose the pairs within the hop horizon (`excl` bit rows = !IsNearbyVox, :2380; loaded when the pair starts, used
//      when it ends) and go to the matrix `mat` [nb][NS] (word-major: conflict-free), one writer per word; a diagonal block
//      contributes the upper triangle of the rows' own words and the lower triangle of the masks;
//   4. after a barrier thread i walks its row's nb words in ascending order -- the creation order of the reference's collision
//      bonds, hence the order CalcContactForce sums them in -- and writes partner and pair stiffness.
// Same arithmetic per pair as the plain scan, hence the same rows (developer build, what-if switch 16: both run, rows compared).
// Measured: DESIGN.md "Broad-phase".  `mat`: nb * NS words; `stg`: fused_sym_stage_doubles(nb) doubles (positions, boxes, the pair
// counter, and per ordinal the local voxel index | class << 10 that step 4 turns bits into partners with).
enum { VXH_A1TAB_CLASSES = 8 };     // robots of up to this many voxel classes: pair stiffnesses from a table built once per run
__device__ __forceinline__ int fused_sym_stage_doubles(int nb) { return 3 * (nb << 6) + 8 * nb + 2 + (nb << 5) + VXH_A1TAB_CLASSES * VXH_A1TAB_CLASSES; }

template <int BLOCK>
__device__ __forceinline__ void fused_rebuild_sym(const DBatch& B, const DRobot& R, DRobotState& rs, const double* ps, unsigned long long* mat,
                                                  double* stg, const DVoxClass* vct)
{
    const int tid = opaque_tid<BLOCK>(), ns = R.nsurf;
    const int nb = (ns + 63) >> 6, NS = nb << 6;
    const int lane = tid & 63;
    double* const box = stg + 3 * NS;                          // per block: min x y z, max x y z, largest thr of its rows, -
    int* const next_pair = (int*)(box + 8 * nb);
    int* const shi = next_pair + 4;                            // [NS] local voxel index | class << 10 of every surface voxel
    double* const a1tab = box + 8 * nb + 2 + (nb << 5);        // [nvc][nvc] contact_a1(class of the earlier voxel, class of the later one)
    const int nvc = R.n_vclass;
    const bool tab = nvc <= VXH_A1TAB_CLASSES;
    if (tab && tid < nvc * nvc) a1tab[tid] = contact_a1(vct[tid / nvc], vct[tid % nvc]);
    const double H = R.col_horizon, filter2 = R.filter_dist2;
#ifdef VXH_PHASE_TIMING
    unsigned long long t_rb = __builtin_readcyclecounter();
#endif
    if (tid == 0) *next_pair = 0;
    if (tid < NS) {                                            // (NS <= BLOCK: one slot per thread, wavefront b stages block b)
        const int k = tid;
        double x = 1.0e150, y = 1.0e150, z = 1.0e150, thr = -1.0;   // a slot past the list: farther than any bound from every voxel; accepts nothing
        int code = 0;
        if (k < ns) {
            code = B.surf_code[R.surf_begin + k];
            const int l = code & 1023;
            x = ps[l]; y = ps[BLOCK + l]; z = ps[2 * BLOCK + l];
            const double sk = ps[3 * BLOCK + l];
            const double act = H * (sk + sk) * 0.5, act2 = act * act;
            thr = act2 < filter2 ? act2 : filter2;
        }
        stg[k] = x; stg[NS + k] = y; stg[2 * NS + k] = z; shi[k] = code;
        const bool real = k < ns;
        const double lox = wave_minmax<false>(real ? x : 1.0e300), loy = wave_minmax<false>(real ? y : 1.0e300), loz = wave_minmax<false>(real ? z : 1.0e300);
        const double hix = wave_minmax<true>(real ? x : -1.0e300), hiy = wave_minmax<true>(real ? y : -1.0e300), hiz = wave_minmax<true>(real ? z : -1.0e300);
        const double tmax = wave_minmax<true>(thr);
        if (lane == 0) { double* e = box + 8 * (tid >> 6); e[0] = lox; e[1] = loy; e[2] = loz; e[3] = hix; e[4] = hiy; e[5] = hiz; e[6] = tmax; }
    }
    __syncthreads();
    VXH_RB_MARK(2100)      // staging
    const int npairs = nb * (nb + 1) / 2;
    for (;;) {
        int p = 0;
        if (lane == 0) p = atomicAdd(next_pair, 1);
        p = __builtin_amdgcn_readfirstlane(p);
        if (p >= npairs) break;
        int I = 0, rem = p;
        while (rem >= nb - I) { rem -= nb - I; ++I; }
        const int J = I + rem;
        const int i = (I << 6) + lane, jrow = (J << 6) + lane;
        // exclusion words of my two rows for this pair: requested now, needed after the candidate loop
        unsigned long long e_own = ~0ull, e_col = ~0ull;
        if (i < ns) e_own = B.excl[R.excl_begin + (long long)i * R.excl_wpr + J];
        if (jrow < ns) e_col = B.excl[R.excl_begin + (long long)jrow * R.excl_wpr + I];
        if (I != J) {
            const double* bi = box + 8 * I; const double* bj = box + 8 * J;
            double gap2 = 0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double g1 = bi[a] - bj[3 + a], g2 = bj[a] - bi[3 + a];
                const double g = g1 > g2 ? g1 : g2;
                if (g > 0) gap2 += g * g;
            }
            if (gap2 > bi[6] * (1.0 + 1.0e-12)) {              // no row of I reaches any voxel of J
                mat[(size_t)J * NS + i] = 0ull;
                mat[(size_t)I * NS + jrow] = 0ull;
                continue;
            }
        }
        const d3 pi = mk3(stg[i], stg[NS + i], stg[2 * NS + i]);
        double thr = -1.0;                                      // (rows past the list accept nothing)
        if (i < ns) {
            const double si = ps[3 * BLOCK + (shi[i] & 1023)];
            const double act = H * (si + si) * 0.5;             // the row is Vox1 = the earlier voxel of every pair this block pair keeps
            const double act2 = act * act;
            thr = act2 < filter2 ? act2 : filter2;
        }
        unsigned own_w[2] = {0, 0}, col_lo = 0, col_hi = 0;
        const double* q = stg + (J << 6);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            unsigned own = 0;
#pragma unroll 1
            for (int u0 = 0; u0 < 32; u0 += 8) {           // (eight candidates in flight: unrolled further the compiler keeps all 64 lane masks alive and spills them)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int u = 32 * half + u0 + k;
                    const d3 d = pi - mk3(q[u], q[NS + u], q[2 * NS + u]);
                    const bool in = len2(d) < thr;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
                    own |= in ? (1u << (u0 + k)) : 0u;
                    col_lo = writelane_u32(col_lo, (unsigned)m, u);
                    col_hi = writelane_u32(col_hi, (unsigned)(m >> 32), u);
                }
            }
            own_w[half] = own;
        }
        const unsigned long long own = ((unsigned long long)own_w[1] << 32) | own_w[0], col = ((unsigned long long)col_hi << 32) | col_lo;
        if (I == J) {
            const unsigned long long below = (1ull << lane) - 1ull;          // partners before me / after me inside the block
            mat[(size_t)I * NS + i] = ((own & ~(below | (1ull << lane))) | (col & below)) & ~e_own;
        } else {
            mat[(size_t)J * NS + i] = own & ~e_own;                          // row i, partners of block J
            mat[(size_t)I * NS + jrow] = col & ~e_col;                       // row 64 J + lane, partners of block I
        }
    }
    VXH_RB_MARK(2101)      // wave 0's block pairs
    __syncthreads();
    VXH_RB_MARK(2102)      // waiting for the slowest wavefront
    if (tid < ns) {
        const int i = tid;
        const int ci = shi[i] >> 10;
        const DVoxClass& Ci = vct[ci];
        int cnt = 0;
        for (int w = 0; w < nb; ++w) {
            unsigned long long word = mat[(size_t)w * NS + i];
            while (word) {
                const int u = __builtin_ctzll(word);
                word &= word - 1;
                const int j = (w << 6) + u;
                if (cnt < R.col_cap) {
                    const int other = shi[j], cj = other >> 10;
                    const size_t at = col_at(R, cnt, R.surf_begin + i);
                    B.col_partner[at] = R.vox_begin + (other & 1023);
                    double a1;
                    if (tab) a1 = (j > i) ? a1tab[ci * nvc + cj] : a1tab[cj * nvc + ci];       // (the same function of the same two classes: the same bits)
                    else { const DVoxClass& Cj = vct[cj]; a1 = (j > i) ? contact_a1(Ci, Cj) : contact_a1(Cj, Ci); }
                    B.col_a1[at] = a1;
                }
                ++cnt;
            }
        }
        if (cnt > R.col_cap) { cnt = R.col_cap; atomicOr(&rs.col_overflow, 1); }
        B.col_cnt[R.surf_begin + i] = cnt;
    }
    __syncthreads();
    VXH_RB_MARK(2126)      // rows written
}

// The broad-phase of a resident robot: the bit-matrix scan when its matrix (nb * NS words) fits the scratch tile and the staged
// positions (fused_sym_stage_doubles) fit behind it or in the contact-row pool (`pool`, dead during a run: rows_to_lds refills it afterwards);
// else the staged / plain scans.  `scratch` is left dirty: the caller re-zeroes what it needs zero.
template <int BLOCK>
__device__ __forceinline__ void fused_rebuild(const DBatch& B, const DRobot& R, DRobotState& rs, const double* ps, double* scratch, int scratch_doubles,
                                              double* pool, int pool_doubles, const DVoxClass* vct)
{
    if constexpr (BLOCK == 1024) {
        // (the 1024-thread variant has one accumulator tile, 6 * BLOCK doubles: the matrix of a robot of more than 768 voxels rarely
        // fits, and the extra code costs this variant, which is short of registers, scratch in its hot phases: 180 -> 252 bytes)
        fused_rebuild_staged<BLOCK>(B, R, rs, ps, (int*)scratch, vct);
    } else {
        const int nb = (R.nsurf + 63) >> 6, NS = nb << 6;
        const int need_mat = nb * NS, need_stg = fused_sym_stage_doubles(nb);
#ifdef VXH_NO_SYM          // (developer what-if: the scan of round 2)
        if (true) fused_rebuild_plain<BLOCK>(B, R, rs, ps, (int*)scratch, vct);
        else
#endif
        if (need_mat + need_stg <= scratch_doubles) fused_rebuild_sym<BLOCK>(B, R, rs, ps, (unsigned long long*)scratch, scratch + need_mat, vct);
        else if (need_mat <= scratch_doubles && need_stg <= pool_doubles) fused_rebuild_sym<BLOCK>(B, R, rs, ps, (unsigned long long*)scratch, pool, vct);
        else fused_rebuild_plain<BLOCK>(B, R, rs, ps, (int*)scratch, vct);
    }
}

// land_water fluid drag (LW/VX_Sim.cpp:1516-1597) inside the resident kernel; the per-corner and per-facet arithmetic is
// shared with the streaming kernels (kernels.hpp: RotFwd, mesh_corner_offset, facet_drag_force).
// Constant index data of the drag pass that a thread needs at the head of its loops (its first two mesh vertices, its first
// four facets).  Workgroups of up to 512 threads have the registers to keep it for the whole launch; the larger ones
// reload it every step.
struct VertRec { unsigned w0, w1, w2; };
struct FacetRec { int u, ia, ib, ic; };
__device__ __forceinline__ VertRec load_vert(const DBatch& B, const DRobot& R, int i)
{
    VertRec r;
    const unsigned tm = B.total_mv, gi = R.vert_begin + min(i, R.nmv - 1);
    r.w0 = B.vert_pack[gi]; r.w1 = B.vert_pack[tm + gi]; r.w2 = B.vert_pack[2u * tm + gi];
    return r;
}
__device__ __forceinline__ d3 load_vert_v0(const DBatch& B, const DRobot& R, int i)
{
    const unsigned tm = B.total_mv, gi = R.vert_begin + min(i, R.nmv - 1);
    return mk3(B.vert_v0[gi], B.vert_v0[tm + gi], B.vert_v0[2u * tm + gi]);
}
__device__ __forceinline__ FacetRec load_facet(const DBatch& B, const DRobot& R, int f)   // the four indices of facet f
{
    FacetRec r;
    const unsigned tf = B.total_facet, gf = R.facet_begin + min(f, R.nfacet - 1);
    r.u = B.facet_vox[gf]; r.ia = B.facet_vert[gf]; r.ib = B.facet_vert[tf + gf]; r.ic = B.facet_vert[2u * tf + gf];
    return r;
}
template <int BLOCK>
struct DragCache {
    static constexpr bool KEEP = BLOCK <= 512;
    VertRec vert0, vert1; d3 v00, v01; FacetRec f0, f1, f2, f3;
    int my_first = 0, my_count = 0;                 // the voxel's own facets (set by the kernel: it knows the thread's voxel)
    __device__ __forceinline__ void load(const DBatch& B, const DRobot& R, int tid)
    {
        vert0 = load_vert(B, R, tid); vert1 = load_vert(B, R, tid + BLOCK);
        v00 = load_vert_v0(B, R, tid); v01 = load_vert_v0(B, R, tid + BLOCK);
        f0 = load_facet(B, R, tid); f1 = load_facet(B, R, tid + BLOCK); f2 = load_facet(B, R, tid + 2 * BLOCK); f3 = load_facet(B, R, tid + 3 * BLOCK);
    }
};

// `st`: the voxels' directional strains of the previous step, [6][BLOCK] in LDS (robots up to 768 voxels) or the
// robot's slice of DBatch::strain with plane stride nv (1024-thread variant, where LDS is full).
// `scr`: the accumulator tile, idle until the bond rounds: first the voxels' corner positions (four or two of the eight
// corners at a time: twelve or six planes), then every voxel's velocity (and its direction), then the facets' drag.
// LEAN (the wide kernel, where a step is latency, not instruction count): the voxels' velocities live in planes of their own
// (`spd_ext`, plane stride `spd_stride`) and are published before the vertex pass instead of behind a barrier of their own; the whole
// tile then takes the facets' contributions -- one chunk of up to SCR_DOUBLES / 3 facets, where the resident kernel needs two for a
// robot of 8^3 --; a voxel's sum over its facets requests four contributions at a time (same order of additions); the tile is not
// re-zeroed (the wide kernel's bond records are written, not accumulated).  Same operations per vertex and facet, same sums: same bits.
template <int BLOCK, int SCR_DOUBLES, bool LEAN = false>
__device__ __forceinline__ d3 fused_drag(const DBatch& B, const DRobot& R, const double* ps, const double* st, unsigned st_stride, double* sh,
                                         double* scr, bool valid, int v, d3 lm, double mass_inv, const DragCache<BLOCK>& kept,
                                         double* spd_ext = nullptr, int spd_stride = 0)
{
    constexpr bool WIDE = SCR_DOUBLES >= 12 * BLOCK;
    constexpr int CPP = WIDE ? 4 : 2;               // voxel corners per pass
    static_assert(SCR_DOUBLES >= 3 * CPP * BLOCK, "accumulator tile too small for the corner passes");
    const double nom = R.lat;
    const int nmv = R.nmv;
    const int tid = threadIdx.x;
    VXH_T_SUB_BEGIN
    // Phase 1, voxel-parallel then vertex-parallel: every voxel rotates its own eight corner offsets (one rotation matrix),
    // every vertex adds up the corners that meet in it, in corner-code order (the reference: in voxel order; § Numerics)
    constexpr bool KEEP = DragCache<BLOCK>::KEEP;
    constexpr int NKV = KEEP ? 2 : 1, NKF = KEEP ? 4 : 0;           // leading vertex / facet records held in registers
    VertRec vert0, vert1 = {0, 0, 0};
    d3 v00 = mk3(0, 0, 0), v01 = mk3(0, 0, 0);
    if constexpr (KEEP) { vert0 = kept.vert0; vert1 = kept.vert1; v00 = kept.v00; v01 = kept.v01; }
    else vert0 = load_vert(B, R, tid);
    // the kept facet records as local scalars, selected by value below (a conditional between members reached through the
    // reference becomes a load from a selected ADDRESS, which pins the whole cache in scratch memory)
    const int ku[4] = {kept.f0.u, kept.f1.u, kept.f2.u, kept.f3.u}, ka[4] = {kept.f0.ia, kept.f1.ia, kept.f2.ia, kept.f3.ia};
    const int kb[4] = {kept.f0.ib, kept.f1.ib, kept.f2.ib, kept.f3.ib}, kc[4] = {kept.f0.ic, kept.f1.ic, kept.f2.ic, kept.f3.ic};
    // Phase 1.  Every vertex is the mean of the voxel corners that meet in it, corner = Pos + R(Angle) * offset, summed in
    // corner-code order (the reference: in voxel order; § Numerics).  GATHER: every voxel rotates its own eight corner
    // offsets into the tile, four (twelve planes) or two (six planes) at a time, and the vertices gather them.  The
    // 1024-thread variant (six planes, short of registers) is faster when every vertex recomputes its corners itself (same
    // operations): measured 42.4 against 46.5 us per step on dense 10^3 swimmers.
    constexpr bool GATHER = BLOCK < 1024;
    constexpr bool SDIR = WIDE && !LEAN;                            // the velocity's direction published next to it (else: per facet)
    double* const spd = LEAN ? spd_ext : scr;                       // [VPL][sstr]
    const int sstr = LEAN ? spd_stride : BLOCK;
    auto publish_speed = [&]() {                                    // every voxel's velocity for the facet pass
        if (valid) {
            const d3 sp = lm * mass_inv;
            spd[tid] = sp.x; spd[sstr + tid] = sp.y; spd[2 * sstr + tid] = sp.z;
            if constexpr (SDIR) { const d3 sd = normalized3(sp); spd[3 * BLOCK + tid] = sd.x; spd[4 * BLOCK + tid] = sd.y; spd[5 * BLOCK + tid] = sd.z; }
        }
    };
    if constexpr (LEAN) publish_speed();
    if constexpr (GATHER) {
        d3 vp = mk3(0, 0, 0), hp = mk3(0, 0, 0), hn = mk3(0, 0, 0);
        RotFwd M;
        if (valid) {
            vp = mk3(ps[tid], ps[BLOCK + tid], ps[2 * BLOCK + tid]);
            M = RotFwd(mkq(ps[4 * BLOCK + tid], ps[5 * BLOCK + tid], ps[6 * BLOCK + tid], ps[7 * BLOCK + tid]));
            // CornerPosCur / CornerNegCur (LW/VXS_Voxel.cpp:472-475)
            hp = mk3((1 + st[tid]) * nom * 0.5, (1 + st[st_stride + tid]) * nom * 0.5, (1 + st[2 * st_stride + tid]) * nom * 0.5);
            hn = mk3(-(1 + st[3 * st_stride + tid]) * nom * 0.5, -(1 + st[4 * st_stride + tid]) * nom * 0.5, -(1 + st[5 * st_stride + tid]) * nom * 0.5);
        }
    #pragma unroll
        for (int p0 = 0; p0 < 8; p0 += CPP) {
            if constexpr (!KEEP) { if (p0 + CPP == 8) v00 = load_vert_v0(B, R, tid); }
            if (valid) {
    #pragma unroll
                for (int c = 0; c < CPP; ++c) {
                    const int corner = p0 + c;
                    const d3 p = vp + M(mk3((corner & 4) ? hp.x : hn.x, (corner & 2) ? hp.y : hn.y, (corner & 1) ? hp.z : hn.z));
                    scr[(3 * c) * BLOCK + tid] = p.x; scr[(3 * c + 1) * BLOCK + tid] = p.y; scr[(3 * c + 2) * BLOCK + tid] = p.z;
                }
            }
            __syncthreads();
            int k = 0;
            VertRec vnext = vert1;
            d3 v0next = v01;
            for (int i = tid; i < nmv; i += BLOCK, ++k) {
                const bool s0 = k == 0, s1 = KEEP && k == 1;          // (selects field by field, like the facet records below)
                VertRec vr;
                vr.w0 = s0 ? vert0.w0 : (s1 ? vert1.w0 : vnext.w0); vr.w1 = s0 ? vert0.w1 : (s1 ? vert1.w1 : vnext.w1); vr.w2 = s0 ? vert0.w2 : (s1 ? vert1.w2 : vnext.w2);
                const d3 v0 = mk3(s0 ? v00.x : (s1 ? v01.x : v0next.x), s0 ? v00.y : (s1 ? v01.y : v0next.y), s0 ? v00.z : (s1 ? v01.z : v0next.z));
                if (k + 1 >= NKV && i + BLOCK < nmv) { vnext = load_vert(B, R, i + BLOCK); if (p0 + CPP == 8) v0next = load_vert_v0(B, R, i + BLOCK); }   // one iteration ahead (a request for
                                                                      // nothing would still be waited for at the barrier below)
                d3 part = p0 == 0 ? mk3(0, 0, 0) : mk3(sh[i], sh[nmv + i], sh[2 * nmv + i]);
    #pragma unroll
                for (int c = 0; c < CPP; ++c) {
                    const int corner = p0 + c;
                    if (!((vr.w2 >> (20 + corner)) & 1u)) continue;
                    const unsigned word = corner < 3 ? vr.w0 : (corner < 6 ? vr.w1 : vr.w2);
                    const unsigned l = (word >> (10 * (corner % 3))) & 1023u;
                    part = part + mk3(scr[(3 * c) * BLOCK + l], scr[(3 * c + 1) * BLOCK + l], scr[(3 * c + 2) * BLOCK + l]);
                }
                if (p0 + CPP == 8) {
                    const double inv = vrcp((double)__popc((vr.w2 >> 20) & 255u));
                    const d3 np = part * inv;
                    part = v0 + (np - v0);                               // v + DrawOffset, as the reference stores it
                }
                sh[i] = part.x; sh[nmv + i] = part.y; sh[2 * nmv + i] = part.z;
            }
            __syncthreads();                                             // the corners have been read: scr changes hands
        }
    } else {
        if (nmv > 0) v00 = KEEP ? v00 : load_vert_v0(B, R, tid);
        int k = 0;
        VertRec vnext = vert1;
        d3 v0next = v01;
        for (int i = tid; i < nmv; i += BLOCK, ++k) {
            const bool s0 = k == 0, s1 = KEEP && k == 1;
            VertRec vr;
            vr.w0 = s0 ? vert0.w0 : (s1 ? vert1.w0 : vnext.w0); vr.w1 = s0 ? vert0.w1 : (s1 ? vert1.w1 : vnext.w1); vr.w2 = s0 ? vert0.w2 : (s1 ? vert1.w2 : vnext.w2);
            const d3 v0 = mk3(s0 ? v00.x : (s1 ? v01.x : v0next.x), s0 ? v00.y : (s1 ? v01.y : v0next.y), s0 ? v00.z : (s1 ? v01.z : v0next.z));
            if (k + 1 >= NKV && i + BLOCK < nmv) { vnext = load_vert(B, R, i + BLOCK); v0next = load_vert_v0(B, R, i + BLOCK); }   // one iteration ahead
            d3 part = mk3(0, 0, 0);
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                if (!((vr.w2 >> (20 + corner)) & 1u)) continue;
                const unsigned word = corner < 3 ? vr.w0 : (corner < 6 ? vr.w1 : vr.w2);
                const unsigned l = (word >> (10 * (corner % 3))) & 1023u;
                const double hx = (1 + st[((corner & 4) ? 0u : 3u) * st_stride + l]) * nom * 0.5;     // CornerPosCur / CornerNegCur
                const double hy = (1 + st[((corner & 2) ? 1u : 4u) * st_stride + l]) * nom * 0.5;
                const double hz = (1 + st[((corner & 1) ? 2u : 5u) * st_stride + l]) * nom * 0.5;
                const RotFwd Ml(mkq(ps[4 * BLOCK + l], ps[5 * BLOCK + l], ps[6 * BLOCK + l], ps[7 * BLOCK + l]));
                part = part + (mk3(ps[l], ps[BLOCK + l], ps[2 * BLOCK + l]) + Ml(mk3((corner & 4) ? hx : -hx, (corner & 2) ? hy : -hy, (corner & 1) ? hz : -hz)));
            }
            const double inv = vrcp((double)__popc((vr.w2 >> 20) & 255u));
            const d3 np = part * inv;
            const d3 now = v0 + (np - v0);                           // v + DrawOffset, as the reference stores it
            sh[i] = now.x; sh[nmv + i] = now.y; sh[2 * nmv + i] = now.z;
        }
    }
    VXH_T_SUB(6)
    VXH_T_DRAG_BEGIN
    // Phase 2, one thread per FACET (the robot's facets in the reference's order: per voxel, faces +X,-X,+Y,-Y,+Z,-Z, two
    // triangles each), so the wavefronts are full whatever the number of exposed faces of a voxel.  `scr` holds every
    // voxel's velocity (and its direction), then chunk after chunk the facets' contributions, which each voxel sums in
    // facet order.
    const int nfac = R.nfacet;
    constexpr int VPL = LEAN ? 0 : (WIDE ? 6 : 3);                  // planes of per-voxel data inside the tile
    constexpr int CHF = (SCR_DOUBLES / BLOCK - VPL) * BLOCK / 3;    // facets per chunk
    double* const fd = scr + VPL * BLOCK;                           // [3][CHF] drag of the facets of the current chunk
    if constexpr (!LEAN) publish_speed();
    if constexpr (!(LEAN && GATHER)) __syncthreads();               // (LEAN: published before the vertex pass, whose last barrier is enough)
    VXH_T_DRAG(0)
    d3 drag = mk3(0, 0, 0);
    const int my_first = KEEP ? kept.my_first : (valid ? B.facet_first[v] : 0), my_count = KEEP ? kept.my_count : (valid ? (int)B.facet_count[v] : 0);
    static_assert(CHF % BLOCK == 0, "a thread's facets are tid + m * BLOCK in every chunk");
    for (int c0 = 0; c0 < nfac; c0 += CHF) {
        int m = c0 / BLOCK;
        FacetRec next = {0, 0, 0, 0};
        if (m >= NKF) next = load_facet(B, R, c0 + tid);
        for (int f = c0 + tid; f < min(nfac, c0 + CHF); f += BLOCK, ++m) {
            FacetRec rec = next;
            if constexpr (KEEP) {
                if (m < 4) {
                    rec.u = m == 0 ? ku[0] : (m == 1 ? ku[1] : (m == 2 ? ku[2] : ku[3]));
                    rec.ia = m == 0 ? ka[0] : (m == 1 ? ka[1] : (m == 2 ? ka[2] : ka[3]));
                    rec.ib = m == 0 ? kb[0] : (m == 1 ? kb[1] : (m == 2 ? kb[2] : kb[3]));
                    rec.ic = m == 0 ? kc[0] : (m == 1 ? kc[1] : (m == 2 ? kc[2] : kc[3]));
                }
            }
            if (m + 1 >= NKF && f + BLOCK < min(nfac, c0 + CHF)) next = load_facet(B, R, f + BLOCK);     // one iteration ahead
            const int u = rec.u, ia = rec.ia, ib = rec.ib, ic = rec.ic;
            const d3 speed = mk3(spd[u], spd[sstr + u], spd[2 * sstr + u]);
            d3 sdir;
            if constexpr (SDIR) sdir = mk3(spd[3 * BLOCK + u], spd[4 * BLOCK + u], spd[5 * BLOCK + u]);
            else sdir = normalized3(speed);
            const d3 A = mk3(sh[ia], sh[nmv + ia], sh[2 * nmv + ia]);
            const d3 contrib = facet_drag_force(speed, sdir, A, mk3(sh[ib], sh[nmv + ib], sh[2 * nmv + ib]), mk3(sh[ic], sh[nmv + ic], sh[2 * nmv + ic]), R.drag_coef);
            fd[f - c0] = contrib.x; fd[CHF + (f - c0)] = contrib.y; fd[2 * CHF + (f - c0)] = contrib.z;
        }
        VXH_T_DRAG(1)
        __syncthreads();
        VXH_T_DRAG(2)
        if constexpr (LEAN) {
            const int ke = min(my_first + my_count, c0 + CHF);
            for (int k0 = max(my_first, c0); k0 < ke; k0 += 4) {                         // my facets of this chunk, in order, four requests in flight
                d3 t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int at = min(k0 + j, ke - 1) - c0; t[j] = mk3(fd[at], fd[CHF + at], fd[2 * CHF + at]); }
#pragma unroll
                for (int j = 0; j < 4; ++j) if (k0 + j < ke) drag = drag + t[j];
            }
        } else {
            for (int k = max(my_first, c0); k < min(my_first + my_count, c0 + CHF); ++k)     // my facets of this chunk, in order
                drag = drag + mk3(fd[k - c0], fd[CHF + (k - c0)], fd[2 * CHF + (k - c0)]);
        }
        VXH_T_DRAG(3)
        __syncthreads();
        VXH_T_DRAG(4)
    }
    if constexpr (!LEAN) {
        for (int k = tid; k < SCR_DOUBLES; k += BLOCK) scr[k] = 0.0;    // the accumulators must be zero when the bond rounds start
        __syncthreads();
    }
    VXH_T_SUB(7)
    return drag;
}

// A bond of axis A (packed entry of DBatch::bsched: negative-end voxel | positive-end voxel << 10 | class << 20):
// both poses from the pose tile, history from/to HBM.  Returns the outputs; the caller adds them to the accumulators.
template <int A, int BLOCK, bool MESH>
__device__ __forceinline__ BondOut fused_bond(const DBatch& B, const DRobot& R, const DBondClass* bct, const double* ps, int entry,
                                              unsigned& modebits, bool damp_on, double* st, unsigned st_stride)
{
    unsigned nv = B.nv;
    asm volatile("" : "+s"(nv));              // plane addresses are rebuilt here by the scalar unit: hoisted out of the step
                                              // loop they cost 36 scalar registers and come back as v_readlane spills
    const int l1 = entry & 1023, l2 = (entry >> 10) & 1023;
    const unsigned voff = (unsigned)(R.vox_begin + l1) * 8u;
    BondHist H;                               // history first: the only HBM/L2 round trip of the bond; one 48-byte record, three 16-byte loads
    double2* const hrec = (double2*)(B.hist_aos + ((size_t)((unsigned)A * nv) + (unsigned)(R.vox_begin + l1)) * 6);
    (void)voff;
    { const double2 h0 = hrec[0], h1 = hrec[1], h2 = hrec[2]; H.p0 = h0.x; H.p1 = h0.y; H.p2 = h1.x; H.g0 = h1.y; H.g1 = h2.x; H.g2 = h2.y; }
    H.flags = (modebits >> (2 * A)) & 3u;
    H.store_hist = false;
    const d3 p1 = mk3(ps[l1], ps[BLOCK + l1], ps[2 * BLOCK + l1]);
    const double s1 = ps[3 * BLOCK + l1];
    const dq q1 = mkq(ps[4 * BLOCK + l1], ps[5 * BLOCK + l1], ps[6 * BLOCK + l1], ps[7 * BLOCK + l1]);
    const d3 p2 = mk3(ps[l2], ps[BLOCK + l2], ps[2 * BLOCK + l2]);
    const double s2 = ps[3 * BLOCK + l2];
    const dq q2 = mkq(ps[4 * BLOCK + l2], ps[5 * BLOCK + l2], ps[6 * BLOCK + l2], ps[7 * BLOCK + l2]);
    // (rotation-vector factor in select form where it was measured faster: the 768-thread variant, 25.15 -> 24.96 us per step; on the
    // 1024-thread one it costs 5 %, dense 10^3 lattices 33.6 -> 35.3 -- same bits either way, kernels.hpp rotvec_factor)
    BondOut o = bond_compute<A, BLOCK == 768>(B, bct[(unsigned)entry >> 20], H, p1, q1, s1, p2, q2, s2, damp_on);
    if (H.store_hist) { hrec[0] = make_double2(H.p0, H.p1); hrec[1] = make_double2(H.p2, H.g0); hrec[2] = make_double2(H.g1, H.g2); }
    modebits = (modebits & ~(3u << (2 * A))) | (H.flags << (2 * A));
    if constexpr (MESH) {                     // SetStrainDir (VXS_BondInternal.cpp:300-304): +A side of voxel 1, -A side of voxel 2
        st[(unsigned)A * st_stride + l1] = o.strain1;
        st[(unsigned)(3 + A) * st_stride + l2] = o.strain2;
    }
    return o;
}

// acc[.][l] += (f, -m): the calling thread is the only writer of voxel l's entry between two barriers.  FIRST: the entry
// still holds the zero the voxel phase left there, so the sum is the value itself (0 + x == x: same bits, no read).
__device__ __forceinline__ void lds_add(double* p, double x)
{
#ifdef VXH_NO_LDS_ADD          // (developer what-if: the read-modify-write of round 2)
    *p += x;
#else
    __hip_atomic_fetch_add(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
template <int BLOCK, bool FIRST>
__device__ __forceinline__ void fused_accumulate(double* acc, int l, d3 f, d3 m)
{
    double* e = acc + l;
    if constexpr (FIRST) {
        e[0] = f.x; e[BLOCK] = f.y; e[2 * BLOCK] = f.z; e[3 * BLOCK] = -m.x; e[4 * BLOCK] = -m.y; e[5 * BLOCK] = -m.z;
    } else {
        // one ds_add_f64 each (no value returned) instead of read + add + write: the same addition of the same two operands,
        // a third of the LDS instructions and no round trip inside the bond's chain
        lds_add(e, f.x); lds_add(e + BLOCK, f.y); lds_add(e + 2 * BLOCK, f.z); lds_add(e + 3 * BLOCK, -m.x); lds_add(e + 4 * BLOCK, -m.y); lds_add(e + 5 * BLOCK, -m.z);
    }
}

// one axis round: bond, then both ends into the accumulators
template <int A, int BLOCK, int NACC, bool MESH>
__device__ __forceinline__ void fused_round(const DBatch& B, const DRobot& R, const DBondClass* bct, const double* ps, double* acc, int entry,
                                            unsigned& modebits, bool damp_on, bool& div, double* st, unsigned st_stride)
{
    const bool has = entry != -1;
    BondOut o;
    if (has) {
        o = fused_bond<A, BLOCK, MESH>(B, R, bct, ps, entry, modebits, damp_on, st, st_stride);
        div = div || o.diverged;
        fused_accumulate<BLOCK, false>(acc, entry & 1023, o.f1, o.m1);
    }
    // one tile: an entry gets Force1 of the voxel's +A bond and Force2 of its -A bond.  In the X round both are ADDED to the zero the
    // voxel phase left and commute exactly; from Y on the second must follow the first (the reference's +A -A order): a barrier
    if constexpr (NACC == 1 && A != 0) __syncthreads();
    if (has) fused_accumulate<BLOCK, false>(acc + (NACC - 1) * 6 * BLOCK, (entry >> 10) & 1023, o.f2, o.m2);
}

// Contact forces of my voxel (the head of voxel_update: same arithmetic, same order), in two passes over the wavefront's segment
// of the workgroup's LDS copy of the contact rows (`seg`, `nseg` pairs: the rows of its lanes one after the other).  Contact rows
// are short and very uneven (most hold none or one partner, a few a dozen) and nearly all listed pairs are out of reach, so
//   pass 1  the lanes share the wavefront's pairs evenly: reach test of pair p (CalcContactForce's cheap reject, the same
//           arithmetic), a bit in the owner's mask for the pairs in reach;
//   pass 2  every lane adds the forces of its pairs in reach, in list order (the order the reference adds its collision bonds).
// A pair: partner | owner << 10 | place in the owner's row << 20 (rc_code), pair stiffness (rc_a1).  `rowd`: partner count |
// (start of my row in the copy + 1) << VXH_ROWD_BITS; a wavefront whose rows did not fit (nseg < 0) reads them from memory, and so does
// a row of more than 64 partners (the mask of the pairs in reach has 64 bits; rows are as long as the physics makes them, DRobot::col_cap).
enum { VXH_ROWD_BITS = 11, VXH_ROWD_MASK = (1 << VXH_ROWD_BITS) - 1 };
template <int BLOCK>
__device__ __forceinline__ void fused_contact_reach(const double* ps, int seg, int nseg, unsigned long long* mask, const int* rc_code)
{
    for (int p = (int)(threadIdx.x & 63); p < nseg; p += 64) {
        const int code = rc_code[seg + p];
        const int l2 = code & 1023, l1 = (code >> 10) & 1023;
        const double nom = (ps[3 * BLOCK + l2] + ps[3 * BLOCK + l1]) * 0.75;
        if (contact_in_reach(ps[l2] - ps[l1], ps[BLOCK + l2] - ps[BLOCK + l1], ps[2 * BLOCK + l2] - ps[2 * BLOCK + l1], nom))
            atomicOr(&mask[l1], 1ull << ((unsigned)code >> 20));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // (LDS of one wavefront: in order; this keeps the compiler from moving the reads up)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ... pass 1 for the WHOLE workgroup by the wavefronts that hold no bond of the Z slot (resident kernel, two accumulator tiles): they
// idle there, the poses are those the voxel phase is going to test, and barrier (B) stands between these bits and their readers.
// Pairs [first, end) of the copy, shared by `nhelp` wavefronts, this one the `help`-th.  Same test, same bits as fused_contact_reach.
template <int BLOCK>
__device__ __forceinline__ void fused_contact_reach_all(const double* ps, int first, int end, int help, int nhelp, unsigned long long* mask, const int* rc_code)
{
    for (int p = first + (help << 6) + (int)(threadIdx.x & 63); p < end; p += nhelp << 6) {
        const int code = rc_code[p];
        const int l2 = code & 1023, l1 = (code >> 10) & 1023;
        const double nom = (ps[3 * BLOCK + l2] + ps[3 * BLOCK + l1]) * 0.75;
        if (contact_in_reach(ps[l2] - ps[l1], ps[BLOCK + l2] - ps[BLOCK + l1], ps[2 * BLOCK + l2] - ps[2 * BLOCK + l1], nom))
            atomicOr(&mask[l1], 1ull << ((unsigned)code >> 20));
    }
}

template <int BLOCK>
__device__ __forceinline__ d3 fused_contact_forces(const DBatch& B, const DRobot& R, const double* ps, d3 F, d3 pos, double scale, int self, int v, int rowd,
                                                   unsigned long long* mask, const int* rc_code, const double* rc_a1)
{
    if ((rowd >> VXH_ROWD_BITS) != 0) {         // my row is in the LDS copy
        unsigned long long m = mask[self];
        if (m) {
            mask[self] = 0;
            const int roff = (rowd >> VXH_ROWD_BITS) - 1;
            do {
                const int k = __builtin_ctzll(m);
                m &= m - 1;
                const int q = rc_code[roff + k] & 1023;
                F = contact_force_add(F, pos, scale, ps[q], ps[BLOCK + q], ps[2 * BLOCK + q], ps[3 * BLOCK + q], rc_a1[roff + k]);
            } while (m);
        }
        return F;
    }
    const int ccnt = rowd & VXH_ROWD_MASK;
    const int row = R.surf_begin + B.surf_ord[v];
    for (int k0 = 0; k0 < ccnt; k0 += 2) {
        int l[2]; double a1[2], qx[2], qy[2], qz[2], qs[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool on = k0 + j < ccnt;
            const size_t at = col_at(R, k0 + j, row);   // partner-major: coalesced across the wave
            l[j] = on ? B.col_partner[at] - R.vox_begin : -1;
            a1[j] = on ? B.col_a1[at] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int q = l[j] < 0 ? self : l[j]; qx[j] = ps[q]; qy[j] = ps[BLOCK + q]; qz[j] = ps[2 * BLOCK + q]; qs[j] = ps[3 * BLOCK + q]; }
#pragma unroll
        for (int j = 0; j < 2; ++j) if (l[j] >= 0) F = contact_force_add(F, pos, scale, qx[j], qy[j], qz[j], qs[j], a1[j]);
    }
    return F;
}

// `flags`: go | latch << 1 | eol << 2 | rebuild << 3 | trace << 4 | damp_on << 5 -- the resident kernel reads the step's decisions with one
// LDS access instead of a chain of them
struct FusedCtl { double time, act_sin, act_cos, prenatal_c; int go, latch, eol, rebuild, damp_on, trace, trace_index, flags; };

__device__ __forceinline__ void fused_control_begin(const DRobot& R, DRobotState& rs, long long step_cap, int begin_new_step, FusedCtl& K)
{
    const StepCtl c = step_control_begin(R, rs, step_cap, begin_new_step);
    K.go = c.go; K.latch = c.latch; K.eol = c.eol; K.rebuild = 0; K.trace = c.trace; K.trace_index = c.trace_index;
    K.time = rs.cur_time; K.damp_on = rs.dt_prev != 0;
    K.flags = (c.go ? 1 : 0) | (c.latch ? 2 : 0) | (c.eol ? 4 : 0) | (c.trace ? 16 : 0) | (K.damp_on ? 32 : 0);
    K.prenatal_c = actuation_prenatal_c(R, rs.cur_time);
    K.act_sin = K.act_cos = 0;
    // sincos of the actuation phase: a chain of ~200 dependent FP64 instructions on one lane.  During a launch it hides behind the
    // voxel phase; at the START of a launch the whole workgroup waits for it (the prologue's first control, ~5 k cycles per robot).
    // So the call that ends a launch (go = 0 with the robot still pending) computes the values of the time the next launch will start
    // at and leaves them in the control block; the same function of the same argument, hence the same bits.
    if (c.go) {
        if (rs.act_time == rs.cur_time) { K.act_sin = rs.act_sin; K.act_cos = rs.act_cos; }
        else actuation_sincos(R, rs.cur_time, K.act_sin, K.act_cos);
    } else if (rs.status == 0) {
        actuation_sincos(R, rs.cur_time, rs.act_sin, rs.act_cos);
        rs.act_time = rs.cur_time;
    }
}
__device__ __forceinline__ void fused_control_horizon(const DRobot& R, DRobotState& rs, FusedCtl& K, double dt_prev)
{
    StepCtl c; c.go = K.go; c.latch = c.eol = c.rebuild = c.trace = c.trace_index = 0;
    step_control_horizon(R, rs, c, dt_prev);
    K.rebuild = c.rebuild;
    if (c.rebuild) K.flags |= 8;
}
__device__ __forceinline__ void fused_control_horizon(const DRobot& R, DRobotState& rs, FusedCtl& K) { fused_control_horizon(R, rs, K, rs.dt_prev); }

// The same decision in two halves (round 6; k_robot_steps): the control lane stands between barriers (C) and (A) with the whole workgroup
// waiting for it, so what the decision reads of the control block -- everything but the step's max |v|^2, which the voxel phase is still
// collecting -- is fetched BEFORE (C), right behind the next step's control (this lane's own stores), and what it leaves is stored without
// being read back: behind (C) stand one LDS read, a square root, a division and stores instead of six dependent LDS round trips.
// The arithmetic is step_control_horizon's (kernels.hpp; UpdateCollisions, VX_Sim.cpp:1729-1755), operation for operation: same bits.
struct HorizonInputs { double max_disp, dt_prev; int go, rebuilds, steps, flags; };
__device__ __forceinline__ HorizonInputs fused_horizon_prefetch(const DRobotState& rs, const FusedCtl& K)
{
    HorizonInputs in;
    in.max_disp = rs.max_disp; in.dt_prev = rs.dt_prev; in.go = K.go; in.rebuilds = rs.rebuilds; in.steps = rs.steps; in.flags = K.flags;
    return in;
}
__device__ __forceinline__ void fused_horizon_decide(const DRobot& R, DRobotState& rs, FusedCtl& K, const HorizonInputs& in)
{
    int rebuild = 0;
    if (in.go && (R.flags & RF_SELF_COL)) {
        const double mv = vsqrt_nn(__longlong_as_double((long long)rs.maxvel2_bits));
        double disp = in.max_disp + fabs(vdiv(mv * in.dt_prev, R.lat));
        rs.maxvel2_bits = 0ull;
        if (!(R.flags & RF_HORIZON_COL) || disp > (R.col_horizon - 1.0) / 2) { rebuild = 1; disp = 0.0; rs.rebuilds = in.rebuilds + 1; rs.col_tiled = 0; rs.reb_step = in.steps; }
        rs.max_disp = disp;
    }
    rs.rebuild_now = rebuild;
    K.rebuild = rebuild;
    if (rebuild) K.flags = in.flags | 8;
}

// Dispatch order of SHORT launches (a call of a few dozen steps: one launch, two robots per CU one after the other).  A broad-phase run
// is ~42 us, a sixth of a 20-step launch, and the launch ends with its slowest CU: with the robots in their fixed list order the last
// workgroups to be dispatched -- which land on the CUs that a run has already delayed -- bring runs of their own four times out of five
// (scripts/dev_gpu_diag.py launchcost: 69 us of fixed cost per launch against 20 without collisions).  So every workgroup leaves a bit at
// the end of EVERY launch: "my robot is due for a run within a launch like this one, or within VXH_ORDER_HORIZON steps if this one was
// longer" (its displacement since the last run, extrapolated at the average rate since then), and a SHORT launch takes the flagged robots FIRST: workgroup i steps the i-th flagged robot of the list,
// or the (i - flagged)-th unflagged one.  Every workgroup reads the same words (written by the PREVIOUS launch, two buffers), so the
// assignment is a permutation whatever the bits are; the robots do not interact, so the order changes no result.
__device__ __forceinline__ int fused_dispatch_slot(const unsigned long long* __restrict__ bits, int count, int i)
{
    const int nw = (count + 63) >> 6;
    int flagged = 0;
    for (int w = 0; w < nw; ++w) flagged += __builtin_popcountll(bits[w]);
    const bool set = i < flagged;
    int k = set ? i : i - flagged;
    for (int w = 0; w < nw; ++w) {
        unsigned long long x = set ? bits[w] : ~bits[w];
        if (!set && w == nw - 1 && (count & 63)) x &= (1ull << (count & 63)) - 1;
        const int c = __builtin_popcountll(x);
        if (k < c) { for (int j = 0; j < k; ++j) x &= x - 1; return (w << 6) + __builtin_ctzll(x); }
        k -= c;
    }
    return i;      // (not reached: flagged + unflagged == count)
}

template <int BLOCK, int NACC, bool MESH, bool TABG>
__global__ __launch_bounds__(BLOCK, (BLOCK + 255) / 256) void k_robot_steps(DBatch B, const DRobot* __restrict__ robots,
                                                                            const int* __restrict__ robot_list, long long step_cap, int iters,
                                                                            int lds_doubles, const unsigned long long* __restrict__ order_in,
                                                                            unsigned long long* order_out)
{
    extern __shared__ __align__(16) double lds[];
    double* const ps = lds;
    double* const acc = lds + 8 * BLOCK;
    // the 768-thread MESH variant is SLIM: actuation phases and strains stay in HBM, so that two accumulator tiles and the
    // mesh vertices fit into the 160 KB
    constexpr bool SLIM = MESH && BLOCK == 768;
    double* const pht = acc + NACC * 6 * BLOCK;    // sin / cos of every voxel's actuation phase, [2][BLOCK] (not SLIM)
    double* const tabs = pht + (SLIM ? 0 : 2 * BLOCK);
    // the robot's mutable control block lives in LDS for the whole launch: the per-step control is a serial chain
    // of ~20 dependent accesses executed by one thread while the workgroup waits, so it must not touch HBM
    __shared__ DRobotState rs;
    // per-step control words, double-buffered by step parity: thread 0 prepares step n+1 while the slower waves are
    // still in the voxel phase of step n (it idles at the barrier otherwise); only the collision-horizon decision,
    // which needs every voxel's new velocity, stays between the barriers
    __shared__ FusedCtl s_ctl[2];
    __shared__ int s_div, s_na[3], s_seg[2 * (BLOCK / 64)];
    static_assert(sizeof(DRobotState) + 2 * sizeof(FusedCtl) + 5 * sizeof(int) + 2 * 16 * sizeof(int) + 16 <= VXH_FUSED_STATIC_LDS, "static LDS bound");

    const int tid = threadIdx.x;
#ifdef VXH_PHASE_TIMING
    const unsigned long long t_entry = __builtin_readcyclecounter();
#endif
    const int slot = order_in ? __builtin_amdgcn_readfirstlane(fused_dispatch_slot(order_in, (int)gridDim.x, (int)blockIdx.x)) : (int)blockIdx.x;
    const int r = __builtin_amdgcn_readfirstlane(robot_list[slot]);   // robots of one launch group, longest-running first; uniform -> scalar loads of R
    const DRobot& R = robots[r];            // separate noalias argument: its loads stay scalar although the kernel stores to HBM
    const unsigned nv = B.nv;
    const int base = R.vox_begin;
    const bool valid = tid < R.nvox;
    const int v = base + tid;
    if (tid == 0) { rs = B.rstate[r]; s_na[0] = s_na[1] = s_na[2] = 0; }
    // the robot's class tables: copied into LDS, or (TABG: robots with per-voxel evolved stiffness, where nearly every
    // bond and voxel is a class of its own and the tables outgrow the LDS) read from HBM / L2 where they lie
    const int nbd = TABG ? 0 : R.n_bclass * (int)(sizeof(DBondClass) / 8), nvd = TABG ? 0 : R.n_vclass * (int)(sizeof(DVoxClass) / 8);
    const DBondClass* bct;
    const DVoxClass* vct;
    if constexpr (TABG) {
        bct = B.bclass_tab + R.btab_begin;
        vct = B.vclass_tab + R.vtab_begin;
    } else {
        for (int k = tid; k < nbd; k += BLOCK) tabs[k] = ((const double*)(B.bclass_tab + R.btab_begin))[k];
        for (int k = tid; k < nvd; k += BLOCK) tabs[nbd + k] = ((const double*)(B.vclass_tab + R.vtab_begin))[k];
        bct = (const DBondClass*)tabs;
        vct = (const DVoxClass*)(tabs + nbd);
    }
    // MESH (land_water robots): directional strains of the previous step (inputs of the surface mesh: fluid drag, and the
    // RobotVolumeEnd tag on the host) in LDS next to the tables, then the mesh vertices; the 1024-thread variant has no
    // room for the strains and keeps them in HBM
    constexpr bool STRAIN_LDS = MESH && NACC == 2 && !SLIM;
    double* const st = STRAIN_LDS ? tabs + nbd + nvd : B.strain + R.vox_begin;
    const unsigned st_stride = STRAIN_LDS ? (unsigned)BLOCK : nv;
    double* const mesh = tabs + nbd + nvd + (STRAIN_LDS ? 6 * BLOCK : 0);
    // what is left of the dynamic LDS (lds_doubles in all) holds the contact rows of colliding robots: a mask per voxel (fused_contacts),
    // then the pairs: stiffnesses, codes
    unsigned long long* const cmask = (unsigned long long*)(mesh + ((MESH && (R.flags & RF_FLUID)) ? 3 * R.nmv : 0));
    double* const rc_a1 = (double*)cmask + BLOCK;
    const int pool_cap = ((R.flags & RF_SELF_COL) && !VXH_DBG(4)) ? max(0, (int)((lds_doubles - (int)(rc_a1 - lds)) * 2 / 3) - 1) : 0;
    int* const rc_code = (int*)(rc_a1 + pool_cap);
    if constexpr (STRAIN_LDS) {
#pragma unroll
        for (int k = 0; k < 6; ++k) st[k * BLOCK + tid] = tid < R.nvox ? B.strain[(unsigned)k * nv + (unsigned)(R.vox_begin + tid)] : 0.0;
    }
    __syncthreads();

#ifdef VXH_PHASE_TIMING
    unsigned long long t_pro = __builtin_readcyclecounter();
    if (!MESH && B.prof && tid == 0) atomicAdd(&B.prof[2103], t_pro - t_entry);          // tables -> LDS, first barrier
#endif
    // ---- this thread's voxel (momenta -> registers, pose -> LDS, accumulators zeroed) and its three bonds
    const DVoxClass& C = vct[valid ? B.vclass[v] : 0];
    int entry[3];                             // my bond of the X, Y and Z slot of a step (DBatch::bsched), -1 = none
    unsigned modebits = 0;                    // 2 bits per bond: SmallAngle, history layout (DBatch::hist)
    float amp_damp = 1.f;
    d3 lm = mk3(0, 0, 0), am = mk3(0, 0, 0);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        entry[a] = B.bsched[R.sched_begin + a * BLOCK + tid];      // (also threads without a voxel: the Y chunks are dealt to the idle wavefronts)
        if (entry[a] != -1) modebits |= (unsigned)(B.small_angle[(unsigned)a * nv + (base + (entry[a] & 1023))] & 3) << (2 * a);
    }
    // wavefronts that hold bonds of a slot (two tiles: that matters for Z, whose list goes to the threads in order -- the first s_na[2]
    // wavefronts; one tile: X, Y and Z all do); the others run the contact reach test of the whole workgroup there (fused_contact_reach_all)
    if ((tid & 63) == 0) {
#pragma unroll
        for (int a = (NACC == 2 ? 2 : 0); a < 3; ++a) if (entry[a] != -1) atomicAdd(&s_na[a], 1);
    }
    if (valid) {
        const int b0 = rs.steps & 1;
        amp_damp = B.amp_damp[v];
        if constexpr (!SLIM) { pht[tid] = B.act_sb[v]; pht[BLOCK + tid] = B.act_cb[v]; }
        lm = mk3(LINMOM(0, v), LINMOM(1, v), LINMOM(2, v));
        am = mk3(ANGMOM(0, v), ANGMOM(1, v), ANGMOM(2, v));
        ps[tid] = POS(b0, 0, v); ps[BLOCK + tid] = POS(b0, 1, v); ps[2 * BLOCK + tid] = POS(b0, 2, v); ps[3 * BLOCK + tid] = SCALE(b0, v);
        ps[4 * BLOCK + tid] = QUAT(0, v); ps[5 * BLOCK + tid] = QUAT(1, v); ps[6 * BLOCK + tid] = QUAT(2, v); ps[7 * BLOCK + tid] = QUAT(3, v);
    }
#pragma unroll
    for (int k = 0; k < NACC * 6; ++k) acc[k * BLOCK + tid] = 0.0;
    const FetchLds<BLOCK> fetch{ps, base};
    DragCache<BLOCK> dcache;
    if constexpr (MESH && DragCache<BLOCK>::KEEP) { if ((R.flags & RF_FLUID) && R.nmv > 0 && R.nfacet > 0) { dcache.load(B, R, tid); if (valid) { dcache.my_first = B.facet_first[v]; dcache.my_count = (int)B.facet_count[v]; } } }
    // my contact row: partner count | (start of the workgroup's LDS copy of it + 1) << VXH_ROWD_BITS; s_seg: where the rows of each wavefront's
    // lanes start in the copy and how many pairs they hold (-1: they did not fit, that wavefront reads its rows from memory).
    // Refreshed after every broad-phase run.  (every thread calls: barriers inside)
    int rowd = 0;
    auto rows_to_lds = [&](bool at_launch) {
        rowd = 0;
        if (!(R.flags & RF_SELF_COL)) return;
        // (see opaque_tid.  The 1024-thread variant uses the kernel's own index: with a local `tid` re-read from threadIdx.x here -- the
        // same value -- dense 10^3 robots ran 4-5 % slower, 56.1-56.7 against 53.3-54.5 us per step, same box, interleaved; without it
        // they run as before, 52.9-53.6 against 53.2-53.5.  Why is not known: the ISA of the two differs by register numbering.)
        int tid_r = tid;
        if constexpr (BLOCK != 1024) tid_r = opaque_tid<BLOCK>();
#ifdef VXH_PHASE_TIMING
        unsigned long long t_r2l = __builtin_readcyclecounter();
#define VXH_R2L_MARK(slot) { const unsigned long long t_now = __builtin_readcyclecounter(); if (B.prof && tid == 0) atomicAdd(&B.prof[slot], t_now - t_r2l); t_r2l = t_now; }
#else
#define VXH_R2L_MARK(slot)
#endif
        const int img = R.img_index;
        constexpr int NW = BLOCK / 64;
        if (at_launch && img >= 0 && rs.rows_img != 0 && !VXH_DBG(1)) {
            // the copy as the previous launch left it (DBatch::rimg_*): one round trip of coalesced loads.  The rows only change in a
            // broad-phase run, and every run ends in the save below.
            const int* const seg = B.rimg_seg + (size_t)img * 64;
            const int used = __builtin_amdgcn_readfirstlane(seg[2 * NW]);
            if (valid) rowd = B.rimg_rowd[v];
            if (tid_r < 2 * NW) s_seg[tid_r] = seg[tid_r];
            for (int k = tid_r; k < used; k += BLOCK) { rc_code[k] = B.rimg_code[(size_t)img * VXH_RIMG_CAP + k]; rc_a1[k] = B.rimg_a1[(size_t)img * VXH_RIMG_CAP + k]; }
            if (pool_cap > 0) cmask[tid_r] = 0;
            __syncthreads();
            VXH_R2L_MARK(2116)
            return;
        }
        int row = -1;
        if (valid) { const int so = B.surf_ord[v]; if (so >= 0) row = R.surf_begin + so; }
        const int ccnt = (row >= 0 && !VXH_DBG(1)) ? B.col_cnt[row] : 0;
        if (pool_cap > 0) cmask[tid_r] = 0;
        __syncthreads();
        VXH_R2L_MARK(2114)     // surface ordinal -> row count (two dependent loads), first barrier
        const int ccnt_l = ccnt <= 64 ? ccnt : 0;     // (a longer row stays in memory: 64 mask bits per voxel)
        int incl = ccnt_l;                    // places in the copy: prefix sum within the wavefront, one atomic per wavefront
        const int lane = tid_r & 63;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        const int wave_total = __shfl(incl, 63);
        // the wavefronts' segments follow each other in wavefront order: which rows fit (and which are read from memory) must not
        // depend on who got there first, the two paths need not round alike
        if (lane == 0) s_seg[2 * (tid_r >> 6) + 1] = wave_total;
        __syncthreads();
        int wave_base = 0, all_total = 0;
        for (int w = 0; w < NW; ++w) { const int t = s_seg[2 * w + 1]; if (w < (tid_r >> 6)) wave_base += t; all_total += t; }
        __syncthreads();                      // (s_seg is rewritten below)
        VXH_R2L_MARK(2115)     // scan, totals, two barriers, prefix over the wavefronts
        const bool fits = wave_base + wave_total <= pool_cap;
        if (lane == 0) { s_seg[2 * (tid_r >> 6)] = wave_base; s_seg[2 * (tid_r >> 6) + 1] = fits ? wave_total : -1; }
        // (the developer build used to count wavefront copies and overflows here with global atomics from every wavefront: contended, and the
        // barrier below waits for them -- this function took 52 k cycles per call with them and takes 13.5 k without, and the difference
        // was for a while mistaken for a cost of copying the rows; removed)
        const int off = wave_base + incl - ccnt_l;
        rowd = ccnt;
        if (fits && ccnt_l > 0) {
            rowd = ccnt | ((off + 1) << VXH_ROWD_BITS);
            for (int k0 = 0; k0 < ccnt; k0 += 4) {           // (four entries' loads in flight: a row of 16 costs four round trips, not sixteen)
                int pj[4]; double aj[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const size_t at = col_at(R, min(k0 + j, ccnt - 1), row); pj[j] = B.col_partner[at]; aj[j] = B.col_a1[at]; }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k0 + j < ccnt) { rc_code[off + k0 + j] = (pj[j] - base) | (tid_r << 10) | ((k0 + j) << 20); rc_a1[off + k0 + j] = aj[j]; }
            }
        }
        __syncthreads();
        VXH_R2L_MARK(2116)     // the copy itself, last barrier
        if (img >= 0 && !VXH_DBG(1)) {
            // save the copy for the next launches (stores only: nothing waits for them)
            const int used = min(min(all_total, pool_cap), (int)VXH_RIMG_CAP);
            int* const seg = B.rimg_seg + (size_t)img * 64;
            if (valid) B.rimg_rowd[v] = rowd;
            if (tid_r < 2 * NW) seg[tid_r] = s_seg[tid_r];
            if (tid_r == 0) { seg[2 * NW] = used; rs.rows_img = (all_total <= VXH_RIMG_CAP || pool_cap <= VXH_RIMG_CAP) ? 1 : 0; }
            for (int k = tid_r; k < used; k += BLOCK) { B.rimg_code[(size_t)img * VXH_RIMG_CAP + k] = rc_code[k]; B.rimg_a1[(size_t)img * VXH_RIMG_CAP + k] = rc_a1[k]; }
        }
    };

    // the control thread sits in the LAST wave: the one with the fewest (often no) voxels, so its serial work hides
    // behind the other waves' voxel phase
    const bool ctl_thread = tid == BLOCK - 64;
    if (ctl_thread) { fused_control_begin(R, rs, step_cap, iters > 0, s_ctl[0]); fused_control_horizon(R, rs, s_ctl[0]); s_div = 0; }
#ifdef VXH_PHASE_TIMING
    { const unsigned long long t_now = __builtin_readcyclecounter(); if (!MESH && B.prof && tid == 0) atomicAdd(&B.prof[2106], t_now - t_pro); t_pro = t_now; }   // state load, zeroing, control
#endif
    rows_to_lds(true);
#ifdef VXH_PHASE_TIMING
    { const unsigned long long t_now = __builtin_readcyclecounter(); if (!MESH && B.prof && tid == 0) atomicAdd(&B.prof[2113], t_now - t_pro); t_pro = t_now; }   // rows_to_lds as a whole
#endif
    __syncthreads();                           // control of the first step + every voxel's pose visible
    constexpr int NWAVES = BLOCK / 64;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (one tile, 1024 threads: a third of the pairs in each of the three rounds -- a dense lattice leaves ONE wavefront idle per round)
    const int nz = __builtin_amdgcn_readfirstlane(s_na[2]);
    const int nx = NACC == 1 ? __builtin_amdgcn_readfirstlane(s_na[0]) : 0, ny = NACC == 1 ? __builtin_amdgcn_readfirstlane(s_na[1]) : 0;
    const bool reach_early = (R.flags & RF_SELF_COL) && nz < NWAVES && nx < NWAVES && ny < NWAVES && !VXH_DBG(32);     // (uniform)
    // pairs of the LDS copy that are really there: the wavefronts whose rows fit are a prefix (a wavefront's place is the sum of ALL
    // earlier totals), so the sum of their totals is the end of the valid stretch
    auto pairs_in_copy = [&]() { int n = 0; for (int w = 0; w < NWAVES; ++w) n += max(s_seg[2 * w + 1], 0); return __builtin_amdgcn_readfirstlane(n); };
    int npairs = reach_early ? pairs_in_copy() : 0;
    VXH_T_DECL
#ifdef VXH_PHASE_TIMING
    unsigned long long reb_cycles = 0;
    if (!MESH && B.prof && (tid & 63) == 0) atomicAdd(&B.prof[(tid >> 6) * 8 + 6], __builtin_readcyclecounter() - t_entry);   // prologue
#endif
    int it = 0;
    for (;; ++it) {
        const FusedCtl& K = s_ctl[it & 1];
        FusedCtl& Knext = s_ctl[(it + 1) & 1];
        const int kf = __builtin_amdgcn_readfirstlane(K.flags);
        const bool k_go = kf & 1, k_latch = kf & 2, k_eol = kf & 4, k_rebuild = kf & 8, k_trace = kf & 16;
        if (!k_go && !k_trace) break;
        // opaque per-step copies: keeps the compiler from hoisting every address of the step out of the loop (dozens of
        // loop-invariant 64-bit pointers, which it then spills)
        int vv = v;
        asm volatile("" : "+v"(vv));
        bool scratch_used = false;            // latch / broad-phase borrow the accumulator tile
        if (k_latch || k_eol || k_trace) {
            fused_latch_cm<BLOCK>(R, rs, ps, acc, valid, C, k_latch, k_eol, k_trace, B.trace + (size_t)(R.trace_begin + K.trace_index) * 4);
            scratch_used = true;
        }
        if (!k_go) break;                      // (the robot has stopped; the last step's trace point, if one was due, is in)
#ifdef VXH_PHASE_TIMING
        const unsigned long long t_reb0 = __builtin_readcyclecounter();
#endif
        if (__builtin_expect(k_rebuild, 0)) {
            fused_rebuild<BLOCK>(B, R, rs, ps, acc, NACC * 6 * BLOCK, (double*)cmask, max(0, lds_doubles - (int)((double*)cmask - lds)), vct);
#ifdef VXH_PHASE_TIMING
            if (VXH_DBG(16) && B.prof) {     // cross-check: the plain scan must leave the same rows (count, partners, stiffness bits)
                int my_row = -1, n_new = 0; unsigned long long sum_new = 0;
                if (tid < R.nsurf) {
                    my_row = R.surf_begin + tid; n_new = B.col_cnt[my_row];
                    for (int k = 0; k < n_new; ++k) { const size_t at = col_at(R, k, my_row); sum_new += (unsigned long long)(k + 1) * ((unsigned long long)B.col_partner[at] * 1000003ull + (unsigned long long)__double_as_longlong(B.col_a1[at])); }
                }
                __syncthreads();
                fused_rebuild_plain<BLOCK>(B, R, rs, ps, (int*)acc, vct);
                if (tid < R.nsurf) {
                    const int n_old = B.col_cnt[my_row]; unsigned long long sum_old = 0;
                    for (int k = 0; k < n_old; ++k) { const size_t at = col_at(R, k, my_row); sum_old += (unsigned long long)(k + 1) * ((unsigned long long)B.col_partner[at] * 1000003ull + (unsigned long long)__double_as_longlong(B.col_a1[at])); }
                    atomicAdd(&B.prof[2104], 1ull);
                    if (n_old != n_new || sum_old != sum_new) atomicAdd(&B.prof[2105], 1ull);
                    if (n_new > n_old) atomicAdd(&B.prof[2120], 1ull);
                    if (n_new < n_old) atomicAdd(&B.prof[2121], 1ull);
                    atomicMax(&B.prof[2122], (unsigned long long)n_new);
                    atomicMax(&B.prof[2123], (unsigned long long)n_old);
                    atomicAdd(&B.prof[2124], (unsigned long long)n_new);
                    atomicAdd(&B.prof[2125], (unsigned long long)n_old);
                }
                __syncthreads();
            }
#endif
            rows_to_lds(false);
            if (reach_early) npairs = pairs_in_copy();
#pragma unroll
            for (int k = 1; k < NACC * 6; ++k) acc[k * BLOCK + tid] = 0.0;      // the scratch of the broad-phase (plane 0: below)
            scratch_used = true;
        }
#ifdef VXH_PHASE_TIMING
        // developer build: broad-phase runs (incl. the copy of the rows) and their cycles, wave 0 of every workgroup; the largest
        // number of cycles any one workgroup spent in them during its launch goes to slot 110 (atomicMax at the kernel's end)
        if (k_rebuild && B.prof && tid == 0) { const unsigned long long dtc = __builtin_readcyclecounter() - t_reb0; atomicAdd(&B.prof[2108], 1ull); atomicAdd(&B.prof[2109], dtc); reb_cycles += dtc; }
#endif
        if (scratch_used) { acc[tid] = 0.0; __syncthreads(); }
        d3 drag = mk3(0, 0, 0);
        const bool fluid = MESH && (R.flags & RF_FLUID) != 0;
        if constexpr (MESH) { if (fluid) drag = fused_drag<BLOCK, NACC * 6 * BLOCK>(B, R, ps, st, st_stride, mesh, acc, valid, vv, lm, C.mass_inv, dcache); }
        const bool damp_on = (kf & 32) != 0;
        VXH_T_MARK(1)

        // ---- bond phase.  Two accumulator tiles: the X and Y slots back to back, no barrier between them (every accumulator entry
        // gets at most one X and one Y contribution, both ADDED to the zero the voxel phase left: a + b == b + a, the sums are those of
        // barrier-separated rounds bit for bit; the Y chunks sit on the wavefronts X leaves idle: DBatch::bsched), barrier, the Z slot
        // on top.  One tile (1024 threads): three rounds, each in two barrier-separated sub-steps.
        // (Measured and not kept, round 4: Z EVALUATED in the same barrier-free stretch, its chunks dealt like Y's -- 21 chunks on 12
        // wavefronts, 6 / 5 / 5 / 5 per SIMD -- and its twelve outputs held in registers across the barrier that orders them behind
        // X + Y: the held outputs raise the peak of the bond's register need, 28 -> 124 B of scratch, 25.5 -> 25.9 us per step.)
        bool div = false;
        fused_round<0, BLOCK, NACC, MESH>(B, R, bct, ps, acc, entry[0], modebits, damp_on, div, st, st_stride);
        if constexpr (NACC == 1) {
            if (reach_early && wave >= nx) fused_contact_reach_all<BLOCK>(ps, 0, npairs / 3, wave - nx, NWAVES - nx, cmask, rc_code);
            __syncthreads();
        }
        fused_round<1, BLOCK, NACC, MESH>(B, R, bct, ps, acc, entry[1], modebits, damp_on, div, st, st_stride);
        if constexpr (NACC == 1) { if (reach_early && wave >= ny) fused_contact_reach_all<BLOCK>(ps, npairs / 3, 2 * (npairs / 3), wave - ny, NWAVES - ny, cmask, rc_code); }
        __syncthreads();
        fused_round<2, BLOCK, NACC, MESH>(B, R, bct, ps, acc, entry[2], modebits, damp_on, div, st, st_stride);
        if (reach_early && wave >= nz) fused_contact_reach_all<BLOCK>(ps, NACC == 1 ? 2 * (npairs / 3) : 0, npairs, wave - nz, NWAVES - nz, cmask, rc_code);
        if (div) s_div = 1;
        VXH_T_MARK(2)
        __syncthreads();                       // (B)
        VXH_T_MARK(3)
        if (s_div) {                           // Integrate() returns before the voxel loop (VX_Sim.cpp:1777)
            __syncthreads();                   // everyone has read s_div
            if (ctl_thread) { rs.diverged = 1; fused_control_begin(R, rs, step_cap, 0, Knext); s_div = 0; }   // -> status diverged, go = 0
            __syncthreads();
            continue;
        }
        // ---- voxel phase: sums from the accumulators (zeroed again for the next step), pose from the tile, momenta in
        // registers; contact partners from the pose tile
        double vel2 = 0;
        VoxState S;
        if (R.flags & RF_SELF_COL) {
            if (!reach_early) {                             // (else: done in the Z slot by the wavefronts without a Z bond)
                const int nseg = s_seg[2 * (tid >> 6) + 1];     // pairs in the LDS copy of my wavefront's contact rows (-1: the rows are read from memory)
                if (nseg > 0) fused_contact_reach<BLOCK>(ps, s_seg[2 * (tid >> 6)], nseg, cmask, rc_code);   // every lane: the wavefront shares the pairs
            }
        }
        if (valid) {
            d3 F = mk3(acc[tid], acc[BLOCK + tid], acc[2 * BLOCK + tid]), M = mk3(acc[3 * BLOCK + tid], acc[4 * BLOCK + tid], acc[5 * BLOCK + tid]);
            if constexpr (NACC == 2) {
                F = F + mk3(acc[6 * BLOCK + tid], acc[7 * BLOCK + tid], acc[8 * BLOCK + tid]);
                M = M + mk3(acc[9 * BLOCK + tid], acc[10 * BLOCK + tid], acc[11 * BLOCK + tid]);
            }
            S.pos = mk3(ps[tid], ps[BLOCK + tid], ps[2 * BLOCK + tid]); S.scale = ps[3 * BLOCK + tid];
            S.ang = mkq(ps[4 * BLOCK + tid], ps[5 * BLOCK + tid], ps[6 * BLOCK + tid], ps[7 * BLOCK + tid]);
            S.lm = lm; S.am = am;
            // (measured and not kept: the class constants copied by value here, so that their LDS reads go out with the sums and the pose
            // instead of one by one inside the update -- 23.45 -> 23.68 us per step, and other contraction choices, i.e. other last bits)
            const d3 vel = S.lm * C.mass_inv;
            F = F + (vel * (-R.slow_z)) * C.c_lin;
#ifdef VXH_PHASE_TIMING
            const d3 F_before = F;
            const unsigned long long mask_seen = ((rowd >> VXH_ROWD_BITS) != 0) ? cmask[tid] : 0ull;     // what pass 2 is about to consume
#endif
            if (rowd != 0) F = fused_contact_forces<BLOCK>(B, R, ps, F, S.pos, S.scale, tid, vv, rowd, cmask, rc_code, rc_a1);
#ifdef VXH_PHASE_TIMING
            if (VXH_DBG(8) && B.prof && (rowd >> VXH_ROWD_BITS) != 0) {     // the same sum through the rows in memory: must give the same bits
                const d3 G = fused_contact_forces<BLOCK>(B, R, ps, F_before, S.pos, S.scale, tid, vv, rowd & VXH_ROWD_MASK, cmask, rc_code, rc_a1);
                atomicAdd(&B.prof[2115], 1ull);
                if (!(G.x == F.x && G.y == F.y && G.z == F.z)) {
                    atomicAdd(&B.prof[2116], 1ull);
                    // pairs of my row in reach by CalcContactForce's own test, against the bits pass 1 set for me
                    int reach = 0;
                    const int crow0 = R.surf_begin + B.surf_ord[vv];
                    for (int k = 0; k < (rowd & VXH_ROWD_MASK); ++k) {
                        const int q = B.col_partner[col_at(R, k, crow0)] - base;
                        if (contact_in_reach(ps[q] - S.pos.x, ps[BLOCK + q] - S.pos.y, ps[2 * BLOCK + q] - S.pos.z, (ps[3 * BLOCK + q] + S.scale) * 0.75)) ++reach;
                    }
                    const int bits = __builtin_popcountll(mask_seen);
                    atomicAdd(&B.prof[bits < reach ? 2118 : (bits > reach ? 2119 : 2111)], 1ull);
                }
                // is the LDS copy of my row what memory holds now?  (partner and stiffness of every entry, bit for bit)
                const int crow = R.surf_begin + B.surf_ord[vv], coff = (rowd >> VXH_ROWD_BITS) - 1;
                bool same = true;
                for (int k = 0; k < (rowd & VXH_ROWD_MASK); ++k) {
                    const size_t at = col_at(R, k, crow);
                    same = same && (rc_code[coff + k] & 1023) == B.col_partner[at] - base && rc_a1[coff + k] == B.col_a1[at];
                }
                if (B.col_cnt[crow] != (rowd & VXH_ROWD_MASK)) same = false;
                if (!same) atomicAdd(&B.prof[2117], 1ull);
            }
#endif
            vel2 = voxel_update(B, R, C, vv, fetch, K.time, K.act_sin, K.act_cos, K.prenatal_c, F, M, vel, S, -1, 0, fluid, drag,
                                SLIM ? B.act_sb[vv] : pht[tid], SLIM ? B.act_cb[vv] : pht[BLOCK + tid], amp_damp);
            lm = S.lm; am = S.am;
            // the accumulators zeroed for the next step -- HERE, behind the update: stores into the LDS between the reads above and the
            // update's own (class table, contact rows) kept the compiler from issuing those reads together (23.75 -> 23.43 us per step)
#pragma unroll
            for (int k = 0; k < NACC * 6; ++k) acc[k * BLOCK + tid] = 0.0;
        }
        // next step's control, off the critical path.  (Round 4, measured and taken out again: the same call issued earlier -- next to the
        // X / Y chunks with one Y chunk fewer on this wavefront, 25.2 -> 26.0 us per step: that slot is issue-bound on every SIMD; or
        // during the Z slot, which leaves this wavefront idle, 25.3 -> 25.45: the developer timers show this wavefront last at barrier
        // (C), but the voxel wavefronts sharing its SIMD finish at the same time with or without it.)
        HorizonInputs hz = {0.0, 0.0, 0, 0, 0, 0};
        if (ctl_thread) { fused_control_begin(R, rs, step_cap, it + 1 < iters, Knext); hz = fused_horizon_prefetch(rs, Knext); }
        if ((R.flags & RF_SELF_COL) && !VXH_DBG(2)) {            // SS.MaxVoxVel for the collision horizon (VX_Sim.cpp:1625-1649)
            vel2 = wave_max_nonneg(vel2);       // (DPP: VALU speed)
            if ((tid & 63) == 0) atomicMax(&rs.maxvel2_bits, (unsigned long long)__double_as_longlong(vel2));
        }
        VXH_T_MARK(4)
        __syncthreads();                       // (C) every read of the old poses is done
        if (valid) {
            ps[tid] = S.pos.x; ps[BLOCK + tid] = S.pos.y; ps[2 * BLOCK + tid] = S.pos.z; ps[3 * BLOCK + tid] = S.scale;
            ps[4 * BLOCK + tid] = S.ang.w; ps[5 * BLOCK + tid] = S.ang.x; ps[6 * BLOCK + tid] = S.ang.y; ps[7 * BLOCK + tid] = S.ang.z;
        }
        VXH_T_MARK(5)
        if (ctl_thread) { fused_horizon_decide(R, rs, Knext, hz); s_div = 0; }
        __syncthreads();                       // (A) control + every voxel's published pose visible
        VXH_T_MARK(0)
    }
    VXH_T_FLUSH
#ifdef VXH_PHASE_TIMING
    const unsigned long long t_loop_end = __builtin_readcyclecounter();
    if (B.prof && tid == 0) { atomicMax(&B.prof[2110], reb_cycles); atomicAdd(&B.prof[2107], reb_cycles > 0 ? 1ull : 0ull); }
#endif
    // ---- back to HBM: state into the buffer the step count selects, flag bits, control block
    if (valid) {
        const int b1 = rs.steps & 1;
        POS(b1, 0, v) = ps[tid]; POS(b1, 1, v) = ps[BLOCK + tid]; POS(b1, 2, v) = ps[2 * BLOCK + tid];
        SCALE(b1, v) = ps[3 * BLOCK + tid];
        QUAT(0, v) = ps[4 * BLOCK + tid]; QUAT(1, v) = ps[5 * BLOCK + tid]; QUAT(2, v) = ps[6 * BLOCK + tid]; QUAT(3, v) = ps[7 * BLOCK + tid];
        LINMOM(0, v) = lm.x; LINMOM(1, v) = lm.y; LINMOM(2, v) = lm.z;
        ANGMOM(0, v) = am.x; ANGMOM(1, v) = am.y; ANGMOM(2, v) = am.z;
    }
    if constexpr (STRAIN_LDS) {
        if (valid) {
#pragma unroll
            for (int k = 0; k < 6; ++k) B.strain[(unsigned)k * nv + (unsigned)v] = st[k * BLOCK + tid];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        if (entry[a] != -1) B.small_angle[(unsigned)a * nv + (base + (entry[a] & 1023))] = (unsigned char)((modebits >> (2 * a)) & 3u);
    if (tid == 0) {
        B.rstate[r] = rs; if (B.rstate_mirror) B.rstate_mirror[r] = rs;
        if (order_out) {       // due for a broad-phase run within a launch like this one?  (see fused_dispatch_slot)
            bool soon = false;
            if (rs.status == 0 && (R.flags & RF_SELF_COL) && (R.flags & RF_HORIZON_COL)) {
                const double rate = rs.max_disp / (double)max(1, rs.steps - rs.reb_step);
                // (`it`: the steps of this launch; a long launch -- whose own order does not matter -- leaves the flags for a short one behind it.
                // Generous: a robot flagged for nothing costs nothing, a run in the second round of a late CU costs the launch 42 us; 2.5 against
                // 1.25 launch lengths: the driver's command 1.383-1.405e10 against 1.380-1.403e10, same box, alternating)
                soon = rs.max_disp + rate * (2.5 * (double)min(max(1, it), (int)VXH_ORDER_HORIZON)) > (R.col_horizon - 1.0) / 2;
            }
            // (the slot is worked out again rather than kept: a scalar register held across the step loop costs the loop a spill)
            const int my = order_in ? fused_dispatch_slot(order_in, (int)gridDim.x, (int)blockIdx.x) : (int)blockIdx.x;
            const unsigned long long bit = 1ull << (my & 63);
            if (soon) atomicOr(&order_out[my >> 6], bit); else atomicAnd(&order_out[my >> 6], ~bit);
        }
    }
#ifdef VXH_PHASE_TIMING
    __builtin_amdgcn_s_waitcnt(0);          // (the write-back has left the wavefront)
    if (!MESH && B.prof && (tid & 63) == 0) atomicAdd(&B.prof[(tid >> 6) * 8 + 7], __builtin_readcyclecounter() - t_loop_end);   // epilogue
#endif
}

}  // namespace vxh
